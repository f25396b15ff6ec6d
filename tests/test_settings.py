"""`CircuitSettings` with lookups (crates/air/src/settings.rs, preprocessed.rs:34-46, lookups/sin/mod.rs:20-24,
lookups/range_check/mod.rs:23-37): bincode / JSON round trips of the reference's own layout form, and the LUT columns
generated behind the C ABI (`lmn_lut_from_ranges`, restating `SinPreProcessed::gen_column`, preprocessed.rs:351-383)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import luminair_amd
from luminair_amd import backend, synthetic as syn
from luminair_amd.pie import CircuitSettings, Lookup, LookupLayout, RangeCheckLookup


@pytest.fixture(scope="module")
def lib(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Library(so)


def _settings():
    return CircuitSettings(
        layouts={"sin": Lookup(LookupLayout([(-300, -100), (5, 40)], 8), list(range(256))),
                 "log2": Lookup(LookupLayout([(1, 16)], 4), [3] * 16)},
        range_check=RangeCheckLookup([8], 8, [1] * 256))


def test_settings_bincode_layout_and_round_trip():
    s = _settings()
    b = s.to_bincode()
    # Lookups { sin: Some(..), exp2: None, log2: Some(..), range_check: Some(..) }, field by field
    assert b[0] == 1 and struct.unpack_from("<Q", b, 1)[0] == 2                      # Some, 2 ranges
    assert struct.unpack_from("<qqqq", b, 9) == (-300, -100, 5, 40)                 # Range(Fixed, Fixed) = two i64
    assert struct.unpack_from("<I", b, 41)[0] == 8                                  # log_size
    assert struct.unpack_from("<Q", b, 45)[0] == 256                                # multiplicities.data: Vec<AtomicU32>
    off = 53 + 4 * 256
    assert b[off] == 0                                                              # exp2: None
    back = CircuitSettings.from_bincode(b)
    assert back.layouts["sin"].layout.ranges == [(-300, -100), (5, 40)] and back.layouts["sin"].multiplicities == list(range(256))
    assert "exp2" not in back.layouts and back.layouts["log2"].layout.log_size == 4
    assert back.range_check.ranges == [8] and back.range_check.log_size == 8 and len(back.range_check.multiplicities) == 256
    assert back.to_bincode() == b
    for bad in (b[:-1], b + b"\x00", b"\x02" + b[1:]):
        with pytest.raises(luminair_amd.LuminairError):
            CircuitSettings.from_bincode(bad)
    assert CircuitSettings().to_bincode() == bytes(4) and CircuitSettings.from_bincode(bytes(4)).layouts is None


def test_settings_json_round_trip():
    s = _settings()
    j = s.to_json()
    d = json.loads(j)
    assert d["lookups"]["exp2"] is None
    assert d["lookups"]["sin"]["layout"] == {"ranges": [[-300, -100], [5, 40]], "log_size": 8}
    assert d["lookups"]["range_check"]["layout"] == {"ranges": [8], "log_size": 8}
    assert CircuitSettings.from_json(j).to_bincode() == s.to_bincode()
    with pytest.raises(luminair_amd.LuminairError):
        CircuitSettings.from_json('{"lookups": {"sin": {"layout": 3}}}')


@pytest.mark.parametrize("name,lo,hi", [("sin", -4 * 4096, 4 * 4096), ("exp2", -2 * 4096, 2 * 4096), ("log2", 1, 4 * 4096),
                                        ("sin", -7, 9)])
def test_lut_from_ranges_matches_the_host_generator(lib, name, lo, hi):
    want0, want1 = syn.make_lut(name, lo, hi)
    assert lib.lut_log_size([(lo, hi)]) == len(want0).bit_length() - 1
    c0, c1 = lib.lut_from_ranges(name, [(lo, hi)])
    assert np.array_equal(c0, want0) and np.array_equal(c1, want1)
    # ranges in any order, overlapping: values are sorted and de-duplicated (gen_column: sort_unstable + dedup)
    mid = (lo + hi) // 2
    d0, d1 = lib.lut_from_ranges(name, [(mid, hi), (lo, mid + 3 if mid + 3 <= hi else mid)], len(want0).bit_length() - 1)
    assert np.array_equal(d0, want0) and np.array_equal(d1, want1)


def test_lut_from_ranges_rejects_bad_input(lib):
    for name, ranges, log in (("sin", [(5, 4)], 4), ("log2", [(0, 5)], 4), ("sin", [(0, 100)], 4), ("sin", [], 4)):
        with pytest.raises(backend.LuminairBackendError):
            lib.lut_from_ranges(name, ranges, log)


def test_prove_from_layout_settings_equals_prove_from_columns(lib):
    """`lmn_prove` fed by the reference's own settings form (ranges -> columns behind the boundary) gives the bytes
    of the same proof fed by pre-expanded LUT columns, and the verifier accepts it together with those settings."""
    ranges = {"sin": (-700, 900), "exp2": (-300, 200)}        # small LUTs (2^11 and 2^9 rows): the emulation is slow
    tabs, luts = syn.activation_graph(30, 4, names=("sin", "exp2"), ranges=ranges)
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    want = ctx.prove_tables([(k, r, len(r)) for k, r in tabs], luts)
    mult = {4: None, 10: None}
    for k, r in tabs:
        if k in mult:
            mult[k] = [int(v) for v in r[:, 0]]
    settings = CircuitSettings(layouts={
        "sin": Lookup(LookupLayout([ranges["sin"]], lib.lut_log_size([ranges["sin"]])), mult[4]),
        "exp2": Lookup(LookupLayout([ranges["exp2"]], lib.lut_log_size([ranges["exp2"]])), mult[10])})
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs], settings.lut_columns(lib))
    assert got == want
    settings2 = CircuitSettings.from_bincode(settings.to_bincode())
    luminair_amd.verify(luminair_amd.LuminairProof(got), settings2, backend.VARIANT_PINNED, library=lib)
    wrong = CircuitSettings(layouts={"log2": Lookup(LookupLayout([(1, 16)], 4), [0] * 16)})
    with pytest.raises(luminair_amd.LuminairError):
        luminair_amd.verify(luminair_amd.LuminairProof(got), wrong, backend.VARIANT_PINNED, library=lib)
    ctx.close()


def test_proof_json_round_trip(kat_bytes, lib):
    """`LuminairProof::to_json / from_json` mirror (crates/prover/src/lib.rs:62-106): the reference's known-answer
    proof (KAT-era 8-slot claim) and a HEAD-layout proof survive bincode -> JSON -> bincode unchanged."""
    p = luminair_amd.LuminairProof(kat_bytes)
    d = json.loads(p.to_json())
    assert list(d["claim"]) == ["add", "mul", "recip", "sin", "sin_lookup", "sum_reduce", "max_reduce", "sqrt"]
    # serde writes `_marker: PhantomData<T>` of `Claim<T>` (components/mod.rs:149-152) as null and requires it back
    assert d["claim"]["add"] == {"log_size": 4, "_marker": None} and d["claim"]["recip"] is None
    assert '"_marker": null' in p.to_json()
    stripped = json.loads(p.to_json())
    for c in stripped["claim"].values():
        if c is not None:
            del c["_marker"]            # JSON written by round 2's mirror (no _marker) is still accepted
    assert luminair_amd.LuminairProof.from_json(json.dumps(stripped)).to_bincode() == kat_bytes
    assert d["proof"]["config"] == {"pow_bits": 5, "fri_config": {"log_blowup_factor": 1, "log_last_layer_degree_bound": 0,
                                                                   "n_queries": 3}}
    assert len(d["proof"]["commitments"]) == 4 and len(d["proof"]["commitments"][0]) == 32
    assert luminair_amd.LuminairProof.from_json(p.to_json()).to_bincode() == kat_bytes
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    tabs = syn.config2_graph_faithful(40, 3)
    head = luminair_amd.LuminairProof(ctx.prove_tables([(k, r, len(r)) for k, r in tabs]))
    dh = head.to_dict()
    assert len(dh["claim"]) == 17 and dh["claim"]["inputs"] is not None
    assert luminair_amd.LuminairProof.from_json(head.to_json()).to_bincode() == head.to_bincode()
    with pytest.raises(luminair_amd.LuminairError):
        luminair_amd.LuminairProof.from_json('{"claim": {}}')
    with pytest.raises(luminair_amd.LuminairError):
        luminair_amd.LuminairProof(kat_bytes[:-1]).to_json()
    ctx.close()


def test_verify_with_lookups_form_settings_and_a_range_check_claim(lib):
    """a sin + LessThan proof verified with the pre-expanded `lookups` form of the settings - which
    names the sin LUT but not the library-generated 8-bit range-check LUT - must be accepted, as must settings that do
    announce the range check; settings announcing a lookup the claim lacks are still rejected."""
    ranges = {"sin": (-300, 200)}
    tabs, luts = syn.activation_graph(30, 4, names=("sin",), ranges=ranges)
    lt = syn.less_than_graph(30, 5)
    # one pie: the sin graph's tables + the LessThan graph's (each graph's logup sums cancel on their own, and the
    # relation is linear, so shared tensor ids do no harm)
    merged = {}
    for k, r in tabs + lt:
        merged[k] = np.concatenate([merged[k], r]) if k in merged else r
    tables = sorted(merged.items())
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    proof = luminair_amd.LuminairProof(ctx.prove_tables([(k, r, len(r)) for k, r in tables], luts))
    ctx.close()
    luminair_amd.verify(proof, None, backend.VARIANT_PINNED, library=lib)
    luminair_amd.verify(proof, CircuitSettings(lookups={"sin": luts["sin"]}), backend.VARIANT_PINNED, library=lib)
    luminair_amd.verify(proof, CircuitSettings(lookups={"sin": luts["sin"]}, range_check=RangeCheckLookup([8], 8, [0] * 256)),
                        backend.VARIANT_PINNED, library=lib)
    with pytest.raises(luminair_amd.LuminairError):     # exp2 announced, not in the claim
        luminair_amd.verify(proof, CircuitSettings(lookups={"sin": luts["sin"], "exp2": luts["sin"]}),
                            backend.VARIANT_PINNED, library=lib)
    # a proof WITHOUT LessThan, verified with settings that announce the range check: rejected
    ctx = backend.Context(0, cfg, lib)
    p2 = luminair_amd.LuminairProof(ctx.prove_tables([(k, r, len(r)) for k, r in tabs], luts))
    ctx.close()
    with pytest.raises(luminair_amd.LuminairError):
        luminair_amd.verify(p2, CircuitSettings(lookups={"sin": luts["sin"]}, range_check=RangeCheckLookup([8], 8, [0] * 256)),
                            backend.VARIANT_PINNED, library=lib)


def test_lut_log_size_bounds_its_ranges(lib):
    """`hi - lo + 1` on unbounded i64 ranges overflowed before the 2^26 cap applied."""
    for rng in ([(-(1 << 62), (1 << 62))], [(-(1 << 63), (1 << 63) - 1)], [(0, 1 << 30)], [(-(1 << 30), 0)]):
        with pytest.raises(backend.LuminairBackendError):
            lib.lut_log_size(rng)
    assert lib.lut_log_size([(-(1 << 30) + 1, -(1 << 30) + 16)]) == 4
