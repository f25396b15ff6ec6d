"""The reference's producer edge cases on the test-only emulation build (CPU); tests/test_gpu_parity.py runs the same
scenarios on the MI355X.  See tests/producer_scenarios.py."""
import os
import subprocess

import pytest

from luminair_amd import backend

import producer_scenarios as ps


@pytest.fixture(scope="module")
def lib(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Library(so)


@pytest.mark.parametrize("build", ps.EXPANSIONS, ids=lambda f: f.__name__)
def test_expansion_scenarios(lib, build):
    ps.run_scenario(lib, build, 42 + ps.EXPANSIONS.index(build))


# the 32x32 shapes and the wide LUT cases are left to the GPU run (the emulation runs one fibre per GPU thread)
_CPU_OPS = [f for f in ps.OPS if "32x32" not in f.__name__]


@pytest.mark.parametrize("build", _CPU_OPS, ids=lambda f: f.__name__)
def test_op_shape_matrix(lib, build):
    ps.run_scenario(lib, build, 7)


def test_lut_index_matches_the_reference_unit_test(lib):
    """`LookupLayout::find_index` on the device (`lmn_trace_lut_ranges`) against the reference's own unit-test vector
    (crates/air/src/preprocessed.rs:581-634: ranges (-100,-50), (0,10), (200,210); values in the gaps have no index)."""
    import numpy as np
    from luminair_amd.graph import DeviceGraph
    ranges = [(-100, -50), (0, 10), (200, 210)]
    expected = {-100: 0, -75: 25, -50: 50, 0: 51, 5: 56, 10: 61, 200: 62, 205: 67, 210: 72}
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    g = DeviceGraph(ctx)
    g.set_lut_ranges("sin", ranges)
    vals = np.array(sorted(expected), dtype=np.int64)
    g.output(g.sin(g.input(vals)))
    tables, luts, bufs = g.gen_trace()
    mult = {k: ctx.download(b.view(0, n * 4)) for k, b, n in tables}[4]          # the SinLookup table
    assert len(mult) == 128 and int(mult.sum()) == len(vals)                       # 73 values -> 2^7 rows
    assert sorted(np.nonzero(mult)[0].tolist()) == sorted(expected.values())
    c0 = luts["sin"][0]
    for v, idx in expected.items():
        assert int(c0[idx]) == (v % ((1 << 31) - 1))                               # the LUT's value column agrees
    for b in bufs:
        b.free()
    for gap in (-49, 11, 199, -101, 211):
        g2 = DeviceGraph(ctx)
        g2.set_lut_ranges("sin", ranges)
        g2.output(g2.sin(g2.input(np.array([gap, 0, 5]))))
        with pytest.raises(backend.LuminairBackendError) as e:
            g2.gen_trace()
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
    # ranges out of order / overlapping / too many are rejected at the boundary
    for bad in ([(0, 10), (-5, -1)], [(0, 10), (10, 20)], [(i * 10, i * 10 + 5) for i in range(17)]):
        g3 = DeviceGraph(ctx)
        g3.luts["sin"] = (bad[0][0], bad[-1][1], luts["sin"])
        g3.lut_ranges["sin"] = bad
        g3.output(g3.sin(g3.input(np.array([bad[0][0]]))))
        with pytest.raises(backend.LuminairBackendError):
            g3.gen_trace()
    ctx.close()


def test_gen_circuit_settings_mirror(lib):
    """`gen_circuit_settings` -> `gen_trace(&mut settings)` -> `prove(trace, settings)` -> `verify(proof, settings)` as
    the reference's tests drive it (crates/graph/src/tests/mod.rs:26-44), on a graph whose two Sin nodes see disjoint
    input ranges (-> one LUT over two coalesced ranges), an Exp2 node and a LessThan node (-> the range check)."""
    import numpy as np
    import luminair_amd
    from luminair_amd.graph import DeviceGraph
    from luminair_amd.pie import CircuitSettings
    from host_graph import host_tables
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    rng = np.random.default_rng(3)
    g = DeviceGraph(ctx)
    x = g.input(rng.integers(-900, -600, size=(3, 4)))
    y = g.input(rng.integers(700, 950, size=(3, 4)))
    s1, s2 = g.sin(x), g.sin(y)
    e = g.exp2(g.add(s1, s2))
    out = g.output(g.less_than(e, g.input(rng.integers(0, 8192, size=(3, 4)))))
    settings = g.gen_circuit_settings()
    sin_ranges = settings.layouts["sin"].layout.ranges
    assert len(sin_ranges) == 2 and sin_ranges[0][1] < sin_ranges[1][0]          # disjoint: not coalesced
    for (lo, hi), src in zip(sin_ranges, (x, y)):                                 # 10 % padding around the buffer's span
        v = g.nodes[src.node_id].host
        span = int(v.max()) - int(v.min())
        assert lo <= int(v.min()) - span // 10 + 1 and hi >= int(v.max()) + span // 10 - 1
        assert lo >= int(v.min()) - span // 10 - 1 and hi <= int(v.max()) + span // 10 + 1
    assert len(settings.layouts["exp2"].layout.ranges) == 1 and settings.range_check is not None
    tables, luts, bufs = g.gen_trace()
    want, vals = host_tables(g)
    for k, buf, n in tables:
        assert np.array_equal(ctx.download(buf.view(0, n * want[k].shape[1] * 4)).reshape(n, -1), want[k]), k
    assert np.array_equal(g.read(out).reshape(-1), vals[out.node_id])
    g.fill_multiplicities(settings, tables)
    assert sum(settings.layouts["sin"].multiplicities) == 24 and sum(settings.range_check.multiplicities) == 48
    # the settings (reference form: ranges + multiplicities) survive bincode, feed the prover and the verifier
    settings2 = CircuitSettings.from_bincode(settings.to_bincode())
    assert settings2.layouts["sin"].layout.ranges == sin_ranges
    proof = ctx.prove_tables(tables, settings2.lut_columns(lib))
    assert proof == ctx.prove_tables(tables, luts)
    luminair_amd.verify(luminair_amd.LuminairProof(proof), settings2, backend.VARIANT_PINNED, library=lib)
    for b in bufs:
        b.free()
    ctx.close()
