"""Host orchestration + kernel indexing logic, run through the TEST-ONLY kernel emulation build
(tests/emu: the same HIP sources compiled with g++ -DLMN_EMU, one fiber per GPU thread) and compared
byte-for-byte with the oracle.  This is CPU-side debugging coverage, not the parity proof: the
parity tests proper are tests/test_gpu_parity.py (-m gpu), which call the real HIP library."""
import os
import subprocess

import numpy as np
import pytest

from luminair_amd import backend, synthetic as syn
from oracle.proof import to_bincode
from oracle.prover import prove as oracle_prove


@pytest.fixture(scope="module")
def emu_ctx(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    srcs = [os.path.join(root, "luminair_amd", "csrc", f) for f in os.listdir(os.path.join(root, "luminair_amd", "csrc"))
            if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(root, "tests", "emu", "emu_runtime.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Context(0, None, backend.Library(so))


def _both(ctx, tabs):
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs]))
    return got, want


def test_emu_reproduces_kat(emu_ctx, kat_bytes):
    tabs = syn.simple_example()
    assert emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == kat_bytes


@pytest.mark.parametrize("name,tabs", [
    ("add-ragged-100", syn.config2_add_only(100, 1)),
    ("chain-300", syn.chain_graph(300, 3)),
    ("mixed-sizes", [(0, syn.chain_graph(64, 4)[0][1]), (1, syn.chain_graph(500, 5)[1][1])]),
    ("single-row", syn.chain_graph(1, 6)),
    ("linear-layer+max", syn.linear_layer(20, 7, 2, True)),
    ("config5-3-layers", syn.config5_linear_layers(3, 4, 5, 8)),
    # FRI layers above 2^10 rows: the leaf hashing computes the fold itself (MerkleFold), except where a smaller
    # quotient column joins the layer
    ("add-2^11 (fold fused into leaf hashing)", syn.config2_add_only(1 << 11, 5)),
    ("mixed 2^11 + 2^10 (fused and plain folds)", [(0, syn.chain_graph(2000, 4)[0][1]), (1, syn.chain_graph(1000, 5)[1][1])]),
    # traces of 2^12 rows and more: the coalesced coset-order prefix sum (k_logup_scan2) instead of the scattered one
    ("chain-2^12 (coalesced logup scan)", syn.chain_graph(1 << 12, 7)),
    # columns of 2^13 rows and more: interpolation and extension share their strided pass (k_fft_interp_extend)
    ("mixed 2^13 + 2^10 (fused interpolate+extend next to the plain passes)",
     [(0, syn.chain_graph(5000, 4)[0][1]), (1, syn.chain_graph(1000, 5)[1][1])]),
])
def test_emu_matches_oracle(emu_ctx, name, tabs):
    got, want = _both(emu_ctx, tabs)
    assert got == want, name


@pytest.mark.parametrize("name,tabs", [
    ("chain-2^12", syn.chain_graph(1 << 12, 7)),
    ("mixed 2^13 + 2^11 (a level with children and columns is a skipped start level)",
     [(0, syn.chain_graph(5000, 4)[0][1]), (1, syn.chain_graph(2000, 5)[1][1])]),
    ("mixed 2^13 + 2^12 (adjacent sizes: the larger component's leaf level sits under the smaller one's)",
     [(0, syn.chain_graph(5000, 4)[0][1]), (1, syn.chain_graph(3000, 5)[1][1])]),
])
def test_emu_trees_stored_without_their_register_levels(emu_ctx, name, tabs, monkeypatch):
    """Big trees are stored without the levels a fused launch keeps in registers (MerkleCut, prover.h); what the
    decommitment needs of them is recomputed inside the gather launch (MerkleRecompute).  On the GPU this starts at 2^18
    leaves; LMN_MERKLE_SUB=3 makes the emulation build take the same path from 2^13 leaves on.  Same bytes as the oracle,
    and as the whole-tree storage (LMN_MERKLE_FULL)."""
    monkeypatch.setenv("LMN_MERKLE_SUB", "3")
    got, want = _both(emu_ctx, tabs)
    assert got == want, name
    # the leaf level of a tree whose next level has columns too (the first FRI tree of every proof; mixed-size trees) is
    # hashed by that level's launch and never stored (MerkleFold::below); on the GPU from 2^19 leaves on
    monkeypatch.setenv("LMN_MERKLE_BELOW_MIN_LOG", "12")
    assert emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == want
    monkeypatch.setenv("LMN_MERKLE_FULL", "1")
    assert emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == want
    # every FRI fold as a launch of its own (k_fold) instead of inside the next layer's leaf hashing / the tail's launch
    monkeypatch.setenv("LMN_NO_FOLD_FUSION", "1")
    assert emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == want


@pytest.mark.parametrize("cols", ["8", "4"])
def test_emu_transpose_inside_the_interpolation(emu_ctx, cols, monkeypatch):
    """LMN_ROWS_FUSION=1 (a switch; on the GPU from 2^18 rows): no transpose launch and no column-major evaluations in memory -
    the first inverse pass reads the table's rows itself (k_fft_rows_fx, 8 or 4 columns per workgroup), pads, checks
    canonicity, and the logup fractions read the rows.  Same bytes as the oracle for ragged, mixed-size and 16-column
    tables; the rejections of the transpose still happen."""
    monkeypatch.setenv("LMN_ROWS_FUSION", "1")
    monkeypatch.setenv("LMN_ROWS_FUSION_MIN_LOG", "13")
    monkeypatch.setenv("LMN_ROWS_FX_COLS", cols)
    for name, tabs in (("chain-5000 (2^13 rows, ragged)", syn.chain_graph(5000, 4)),
                       ("mixed 2^13 + 2^10", [(0, syn.chain_graph(5000, 4)[0][1]), (1, syn.chain_graph(1000, 5)[1][1])]),
                       ("mul-only 2^13 (16 columns)", syn.config2_mul_only(1 << 13, 3))):
        got, want = _both(emu_ctx, tabs)
        assert got == want, name
        assert emu_ctx.timings()["transpose_ms"] >= 0
    bad = syn.config2_add_only(1 << 13, 9)[0][1].copy()
    bad[5000, 9] = 0x7fffffff
    with pytest.raises(backend.LuminairBackendError) as e:
        emu_ctx.prove_tables([(0, bad, len(bad))])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    bad[5000, 9] = 5
    bad[3, 11] ^= 1
    with pytest.raises(backend.LuminairBackendError) as e:
        emu_ctx.prove_tables([(0, bad, len(bad))])
    assert e.value.code == backend.ERR_CONSTRAINTS


def test_emu_error_codes(emu_ctx):
    with pytest.raises(backend.LuminairBackendError) as e:
        emu_ctx.prove_tables([(0, np.zeros((0, 15), np.uint32), 0)])
    assert e.value.code == backend.ERR_EMPTY_TRACE
    bad = syn.config2_add_only(64, 9)[0][1].copy()
    bad[3, 11] ^= 1
    with pytest.raises(backend.LuminairBackendError) as e:
        emu_ctx.prove_tables([(0, bad, len(bad))])
    assert e.value.code == backend.ERR_CONSTRAINTS
    bad = syn.config2_add_only(64, 9)[0][1].copy()
    bad[5, 9] = 0x7fffffff                                    # P itself is not a canonical M31 value
    with pytest.raises(backend.LuminairBackendError) as e:
        emu_ctx.prove_tables([(0, bad, len(bad))])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    with pytest.raises(backend.LuminairBackendError) as e:   # no such TraceTable variant
        emu_ctx.prove_tables([(17, np.zeros((4, 12), np.uint32), 4)])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    with pytest.raises(backend.LuminairBackendError) as e:   # SinLookup table without its LUT columns in the settings
        emu_ctx.prove_tables([(4, np.zeros((16, 1), np.uint32), 16)])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    with pytest.raises(backend.LuminairBackendError) as e:   # wrong table order
        t = syn.chain_graph(16, 1)
        emu_ctx.prove_tables([(t[1][0], t[1][1], 16), (t[0][0], t[0][1], 16)])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT


def test_emu_table_size_limits_are_rejected_before_the_rows_are_read(emu_ctx):
    """The largest table PcsConfig::default() leaves room for is 2^25 rows (composition LDE 2^27, tests/test_gpu_parity.py
    proves one on the GPU); 2^25 + 1 .. 2^26 rows pad to 2^26 and are refused as too large, more than 2^26 rows as an
    invalid table - both on the row COUNT, before any row is touched (the buffer here holds 16 rows)."""
    rows = syn.config2_add_only(16, 3)[0][1]
    for n_rows, text in (((1 << 25) + 1, "too large"), (1 << 26, "too large"), ((1 << 26) + 1, "2^26")):
        with pytest.raises(backend.LuminairBackendError) as e:
            emu_ctx.prove_tables([(0, rows, n_rows)])
        assert e.value.code == backend.ERR_INVALID_ARGUMENT and text in str(e.value), (n_rows, str(e.value))


def test_emu_level2_ops(emu_ctx):
    from oracle import fft
    from oracle.field import P, QM31
    from oracle.merkle import MerkleTree
    rng = np.random.default_rng(5)
    ev = rng.integers(0, P, size=(3, 1 << 7), dtype=np.uint64)
    co = emu_ctx.interpolate(ev.astype(np.uint32))
    assert np.array_equal(co, fft.interpolate(ev).astype(np.uint32))
    lde = emu_ctx.evaluate(co, 9)
    assert np.array_equal(lde, fft.evaluate(co.astype(np.uint64), 9).astype(np.uint32))
    cols = [rng.integers(0, P, size=1 << k, dtype=np.uint64).astype(np.uint32) for k in (6, 6, 5, 6, 3)]
    assert emu_ctx.merkle_root(cols) == MerkleTree(cols).root()
    pt = [int(v) for v in rng.integers(0, P, size=8)]
    c = rng.integers(0, P, size=1 << 12, dtype=np.uint64)
    want = fft.eval_at_point(c, (QM31(*pt[:4]), QM31(*pt[4:])))
    assert emu_ctx.eval_at_point(c.astype(np.uint32), pt) == want.v
    emu_ctx.fft_selftest(13, 1)
    from level2_checks import check_evaluate_block, check_quotient_fold_grind_ops
    check_quotient_fold_grind_ops(emu_ctx, 7)
    check_evaluate_block(emu_ctx)


def test_emu_device_handle_ops(emu_ctx):
    from level2_checks import check_device_handle_ops
    check_device_handle_ops(emu_ctx, 7)
    check_device_handle_ops(emu_ctx, 4)


def test_emu_device_trace_generation(emu_ctx):
    from level2_checks import check_device_trace_generation
    check_device_trace_generation(emu_ctx, 300)
    check_device_trace_generation(emu_ctx, 1)
    from level2_checks import check_device_linear_layer
    check_device_linear_layer(emu_ctx)


def test_emu_device_graph(root):
    from level2_checks import check_device_graph
    check_device_graph(backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so")))


def test_emu_device_remaining_ops(root):
    from level2_checks import check_device_remaining_ops
    check_device_remaining_ops(backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so")))


def test_emu_pinned_variant_with_inputs_component(root):
    """17-slot claim + Inputs table (mixed column sizes inside one Merkle tree)."""
    from oracle.channel import ProtocolVariant
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    lib = backend.Library(so)
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    tabs = syn.config2_graph_faithful(100, 3)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED))
    assert got == want
    # LessThan + RangeCheckLookup: preprocessed tree 0, width-1 relations with their own element set
    tabs2 = syn.less_than_graph(40, 5)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs2])
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs2], variant=ProtocolVariant.PINNED))
    assert got == want
    tabs3 = syn.sqrt_rem_graph(30, 6)     # Sqrt + Rem (numerair forms unpinned, self-consistent)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs3])
    assert got == to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs3], variant=ProtocolVariant.PINNED))
    lib.verify(got, backend.VARIANT_PINNED)
    # Sin / Exp2 / Log2 + their lookup components: two-column LUTs of different sizes in tree 0 (passed in as
    # settings data), width-2 LUT relations with three more element sets
    tabs4, luts = syn.activation_graph(50, 8, ranges={"sin": (-3000, 2500), "exp2": (-900, 1100), "log2": (1, 700)})
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs4], luts)
    assert got == to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs4], variant=ProtocolVariant.PINNED,
                                          luts=luts))
    lib.verify(got, backend.VARIANT_PINNED)
    bad = [(k, r.copy()) for k, r in tabs4]
    bad[0][1][7, 8] ^= 1                  # a sin output that is not in the LUT: logup sums no longer cancel
    got = ctx.prove_tables([(k, r, len(r)) for k, r in bad], luts)
    with pytest.raises(backend.LuminairBackendError) as e:
        lib.verify(got, backend.VARIANT_PINNED)
    assert e.value.code == backend.ERR_INVALID_LOGUP
    # the KAT era drew one LUT relation (sin): Sin + SinLookup fit its 8-slot claim
    kat_ctx = backend.Context(0, None, lib)
    tabs5, luts5 = syn.activation_graph(20, 9, names=("sin",), ranges={"sin": (-800, 800)})
    tabs5 = [t for t in tabs5 if t[0] != 15]       # no Inputs component in the KAT era
    got = kat_ctx.prove_tables([(k, r, len(r)) for k, r in tabs5], luts5)
    assert got == to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs5], luts=luts5))
    # the KAT-variant context has no claim slot for kind 15
    with pytest.raises(backend.LuminairBackendError) as e:
        kat_ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    assert e.value.code == backend.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("seed", range(6))
def test_emu_random_pies_match_oracle(root, seed):
    """Differential test over random component mixes and ragged sizes (mixed-size Merkle trees, several
    composition sizes, tree-0 layouts), PINNED variant so that every component has a claim slot."""
    from oracle.channel import ProtocolVariant
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    from level2_checks import random_pie
    tabs, luts = random_pie(seed)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs], luts)
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED, luts=luts))
    assert got == want, [k for k, _ in tabs]


def test_emu_trace_generation_error_paths(emu_ctx):
    """lmn_trace_*: argument validation mirrors the reference's failure modes (empty tensors -> EmptyTrace)."""
    a = emu_ctx.upload(np.arange(1, 9, dtype=np.int32))
    with pytest.raises(backend.LuminairBackendError) as e:     # view shape does not match the element count
        emu_ctx.trace_elementwise(0, a, a, 8, node_id=1, input_ids=(0, 0), num_consumers=0,
                                  lhs_view=backend.LmnView.make((3, 2), (2, 1)))
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    with pytest.raises(backend.LuminairBackendError) as e:     # zero elements
        emu_ctx.trace_elementwise(0, a, a, 0, node_id=1, input_ids=(0, 0), num_consumers=0)
    assert e.value.code == backend.ERR_EMPTY_TRACE
    with pytest.raises(backend.LuminairBackendError) as e:     # SumReduce is not an elementwise kind
        emu_ctx.trace_elementwise(5, a, a, 8, node_id=1, input_ids=(0, 0), num_consumers=0)
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    with pytest.raises(backend.LuminairBackendError) as e:     # binary op without a right operand
        emu_ctx.trace_elementwise(1, a, None, 8, node_id=1, input_ids=(0, 0), num_consumers=0)
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    lut = syn.make_lut("exp2", -4, 4)
    c1, mult = emu_ctx.upload(lut[1]), emu_ctx.upload(np.zeros(len(lut[0]), dtype=np.uint32))
    with pytest.raises(backend.LuminairBackendError) as e:     # inputs 5..8 lie outside the LUT range [-4, 4]
        emu_ctx.trace_lut(9, a, 8, node_id=2, input_id=0, num_consumers=0, lut_col1=c1, lo=-4, lut_len=9, mult=mult)
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    # an expanded view: every row reads element r % 2 of a two-element tensor
    rows, out = emu_ctx.trace_elementwise(16, a, None, 6, node_id=3, input_ids=(0,), num_consumers=1, input_mults=(-1,),
                                          lhs_view=backend.LmnView.make((3, 2), (0, 1)))
    assert emu_ctx.download(out, np.int32).tolist() == [1, 2, 1, 2, 1, 2]
    for b in (a, c1, mult, rows, out):
        b.free()


def test_emu_shard_and_handle_argument_validation(emu_ctx):
    """Error behaviour of the round-2 entry points: bad shard geometry, shape mismatches on device handles."""
    E = backend.LuminairBackendError
    noop = lambda buf, nbytes, stream: None
    for rank, world, t in ((0, 3, 0), (2, 2, 0), (0, 16, 0), (0, 0, 0), (0, 8, 2), (0, 4, 40)):
        with pytest.raises(E) as e:
            emu_ctx.set_shard(rank, world, noop, t)
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
    emu_ctx.clear_shard()                                  # a failed set_shard leaves the context unsharded
    tabs = syn.config2_add_only(40, 2)
    want = emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    emu_ctx.set_shard(0, 1, noop)                          # world 1: the collective is called but has nothing to move
    assert emu_ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == want
    emu_ctx.clear_shard()
    with pytest.raises(E):                                 # no RCCL in the emulation build
        emu_ctx.set_shard_rccl(0, 1, bytes(128))
    a = emu_ctx.col_from_cpu(np.zeros((4, 64), np.uint32))
    b = emu_ctx.col_from_cpu(np.zeros((3, 64), np.uint32))
    c = emu_ctx.col_from_cpu(np.zeros((4, 16), np.uint32))
    for fn in (lambda: a.accumulate(b), lambda: a.accumulate(c), lambda: b.fold_line((1, 0, 0, 0)),
               lambda: c.fold_circle_into_line(a, (1, 0, 0, 0)), lambda: b.decompose(), lambda: a.evaluate(5),
               lambda: a.extend(5), lambda: a.eval_at_point(4, [0] * 8), lambda: a.evaluate_block(8, 4, 0),
               lambda: a.evaluate_block(8, 1, 2), lambda: emu_ctx.commit([]), lambda: emu_ctx.col_zeros(0, 4),
               lambda: emu_ctx.col_zeros(1, 40),
               lambda: emu_ctx.col_accumulate_quotients([a, c], [(0, 0, (0, 0, 0, 0))], [[0] * 8], (1, 0, 0, 0))):
        with pytest.raises(E) as e:
            fn()
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
    for x in (a, b, c):
        x.free()


def test_emu_one_context_from_several_threads_is_serialised(emu_ctx):
    """Calls on one context from several threads are serialised by the context's lock (include/luminair_hip.h):
    the misuse that corrupted round 2's `host_rows` sub-result (two pool workers inside one context) is now safe."""
    import threading
    tabs = [(k, r, len(r)) for k, r in syn.chain_graph(40, 11)]
    want = emu_ctx.prove_tables(tabs)
    bad = []

    def work():
        for _ in range(3):
            if emu_ctx.prove_tables(tabs) != want:
                bad.append(1)

    ths = [threading.Thread(target=work) for _ in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad


def test_emu_advice_r2_boundary_fixes(emu_ctx):
    """a rejected lmn_ctx_set_shard keeps the context's current sharding; a NULL first column handle
    and non-canonical field words at the level-2 boundary are INVALID_ARGUMENT, not a crash / silent wrap."""
    import ctypes as C
    E = backend.LuminairBackendError
    calls = []
    tabs = [(k, r, len(r)) for k, r in syn.config2_add_only(40, 2)]
    want = emu_ctx.prove_tables(tabs)
    emu_ctx.set_shard(0, 1, lambda buf, nbytes, stream: calls.append(nbytes))
    with pytest.raises(E):
        emu_ctx.set_shard(5, 3, lambda *a: None)            # rejected ...
    n0 = len(calls)
    assert emu_ctx.prove_tables(tabs) == want and len(calls) > n0   # ... and the world-1 shard is still active
    emu_ctx.clear_shard()
    a = emu_ctx.col_from_cpu(np.zeros((4, 64), np.uint32))
    lib = emu_ctx.lib.lib
    arr = (C.c_void_p * 2)(None, a.handle)
    sc = (C.c_uint32 * 1)(0)
    sv = (C.c_uint32 * 4)(0, 0, 0, 0)
    pts = (C.c_uint32 * 8)(*([0] * 8))
    al = (C.c_uint32 * 4)(1, 0, 0, 0)
    out = C.c_void_p()
    rc = lib.lmn_col_accumulate_quotients(emu_ctx.handle, arr, 2, sc, sc, sv, 1, pts, 1, al, C.byref(out))
    assert rc == backend.ERR_INVALID_ARGUMENT and not out.value
    P = (1 << 31) - 1
    for fn in (lambda: a.eval_at_point(0, [P] + [0] * 7), lambda: a.fold_line((P, 0, 0, 0)),
               lambda: emu_ctx.col_accumulate_quotients([a], [(0, 0, (0, P + 5, 0, 0))], [[0] * 8], (1, 0, 0, 0)),
               lambda: emu_ctx.col_accumulate_quotients([a], [(0, 0, (0, 0, 0, 0))], [[0] * 8], (1, 0, 0, 1 << 31))):
        with pytest.raises(E) as e:
            fn()
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
    a.free()


@pytest.mark.parametrize("pow_bits,log_last_layer,n_queries,tabs", [
    (10, 3, 20, syn.chain_graph(300, 3)),
    (0, 5, 1, syn.config2_add_only(100, 1)),
    (7, 0, 64, syn.config3_mixed(10, 9, 9, 8)),       # mixed-size trees
    (5, 10, 3, syn.config2_add_only(1 << 11, 5)),      # last layer of 2^11 evaluations: the FRI loop ends above the tail kernel
    (5, 9, 5, syn.config2_add_only(1 << 11, 6)),
    (12, 2, 200, syn.linear_layer(20, 7, 2, True)),    # more queries than rows in the smallest tree
])
def test_emu_non_default_pcs_config_matches_oracle(root, pow_bits, log_last_layer, n_queries, tabs):
    """`lmn_config` other than PcsConfig::default(): PoW bits, FRI last-layer degree bound and query count reach the
    transcript, the FRI loop's end, the grind and the decommitment - byte-equal to the oracle under the same config,
    and accepted by the product verifier only when it expects that config."""
    from oracle.prover import PcsConfig
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    cfg = lib.default_config()
    cfg.pow_bits, cfg.log_last_layer, cfg.n_queries = pow_bits, log_last_layer, n_queries
    ctx = backend.Context(0, cfg, lib)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    ctx.close()
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs],
                                   PcsConfig(pow_bits=pow_bits, log_last_layer=log_last_layer, n_queries=n_queries)))
    assert got == want
    lib.verify(got, backend.VARIANT_KAT, config=cfg)
    if (pow_bits, log_last_layer, n_queries) != (5, 0, 3):
        with pytest.raises(backend.LuminairBackendError):
            lib.verify(got, backend.VARIANT_KAT)           # the default verifier expects PcsConfig::default()


def test_emu_trace_smaller_than_the_last_fri_layer_is_refused(root):
    """a 16-row table under log_last_layer = 10 has a first line layer (2^5) below the configured last layer
    (2^11), and a 2^10-row table's columns would join a line layer below the last one: stwo's FRI commit asserts both and
    the reference panics; the library refuses such a pie up front (the plan and the loop used to disagree about a
    clamped size) and the oracle raises too."""
    from oracle.prover import PcsConfig, ProvingError
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    cfg = lib.default_config()
    cfg.log_last_layer = 10
    ctx = backend.Context(0, cfg, lib)
    tabs = syn.simple_example()
    for bad in (tabs, syn.config2_add_only(1 << 10, 3)):
        with pytest.raises(backend.LuminairBackendError) as e:
            ctx.prove_tables([(k, r, len(r)) for k, r in bad])
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
        with pytest.raises(ProvingError):
            oracle_prove([(k, r.astype(np.uint64)) for k, r in bad], PcsConfig(log_last_layer=10))
    # the context stays usable, and the smallest admissible table (2^11 rows: its columns join the last layer) proves
    tabs_ok = syn.config2_add_only(1 << 11, 3)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs_ok])
    ctx.close()
    assert got == to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs_ok], PcsConfig(log_last_layer=10)))
    lib.verify(got, backend.VARIANT_KAT, config=cfg)


def test_emu_prove_submit_wait(root):
    """`lmn_prove_submit` / `lmn_prove_wait`: the asynchronous form for single-threaded callers (the reference's
    `prove` is a plain function call, prover.rs:28) - same bytes and same error codes as `lmn_prove`, one outstanding
    submission per context, a context can be destroyed with an uncollected proof."""
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    ctx = backend.Context(0, None, lib)
    tabs = [(k, r, len(r)) for k, r in syn.chain_graph(100, 3)]
    want = ctx.prove_tables(tabs)
    for _ in range(2):
        ctx.prove_submit(tabs)
        assert ctx.prove_wait() == want
    ctx.prove_submit(tabs)
    with pytest.raises(backend.LuminairBackendError) as e:      # one outstanding submission per context
        ctx.prove_submit(tabs)
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    assert ctx.prove_wait() == want
    with pytest.raises(backend.LuminairBackendError) as e:      # nothing to wait for
        ctx.prove_wait()
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
    ctx.prove_submit([(0, np.zeros((0, 15), np.uint32), 0)])    # errors surface at wait, as lmn_prove would report them
    with pytest.raises(backend.LuminairBackendError) as e:
        ctx.prove_wait()
    assert e.value.code == backend.ERR_EMPTY_TRACE
    assert ctx.prove_tables(tabs) == want                       # the synchronous form still works next to it
    ctx.prove_submit(tabs)
    ctx.close()                                                 # destroy with an uncollected proof: no hang, no leak


def test_emu_prover_pool_prove_many(root):
    """`ProverPool.prove_many`: proofs come back in input order, errors surface as LuminairError, the pool stays usable."""
    import luminair_amd
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    pool = luminair_amd.ProverPool(0, n=1, library=lib)       # the emulation runtime is single-context
    pies = [luminair_amd.LuminairPie.from_tables(syn.chain_graph(40 + 7 * i, i)) for i in range(3)]
    want = [pool.provers[0].prove(p).to_bincode() for p in pies]
    assert [p.to_bincode() for p in pool.prove_many(pies)] == want
    bad = luminair_amd.LuminairPie([luminair_amd.TraceTable(luminair_amd.TraceTableKind.Add, np.zeros((0, 15), np.uint32))])
    with pytest.raises(luminair_amd.LuminairError):
        pool.prove_many([pies[0], bad, pies[1]])
    assert [p.to_bincode() for p in pool.prove_many(pies[:2])] == want[:2]
    pool.close()


def test_emu_rows_in_lmn_host_alloc_memory(emu_ctx):
    """`lmn_host_alloc` / `lmn_host_register` (page-locked host memory for host-resident trace rows; plain memory in the
    emulation build): same proof as from an ordinary numpy buffer, argument checks."""
    import ctypes as C
    from luminair_amd import backend as bk
    ctx, emu_lib = emu_ctx, emu_ctx.lib
    tabs = syn.config2_add_only(1 << 6, 3)
    want = ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    pinned = []
    for k, r in tabs:
        a = emu_lib.host_rows(r.shape, r.dtype)
        a.array[...] = r
        pinned.append((k, a))
    assert ctx.prove_tables([(k, a.array, len(a.array)) for k, a in pinned]) == want
    own = [np.array(r, copy=True) for _, r in tabs]
    for x in own:
        emu_lib.host_register(x)
    assert ctx.prove_tables([(k, x, len(x)) for (k, _), x in zip(tabs, own)]) == want
    for x in own:
        emu_lib.host_unregister(x)
    for _, a in pinned:
        a.free()
        a.free()                                   # idempotent
    p = C.c_void_p()
    assert emu_lib.lib.lmn_host_alloc(0, C.byref(p)) == bk.ERR_INVALID_ARGUMENT
    assert emu_lib.lib.lmn_host_alloc(64, None) == bk.ERR_INVALID_ARGUMENT
    assert emu_lib.lib.lmn_host_register(None, 64) == bk.ERR_INVALID_ARGUMENT
    assert emu_lib.lib.lmn_host_unregister(None) == bk.ERR_INVALID_ARGUMENT


def test_emu_device_and_host_transcript_give_the_same_bytes_and_errors(root, monkeypatch):
    """The commitment phases' Fiat-Shamir steps run on the device by default (ChanStep: made by the launch that produces
    each tree's root, or by k_chan_step under LMN_CHAN_STEP_SEPARATE=1; no host wait before the sampled values) and on the host under LMN_HOST_FS=1: same proof bytes for a multi-component pie with LUT relations
    and a non-default constraint form, same rejection of a non-canonical word and of an unsatisfied constraint."""
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED | backend.PV_MUL_ONE_SLOT | backend.PV_RECIP_NEG
    ctx = backend.Context(0, cfg, lib)
    cases = [(syn.chain_graph(300, 3) + [], None)]
    tabs, luts = syn.activation_graph(40, 3)
    cases.append((tabs, luts))
    for tabs, luts in cases:
        t = [(k, r, len(r)) for k, r in tabs]
        monkeypatch.delenv("LMN_HOST_FS", raising=False)
        dev = ctx.prove_tables(t, luts)
        monkeypatch.setenv("LMN_HOST_FS", "1")
        assert ctx.prove_tables(t, luts) == dev
        monkeypatch.delenv("LMN_HOST_FS")
        monkeypatch.setenv("LMN_CHAN_STEP_SEPARATE", "1")      # the steps as launches of their own (k_chan_step)
        assert ctx.prove_tables(t, luts) == dev
        monkeypatch.delenv("LMN_CHAN_STEP_SEPARATE")
        # round 6: the step in front of the FRI quotient kernels (mix of the sampled values, quotient randomness, line
        # coefficients and tables) is made by k_quot_prepare by default; LMN_HOST_QUOT=1 puts the host's wait back
        monkeypatch.setenv("LMN_HOST_QUOT", "1")
        assert ctx.prove_tables(t, luts) == dev
        monkeypatch.delenv("LMN_HOST_QUOT")
    tabs = syn.chain_graph(300, 3)
    for env in (None, "1", "quot"):
        monkeypatch.delenv("LMN_HOST_FS", raising=False)
        monkeypatch.delenv("LMN_HOST_QUOT", raising=False)
        if env == "1":
            monkeypatch.setenv("LMN_HOST_FS", env)
        elif env == "quot":
            monkeypatch.setenv("LMN_HOST_QUOT", "1")
        bad = [(k, r.copy(), len(r)) for k, r in tabs]
        bad[0][1][5, 9] = 0x7fffffff
        with pytest.raises(backend.LuminairBackendError) as e:
            ctx.prove_tables(bad)
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
        wrong = [(k, r.copy(), len(r)) for k, r in tabs]
        wrong[0][1][7, 11] = (int(wrong[0][1][7, 11]) + 1) % 0x7fffffff          # out != lhs + rhs
        with pytest.raises(backend.LuminairBackendError) as e:
            ctx.prove_tables(wrong)
        assert e.value.code == backend.ERR_CONSTRAINTS
        assert ctx.prove_tables([(k, r, len(r)) for k, r in tabs])               # the context stays usable
    ctx.close()


@pytest.mark.parametrize("log_blowup,tabs,luts,flags", [
    (2, syn.chain_graph(300, 3), None, 0),
    (2, syn.config3_mixed(9, 8, 8, 8), None, 0),                          # mixed-size trees
    (3, syn.chain_graph(100, 4), None, backend.PV_MIX_U64_HASHED | backend.PV_MUL_ONE_SLOT),
    (2,) + syn.activation_graph(40, 3) + (backend.VARIANT_PINNED,),       # preprocessed LUT columns on the constraint domain
])
def test_emu_larger_blowups_match_oracle(root, log_blowup, tabs, luts, flags):
    """Blow-up 4 and 8 (round 5; the reference itself only uses PcsConfig::default() = blow-up 2): the committed LDE is then no
    longer the constraint evaluation domain, so the composition phase evaluates the columns there from their coefficients;
    LDE sizes, FRI layer count and query domain follow log_blowup.  Byte-equal to the oracle, accepted by the product
    verifier under the same config only; sharded proofs refuse it."""
    from oracle.channel import ProtocolVariant
    from oracle.prover import PcsConfig
    lib = backend.Library(os.path.join(root, "tests", "emu", "libluminair_emu.so"))
    cfg = lib.default_config()
    cfg.log_blowup, cfg.protocol_variant = log_blowup, flags
    ctx = backend.Context(0, cfg, lib)
    got = ctx.prove_tables([(k, r, len(r)) for k, r in tabs], luts)
    ctx.close()
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs], PcsConfig(log_blowup=log_blowup),
                                   variant=ProtocolVariant(flags), luts=luts))
    assert got == want
    lib.verify(got, flags, config=cfg)
    with pytest.raises(backend.LuminairBackendError):
        lib.verify(got, flags)                                  # the default verifier expects blow-up 2
    bad = lib.default_config()
    bad.log_blowup = 4
    with pytest.raises(backend.LuminairBackendError) as e:
        backend.Context(0, bad, lib)
    assert e.value.code == backend.ERR_INVALID_ARGUMENT
