"""Lock-step batches of small proofs (`lmn_batch_*`, libluminair_hip_batch.so; luminair_amd/csrc/batch.h): every proof
of a batch must be byte-identical to what `lmn_prove` returns for its pie - over the operator set of the reference's own
benchmark (crates/graph/benches/ops.rs:92-884: Add, Mul, Recip, SumReduce, MaxReduce, Sqrt, Rem, Sin, Exp2, LessThan at
32x32) and BASELINE config 4's shape - and a bad pie must fail alone."""
import ctypes as C
import os

import numpy as np
import pytest

from luminair_amd import backend, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH_LIB = os.path.join(ROOT, "luminair_amd", "csrc", "libluminair_hip_batch.so")


def test_batch_library_exports_the_batch_abi_and_the_whole_c_abi():
    if not os.path.exists(BATCH_LIB):
        import __graft_entry__
        __graft_entry__.build()
    import re
    lib = backend.Library(BATCH_LIB)                    # binds every symbol of include/luminair_hip.h + checks the ABI version
    hdr = open(os.path.join(ROOT, "include", "luminair_hip_batch.h")).read()
    declared = sorted(set(re.findall(r"\b(lmn_batch_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == ["lmn_batch_counter", "lmn_batch_create", "lmn_batch_destroy", "lmn_batch_last_error", "lmn_batch_prove"]
    for name in backend.EXPORTS + declared:
        getattr(lib.lib, name)
    # no GPU here: creating a batch fails loudly, it never falls back to anything
    if not __import__("torch").cuda.is_available():
        from luminair_amd.batch import BatchProver
        with pytest.raises(backend.LuminairBackendError) as e:
            BatchProver(0, 4)
        assert e.value.code == backend.ERR_NO_DEVICE


def _operator_pies(n=1024):
    """(name, tables(seed), luts) for the ten operators of the reference's benchmark at 32x32 = 1024 elements"""
    def reduce_tabs(fn, kind):
        def mk(seed):
            rng = np.random.default_rng(seed)
            x = rng.integers(-2048, 2048, size=(32, 32))
            return sorted([(kind, fn(x, node=2, input_id=0, input_mult=-1, out_mult=0)),
                           (syn.KIND_INPUTS, syn.inputs_rows(x.reshape(-1), 0, 1))], key=lambda kt: kt[0])
        return mk

    def recip(seed):
        rng = np.random.default_rng(seed)
        a = rng.integers(5, 2048, size=n)
        return [(syn.KIND_RECIP, syn.recip_rows(a, node=2, input_id=0, mults=(-1, 0))), (syn.KIND_INPUTS, syn.inputs_rows(a, 0, 1))]

    def mul(seed):
        rng = np.random.default_rng(seed)
        a, b = rng.integers(0, 2048, size=n), rng.integers(0, 2048, size=n)
        return [(syn.KIND_MUL, syn.mul_rows(a, b, mults=(-1, -1, 0))),
                (syn.KIND_INPUTS, np.concatenate([syn.inputs_rows(a, 0, 1), syn.inputs_rows(b, 1, 1)]))]

    out = [("add", lambda s: syn.config2_graph_faithful(n, s), None), ("mul", mul, None), ("recip", recip, None),
           ("sum_reduce", reduce_tabs(syn.sum_reduce_rows, syn.KIND_SUM_REDUCE), None),
           ("max_reduce", reduce_tabs(syn.max_reduce_rows, syn.KIND_MAX_REDUCE), None),
           ("sqrt+rem", lambda s: syn.sqrt_rem_graph(n, s), None),
           ("less_than", lambda s: syn.less_than_graph(n, s), None)]
    for name in ("sin", "exp2"):
        def mk(seed, name=name):
            return syn.activation_graph(n, seed, names=(name,))[0]
        out.append((name, mk, syn.activation_graph(n, 1, names=(name,))[1]))
    return out


@pytest.mark.gpu
def test_gpu_batched_proofs_equal_lmn_prove_for_the_reference_benchmark_operators(gpu_prover):
    import luminair_amd
    from luminair_amd.batch import BatchProver
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    bp = BatchProver(0, 6, protocol_variant=backend.VARIANT_PINNED)
    try:
        for name, mk, luts in _operator_pies():
            pies = [[(k, r, len(r)) for k, r in mk(50 + i)] for i in range(6)]
            got = bp.prove_batch(pies, luts)
            want = [solo.ctx.prove_tables(p, luts) for p in pies]
            assert got == want, name
        c = bp.counters()
        assert c["host_waits"] < c["launches"] and c["direct_copies"] == 0
    finally:
        bp.close()


@pytest.mark.gpu
def test_gpu_batched_config4_and_kat_variant(gpu_prover, kat_bytes):
    import luminair_amd
    from luminair_amd.batch import BatchProver
    tabs4, luts4 = syn.config4_black_scholes_shape()
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    bp = BatchProver(0, 3, protocol_variant=backend.VARIANT_PINNED)
    try:
        pie = [(k, r, len(r)) for k, r in tabs4]
        assert bp.prove_batch([pie] * 3, luts4) == [solo.ctx.prove_tables(pie, luts4)] * 3
    finally:
        bp.close()
    bk = BatchProver(0, 4)            # KAT variant: the reference's own known-answer proof, four times in one batch
    try:
        kat = [(k, r, len(r)) for k, r in syn.simple_example()]
        assert bk.prove_batch([kat] * 4) == [kat_bytes] * 4
        # a batch smaller than the number of slots, device-resident rows
        ctx = gpu_prover.ctx
        dev = [(k, ctx.upload(r), n) for k, r, n in kat]
        assert bk.prove_batch([dev, dev]) == [kat_bytes] * 2
    finally:
        bk.close()


@pytest.mark.gpu
def test_gpu_batch_rejects_mixed_shapes_and_a_bad_pie_fails_alone(gpu_prover):
    import luminair_amd
    from luminair_amd.batch import BatchProver
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    bp = BatchProver(0, 4, protocol_variant=backend.VARIANT_PINNED)
    try:
        good = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(256, 7 + i)] for i in range(4)]
        other = [(k, r, len(r)) for k, r in syn.config2_graph_faithful(512, 3)]
        with pytest.raises(backend.LuminairBackendError) as e:
            bp.prove_batch(good[:3] + [other])
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
        # pie 2 violates its constraints (out != lhs + rhs): the other three proofs are still produced and correct
        bad = [(k, r.copy(), n) for k, r, n in good[2]]
        bad[0][1][5, 11] = (int(bad[0][1][5, 11]) + 1) % ((1 << 31) - 1)
        lib = bp.lib.lib
        n = 4
        pies = good[:2] + [bad] + good[3:]
        arrs = (C.POINTER(backend.LmnTable) * n)()
        keep = []
        for i, t in enumerate(pies):
            arr, nt, st, k = backend.Context._marshal_tables(None, t, None)
            keep.append(k)
            arrs[i] = C.cast(arr, C.POINTER(backend.LmnTable))
        proofs, lens, rcs = (C.POINTER(C.c_uint8) * n)(), (C.c_size_t * n)(), (C.c_int * n)()
        rc = lib.lmn_batch_prove(bp.handle, n, arrs, nt, C.byref(st), proofs, lens, rcs)
        assert rc == backend.ERR_CONSTRAINTS and list(rcs) == [0, 0, backend.ERR_CONSTRAINTS, 0]
        for i in (0, 1, 3):
            assert C.string_at(proofs[i], lens[i]) == solo.ctx.prove_tables(pies[i])
            lib.lmn_free(proofs[i])
        # and the batch object is still usable
        assert bp.prove_batch(good) == [solo.ctx.prove_tables(p) for p in good]
    finally:
        bp.close()


@pytest.mark.gpu
def test_gpu_batch_grows_across_calls(gpu_prover):
    """A batch that uses more slots, then a larger shape, than the batches before it: the contexts that prove a shape
    for the first time build their twiddle tables outside lock-step (before the members start), so members that have
    the tables and members that do not issue the same launch sequence."""
    import luminair_amd
    from luminair_amd.batch import BatchProver
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    bp = BatchProver(0, 4, protocol_variant=backend.VARIANT_PINNED)
    try:
        small = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(200, 20 + i)] for i in range(4)]
        big = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(3000, 30 + i)] for i in range(4)]
        assert bp.prove_batch(small[:2]) == [solo.ctx.prove_tables(p) for p in small[:2]]      # slots 0-1 only
        assert bp.prove_batch(small) == [solo.ctx.prove_tables(p) for p in small]              # slots 2-3 are new
        assert bp.prove_batch(big[:3]) == [solo.ctx.prove_tables(p) for p in big[:3]]          # larger domain on 0-2
        assert bp.prove_batch(big) == [solo.ctx.prove_tables(p) for p in big]                  # ... and on 3
        assert bp.prove_batch(small) == [solo.ctx.prove_tables(p) for p in small]
    finally:
        bp.close()


@pytest.mark.gpu
def test_gpu_batch_pie_with_a_non_canonical_word_fails_alone(gpu_prover):
    """One pie of a batch holds a word >= 2^31 - 1: that pie is rejected (invalid argument), the other proofs are
    produced, and the slot that saw the bad word proves correctly in the next batch (the verdict word is never reset:
    each proof has its own mark)."""
    import luminair_amd
    from luminair_amd.batch import BatchProver
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    bp = BatchProver(0, 4, protocol_variant=backend.VARIANT_PINNED)
    try:
        good = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(256, 40 + i)] for i in range(4)]
        bad = [(k, r.copy(), n) for k, r, n in good[1]]
        bad[0][1][7, 9] = (1 << 31) - 1
        lib = bp.lib.lib
        n = 4
        pies = [good[0], bad, good[2], good[3]]
        arrs = (C.POINTER(backend.LmnTable) * n)()
        keep = []
        for i, t in enumerate(pies):
            arr, nt, st, k = backend.Context._marshal_tables(None, t, None)
            keep.append(k)
            arrs[i] = C.cast(arr, C.POINTER(backend.LmnTable))
        proofs, lens, rcs = (C.POINTER(C.c_uint8) * n)(), (C.c_size_t * n)(), (C.c_int * n)()
        rc = lib.lmn_batch_prove(bp.handle, n, arrs, nt, C.byref(st), proofs, lens, rcs)
        assert rc == backend.ERR_INVALID_ARGUMENT and list(rcs) == [0, backend.ERR_INVALID_ARGUMENT, 0, 0]
        for i in (0, 2, 3):
            assert C.string_at(proofs[i], lens[i]) == solo.ctx.prove_tables(pies[i])
            lib.lmn_free(proofs[i])
        for _ in range(2):     # the same slots again, all four pies good
            assert bp.prove_batch(good) == [solo.ctx.prove_tables(p) for p in good]
        # the solo prover after a rejected pie
        with pytest.raises(backend.LuminairBackendError):
            solo.ctx.prove_tables(bad)
        assert solo.ctx.prove_tables(good[1]) == bp.prove_batch([good[1]])[0]
    finally:
        bp.close()


@pytest.mark.gpu
def test_gpu_concurrent_batch_groups_equal_lmn_prove(gpu_prover):
    """Three batch groups driven at once (`BatchPool`: one thread per group; while one group's members run their host code
    another group's launches use the GPU): every proof byte-identical to `lmn_prove`'s, in input order, for more pies than
    the groups hold at once and a last batch that is not full."""
    import luminair_amd
    from luminair_amd.batch import BatchPool
    solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
    pool = BatchPool(0, groups=3, slots=8, protocol_variant=backend.VARIANT_PINNED)
    try:
        pies = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 300 + i)] for i in range(45)]
        want = [solo.ctx.prove_tables(p) for p in pies]
        for _ in range(3):
            assert pool.prove_many(pies) == want
        # a pie that violates its constraints fails its own batch call; the pool raises and stays usable
        bad = [(k, r.copy(), n) for k, r, n in pies[9]]
        bad[0][1][5, 11] = (int(bad[0][1][5, 11]) + 1) % ((1 << 31) - 1)
        with pytest.raises(backend.LuminairBackendError) as e:
            pool.prove_many(pies[:9] + [bad] + pies[10:])
        assert e.value.code == backend.ERR_CONSTRAINTS
        assert pool.prove_many(pies) == want
    finally:
        pool.close()
