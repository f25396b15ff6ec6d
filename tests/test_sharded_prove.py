"""Single-proof sharding (BASELINE configs 4/5): `lmn_prove` on world-2, world-4 and world-8 process groups
(gloo, the TEST-ONLY emulation build, the `lmn_collective` callback as transport) must return, on every rank, exactly
the bytes an unsharded context produces - for one table, mixed-size tables, a config-5-shaped pie and a pie with
lookups, with the FRI sharding threshold low enough that split layers, the split -> replicated transition and the
replicated tail are all exercised."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cases():
    """(name, tables, luts, pinned variant?)"""
    sys.path.insert(0, ROOT)
    from luminair_amd import synthetic as syn
    act, luts = syn.activation_graph(40, 8, names=("sin", "exp2"), ranges={"sin": (-500, 400), "exp2": (-100, 90)})
    return [
        ("examples/simple (16-row tables: 4-row blocks at world 8)", syn.simple_example(), None, False),
        ("2a-small", syn.config2_add_only(300, 1), None, False),
        ("config3-small", syn.config3_mixed(8, 7, 7, 2), None, False),
        ("mixed-sizes", [(0, syn.chain_graph(64, 4)[0][1]), (1, syn.chain_graph(500, 5)[1][1])], None, False),
        ("config5-small", syn.config5_linear_layers(3, 4, 5, 8), None, False),
        ("2b-small (Inputs component)", syn.config2_graph_faithful(200, 3), None, True),
        ("less-than + range-check LUT", syn.less_than_graph(40, 5), None, True),
        ("sin/exp2 + LUT tree 0", act, luts, True),
        # PcsConfig other than the default: (pow_bits, log_last_layer, n_queries)
        ("chain, 20 queries, last layer 2^3 coefficients", syn.chain_graph(300, 3), None, False, (10, 3, 20)),
        ("2a-small, 1 query, last layer 2^5 coefficients", syn.config2_add_only(300, 2), None, False, (0, 5, 1)),
        # blow-up 4 and 8 (round 6): the committed LDE is not the constraint domain - every rank evaluates its row block of
        # a component's columns there from the coefficients, which every rank keeps (no column-parallel stage A)
        ("chain, blow-up 4", syn.chain_graph(300, 4), None, False, (5, 0, 3, 2)),
        ("mixed sizes, blow-up 8, last layer 2^2", [(0, syn.chain_graph(64, 4)[0][1]), (1, syn.chain_graph(500, 5)[1][1])], None, False,
         (5, 2, 4, 3)),
    ]


def _prove_all(make_ctx, stats=None):
    out = []
    ctxs = {}
    for case in _cases():
        name, tabs, luts, pinned = case[:4]
        key = (pinned, case[4] if len(case) > 4 else None)
        if key not in ctxs:
            ctxs[key] = make_ctx(*key)
        proof = ctxs[key].prove_tables([(k, r, len(r)) for k, r in tabs], luts)
        out.append((name, hashlib.sha256(proof).hexdigest(), len(proof)))
        if stats is not None:
            t = ctxs[key].timings()
            stats.append((t["shard_a2a_calls"], t["shard_a2a_bytes"], t["shard_gather_calls"]))
    for c in ctxs.values():
        c.close()
    return out


def _make_ctx(pinned, pcs=None):
    from luminair_amd import backend
    lib = backend.Library(EMU)
    cfg = lib.default_config()
    if pinned:
        cfg.protocol_variant = backend.VARIANT_PINNED
    if pcs:
        cfg.pow_bits, cfg.log_last_layer, cfg.n_queries = pcs[:3]
        if len(pcs) > 3:
            cfg.log_blowup = pcs[3]
    return backend.Context(0, cfg, lib)


def _worker(rank, world, port, fri_min_log, a2a, q, rows=False):
    # a2a: SURVEY 8e stages A / B (column-parallel interpolation + all-to-all into row blocks) for every column size;
    # rows: the row-parallel front end as well (transposes and logup fractions of the own row block only, blocks sent to
    # the columns' owners) for every table with at least 64 rows per rank
    os.environ["LMN_SHARD_A2A_MIN_LOG"] = "4"
    os.environ["LMN_SHARD_ROWS_MIN_LOG"] = "4" if rows else "99"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from luminair_amd.sharded import shard_context
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def make(pinned, pcs=None):
            ctx = _make_ctx(pinned, pcs)
            shard_context(ctx, fri_min_log=fri_min_log, all_to_all=a2a)
            return ctx
        # errors surface identically on every rank (same transcript, same checks) and leave the contexts usable
        from luminair_amd import backend, synthetic as syn
        ctx = make(False)
        bad = syn.config2_add_only(64, 9)[0][1].copy()
        bad[3, 11] ^= 1                                  # out != lhs + rhs: ProverError(ConstraintsNotSatisfied)
        try:
            ctx.prove_tables([(0, bad, len(bad))])
            raise AssertionError("a violated constraint went unnoticed")
        except backend.LuminairBackendError as e:
            assert e.code == backend.ERR_CONSTRAINTS
        bad[3, 11] = 0x7fffffff                          # not a canonical M31 word
        try:
            ctx.prove_tables([(0, bad, len(bad))])
            raise AssertionError("a non-canonical word went unnoticed")
        except backend.LuminairBackendError as e:
            assert e.code == backend.ERR_INVALID_ARGUMENT
        ctx.close()
        stats = []
        proofs = _prove_all(make, stats)
        # the exchange mode under test really ran: every proof used the all-to-all (two per interaction run: rows + halo), or
        # none did (blow-ups above 2 never do: their columns' coefficients stay on every rank)
        blowup2 = [len(c) <= 4 or len(c[4]) <= 3 or c[4][3] == 1 for c in _cases()]
        assert all((c > 0) == (bool(a2a) and b2) for (c, _, _), b2 in zip(stats, blowup2)), stats
        assert all(g > 0 for _, _, g in stats)
        q.put((rank, proofs))
    finally:
        dist.destroy_process_group()


def _single():
    return _prove_all(_make_ctx)


@pytest.fixture(scope="module")
def single_rank_proofs():
    if not os.path.exists(EMU):
        import subprocess
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return _single()


@pytest.mark.parametrize("world,fri_min_log,a2a,rows", [(2, 4, True, False), (4, 5, True, True), (8, 0, True, True),
                                                        (2, 5, True, True), (2, 5, False, False), (8, 4, False, False)])
def test_sharded_prove_is_byte_identical(world, fri_min_log, a2a, rows, single_rank_proofs):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fri_min_log, a2a, q, rows)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _ in res] == list(range(world))
    for _, out in res:
        assert out == single_rank_proofs


@pytest.mark.gpu
def test_gpu_sharded_prove_single_rank_over_rccl(hip_lib_path):
    """The sharded code path on the real library with its built-in RCCL transport (librccl dlopen'ed,
    `lmn_ctx_set_shard_rccl`, collectives on the prover's own stream) and one rank: row-block commits, the
    all-gathers of subtree roots / composition evaluations / FRI layers / decommitted values all run (over a
    1-rank communicator) and the proof must equal the unsharded one byte for byte, at two FRI thresholds."""
    import torch.distributed as dist
    from luminair_amd import backend, synthetic as syn
    from luminair_amd.sharded import shard_context
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
    except Exception as e:      # an environment without a usable RCCL transport is not a parity failure
        pytest.skip("nccl process group unavailable: %s" % e)
    try:
        lib = backend.Library(hip_lib_path)
        plain = backend.Context(0, None, lib)
        cases = [syn.config2_add_only(1 << 14, 3), syn.config3_mixed(13, 12, 12, 4), syn.config5_linear_layers(4, 16, 32, 5)]
        want = [plain.prove_tables([(k, r, len(r)) for k, r in tabs]) for tabs in cases]
        for fri_min_log in (0, 6):
            ctx = backend.Context(0, None, lib)
            shard_context(ctx, fri_min_log=fri_min_log)
            for tabs, w in zip(cases, want):
                assert ctx.prove_tables([(k, r, len(r)) for k, r in tabs]) == w
            ctx.clear_shard()
            assert ctx.prove_tables([(k, r, len(r)) for k, r in cases[0]]) == want[0]
            ctx.close()
        plain.close()
    finally:
        dist.destroy_process_group()


def _gpu_cases():
    """(tables, luts, pinned variant?): sizes where every kernel family of the big workloads runs, plus BASELINE
    config 4 (the black-scholes MLP shape with its exp2 LUT in tree 0), which north_star shards over 4 GPUs."""
    from luminair_amd import synthetic as syn
    c4, luts4 = syn.config4_black_scholes_shape()
    return [(syn.config2_add_only(1 << 15, 3), None, False), (syn.config3_mixed(13, 12, 12, 4), None, False),
            (syn.config5_linear_layers(4, 16, 32, 5), None, False), (c4, luts4, True),
            (syn.less_than_graph(3000, 6), None, True), (syn.config2_graph_faithful(1 << 13, 3), None, True)]


def _gpu_prove_all(lib, shard=None):
    from luminair_amd import backend
    out, ctxs = [], {}
    for tabs, luts, pinned in _gpu_cases():
        if pinned not in ctxs:
            cfg = lib.default_config()
            cfg.protocol_variant = backend.VARIANT_PINNED if pinned else backend.VARIANT_KAT
            ctxs[pinned] = backend.Context(0, cfg, lib)
            if shard:
                shard(ctxs[pinned])
        out.append(hashlib.sha256(ctxs[pinned].prove_tables([(k, r, len(r)) for k, r in tabs], luts)).hexdigest())
    for c in ctxs.values():
        c.close()
    return out


def _gpu_worker(rank, world, port, fri_min_log, lib_path, q, a2a=True):
    os.environ["LMN_SHARD_ROWS_MIN_LOG"] = "13" if (world == 4 and a2a) else "99"   # one world with the row-parallel front end
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from luminair_amd import backend, synthetic as syn
    from luminair_amd.sharded import shard_context
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank on GPU 0: the sharded code paths, not the speed-up
        q.put((rank, _gpu_prove_all(backend.Library(lib_path),
                                    lambda c: shard_context(c, fri_min_log=fri_min_log, all_to_all=a2a))))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,fri_min_log,a2a", [(2, 6, True), (4, 0, True), (8, 7, True), (4, 6, False)])
def test_gpu_sharded_prove_multi_rank_on_one_gpu(world, fri_min_log, a2a, hip_lib_path):
    """The world > 1 code paths of the REAL library (row-block LDEs from `k_fft_top_block`, halo blocks of the last
    logup group, row-offset constraint / quotient / fold kernels, owner-aware decommitment; with a2a the
    column-parallel interpolate + extend, the block packing and the all-to-all for columns of 2^13 rows and more) on hardware: `world`
    processes share GPU 0, the collective is the gloo-staged callback, and every rank must return the unsharded
    proof's bytes."""
    from luminair_amd import backend
    want = _gpu_prove_all(backend.Library(hip_lib_path))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, fri_min_log, hip_lib_path, q, a2a)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert out == want
