"""`lmn_ctx_set_shard_rccl` with MORE THAN ONE RANK on a machine without GPUs: the emulation build carries the library's
built-in RCCL transport (shard.cpp RcclApi / RcclTransport), LMN_RCCL_LIB points it at tests/emu/libstub_rccl.so - the
NCCL entry points over POSIX shared memory - and world 2 / 4 / 8 rank processes prove the sharded test pies.  What this
executes before the driver's multi-GPU node does: `lmn_rccl_unique_id` on rank 0 and its hand-over, one communicator per
rank (`ncclCommInitRank` returns when every rank has joined), grouped all-gathers of coordinate columns, the grouped
send / recv all-to-all of SURVEY.md section 8e stage B, communicator teardown - and the proofs must equal the unsharded
bytes on every rank.  (RCCL itself refuses two ranks on one device, so the one-GPU boxes cannot run this on hardware.)"""
import os
import sys

import pytest
import torch.multiprocessing as mp

import test_sharded_prove as tsp

ROOT = tsp.ROOT
STUB = os.path.join(ROOT, "tests", "emu", "libstub_rccl.so")


def _worker(rank, world, fri_min_log, id_q, out_q):
    os.environ["LMN_RCCL_LIB"] = STUB
    os.environ["LMN_SHARD_A2A_MIN_LOG"] = "4"
    os.environ["LMN_SHARD_ROWS_MIN_LOG"] = "4" if world != 2 else "99"   # row-parallel front end at world 4 and 8
    sys.path.insert(0, ROOT)
    from luminair_amd import backend
    boot = tsp._make_ctx(False)
    if rank == 0:
        ident = boot.rccl_unique_id()
        for _ in range(world - 1):
            id_q.put(ident)
    else:
        ident = id_q.get(timeout=120)
    boot.close()
    idents = {}

    def make(pinned, pcs=None):
        ctx = tsp._make_ctx(pinned, pcs)
        ctx.set_shard_rccl(rank, world, ident, fri_min_log)      # a communicator per context, same id: the stub keys its
        return ctx                                               # shared memory by id and is re-joined by every rank in order
    stats = []
    proofs = tsp._prove_all(make, stats)
    # both primitives ran through the built-in transport (blow-ups above 2 - the last cases - keep every column's
    # coefficients on every rank: all-gathers only)
    blowup2 = [len(c) <= 4 or len(c[4]) <= 3 or c[4][3] == 1 for c in tsp._cases()]
    assert all((c > 0) == b2 and g > 0 for (c, _, g), b2 in zip(stats, blowup2)), stats
    # a rejected re-shard keeps the transport; clearing it returns the context to unsharded proofs
    ctx = make(False)
    with pytest.raises(backend.LuminairBackendError):
        ctx.set_shard_rccl(world, world, ident, fri_min_log)
    ctx.clear_shard()
    ctx.close()
    out_q.put((rank, proofs))


@pytest.mark.parametrize("world,fri_min_log", [(2, 4), (4, 0), (8, 5)])
def test_set_shard_rccl_multi_rank_against_the_stub_transport(world, fri_min_log, single_rank_proofs):
    if not os.path.exists(STUB):
        import subprocess
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, fri_min_log, id_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out_q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _ in res] == list(range(world))
    for _, out in res:
        assert out == single_rank_proofs


single_rank_proofs = tsp.single_rank_proofs
