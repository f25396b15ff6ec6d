"""Merkle subtree sharding over ranks (luminair_amd/sharded.py) with the gloo backend on CPU: every rank runs
`lmn_op_merkle_root` on its row block through the TEST-ONLY emulation build, the roots are all-gathered, and the
result equals the oracle's root of the whole tree - for world sizes 2 and 4, equal and mixed column sizes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _columns():
    rng = np.random.default_rng(17)
    P = (1 << 31) - 1
    return [rng.integers(0, P, size=1 << k, dtype=np.uint64).astype(np.uint32) for k in (9, 9, 7, 9, 4, 7, 10)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from luminair_amd import backend
    from luminair_amd.sharded import merkle_root_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = backend.Context(0, None, backend.Library(os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")))
    from luminair_amd.sharded import commit_sharded
    root = merkle_root_sharded(ctx, _columns())
    q.put((rank, root, commit_sharded(ctx, _columns(), 1)))
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_merkle_root_equals_oracle(world):
    from oracle.merkle import MerkleTree
    so = os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    from oracle import fft
    want = MerkleTree(_columns()).root()
    # commit_sharded takes the columns as COEFFICIENTS: the tree is over their blow-up-2 evaluations
    lde = [fft.evaluate(c.astype(np.uint64).reshape(1, -1), len(c).bit_length())[0].astype(np.uint32) for c in _columns()]
    want_commit = MerkleTree(lde).root()
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _, _ in res] == list(range(world))
    assert all(root == want for _, root, _ in res)
    assert all(commit == want_commit for _, _, commit in res)


def test_row_block_validation():
    from luminair_amd.sharded import row_block
    cols = _columns()
    with pytest.raises(ValueError):
        row_block(cols, 0, 3)
    with pytest.raises(ValueError):
        row_block(cols, 0, 32)      # the 16-row column cannot be split 32 ways
    assert [len(c) for c in row_block(cols, 1, 4)] == [128, 128, 32, 128, 4, 32, 256]


@pytest.mark.gpu
def test_gpu_sharded_merkle_root_single_rank_over_rccl(hip_lib_path):
    """The same code path on the GPU box with the nccl (= RCCL) backend and one rank: device subtree root,
    all-gather of a CUDA tensor, top levels."""
    import torch.distributed as dist
    from luminair_amd import backend
    from luminair_amd.sharded import merkle_root_sharded
    from oracle.merkle import MerkleTree
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
    except Exception as e:      # an environment without a usable RCCL transport is not a parity failure
        pytest.skip("nccl process group unavailable: %s" % e)
    try:
        ctx = backend.Context(0, None, backend.Library(hip_lib_path))
        cols = _columns() + [np.arange(1 << 16, dtype=np.uint32)]
        assert merkle_root_sharded(ctx, cols) == MerkleTree(cols).root()
        from luminair_amd.sharded import commit_sharded
        from oracle import fft
        lde = [fft.evaluate(c.astype(np.uint64).reshape(1, -1), len(c).bit_length())[0].astype(np.uint32) for c in cols]
        assert commit_sharded(ctx, cols, 1) == MerkleTree(lde).root()
        ctx.close()
    finally:
        dist.destroy_process_group()
