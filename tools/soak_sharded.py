#!/usr/bin/env python3
"""Soak of the sharded prover through the built-in RCCL transport (one rank): many proofs on one context, set/clear
of the shard in between, device-handle ops in a loop; every proof must equal the unsharded bytes and device memory
must not grow."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from luminair_amd import backend, synthetic as syn
from luminair_amd.sharded import shard_context

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = backend.Context(0)
tabs = syn.config2_add_only(1 << 18, 7)
bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
want = hashlib.sha256(ctx.prove_tables(bufs)).hexdigest()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
n = 0
for rnd in range(6):
    shard_context(ctx, fri_min_log=(0, 6, 9)[rnd % 3])
    for _ in range(60):
        assert hashlib.sha256(ctx.prove_tables(bufs)).hexdigest() == want
        n += 1
    ctx.clear_shard()
    assert hashlib.sha256(ctx.prove_tables(bufs)).hexdigest() == want
    torch.cuda.synchronize()
    print("round %d: device memory in use grew by %.1f MiB since start" % (rnd, (free0 - torch.cuda.mem_get_info()[0]) / 2**20), flush=True)
    if rnd == 0:
        free_r0 = torch.cuda.mem_get_info()[0]   # RCCL's one-time pools and the arena's growth to the sharded layout
free_mid = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(1)
ev = rng.integers(0, (1 << 31) - 1, size=(4, 1 << 14), dtype=np.uint64).astype(np.uint32)
for _ in range(200):
    h = ctx.col_from_cpu(ev)
    h.interpolate()
    lde = h.evaluate(15)
    t = ctx.commit([lde])
    r = t.root()
    g, lam = lde.decompose()
    f = lde.fold_line((1, 2, 3, 4))
    for x in (h, lde, g, f):
        x.free()
    t.free()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("handle-op rounds: delta %.1f MiB" % ((free_mid - free1) / 2**20), flush=True)
print("sharded soak: %d sharded proofs + 200 handle-op rounds in %.1f s, all bytes equal; device memory delta %.1f MiB"
      % (n, time.time() - t0, (free0 - free1) / 2**20))
assert free_r0 - free1 < 64 << 20   # nothing grows after the first shard set-up
dist.destroy_process_group()
