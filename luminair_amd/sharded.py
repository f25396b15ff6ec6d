"""Single-commitment sharding across GPUs: the one part of the path that partitions without moving column data
(SURVEY.md §8e stage C/D).  Every rank hashes the aligned Merkle subtree over its block of rows on its own GPU;
the only exchange is an all-gather of the subtree roots (32 bytes per rank, RCCL under the `nccl` backend), after
which every rank hashes the top log2(world) levels itself (deterministic, identical everywhere).

This is a level-2 building block (`lmn_op_merkle_root` per rank), not a sharded `prove`: DESIGN.md §6 says what a
full single-proof sharding needs beyond it."""
from __future__ import annotations

import hashlib
import os
from typing import List, Optional, Sequence

import numpy as np


def shard_context(ctx, group=None, fri_min_log: int = 0, all_to_all: bool = True):
    """Make `ctx.prove_tables` a single proof sharded over the ranks of `group` (torch.distributed must be
    initialised; one context per rank; every rank passes the same tables and receives the same proof bytes).
    `nccl` backend: the library's own RCCL communicator on the prover's stream (`lmn_ctx_set_shard_rccl`), the
    128-byte unique id travels through a torch broadcast.  `gloo` backend (CPU tests over the emulation build, where
    "device" pointers are host pointers): the `lmn_collective` callback form over `dist.all_gather`."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            ident = torch.frombuffer(bytearray(ctx.rccl_unique_id()), dtype=torch.uint8).cuda()
        dist.broadcast(ident, 0, group=group)
        ctx.set_shard_rccl(rank, world, bytes(ident.cpu().numpy().tobytes()), fri_min_log)
        return

    device_memory = "emu" not in os.path.basename(ctx.lib.path)     # the emulation build's "device" memory is host memory

    def all_gather(buf, nbytes, _stream):
        if device_memory:
            # a transport that only speaks host memory (gloo) under the real library: stage this rank's part through
            # the host on the prover's own stream (lmn_download / lmn_upload_to are stream-ordered and synchronous).
            # This is the test path that runs several ranks on ONE GPU; production uses the RCCL transport above.
            from .backend import DeviceBuffer
            mine = ctx.download(DeviceBuffer(ctx, buf + rank * nbytes, nbytes, owned=False), np.uint8)
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine), group=group)
            whole = np.concatenate([p_.numpy() for p_ in parts])
            ctx.upload_to(DeviceBuffer(ctx, buf, nbytes * world, owned=False), whole)
            return
        whole = np.ctypeslib.as_array((C.c_uint8 * (nbytes * world)).from_address(buf))
        t = torch.from_numpy(whole)
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, t[rank * nbytes:(rank + 1) * nbytes].clone(), group=group)
        for r, part in enumerate(parts):
            if r != rank:
                t[r * nbytes:(r + 1) * nbytes] = part
    def a2a(send, send_off, send_bytes, recv, recv_off, recv_bytes, _stream):
        # the all-to-all of SURVEY.md section 8e stage B over the same host-staged transport (tests only)
        def read(ptr, nbytes):
            if nbytes == 0:
                return np.empty(0, np.uint8)
            if device_memory:
                from .backend import DeviceBuffer
                return ctx.download(DeviceBuffer(ctx, ptr, nbytes, owned=False), np.uint8)
            return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).copy()

        def write(ptr, data):
            if len(data) == 0:
                return
            if device_memory:
                from .backend import DeviceBuffer
                ctx.upload_to(DeviceBuffer(ctx, ptr, len(data), owned=False), data)
            else:
                np.ctypeslib.as_array((C.c_uint8 * len(data)).from_address(ptr))[:] = data
        outs = [torch.from_numpy(read(send + send_off[p], send_bytes[p])) for p in range(world)]
        ins = [torch.empty(recv_bytes[p], dtype=torch.uint8) for p in range(world)]
        dist.all_to_all(ins, outs, group=group) if dist.get_backend(group) != "gloo" else _gloo_all_to_all(ins, outs, rank, world, group)
        for p in range(world):
            write(recv + recv_off[p], ins[p].numpy())
    ctx.set_shard(rank, world, all_gather, fri_min_log, a2a if all_to_all else None)


def _gloo_all_to_all(ins, outs, rank, world, group):
    """gloo has no all_to_all: pairwise exchange (isend / irecv), the own part is a copy"""
    import torch.distributed as dist
    reqs = []
    for p in range(world):
        if p == rank:
            ins[p].copy_(outs[p])
            continue
        if outs[p].numel():
            reqs.append(dist.isend(outs[p], p, group=group))
        if ins[p].numel():
            reqs.append(dist.irecv(ins[p], p, group=group))
    for r in reqs:
        r.wait()


def _parent(left: bytes, right: bytes) -> bytes:
    # Blake2sMerkleHasher::hash_node(children, no column values): blake2s(left || right), SURVEY.md Appendix A.4
    return hashlib.blake2s(left + right, digest_size=32).digest()


def row_block(cols: Sequence[np.ndarray], rank: int, world: int) -> List[np.ndarray]:
    """Rank `rank`'s share of every column: rows [rank*n/world, (rank+1)*n/world) of a column of n rows (columns
    are in bit-reversed domain order, where an aligned block of rows is an aligned Merkle subtree)."""
    if world & (world - 1):
        raise ValueError("world size must be a power of two")
    out = []
    for c in cols:
        n = len(c)
        if n < world or n & (n - 1):
            raise ValueError("every column needs a power-of-two length of at least world_size rows")
        out.append(c[rank * n // world:(rank + 1) * n // world])
    return out


def commit_sharded(ctx, coeff_cols: Sequence[np.ndarray], log_blowup: int = 1, group=None) -> bytes:
    """`tree_builder.extend_polys(..).commit()` sharded over the ranks of `group` with no bulk exchange
    (DESIGN.md §6), on device handles: the coefficient columns are uploaded ONCE (`lmn_col_from_cpu`), every rank
    evaluates ONLY its block of rows of every column's LDE (`lmn_col_evaluate_block`) and hashes the Merkle subtree
    over those blocks (`lmn_col_commit`) without the data leaving HBM; the 32-byte subtree roots are all-gathered
    and the top log2(world) levels hashed on every rank.  Returns the same root as a single-GPU commit of all
    columns (world sizes 1, 2, 4, 8).  (`lmn_prove` on a context with `shard_context` does all of this - and the
    rest of the proof - natively; this is the level-2 building block.)"""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world & (world - 1) or world > 8:
        raise ValueError("world size must be 1, 2, 4 or 8")
    g = world.bit_length() - 1
    blocks, handles = [], []
    for c in coeff_cols:
        h = ctx.col_from_cpu(np.ascontiguousarray(c, dtype=np.uint32).reshape(1, -1))
        log_domain = h.log_size + log_blowup
        blocks.append(h.evaluate(log_domain) if g == 0 else h.evaluate_block(log_domain, g, rank))
        handles.append(h)
    tree = ctx.commit(blocks)
    sub = tree.root()
    tree.free()
    for h in handles + blocks:
        h.free()
    return _gather_root(sub, group)


def merkle_root_sharded(ctx, cols: Sequence[np.ndarray], group=None) -> bytes:
    """Root of the mixed-size-column Merkle tree over `cols`, subtrees sharded over the ranks of `group`
    (torch.distributed must be initialised; all ranks pass the same columns or at least their own rows)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    return _gather_root(ctx.merkle_root(row_block(cols, rank, world)), group)   # node `rank` of level log2(world)


def _gather_root(sub: bytes, group=None) -> bytes:
    """All-gather the per-rank subtree roots and hash the top log2(world) levels (identically on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.frombuffer(bytearray(sub), dtype=torch.uint8).to(dev)
    parts = [torch.empty(32, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    nodes = [bytes(p.cpu().numpy().tobytes()) for p in parts]
    while len(nodes) > 1:
        nodes = [_parent(nodes[2 * i], nodes[2 * i + 1]) for i in range(len(nodes) // 2)]
    return nodes[0]
