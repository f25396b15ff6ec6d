#!/usr/bin/env python3
"""Per-rank critical path of a sharded proof, measured on ONE GPU.

The test boxes have a single MI355X, so `bench.py --shard-proof` can only be run at G = 1 there.  This tool measures
what one rank of a G-rank sharded proof does, on real hardware, in two steps:
  record:  G processes share the GPU (gloo-staged collective, as tests/test_sharded_prove.py) and prove the workload
           once; rank 0 saves the contents of every all-gather result, in call order;
  replay:  ONE process is rank 0 of G with a collective that uploads the recorded buffers instead of talking to peers,
           alone on the GPU: its kernels are exactly rank 0's share of the sharded proof.  The proof bytes must equal
           the unsharded proof's.
Reported per G: wall time per proof, time spent inside the collective callbacks (H2D uploads over PCIe stand in for
the xGMI all-gather and, unlike RCCL, stall the stream), and their difference = the rank's compute critical path.
An ESTIMATE of the multi-GPU latency, not a multi-GPU measurement.
Usage: shard_replay.py <config2a|config3|config5> [G ...]"""
import ctypes as C
import hashlib
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(name):
    from luminair_amd import synthetic as syn
    if name == "config5":
        return syn.config5_linear_layers()
    if name == "config3":
        return syn.config3_mixed()
    return syn.config2_add_only(1 << 20, 42)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def record_worker(rank, world, port, name, path, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from luminair_amd import backend
    from luminair_amd.backend import DeviceBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = backend.Context(0)
    rec = []

    def all_gather(buf, nbytes, _stream):
        mine = ctx.download(DeviceBuffer(ctx, buf + rank * nbytes, nbytes, owned=False), np.uint8)
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine))
        whole = np.concatenate([p_.numpy() for p_ in parts])
        ctx.upload_to(DeviceBuffer(ctx, buf, nbytes * world, owned=False), whole)
        if rank == 0:
            rec.append(whole)
    def all_to_all(send, so, sb, recv, ro, rb, _stream):
        outs = [torch.from_numpy(ctx.download(DeviceBuffer(ctx, send + so[p], sb[p], owned=False), np.uint8)) if sb[p]
                else torch.empty(0, dtype=torch.uint8) for p in range(world)]
        ins = [torch.empty(rb[p], dtype=torch.uint8) for p in range(world)]
        from luminair_amd.sharded import _gloo_all_to_all
        _gloo_all_to_all(ins, outs, rank, world, None)
        for p in range(world):
            if rb[p]:
                ctx.upload_to(DeviceBuffer(ctx, recv + ro[p], rb[p], owned=False), ins[p].numpy())
        if rank == 0:
            rec.append([ins[p].numpy().copy() for p in range(world)])
    use_a2a = os.environ.get("LMN_REPLAY_A2A", "1") != "0"
    ctx.set_shard(rank, world, all_gather, int(os.environ.get('LMN_FRI_MIN_LOG', '0')), all_to_all if use_a2a else None)
    tabs = workload(name)
    bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
    proof = ctx.prove_tables(bufs)
    if rank == 0:
        flat = {}
        for i, r in enumerate(rec):
            if isinstance(r, list):
                for p_, part in enumerate(r):
                    flat["a2a_%d_%d" % (i, p_)] = part
            else:
                flat["ag_%d" % i] = r
        np.savez(path, n=np.array([len(rec)]), **flat)
    q.put((rank, hashlib.sha256(proof).hexdigest()))
    ctx.close()
    dist.destroy_process_group()


def replay(world, name, path, want_sha, reps):
    """rank 0 of `world`, alone on the GPU.  Two collectives: `pcie` uploads the recorded result over PCIe and waits
    (pessimistic: a host round trip per all-gather); `ideal` copies it from a device-resident staging copy on the
    prover's own stream without waiting (an interconnect of infinite bandwidth and zero latency: what remains is the
    rank's own stream-ordered work)."""
    from luminair_amd import backend
    from luminair_amd.backend import DeviceBuffer
    data = np.load(path)
    rec = []
    for i in range(int(data["n"][0])):
        if "ag_%d" % i in data.files:
            rec.append(data["ag_%d" % i])
        else:
            rec.append([data["a2a_%d_%d" % (i, p_)] for p_ in range(world)])
    ctx = backend.Context(0)
    staged = [ctx.upload(r) if not isinstance(r, list) else [ctx.upload(x) if len(x) else None for x in r] for r in rec]
    state = {"i": 0, "mode": "pcie"}

    def all_to_all(send, so, sb, recv, ro, rb, _stream):
        k = state["i"] % len(rec)
        state["i"] += 1
        parts = rec[k]
        assert isinstance(parts, list) and [len(x) for x in parts] == list(rb)
        for p_ in range(world):
            if not rb[p_]:
                continue
            if p_ == 0:        # the own part is a real device copy out of the send buffer
                ctx.device_copy(recv + ro[0], send + so[0], rb[0])
            elif state["mode"] == "pcie":
                ctx.upload_to(DeviceBuffer(ctx, recv + ro[p_], rb[p_], owned=False), parts[p_])
            else:
                ctx.device_copy(recv + ro[p_], staged[k][p_].ptr, rb[p_])

    def all_gather(buf, nbytes, _stream):
        k = state["i"] % len(rec)
        state["i"] += 1
        assert not isinstance(rec[k], list) and len(rec[k]) == nbytes * world
        if state["mode"] == "pcie":
            ctx.upload_to(DeviceBuffer(ctx, buf, nbytes * world, owned=False), rec[k])
        else:
            ctx.device_copy(buf, staged[k].ptr, nbytes * world)
    use_a2a = os.environ.get("LMN_REPLAY_A2A", "1") != "0"
    ctx.set_shard(0, world, all_gather, int(os.environ.get('LMN_FRI_MIN_LOG', '0')), all_to_all if use_a2a else None)
    tabs = workload(name)
    bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
    ags = [r for r in rec if not isinstance(r, list)]
    a2as = [r for r in rec if isinstance(r, list)]
    res = {"world": world, "exchange": "all_to_all + all_gather" if use_a2a else "all_gather only (replicated interpolation)",
           "all_gathers_per_proof": len(ags), "gathered_bytes_per_proof": int(sum(len(r) for r in ags)),
           "all_to_alls_per_proof": len(a2as),
           "all_to_all_received_bytes_per_proof": int(sum(len(x) for r in a2as for x in r[1:]))}
    for mode in ("pcie", "ideal"):
        state["mode"] = mode
        assert hashlib.sha256(ctx.prove_tables(bufs)).hexdigest() == want_sha, "replayed rank-0 proof differs"
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.prove_tables(bufs)
            ts.append(1e3 * (time.perf_counter() - t0))
        res["rank0_ms_" + mode] = sorted(ts)[reps // 2]
    # where rank 0's time goes (HIP events of one profiled proof, `ideal` collective)
    ctx.set_profiling(True)
    ctx.prove_tables(bufs)
    ctx.set_profiling(False)
    res["rank0_stage_ms"] = {k: round(v, 4) for k, v in ctx.timings().items() if k.endswith("_ms")}
    # exchange models on top of `ideal`: every rank receives (world-1)/world of each all-gather.
    #  ring:   15 us per all-gather call, 150 GB/s receive bandwidth (one xGMI link per ring step: per-link bound)
    #  direct: 15 us per GROUP of calls (the library batches the coordinate columns of one exchange in one RCCL
    #          group: consecutive calls of equal size are counted once), (world-1) peers feeding the rank over their
    #          own links at ~153 GB/s each (xGMI is point-to-point: 7 links per GPU on an 8-GPU node)
    #  an all-to-all is point-to-point by construction: ring = everything a rank receives passes one link; direct = every
    #  peer's part over its own link (the largest part bounds the call)
    recv = res["gathered_bytes_per_proof"] * (world - 1) / world
    groups = sum(1 for i, r in enumerate(ags) if i == 0 or len(r) != len(ags[i - 1]))
    res["all_gather_groups_per_proof"] = groups
    a2a_recv = res["all_to_all_received_bytes_per_proof"]
    a2a_direct = sum(max((len(x) for x in r[1:]), default=0) for r in a2as)
    res["exchanged_bytes_received_per_rank"] = int(recv + a2a_recv)
    res["modelled_exchange_ms"] = 1e3 * ((len(ags) + len(a2as)) * 15e-6 + (recv + a2a_recv) / 150e9)
    res["modelled_exchange_direct_links_ms"] = 1e3 * ((groups + len(a2as)) * 15e-6 + recv / (153e9 * max(1, world - 1))
                                                      + a2a_direct / 153e9)
    res["estimated_latency_ms"] = res["rank0_ms_ideal"] + res["modelled_exchange_ms"]
    res["estimated_latency_direct_links_ms"] = res["rank0_ms_ideal"] + res["modelled_exchange_direct_links_ms"]
    ctx.close()
    return res


def main():
    import torch.multiprocessing as mp
    from luminair_amd import backend
    name = sys.argv[1] if len(sys.argv) > 1 else "config2a"
    worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
    reps = 5 if name == "config5" else 11
    ctx = backend.Context(0)
    tabs = workload(name)
    bufs = [(k, ctx.upload(r), len(r)) for k, r in tabs]
    want = hashlib.sha256(ctx.prove_tables(bufs)).hexdigest()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.prove_tables(bufs)
        ts.append(1e3 * (time.perf_counter() - t0))
    out = {"workload": name, "rows": int(sum(n for _, _, n in bufs)), "unsharded_ms": sorted(ts)[reps // 2], "replay": []}
    for _, b, _ in bufs:
        b.free()
    ctx.close()
    for world in worlds:
        path = "/tmp/shard_rec_%s_%d.npz" % (name, world)
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        port = _free_port()
        procs = [mpc.Process(target=record_worker, args=(r, world, port, name, path, q)) for r in range(world)]
        for p in procs:
            p.start()
        shas = [q.get(timeout=1200)[1] for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
        assert all(s == want for s in shas), "sharded proof differs from the unsharded one"
        out["replay"].append(replay(world, name, path, want, reps))
        os.remove(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
