#!/usr/bin/env python3
"""bench.py — proofs/sec of the MI355X prove hot path on BASELINE.json config 2
(single Add-op AIR, 2^20 trace rows per proof), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one whole proof (AoS trace rows resident in HBM -> bincode proof bytes on the host).
Proofs are independent, so ranks shard proofs with no data-path collective ("weak" scaling);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks timing.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak
# ALU ceilings measured on MI355X with tools/microbench.hip; informational only
BLAKE2S_PEAK_GCOMP = 39.0                                  # G compressions/s, chip-wide
BUTTERFLY_PEAK_G = 1.0 / (1.0 / 6326.0 + 2.0 / 12800.0)    # G butterflies/s = 1 M31 mul + 2 add|sub


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--log-rows", type=int, default=20, help="log2 rows of the Add trace (default 20 = BASELINE config 2)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent proofs in flight per GPU (one prover context + HIP stream each)")
    ap.add_argument("--host-rows", action="store_true",
                    help="hand the trace rows over as host buffers (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-anchor", action="store_true", help="skip the 32x32 reference-shape latency anchor")
    ap.add_argument("--cpu-sample-log", type=int, default=20)
    ap.add_argument("--shard-proof", action="store_true",
                    help="latency of ONE proof sharded over the --gpus ranks (lmn_ctx_set_shard_rccl) instead of "
                         "proofs/s of independent proofs; --shard-workload picks the pie")
    ap.add_argument("--shard-workload", default="config5", choices=["config5", "config2a", "config3"],
                    help="config5: 256 x (Mul + SumReduce + Add), 2^24 rows; config2a: Add 2^log-rows; config3: 2^22 rows")
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-results (host_rows, config_2b, sharded_proof) and print the headline only")
    return ap.parse_args(argv)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def timed_region(step_fn, steps, warmup, barrier, device_sync):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides."""
    for _ in range(warmup):
        step_fn()
    device_sync()
    barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    device_sync()
    barrier()
    device_sync()
    return time.perf_counter() - t0


def aggregate(elapsed, world, steps_per_rank, reduce_max):
    """value = units all ranks processed / max-over-ranks time."""
    tmax = reduce_max(elapsed)
    return {"seconds": tmax, "value": world * steps_per_rank / tmax, "ms_per_step": 1e3 * tmax / steps_per_rank}


def cpu_baseline(sample_log, full_log):
    """The oracle's plain-C restatement (oracle/c/stark_kernels.c, OpenMP, driven by oracle/prover.py) timed on this
    box on the same workload, in its own process (oracle/cpu_baseline.py says why): one cold proof (builds the
    twiddle/domain tables, as the reference does per proof), three warm ones, and the reference's own published
    32x32 Add shape.  Bounded: a few seconds of CPU work."""
    import subprocess
    threads = max(1, min(64, (os.cpu_count() or 2) // 2))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="passive", OMP_PROC_BIND="close",
               OMP_PLACES="cores", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", str(sample_log), str(full_log)], cwd=ROOT, env=env,
                         check=True, capture_output=True, text=True, timeout=900)
    return json.loads(out.stdout.strip().splitlines()[-1])


def throughput(provers, bufs, steps, warmup, luts=None):
    """proofs/s of `steps` proofs dealt round-robin over the in-flight contexts (no barrier: sub-results only)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(provers)
    with ThreadPoolExecutor(max_workers=n) as pool:
        def one(i):
            return provers[i].ctx.prove_tables(bufs[i], luts)
        for f in [pool.submit(one, i % n) for i in range(warmup)]:
            f.result()
        t0 = time.perf_counter()
        pending = []
        for k in range(steps):
            if len(pending) >= n:
                pending.pop(0).result()
            pending.append(pool.submit(one, k % n))
        for f in pending:
            f.result()
        dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "proofs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps}


def solo_latency(ctx, tables, n=9, luts=None):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        ctx.prove_tables(tables, luts)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts[len(ts) // 2]


def shard_workload(name, log_rows):
    from luminair_amd import synthetic as syn
    if name == "config5":
        return syn.config5_linear_layers(), "BASELINE config 5: 256 x (Mul + SumReduce + Add), Mul 2^23 + SumReduce 2^23 + Add 2^15 rows"
    if name == "config3":
        return syn.config3_mixed(), "BASELINE config 3: Add 2^21 + Mul 2^20 + Recip 2^20 rows"
    return syn.config2_add_only(1 << log_rows, 42), "BASELINE config 2a: Add 2^%d rows" % log_rows


def shard_proof_main(args, rank, local_rank, world):
    """ONE proof sharded over the ranks: every rank holds the same tables, evaluates / hashes /
    folds only its row blocks, the library's own RCCL communicator carries the all-gathers on the prover's stream.
    A step = one whole sharded proof; value = proofs/s of that single stream of proofs (1 / latency)."""
    import hashlib
    import torch
    import torch.distributed as dist
    import luminair_amd
    from luminair_amd.sharded import shard_context
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    prover = luminair_amd.Prover(local_rank)
    tabs, what = shard_workload(args.shard_workload, args.log_rows)        # identical on every rank (fixed seed)
    bufs = [(k, prover.ctx.upload(r), len(r)) for k, r in tabs]
    plain = prover.ctx.prove_tables(bufs)                                   # unsharded reference bytes (+ context setup)
    plain_ms = solo_latency(prover.ctx, bufs, 5) if rank == 0 else None
    shard_context(prover.ctx)
    out = {}

    def step():
        out["proof"] = prover.ctx.prove_tables(bufs)

    def barrier():
        dist.barrier(device_ids=[local_rank])

    steps = min(args.steps, 32)
    elapsed = timed_region(step, steps, min(args.warmup, 4), barrier, torch.cuda.synchronize)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tmax = float(t.item())
    same = torch.tensor([1 if out["proof"] == plain else 0], device="cuda")
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        rows = sum(n for _, _, n in bufs)
        print(json.dumps({
            "metric": "sharded-proof latency (one proof over all GPUs)", "value": steps / tmax, "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": min(args.warmup, 4), "ms_per_step": 1e3 * tmax / steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 (M31/QM31 field arithmetic)", "data": "synthetic",
            "config": {"workload": what + "; ONE proof sharded into row blocks over %d rank(s), RCCL all-gathers of subtree "
                                          "roots / composition evaluations / small FRI layers only" % world,
                       "rows": rows, "parallelism": "single-proof row-block sharding x%d" % world,
                       "proof_bytes": len(out["proof"]), "proof_sha256": hashlib.sha256(out["proof"]).hexdigest()},
            "prove_latency_ms": 1e3 * tmax / steps, "unsharded_latency_ms_rank0": plain_ms,
            "rows_per_s": rows * steps / tmax,
            "bytes_identical_to_unsharded_proof_on_every_rank": bool(int(same.item())),
        }))
    prover.ctx.clear_shard()
    for _, b, _ in bufs:
        b.free()
    dist.destroy_process_group()


def main(argv=None):
    # RCCL prints a version banner on STDOUT when a communicator is created unless told otherwise; the contract is
    # ONE JSON line on stdout
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":   # leave a real debug level alone
        os.environ["NCCL_DEBUG"] = "NONE"
    args = parse_args(argv)
    rank, local_rank, world = dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if args.shard_proof:
        return shard_proof_main(args, rank, local_rank, world)
    import numpy as np
    import torch
    import luminair_amd
    from luminair_amd import synthetic as syn

    use_dist = world > 1 or args.force_dist   # --force-dist: the N > 1 code path on one rank (self-test)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = local_rank if torch.cuda.is_available() else 0

    # `inflight` independent prover contexts per GPU (own HIP stream + device arena each); a step is
    # still one whole proof, steps are dealt round-robin to the contexts and run concurrently
    from concurrent.futures import ThreadPoolExecutor
    inflight = max(1, min(args.inflight, args.steps))
    provers = [luminair_amd.Prover(dev) for _ in range(inflight)]
    prover = provers[0]
    tabs = syn.config2_add_only(1 << args.log_rows, 42 + rank)   # each rank proves its own trace
    if args.host_rows:
        bufs = [[(k, r, len(r)) for k, r in tabs] for p in provers]             # PCIe-inclusive variant
    else:
        bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]   # trace rows resident in HBM
    out = {}
    pool = ThreadPoolExecutor(max_workers=inflight)
    pending = []

    def one(i):
        out["proof"] = provers[i].ctx.prove_tables(bufs[i])

    counter = {"n": 0}

    def step():
        # keep at most `inflight` proofs outstanding (ctypes releases the GIL inside lmn_prove)
        i = counter["n"] % inflight
        counter["n"] += 1
        if len(pending) >= inflight:
            pending.pop(0).result()
        pending.append(pool.submit(one, i))

    def drain():
        while pending:
            pending.pop(0).result()

    # one-time context initialisation (twiddle tables, device arena) for EVERY in-flight context, so the
    # W warmup + K timed steps below never include a context's first-use setup whatever W is
    for i in range(inflight):
        one(i)

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    def device_sync():
        drain()
        torch.cuda.synchronize()

    def reduce_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = timed_region(step, args.steps, args.warmup, barrier, device_sync)
    agg = aggregate(elapsed, world, args.steps, reduce_max)
    # single-proof latency (one proof alone on the GPU) and the per-kernel timings behind `roofline`
    lat = []
    for _ in range(21):
        t0 = time.perf_counter()
        one(0)
        lat.append(1e3 * (time.perf_counter() - t0))
    lat.sort()
    latency_ms = lat[len(lat) // 2]                      # p50 of 21 solo proofs
    latency_p95_ms = lat[int(0.95 * (len(lat) - 1))]
    # one more solo proof with HIP-event profiling switched on: the source of `roofline` and `stage_ms`
    prover.ctx.set_profiling(True)
    one(0)
    prover.ctx.set_profiling(False)

    # roofline per kernel family, from HIP events recorded by the library on the prover's own stream
    # around every transform / tree of the solo proof above (lmn_timings).  `traffic` comes from the
    # committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction
    # applied: 2*FETCH_SIZE + WRITE_SIZE), per launch like `achieved`.
    tm = prover.timings()
    pmc = {}
    pmc_source = None
    try:
        for cand in ("r2_pmc_summary.json", "r1_pmc_summary.json"):
            pth = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(pth):
                with open(pth) as f:
                    pmc = json.load(f)["kernels"]
                pmc_source = "profiles/" + cand
                break
    except Exception:
        pass
    fams = {
        "k_fft_staged": (tm["fft_ms"], tm["fft_bytes"], tm["fft_launches"], ["k_fft_staged<false>", "k_fft_staged<true>"]),
        "k_merkle_fused": (tm["merkle_fused_ms"], tm["merkle_fused_bytes"], tm["merkle_fused_launches"],
                           ["k_merkle_fused", "k_merkle_fused<0>", "k_merkle_fused<1>", "k_merkle_fused<2>", "k_merkle_fused<3>"]),
    }
    alu = {
        "k_fft_staged": (tm["fft_butterflies"], BUTTERFLY_PEAK_G, "G butterflies/s"),
        "k_merkle_fused": (tm["merkle_fused_compressions"], BLAKE2S_PEAK_GCOMP, "G Blake2s compressions/s"),
    }

    def roof(name):
        ms, nbytes, launches, pmc_names = fams[name]
        launches = max(launches, 1)
        achieved = nbytes / (1e-3 * ms) / 1e9 if ms > 0 else 0.0
        traffic = None
        got = [pmc[k] for k in pmc_names if k in pmc]
        if got:
            tot_l = sum(g["launches"] for g in got)
            traffic = sum(g["hbm_bytes_per_launch_corrected"] * g["launches"] for g in got) / max(tot_l, 1)
        ops, alu_peak, alu_unit = alu[name]
        alu_achieved = ops / (1e-3 * ms) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (pmc_source + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE)")
                if traffic is not None and pmc_source else None,
                "launches_per_proof": launches,
                "avg_launch_ms": ms / launches, "algorithmic_bytes_per_launch": nbytes / launches,
                "alu_ceiling": {"achieved": alu_achieved, "peak_measured": alu_peak, "unit": alu_unit,
                                "frac": alu_achieved / alu_peak}}

    dom = max(fams, key=lambda k: fams[k][0])
    roofline = roof(dom)
    roofline_other = [roof(k) for k in fams if k != dom]

    # whole-proof figure against SURVEY.md §8d's minimum-traffic model (48*C*N + 1500*N bytes, C = 27 columns)
    model_bytes = (48 * 27 + 1500) * float(1 << args.log_rows)
    whole = {"model_bytes_per_proof": model_bytes, "achieved": model_bytes / (1e-3 * agg["ms_per_step"]) / 1e9 * world,
             "peak": HBM_PEAK_GBS * world, "unit": "GB/s"}
    whole["frac"] = whole["achieved"] / whole["peak"]

    # the reference's own published shape (32x32 Add, 1 024 Add rows + 2 048 Inputs rows,
    # 13.05 ms on a GitHub Actions runner) as a sanity anchor: solo GPU latency, median of 9
    anchor = None
    if rank == 0 and world == 1 and not args.no_anchor:
        from luminair_amd import backend as _bk
        ap = luminair_amd.Prover(dev, protocol_variant=_bk.VARIANT_PINNED)
        atabs = [(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 42)]
        ap.ctx.prove_tables(atabs)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            ap.ctx.prove_tables(atabs)
            ts.append(1e3 * (time.perf_counter() - t0))
        anchor = {"workload": "32x32 Add graph: Add 2^10 rows + Inputs 2^11 rows, host rows (PCIe-inclusive)",
                  "gpu_latency_ms": sorted(ts)[4], "reference_published_ms": 13.05,
                  "reference_hardware": "GitHub Actions ubuntu-latest CPU (docs/snippets/benchmark-component.mdx:172)"}

    # the step before the path: `process_trace` of the Add node on device tensors
    trace_gen = None
    if rank == 0:
        import numpy as _np
        n_rows = 1 << args.log_rows
        rng = _np.random.default_rng(7)
        dl = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(_np.int32))
        dr = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(_np.int32))
        rows_buf = prover.ctx.alloc(n_rows * 15 * 4)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            _, ob = prover.ctx.trace_elementwise(0, dl, dr, n_rows, node_id=2, input_ids=(0, 1), num_consumers=0,
                                                 is_final_output=True, input_mults=(0, 0), rows=rows_buf)
            ts.append(1e3 * (time.perf_counter() - t0))
            ob.free()
        for b_ in (dl, dr, rows_buf):
            b_.free()
        trace_gen = {"workload": "Add node process_trace on device tensors, 2^%d elements -> 15-column rows in HBM"
                                 % args.log_rows, "ms": sorted(ts)[2],
                     "reference_published_ms": 0.0959, "reference_workload": "32x32 Add trace generation (BASELINE.md §1)"}

    line = {
        "metric": "proofs/sec, 2^%d-row Add trace" % args.log_rows, "value": agg["value"], "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": agg["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (M31/QM31 field arithmetic)",
        "data": "synthetic" + (" (host rows: PCIe-inclusive)" if args.host_rows else ""),
        "config": {"workload": "BASELINE config 2a: single Add-op AIR, 2^%d trace rows per proof, PcsConfig default "
                               "(pow 5, blowup 2x, 3 queries), KAT protocol variant (the variant the reference's only "
                               "known-answer proof pins; Add's constraint forms are KAT-pinned, Mul's second "
                               "eval_fixed_mul slot and the Recip/Sqrt/Rem forms are unpinned and not used here)"
                               % args.log_rows,
                   "rows": 1 << args.log_rows, "proofs_per_rank": args.steps, "parallelism": "proof-sharded x%d" % world,
                   "proofs_in_flight_per_gpu": inflight,
                   "proof_bytes": len(out["proof"])},
        "prove_latency_ms": latency_ms,
        "prove_latency_p95_ms": latency_p95_ms,
        "stage_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        "roofline": roofline,
        "roofline_other": roofline_other,
        "whole_proof_vs_traffic_model": whole,
    }
    if rank == 0 and not args.no_extras and not args.host_rows:
        # sub-results next to the headline: the same workload with the trace rows handed over as
        # host buffers (the reference API takes a host pie: 60 MiB over PCIe per proof), and config 2b = what
        # gen_trace really emits for an Add node at HEAD (Add 2^20 rows consumed with multiplicity -1 + the Inputs
        # table of 2^21 rows; PINNED protocol variant - parity unpinned)
        try:
            hb = [[(k, r, len(r)) for k, r in tabs] for _ in provers]
            line["host_rows"] = dict(throughput(provers, hb, 48, 8), note="trace rows as host buffers: PCIe-inclusive",
                                     prove_latency_ms=solo_latency(prover.ctx, hb[0]))
        except Exception as e:  # never lose the headline over a sub-result
            line["host_rows"] = {"error": str(e)}
        try:
            from luminair_amd import backend as _bk
            p2 = [luminair_amd.Prover(dev, protocol_variant=_bk.VARIANT_PINNED) for _ in range(inflight)]
            t2 = syn.config2_graph_faithful(1 << args.log_rows, 42)
            b2 = [[(k, q.ctx.upload(r), len(r)) for k, r in t2] for q in p2]
            for q, bb in zip(p2, b2):
                q.ctx.prove_tables(bb)
            line["config_2b"] = dict(throughput(p2, b2, 32, 4),
                                     workload="BASELINE config 2b (graph-faithful): Add 2^%d rows + Inputs 2^%d rows, "
                                              "PINNED protocol variant (parity unpinned)" % (args.log_rows, args.log_rows + 1),
                                     prove_latency_ms=solo_latency(p2[0].ctx, b2[0]), proofs_in_flight_per_gpu=len(p2))
            for bb in b2:
                for _, b_, _ in bb:
                    b_.free()
            for q in p2:
                q.ctx.close()
        except Exception as e:
            line["config_2b"] = {"error": str(e)}
    if anchor:
        line["reference_shape_anchor"] = anchor
    if trace_gen:
        line["device_trace_generation"] = trace_gen
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_log, args.log_rows), args.log_rows)
        except Exception as e:  # never lose the headline over the baseline leg
            line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if use_dist and not args.no_extras and os.environ.get("LMN_BENCH_SHARDED_EXTRA", "1") != "0":
        # Sub-result at N > 1: latency of ONE 2^log_rows-row Add proof sharded over all N GPUs (the library's own RCCL
        # communicator on the prover stream), next to the solo latency above.  A watchdog makes sure the headline
        # line is printed even if the multi-GPU transport misbehaves on this node.
        import threading

        def give_up():
            if rank == 0:
                line["sharded_proof"] = {"error": "timed out (watchdog)"}
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("LMN_BENCH_SHARDED_TIMEOUT", "120")), give_up)
        dog.daemon = True
        dog.start()
        res = {}
        try:
            from luminair_amd.sharded import shard_context
            sp = luminair_amd.Prover(dev)
            stabs = syn.config2_add_only(1 << args.log_rows, 42)            # the same table on every rank
            sb = [(k, sp.ctx.upload(r), len(r)) for k, r in stabs]
            want = sp.ctx.prove_tables(sb)
            shard_context(sp.ctx)
            got = sp.ctx.prove_tables(sb)
            el = timed_region(lambda: sp.ctx.prove_tables(sb), 16, 2, barrier, torch.cuda.synchronize)
            tmax = reduce_max(el)
            res = {"workload": "ONE 2^%d-row Add proof sharded into row blocks over %d GPUs" % (args.log_rows, world),
                   "prove_latency_ms": 1e3 * tmax / 16, "solo_unsharded_latency_ms": latency_ms,
                   "bytes_identical_to_unsharded_proof": got == want, "scaling": "strong"}
            sp.ctx.clear_shard()
        except Exception as e:
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
        line["sharded_proof"] = res
    if rank == 0:
        print(json.dumps(line), flush=True)
    pool.shutdown()
    for bl in bufs:
        for _, b, _ in bl:
            if hasattr(b, "free"):
                b.free()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
