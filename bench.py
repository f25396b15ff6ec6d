#!/usr/bin/env python3
"""bench.py — proofs/sec of the MI355X prove hot path on BASELINE.json config 2
(single Add-op AIR, 2^20 trace rows per proof), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one BATCH of synthetic input: `--inflight` (default 24; 8 until round 6) independent 2^20-row
traces per GPU, one per prover context, proved concurrently (AoS trace rows resident in HBM -> bincode proof bytes on
the host).  `value` is proofs/s = steps x batch x ranks / time.  (Rounds 1-3 counted one proof per step; the driver's
20-step command then timed 35 ms, a region that starts and ends drained and is dominated by ramp and tail - that
figure is still reported, as the `short_region` sub-result.)
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

from luminair_amd.roofline import HBM_PEAK_GBS  # noqa: E402  (ceilings and the `roofline` objects: luminair_amd/roofline.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed steps; one step = one batch of --inflight proofs per GPU")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-rows", type=int, default=20, help="log2 rows of the Add trace (default 20 = BASELINE config 2)")
    ap.add_argument("--inflight", type=int, default=24,
                    help="independent proofs in flight per GPU (one prover context + HIP stream + ~1.9 GB arena each); "
                         "measured on MI355X at equal numbers of proofs: 12 make +2 %, 24 +3 % more proofs/s than 8, 36 / 48 "
                         "no more, and 9 / 10 / 15 / 20 less than their neighbours 12 / 24 (DESIGN.md section 8)")
    ap.add_argument("--host-rows", action="store_true",
                    help="hand the trace rows over as host buffers (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-anchor", action="store_true", help="skip the 32x32 reference-shape latency anchor")
    ap.add_argument("--cpu-sample-log", type=int, default=20)
    ap.add_argument("--shard-proof", action="store_true",
                    help="latency of ONE proof sharded over the --gpus ranks (lmn_ctx_set_shard_rccl) instead of "
                         "proofs/s of independent proofs; --shard-workload picks the pie")
    ap.add_argument("--shard-workload", default="config5",
                    choices=["config5", "config2a", "config3", "config4", "config_5", "config_2a", "config_3", "config_4"],
                    help="config5: 256 x (Mul + SumReduce + Add), 2^24 rows; config2a: Add 2^log-rows; config3: 2^22 rows; "
                         "config4: the 2->64->64->1 MLP shape with its exp2 LUT")
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--emu-library", default=None,
                    help="TEST ONLY: path of the host-emulation build (tests/emu): exercises the rank launch / barrier / "
                         "aggregation path over gloo on a machine without a GPU; the line says so in `data`")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-results (host_rows, config_2b, sharded_proof) and print the headline only")
    return ap.parse_args(argv)


from luminair_amd.hostinfo import (cpu_budget, cpu_model, gpu_local_cpus, physical_cores, pin_rank_to_gpu_numa_node,  # noqa: E402,F401
                                   rank_cpu_slice, _format_cpulist, _parse_cpulist)


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def timed_region(step_fn, steps, warmup, barrier, device_sync):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides.  The cyclic garbage
    collector is parked for the timed region (as timeit does): a generation-2 pass of the interpreter that happens to
    fall into it holds the GIL for tens of milliseconds, during which no worker thread can hand its next proof to the
    library - measured on the MI355X box as a one-off ~42 ms stall in a 40 ms region, for some (contexts, warmup)
    pairs and not for others."""
    import gc
    for _ in range(warmup):
        step_fn()
    device_sync()
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        barrier()
        device_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        device_sync()
        barrier()
        device_sync()
        return time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()


def aggregate(elapsed, world, steps_per_rank, reduce_max, units_per_step=1):
    """value = units (proofs) all ranks processed / max-over-ranks time."""
    tmax = reduce_max(elapsed)
    return {"seconds": tmax, "value": world * steps_per_rank * units_per_step / tmax,
            "ms_per_step": 1e3 * tmax / steps_per_rank}


def cpu_baseline_run(threads, sample_log, full_log):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="passive", OMP_PROC_BIND="close",
               OMP_PLACES="cores", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", str(sample_log), str(full_log)], cwd=ROOT, env=env,
                         check=True, capture_output=True, text=True, timeout=900)
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_baseline(sample_log, full_log):
    """The oracle's plain-C restatement (oracle/c/stark_kernels.c, OpenMP, driven by oracle/prover.py) timed on this
    box on the same workload, in its own process (oracle/cpu_baseline.py says why): one cold proof (builds the
    twiddle/domain tables, as the reference does per proof), three warm ones, and the reference's own published
    32x32 Add shape.  kind "port" = the build whose hot loops run 16 M31 lanes per operation (AVX-512 / AVX2 by
    cpuid: the stand-in for stwo's SimdBackend); `port_scalar` = the same source without lanes.  Run twice - on ALL physical cores of the host and on 64 pinned threads (one socket's worth;
    the faster one on a 2-socket EPYC in round 2) - `value` is the better of the two, `cores` the threads it used,
    `by_threads` both.  Bounded: a few seconds of CPU work each."""
    phys = physical_cores()
    runs = {}
    for threads in sorted({phys, max(1, min(64, phys))}, reverse=True):
        runs[threads] = cpu_baseline_run(threads, sample_log, full_log)
    best = max(runs, key=lambda t: runs[t]["value"])
    res = dict(runs[best])
    res["host_physical_cores"] = phys
    res["by_threads"] = {str(t): {"value": r["value"], "sample": r["sample"],
                                  "port_scalar_value": r.get("port_scalar", {}).get("value"), "memory_policy": r.get("memory_policy"),
                                  "reference_shape_32x32_add_ms": r["reference_shape_32x32_add_ms"]} for t, r in runs.items()}
    return res


def throughput(provers, bufs, steps, warmup, luts=None):
    """proofs/s of `steps` proofs dealt round-robin over the in-flight contexts (no barrier: sub-results only).
    ONE driver thread per context: proof k runs on context k % n, and a context only ever sees its own thread
    (round 2 dealt the warm-up proofs to a shared worker pool, which let two workers enter the same context at
    once - the source of the `host_rows` ConstraintsNotSatisfied in BENCH_r02.json)."""
    import threading
    n = len(provers)
    start = threading.Barrier(n + 1)
    errors = []

    def drive(i):
        try:
            for _ in range((warmup + n - 1 - i) // n):
                provers[i].ctx.prove_tables(bufs[i], luts)
            start.wait()
            for _ in range((steps + n - 1 - i) // n):
                provers[i].ctx.prove_tables(bufs[i], luts)
        except BaseException as e:  # noqa: BLE001 - reported by the caller
            errors.append(e)
            start.abort()

    import gc
    threads = [threading.Thread(target=drive, args=(i,)) for i in range(n)]
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()            # as in timed_region: no generation-2 pass (GIL held for tens of ms) inside the measurement
    try:
        for t in threads:
            t.start()
        try:
            start.wait()
        except threading.BrokenBarrierError:
            pass
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    if errors:
        raise errors[0]
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
    """(tables, description, protocol variant, LUT settings) of a sharded-proof workload"""
    from luminair_amd import synthetic as syn
    from luminair_amd import backend as _bk
    if name in ("config5", "config_5"):
        return (syn.config5_linear_layers(), "BASELINE config 5: 256 x (Mul + SumReduce + Add), Mul 2^23 + SumReduce 2^23 + Add 2^15 rows",
                _bk.VARIANT_KAT, None)
    if name in ("config3", "config_3"):
        return syn.config3_mixed(), "BASELINE config 3: Add 2^21 + Mul 2^20 + Recip 2^20 rows", _bk.VARIANT_KAT, None
    if name in ("config4", "config_4"):
        tabs4, luts4 = syn.config4_black_scholes_shape()
        return tabs4, "BASELINE config 4 (2->64->64->1 tanh MLP shape, all tables <= 2^13 rows)", _bk.VARIANT_PINNED, luts4
    return syn.config2_add_only(1 << log_rows, 42), "BASELINE config 2a: Add 2^%d rows" % log_rows, _bk.VARIANT_KAT, None


def shard_proof_main(args, rank, local_rank, world):
    """ONE proof sharded over the ranks: every rank holds the same tables, evaluates / hashes /
    folds only its row blocks, the library's own RCCL communicator carries the all-gathers on the prover's stream.
    A step = one whole sharded proof; value = proofs/s of that single stream of proofs (1 / latency).
    Also the child process `bench.py --gpus N` (N > 1) starts per rank for its `sharded_proof` sub-results: a transport
    that hangs or crashes takes this process down, not the one that holds the headline.  Every rank prints ONE JSON line
    (rank 0 the result, the others their verdict)."""
    import hashlib
    import torch
    import torch.distributed as dist
    import luminair_amd
    from luminair_amd import backend as _bk
    from luminair_amd.sharded import shard_context
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    emu = args.emu_library is not None
    has_cuda = torch.cuda.is_available() and not emu
    if has_cuda:
        torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl" if has_cuda else "gloo", rank=rank, world_size=world)
    tdev = "cuda" if has_cuda else "cpu"
    tabs, what, variant, luts = shard_workload(args.shard_workload, args.log_rows)   # identical on every rank (fixed seed)
    prover = luminair_amd.Prover(local_rank if has_cuda else 0, library=_bk.Library(args.emu_library) if emu else None,
                                 protocol_variant=variant)
    bufs = [(k, r if emu else prover.ctx.upload(r), len(r)) for k, r in tabs]
    plain = prover.ctx.prove_tables(bufs, luts)                             # unsharded reference bytes (+ context setup)
    plain_ms = solo_latency(prover.ctx, bufs, 3 if emu else 5, luts)        # unsharded, every rank on its own GPU
    shard_context(prover.ctx)
    out = {}

    def step():
        out["proof"] = prover.ctx.prove_tables(bufs, luts)

    def barrier():
        if has_cuda:
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    steps = 2 if emu else min(args.steps, 32)
    warm = 1 if emu else min(args.warmup, 4)
    elapsed = timed_region(step, steps, warm, barrier, torch.cuda.synchronize if has_cuda else (lambda: None))
    t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tmax = float(t.item())
    same_here = out["proof"] == plain
    same = torch.tensor([1 if same_here else 0], device=tdev)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        rows = sum(n for _, _, n in bufs)
        print(json.dumps({
            "metric": "sharded-proof latency (one proof over all GPUs)", "value": steps / tmax, "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * tmax / steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 (M31/QM31 field arithmetic)", "data": "synthetic",
            "config": {"workload": what + "; ONE proof sharded into row blocks over %d rank(s), RCCL all-gathers of subtree "
                                          "roots / composition evaluations / small FRI layers only" % world,
                       "rows": rows, "parallelism": "single-proof row-block sharding x%d" % world,
                       "proof_bytes": len(out["proof"]), "proof_sha256": hashlib.sha256(out["proof"]).hexdigest()},
            "prove_latency_ms": 1e3 * tmax / steps, "unsharded_latency_ms_rank0": plain_ms,
            "rows_per_s": rows * steps / tmax,
            "bytes_identical_to_unsharded_proof_on_every_rank": bool(int(same.item())),
        }), flush=True)
    else:
        print(json.dumps({"rank": rank, "bytes_identical_to_unsharded_proof": bool(same_here)}), flush=True)
    prover.ctx.clear_shard()
    for _, b, _ in bufs:
        if hasattr(b, "free"):
            b.free()
    prover.ctx.close()
    dist.destroy_process_group()
    return 0 if int(same.item()) else 3   # 3: the sharded proof's bytes differ on some rank (a parity failure)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per
    GPU, RCCL rendezvous on 127.0.0.1) exactly as the driver's torch.distributed.run line does, and pass their
    single JSON line through.  Fails loudly if the box has fewer than N GPUs."""
    import socket
    import subprocess
    if not args.emu_library:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible on this node" % (args.gpus, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(argv if argv is not None else sys.argv[1:])
    return subprocess.run(cmd, env=dict(os.environ, LMN_BENCH_SELF_LAUNCHED="1")).returncode


def finalize_line(line):
    """Key order of the printed line.  The driver keeps the contract keys, `config`, `roofline` and `cpu_baseline` whole and
    only the NAMES of the other keys, and its log keeps the line's tail: the reference's own calling convention (`prove(pie)`
    takes HOST rows, crates/prover/src/prover.rs:28-31,70) is therefore reported twice - as flat keys right behind the
    headline's latency and inside `config.reference_calling_convention` - next to the device-resident headline.  Never `value`."""
    hr = line.get("host_rows") if isinstance(line.get("host_rows"), dict) else {}
    hp = line.get("host_rows_pinned") if isinstance(line.get("host_rows_pinned"), dict) else {}
    tg = line.get("device_trace_generation") if isinstance(line.get("device_trace_generation"), dict) else {}
    flat = {"host_rows_proofs_per_s": hr.get("value"), "host_rows_prove_latency_ms": hr.get("prove_latency_ms"),
            "host_rows_pinned_proofs_per_s": hp.get("value"), "host_rows_pinned_prove_latency_ms": hp.get("prove_latency_ms")}
    if hr or hp:
        line["config"]["reference_calling_convention"] = dict(
            flat, note="trace rows handed over as HOST buffers per proof (60 MiB over PCIe), what prove(pie) of the reference "
                       "takes; the headline `value` has the rows resident in HBM, which is what the recommended binding "
                       "(INTEGRATION.md: process_trace -> lmn_trace_*) produces",
            device_trace_generation_ms=tg.get("ms"))
    out = {}
    for k, v in line.items():
        out[k] = v
        if k == "prove_latency_p95_ms" and (hr or hp):
            out.update(flat)
    return out


def main(argv=None):
    # RCCL prints a version banner on STDOUT when a communicator is created unless told otherwise; the contract is
    # ONE JSON line on stdout
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":   # leave a real debug level alone
        # warnings only, into a per-process file (stdout carries ONE JSON line): dumped to stderr if a multi-GPU step fails
        os.environ["NCCL_DEBUG"] = "WARN"
        if "NCCL_DEBUG_FILE" not in os.environ:
            # a directory of this run's own (ranks started by this process inherit it): a failure report must not pick up
            # another run's - or another user's - files from a predictable path; removed at exit
            import atexit, shutil, tempfile
            run_dir = tempfile.mkdtemp(prefix="lmn_bench_rccl_")
            os.environ["NCCL_DEBUG_FILE"] = os.path.join(run_dir, "rccl.%h.%p.log")
            os.environ["LMN_BENCH_RCCL_LOG_DIR"] = run_dir
            atexit.register(shutil.rmtree, run_dir, True)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if args.shard_proof:
        return shard_proof_main(args, rank, local_rank, world)
    # N ranks x `inflight` polling contexts in one CPU-limited container starve each other's launch threads (a rank held to
    # 2 CPUs loses 10 - 20 %, profiles/r4_host_cpu_wait_policy.jsonl): when the node's CPU budget is below what the default needs - a
    # rank's ~3.4 busy CPUs, the library's waits sleep between polls from the start (LMN_SPIN_US=0: 1.7 CPUs per rank, - 1 %
    # proofs/s where CPUs are plentiful).  Set before the library is loaded; an explicit LMN_SPIN_US wins.
    affinity = pin_rank_to_gpu_numa_node(local_rank, world) if args.emu_library is None else {"pinned": False, "reason": "emulation"}
    budget = cpu_budget()
    spin_set_by_bench = False
    if world > 1 and budget < 4 * world and "LMN_SPIN_US" not in os.environ:
        os.environ["LMN_SPIN_US"] = "0"
        spin_set_by_bench = True
    import numpy as np
    import torch
    import luminair_amd
    from luminair_amd import backend as _bk
    from luminair_amd import synthetic as syn

    emu = args.emu_library is not None            # test of the launch path only (no GPU, gloo, emulation build)
    library = _bk.Library(args.emu_library) if emu else None
    has_cuda = torch.cuda.is_available() and not emu
    tdev = "cuda" if has_cuda else "cpu"
    use_dist = world > 1 or args.force_dist   # --force-dist: the N > 1 code path on one rank (self-test)
    ranks_seen = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if has_cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl" if has_cuda else "gloo", rank=rank, world_size=world)
        # how many ranks the collective backend (RCCL under "nccl") really connects
        ones = torch.ones(1, dtype=torch.int64, device=tdev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit("bench.py: the process group connects %d ranks, expected %d" % (ranks_seen, world))
    dev = local_rank if has_cuda else 0
    errors = []        # every sub-result failure lands here; a non-empty list makes the process exit non-zero

    def mk_prover(**kw):
        return luminair_amd.Prover(dev, library=library, **kw)

    # `inflight` independent prover contexts per GPU (own HIP stream + device arena each); a step is
    # still one whole proof, steps are dealt round-robin to the contexts and run concurrently
    from concurrent.futures import ThreadPoolExecutor
    inflight = 1 if emu else max(1, args.inflight)   # the emulation runtime is single-context
    provers = [mk_prover() for _ in range(inflight)]
    prover = provers[0]
    tabs = syn.config2_add_only(1 << args.log_rows, 42 + rank)   # each rank proves its own trace
    if args.host_rows:
        bufs = [[(k, r, len(r)) for k, r in tabs] for p in provers]             # PCIe-inclusive variant
    else:
        bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]   # trace rows resident in HBM
    out = {}
    # one worker thread per context (a worker pool shared by the contexts could put two threads into one context)
    pools = [ThreadPoolExecutor(max_workers=1) for _ in range(inflight)]
    pending = []

    def one(i):
        out["proof"] = provers[i].ctx.prove_tables(bufs[i])

    counter = {"n": 0}

    def step_one():
        # proof k is queued on context k % inflight; every context works through its own queue on its own thread
        # (ctypes releases the GIL inside lmn_prove), so a slow proof on one context never holds the others back
        i = counter["n"] % inflight
        counter["n"] += 1
        pending.append(pools[i].submit(one, i))

    def step():
        # one step = one batch: a proof for every context.  Batches are queued back to back (no barrier between
        # steps), so the contexts stay busy across step boundaries
        for _ in range(inflight):
            step_one()

    def drain():
        while pending:
            pending.pop(0).result()

    # one-time initialisation for EVERY in-flight context ON ITS OWN WORKER THREAD (twiddle tables, device arena, and
    # the worker's first HIP call, which sets up the runtime's per-thread state), so the W warmup + K timed steps
    # below never include first-use setup whatever W is: with W < inflight some workers are otherwise first used
    # inside the timed region (measured: 20 proofs on 8 contexts took 80 ms instead of 40)
    for f in [pools[i].submit(one, i) for i in range(inflight)]:
        f.result()

    def barrier():
        if use_dist:
            if has_cuda:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def device_sync():
        drain()
        if has_cuda:
            torch.cuda.synchronize()

    def reduce_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = timed_region(step, args.steps, args.warmup, barrier, device_sync)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    # host CPU this rank spent in the warm-up + timed steps (user + system, all threads): what N ranks need from a
    # CPU-limited container (tools/host_cpu_per_proof.py; the library's waits back off to sleeps under load)
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    agg = aggregate(elapsed, world, args.steps, reduce_max, inflight)
    per_rank = [{"rank": rank, "proofs_per_s": args.steps * inflight / elapsed, "affinity": affinity}]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    ms_per_proof = 1e3 / (agg["value"] / world)        # per-GPU time per proof at this throughput
    # the rounds 1-3 form of the driver's command: 20 single proofs after 5, a 30 ms region that starts and ends drained
    short_elapsed = timed_region(step_one, 20, 5, barrier, device_sync)
    short = aggregate(short_elapsed, world, 20, reduce_max)
    # single-proof latency (one proof alone on the GPU) and the per-kernel timings behind `roofline`
    lat = []
    for _ in range(5 if emu else 21):
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
    # applied: 2*FETCH_SIZE + WRITE_SIZE), per launch like `achieved`.  The objects are built in luminair_amd/roofline.py.
    from luminair_amd import roofline as rf
    tm = prover.timings()
    # the committed counter summary is of the 2^20-row workload on the GPU: no `traffic` for anything else
    pmc, pmc_source = rf.load_pmc(ROOT) if (args.log_rows == 20 and not emu) else ({}, None)
    fams = {name: tm[rf.KERNEL_FAMILIES[name][0] + "_ms"] for name in rf.KERNEL_FAMILIES}
    dom = max(fams, key=lambda k: fams[k])
    roofline = rf.kernel_roofline(dom, tm, pmc, pmc_source)
    roofline_other = [rf.kernel_roofline(k, tm, pmc, pmc_source) for k in fams if k != dom]
    counter_total = None
    if pmc:
        try:
            counter_total = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in pmc.values()) / max(
                1, next(v["launches"] for k, v in pmc.items() if k.startswith("k_transpose_pad")))
        except (KeyError, StopIteration):
            counter_total = None
    whole = rf.whole_proof(args.log_rows, ms_per_proof, world, counter_total)

    def sub_result(name, fn):
        try:
            return fn()
        except BaseException as e:  # noqa: BLE001 - the headline is still printed; the failure is made visible
            errors.append("%s: %s: %s" % (name, type(e).__name__, e))
            return {"error": "%s: %s" % (type(e).__name__, e)}

    def variant_throughput(tables, variant, steps, what, n_ctx=None):
        """proofs/s + solo latency of another workload on its own contexts (device-resident rows)"""
        ps = [mk_prover(protocol_variant=variant) for _ in range(n_ctx or min(inflight, 8))]   # (sub-results: as in rounds 2 - 5)
        bs = [[(k, q.ctx.upload(r), len(r)) for k, r in tables] for q in ps]
        try:
            for q, bb in zip(ps, bs):
                q.ctx.prove_tables(bb)
            return dict(throughput(ps, bs, steps, len(ps)), workload=what,
                        prove_latency_ms=solo_latency(ps[0].ctx, bs[0]), proofs_in_flight_per_gpu=len(ps))
        finally:
            for bb in bs:
                for _, b_, _ in bb:
                    b_.free()
            for q in ps:
                q.ctx.close()

    # the reference's own published shape (32x32 Add, 1 024 Add rows + 2 048 Inputs rows,
    # 13.05 ms on a GitHub Actions runner) as a sanity anchor: solo GPU latency, median of 9
    anchor = None
    if rank == 0 and world == 1 and not args.no_anchor:
        def run_anchor():
            ap = mk_prover(protocol_variant=_bk.VARIANT_PINNED)
            atabs = [(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 42)]
            ap.ctx.prove_tables(atabs)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                ap.ctx.prove_tables(atabs)
                ts.append(1e3 * (time.perf_counter() - t0))
            ap.ctx.close()
            return {"workload": "32x32 Add graph: Add 2^10 rows + Inputs 2^11 rows, host rows (PCIe-inclusive)",
                    "gpu_latency_ms": sorted(ts)[4], "reference_published_ms": 13.05,
                    "reference_hardware": "GitHub Actions ubuntu-latest CPU (docs/snippets/benchmark-component.mdx:172)"}
        anchor = sub_result("reference_shape_anchor", run_anchor)

    # the step before the path: `process_trace` of the Add node on device tensors
    trace_gen = None
    if rank == 0 and not emu:
        def run_trace_gen():
            n_rows = 1 << args.log_rows
            rng = np.random.default_rng(7)
            dl = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(np.int32))
            dr = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(np.int32))
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
            return {"workload": "Add node process_trace on device tensors, 2^%d elements -> 15-column rows in HBM"
                                % args.log_rows, "ms": sorted(ts)[2],
                    "reference_published_ms": 0.0959, "reference_workload": "32x32 Add trace generation (BASELINE.md §1)"}
        trace_gen = sub_result("device_trace_generation", run_trace_gen)

    line = {
        "metric": "proofs/sec, 2^%d-row Add trace" % args.log_rows, "value": agg["value"], "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": agg["ms_per_step"],
        "ms_per_proof": ms_per_proof,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (M31/QM31 field arithmetic)",
        "data": ("EMULATION BUILD ON CPU - test of the rank launch path only, not a measurement" if emu else
                 "synthetic" + (" (host rows: PCIe-inclusive)" if args.host_rows else "")),
        "config": {"workload": "BASELINE config 2a: single Add-op AIR, 2^%d trace rows per proof, PcsConfig default "
                               "(pow 5, blowup 2x, 3 queries), KAT protocol variant (the variant the reference's only "
                               "known-answer proof pins; Add's constraint forms are KAT-pinned, Mul's second "
                               "eval_fixed_mul slot and the Recip/Sqrt/Rem forms are unpinned and not used here)"
                               % args.log_rows,
                   "rows": 1 << args.log_rows, "step": "one batch of %d independent proofs per GPU (one per prover context)" % inflight,
                   "proofs_per_step_per_gpu": inflight, "proofs_per_rank": args.steps * inflight,
                   "parallelism": "proof-sharded x%d" % world,
                   "proofs_in_flight_per_gpu": inflight, "ranks_in_process_group": ranks_seen,
                   "collective_backend": ("nccl (RCCL)" if has_cuda else "gloo") if use_dist else None,
                   "per_rank_proofs_per_s": {"min": min(r["proofs_per_s"] for r in per_rank), "max": max(r["proofs_per_s"] for r in per_rank),
                                             "all": [round(r["proofs_per_s"], 1) for r in per_rank]},
                   "cpu_affinity": [r["affinity"] for r in per_rank],
                   "host_cpu_budget": round(budget, 1), "host_wait_spin_us": os.environ.get("LMN_SPIN_US", "3000 (default; 100 while several proofs are in flight)"),
                   "host_wait_policy_set_by_bench": spin_set_by_bench,   # True: fewer than 4 CPUs per rank, waits sleep between polls
                   "ms_per_step_is": "per batch of %d proofs (rounds 1-3: per proof; compare `ms_per_proof` / `short_region`)" % inflight,
                   "proof_bytes": len(out["proof"])},
        "prove_latency_ms": latency_ms,
        "prove_latency_p95_ms": latency_p95_ms,
        "host_cpu_ms_per_proof": round(1e3 * host_cpu_s / max(1, (args.steps + args.warmup) * inflight), 3),
        "stage_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        "roofline": roofline,
        "roofline_other": roofline_other,
        "whole_proof_vs_traffic_model": whole,
        "short_region": {"value": short["value"], "unit": "proofs/s", "steps": 20, "warmup": 5,
                         "note": "20 single proofs after 5, timed like the headline: the rounds 1-3 reading of the driver's "
                                 "command (a ~30 ms region that starts and ends drained: ramp and tail inside)"},
    }
    if rank == 0 and not args.no_extras and not args.host_rows and not emu:
        # sub-results next to the headline: the same workload with the trace rows handed over as host buffers (the
        # reference API takes a host pie: 60 MiB over PCIe per proof); config 2b = what gen_trace really emits for an
        # Add node at HEAD (Add 2^20 rows consumed with multiplicity -1 + the Inputs table of 2^21 rows; PINNED
        # protocol variant - parity unpinned); a Mul-only trace of the same size (the metric says
        # "Add/Mul"); BASELINE config 3 (Add 2^21 + Mul 2^20 + Recip 2^20 rows, three components in one commitment)
        sub_provers = provers[:8]     # the sub-results keep the 8 contexts of rounds 2 - 5 (comparable figures; 24 threads copying
        # from pageable memory at once made 136 - 564 proofs/s from run to run, gpu_session_r10p)

        def run_host_rows():
            hb = [[(k, r, len(r)) for k, r in tabs] for _ in sub_provers]
            # 192 proofs after 16: the first batch uploads 8 x 60 MiB before any proof can start (a 9 ms ramp, 10 % of a
            # 48-proof region) and some boxes bring the PCIe link up to speed only under sustained traffic
            return dict(throughput(sub_provers, hb, 192, 16), note="trace rows as host buffers: PCIe-inclusive", proofs_in_flight_per_gpu=len(sub_provers),
                        prove_latency_ms=solo_latency(prover.ctx, hb[0]))
        line["host_rows"] = sub_result("host_rows", run_host_rows)

        def run_host_rows_pinned():
            # the same, with the rows in page-locked host memory (lmn_host_alloc: what a binding's `flatten` would write
            # into - include/luminair_hip.h): the GPU's DMA engines fetch them directly
            lib = provers[0].ctx.lib
            pins = []
            try:
                for k, r in tabs:
                    a = lib.host_rows(r.shape, r.dtype)
                    a.array[...] = r
                    pins.append((k, a))
                hb = [[(k, a.array, len(a.array)) for k, a in pins] for _ in sub_provers]
                return dict(throughput(sub_provers, hb, 192, 16),
                            note="trace rows in page-locked host memory (lmn_host_alloc): PCIe-inclusive, direct DMA",
                            proofs_in_flight_per_gpu=len(sub_provers),
                            prove_latency_ms=solo_latency(prover.ctx, hb[0]))
            finally:
                for _, a in pins:
                    a.free()
        line["host_rows_pinned"] = sub_result("host_rows_pinned", run_host_rows_pinned)
        lr = args.log_rows
        line["config_2b"] = sub_result("config_2b", lambda: variant_throughput(
            syn.config2_graph_faithful(1 << lr, 42), _bk.VARIANT_PINNED, 32,
            "BASELINE config 2b (graph-faithful): Add 2^%d rows + Inputs 2^%d rows, PINNED protocol variant (parity "
            "unpinned)" % (lr, lr + 1)))
        line["mul_only"] = sub_result("mul_only", lambda: variant_throughput(
            syn.config2_mul_only(1 << lr, 42), _bk.VARIANT_KAT, 48,
            "single Mul-op AIR, 2^%d trace rows, all multiplicities 0, KAT variant (16 columns; rem != 0, so Mul's "
            "second eval_fixed_mul slot - restated as a zero slot - is unpinned)" % lr))
        line["config_3"] = sub_result("config_3", lambda: variant_throughput(
            syn.config3_mixed(lr + 1, lr, lr), _bk.VARIANT_KAT, 16,
            "BASELINE config 3: Add 2^%d + Mul 2^%d + Recip 2^%d rows in one pie (three components, mixed-size trees; "
            "Recip's constraint form unpinned)" % (lr + 1, lr, lr), n_ctx=min(inflight, 4)))
    if rank == 0 and world == 1 and not args.no_extras and not args.host_rows and not emu:
        # small proofs (the reference's own benchmark shape, BASELINE config 4): lock-step batches (lmn_batch_prove,
        # libluminair_hip_batch.so) next to concurrent contexts of the main library
        def run_small():
            from luminair_amd.batch import BatchProver
            res = {}
            tabs4, luts4 = syn.config4_black_scholes_shape()
            # (G = batch groups driven at once in `concurrent_groups`: 3 pay on the host-bound 32x32 shape - 30 against 21 k
            # proofs/s; config 4's batches are GPU-bound: 1 / 2 / 3 groups = 1 525 / 1 580 / 1 340, tools/small_proof_groups.py)
            # PB = slots per group there: 3 x 64 = 30, 3 x 128 = 36, 3 x 192 or 256 = 40 k proofs/s on the 32x32 shape)
            for name, mk, luts, B, G, PB in (("reference_shape_32x32_add", lambda i: syn.config2_graph_faithful(1024, 100 + i), None, 64, 3, 192),
                                             ("config_4", lambda i: tabs4, luts4, 16, 2, 16)):
                pies = [[(k, r, len(r)) for k, r in mk(i)] for i in range(B)]
                sp = mk_prover(protocol_variant=_bk.VARIANT_PINNED)
                want = sp.ctx.prove_tables(pies[0], luts)
                solo_ms = solo_latency(sp.ctx, pies[0], 9, luts)
                sp.ctx.close()
                bp = BatchProver(dev, B, protocol_variant=_bk.VARIANT_PINNED)
                try:
                    got = bp.prove_batch(pies, luts)
                    if got[0] != want:
                        raise RuntimeError("batched proof bytes differ from lmn_prove")
                    for _ in range(2):
                        bp.prove_batch(pies, luts)
                    reps = max(4, 512 // B)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        bp.prove_batch(pies, luts)
                    dt = time.perf_counter() - t0
                finally:
                    bp.close()
                res[name] = {"value": B * reps / dt, "unit": "proofs/s", "proofs_per_batch": B, "ms_per_batch": 1e3 * dt / reps,
                             "solo_lmn_prove_latency_ms": solo_ms, "bytes_identical_to_lmn_prove": True}
                # several groups at once (BatchPool): the host code of one group's members overlaps the launches of another's
                from luminair_amd.batch import BatchPool
                pool = BatchPool(dev, G, PB, protocol_variant=_bk.VARIANT_PINNED)
                try:
                    many = (pies + [[(k, r, len(r)) for k, r in mk(i)] for i in range(B, PB)]) * (G * 4)
                    got = pool.prove_many(many, luts)
                    if got[0] != want or got[-PB] != want:
                        raise RuntimeError("proof bytes from concurrent batch groups differ from lmn_prove")
                    t0 = time.perf_counter()
                    pool.prove_many(many, luts)
                    dt = time.perf_counter() - t0
                finally:
                    pool.close()
                res[name]["concurrent_groups"] = {"value": len(many) / dt, "unit": "proofs/s", "groups": G, "proofs_per_batch": PB,
                                                  "bytes_identical_to_lmn_prove": True}
            res["note"] = ("lmn_batch_prove: B pies of identical shape in lock-step, one launch per pipeline step for the whole "
                           "batch (host rows, PINNED variant); `concurrent_groups`: several such groups driven at once (BatchPool); config 4 = 2->64->64->1 tanh MLP shape with its 2^17-row exp2 LUT")
            return res
        line["small_proofs"] = sub_result("small_proofs", run_small)
    if anchor:
        line["reference_shape_anchor"] = anchor
    if trace_gen:
        line["device_trace_generation"] = trace_gen
    if use_dist and not args.no_extras and os.environ.get("LMN_BENCH_SHARDED_EXTRA", "1") != "0":
        # Sub-results at N > 1: latency of ONE proof sharded over all N GPUs (the library's own RCCL communicator on the
        # prover stream), next to its unsharded latency.  Every rank runs them in a CHILD process (`bench.py --shard-proof`,
        # its own rendezvous on a fresh port): RCCL with more than one rank first runs on the driver's node, and a transport
        # that hangs or crashes there must not take the process that holds the headline with it.  A child that does not
        # finish within LMN_BENCH_SHARDED_TIMEOUT (300 s) is killed; the line says what happened in `warnings`, a proof
        # whose BYTES differ is an error (non-zero exit).  Still ONE JSON line on stdout.
        import socket
        import subprocess

        def dump_rccl_logs():
            import glob
            log_dir = os.environ.get("LMN_BENCH_RCCL_LOG_DIR")   # only this run's files (main() made the directory)
            for pth in sorted(glob.glob(os.path.join(log_dir, "rccl.*.log"))) if log_dir else []:
                try:
                    txt = open(pth).read().strip()
                except OSError:
                    continue
                if txt:
                    sys.stderr.write("---- RCCL warnings (%s)\n%s\n" % (pth, txt[-4000:]))

        def sharded_workloads():
            """config 2a always; the config BASELINE.json names for this GPU count next to it: config 4 (black-scholes
            MLP shape, every table <= 2^13 rows: latency-bound, sharding is expected to LOSE and the line says so) at
            4 GPUs, config 5 (2^24 rows) at 8"""
            w = [("config_2a", "BASELINE config 2a (Add 2^%d rows)" % args.log_rows, 16)]
            if emu:
                return w
            if world == 4:
                w.append(("config_4", "BASELINE config 4 (2->64->64->1 tanh MLP shape, all tables <= 2^13 rows)", 16))
            if world == 8:
                w.append(("config_5", "BASELINE config 5 (256 x (Mul + SumReduce + Add), 2^24 rows)", 4))
            return w

        def run_sharded_child(name, what, n_sh):
            port = [0]
            if rank == 0:
                sock = socket.socket()
                sock.bind(("127.0.0.1", 0))
                port[0] = sock.getsockname()[1]
                sock.close()
            dist.broadcast_object_list(port, src=0)
            env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "GROUP_", "ROLE_"))}
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), RANK=str(rank), LOCAL_RANK=str(local_rank),
                       WORLD_SIZE=str(world), LMN_BENCH_AFFINITY="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--shard-proof", "--shard-workload", name,
                   "--log-rows", str(args.log_rows), "--steps", str(n_sh), "--warmup", "2"]
            if emu:
                cmd += ["--emu-library", args.emu_library]
            limit = float(os.environ.get("LMN_BENCH_SHARDED_TIMEOUT", "300"))
            verdict = {"rc": None, "timed_out": False, "stdout": "", "stderr": ""}
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
                verdict.update(rc=r.returncode, stdout=r.stdout, stderr=r.stderr)
            except subprocess.TimeoutExpired as e:
                verdict.update(timed_out=True, stdout=(e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                               stderr=(e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))
            # every rank's verdict reaches rank 0 (the children of the other ranks print only their own)
            verdicts = [None] * world
            dist.all_gather_object(verdicts, {"rank": rank, "rc": verdict["rc"], "timed_out": verdict["timed_out"],
                                              "stderr_tail": verdict["stderr"][-1500:]})
            if rank != 0:
                return None
            bad = [v for v in verdicts if v["timed_out"] or v["rc"] != 0]
            if any(v["rc"] == 3 for v in verdicts):      # shard_proof_main: different bytes on some rank
                raise RuntimeError("sharded proof bytes differ from the unsharded proof on some rank")
            if bad:
                dump_rccl_logs()
                why = "timed out after %.0f s (killed)" % limit if any(v["timed_out"] for v in bad) else \
                    "child exit codes %s: %s" % ([v["rc"] for v in verdicts], bad[0]["stderr_tail"].strip().splitlines()[-1:] or "")
                line.setdefault("warnings", []).append("sharded_proof.%s: %s" % (name, why))
                return {"error": why}
            res = json.loads(verdict["stdout"].strip().splitlines()[-1])
            if not res.get("bytes_identical_to_unsharded_proof_on_every_rank"):
                raise RuntimeError("sharded proof bytes differ from the unsharded proof on some rank")
            sharded_ms, solo = res["prove_latency_ms"], res["unsharded_latency_ms_rank0"]
            return {"workload": "ONE proof of %s sharded into row blocks over %d GPUs" % (what, world),
                    "prove_latency_ms": sharded_ms, "solo_unsharded_latency_ms": solo,
                    "speedup_over_one_gpu": solo / sharded_ms, "sharding_wins": bool(sharded_ms < solo),
                    "bytes_identical_to_unsharded_proof": True, "scaling": "strong",
                    "isolated_in_child_process": True, "proof_sha256": res["config"]["proof_sha256"]}
        line["sharded_proof"] = {}
        for name, what, n_sh in sharded_workloads():
            got = sub_result("sharded_proof." + name, lambda: run_sharded_child(name, what, n_sh))
            if rank == 0:
                line["sharded_proof"][name] = got
    if rank == 0 and not args.no_cpu_baseline and not emu:   # also on N > 1 lines: rank 0 times it once, after its GPU work
        line["cpu_baseline"] = sub_result("cpu_baseline",
                                          lambda: cpu_baseline(min(args.cpu_sample_log, args.log_rows), args.log_rows))
    # a failure on any rank must reach rank 0's line and every rank's exit status
    n_err = len(errors)
    if use_dist:
        t = torch.tensor([n_err], dtype=torch.int64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t.item()) and not n_err:
            errors.append("a sub-result failed on another rank")
    line["errors"] = errors
    if rank == 0:
        print(json.dumps(finalize_line(line)), flush=True)
    for pl in pools:
        pl.shutdown()
    for bl in bufs:
        for _, b, _ in bl:
            if hasattr(b, "free"):
                b.free()
    if use_dist:
        dist.destroy_process_group()
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
