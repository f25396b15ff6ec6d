"""bench.py's `cpu_baseline` leg: the oracle (C/OpenMP kernels driven by oracle/prover.py) timed on this host.

Run as its own process (`python -m oracle.cpu_baseline <sample_log> <full_log>`) so that the OpenMP runtime starts with
the settings below instead of whatever an already loaded torch/libgomp picked: on a 2-socket, 256-thread host the
default (all threads, active spin-waiting between parallel regions) makes a 2^20-row proof take 1.3-29 s, while 64
threads pinned to cores with passive waiting take 0.5-0.6 s (measured on the MI355X box, AMD EPYC 9575F).
Test infrastructure only: prints one JSON object."""
import json
import os
import sys
import time


def numa_policy(threads):
    """Memory placement for a run whose threads span more than one NUMA node.  The proof's arrays are allocated by numpy on
    the main thread and a transform's butterflies pair every row with rows from all over the column: with first-touch
    placement a 128-thread run on two sockets sends most accesses of one socket's threads through the other socket's memory
    controllers (BENCH_r05: 128 threads 2.58 proofs/s against 3.49 on 64).  Pages interleaved over the nodes the threads
    run on spread that load evenly (set_mempolicy(MPOL_INTERLEAVE): the same as `numactl --interleave`).  A run that
    fits one node keeps the default (local) policy."""
    import ctypes
    import glob
    nodes = []
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        try:
            cpus = open(os.path.join(d, "cpulist")).read().strip()
            n_cpu = sum((int(b) - int(a) + 1) if b else 1 for a, _, b in (r.partition("-") for r in cpus.split(",") if r))
            nodes.append((int(os.path.basename(d)[4:]), n_cpu))
        except (OSError, ValueError):
            continue
    if len(nodes) < 2:
        return "default (single NUMA node)"
    try:
        smt = 2 if open("/sys/devices/system/cpu/smt/active").read().strip() == "1" else 1
    except OSError:
        smt = 1
    cores_per_node = max(1, min(n for _, n in nodes) // smt)     # OMP_PLACES=cores: one thread per physical core
    if threads <= cores_per_node:
        return "default (the %d threads fit one of %d NUMA nodes of %d cores)" % (threads, len(nodes), cores_per_node)
    mask = 0
    for nid, _ in nodes:
        mask |= 1 << nid
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        nodemask = ctypes.c_ulong(mask)
        MPOL_INTERLEAVE, SYS_set_mempolicy = 3, 238      # x86_64
        rc = libc.syscall(SYS_set_mempolicy, MPOL_INTERLEAVE, ctypes.byref(nodemask), ctypes.c_ulong(8 * ctypes.sizeof(nodemask) + 1))
        if rc != 0:
            return "default (set_mempolicy failed: errno %d)" % ctypes.get_errno()
    except (OSError, AttributeError) as e:
        return "default (set_mempolicy unavailable: %s)" % e
    return "interleaved over %d NUMA nodes (set_mempolicy MPOL_INTERLEAVE)" % len(nodes)


def main():
    sample_log, full_log = int(sys.argv[1]), int(sys.argv[2])
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    mempolicy = numa_policy(threads)          # before the first large allocation
    from luminair_amd import synthetic as syn
    from oracle.cbackend import CKernels
    from oracle.channel import ProtocolVariant
    from oracle.prover import prove
    K = CKernels()
    tabs = syn.config2_add_only(1 << sample_log, 42)
    t0 = time.perf_counter()
    prove(tabs, kernels=K)
    cold = time.perf_counter() - t0
    warm = []
    for _ in range(3):
        t0 = time.perf_counter()
        prove(tabs, kernels=K)
        warm.append(time.perf_counter() - t0)
    warm = sorted(warm)[1]
    # the same source without lanes / vectoriser (oracle/c/Makefile): what the 16 lanes buy on this host
    from oracle.proof import to_bincode
    KS = CKernels(scalar=True)
    ref = to_bincode(prove(tabs, kernels=K))
    scalar = []
    for i in range(3):
        t0 = time.perf_counter()
        p = prove(tabs, kernels=KS)
        scalar.append(time.perf_counter() - t0)
        if i == 0 and to_bincode(p) != ref:
            raise SystemExit("scalar and 16-lane builds of the oracle disagree")
    scalar = sorted(scalar[1:])[0]
    small = syn.config2_graph_faithful(1024, 42)     # the reference's published shape: 32x32 Add
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        prove(small, kernels=K, variant=ProtocolVariant.PINNED)
        ts.append(1e3 * (time.perf_counter() - t0))
    scale = float(1 << (full_log - sample_log))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    print(json.dumps({
        "value": 1.0 / (warm * scale), "unit": "proofs/s", "cores": threads, "kind": "port", "cpu_model": model,
        "lanes": "16 x M31 per operation in the transforms, Blake2s, logup, FRI quotients and point evaluation (gcc "
                 "target_clones: AVX-512 / AVX2 / SSE2 picked by cpuid), batched field inversions",
        "port_scalar": {"value": 1.0 / (scalar * scale), "unit": "proofs/s", "kind": "port-scalar",
                        "sample": "the same C source built without lanes and with the vectoriser off: %.2f s warm; proof bytes "
                                  "identical to the 16-lane build's" % scalar},
        "host_logical_cpus": os.cpu_count(), "memory_policy": mempolicy,
        "sample": "C/OpenMP oracle proof of a 2^%d-row Add trace: %.2f s warm (median of 3, tables cached), %.2f s cold%s; "
                  "%d OpenMP threads pinned to cores, passive waiting"
                  % (sample_log, warm, cold, "" if scale == 1 else "; scaled x%d to 2^%d rows" % (scale, full_log), threads),
        "reference_shape_32x32_add_ms": sorted(ts)[len(ts) // 2],
        "reference_shape_published_ms": 13.05,
    }))


if __name__ == "__main__":
    main()
