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


def main():
    sample_log, full_log = int(sys.argv[1]), int(sys.argv[2])
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
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
        "host_logical_cpus": os.cpu_count(),
        "sample": "C/OpenMP oracle proof of a 2^%d-row Add trace: %.2f s warm (median of 3, tables cached), %.2f s cold%s; "
                  "%d OpenMP threads pinned to cores, passive waiting"
                  % (sample_log, warm, cold, "" if scale == 1 else "; scaled x%d to 2^%d rows" % (scale, full_log), threads),
        "reference_shape_32x32_add_ms": sorted(ts)[len(ts) // 2],
        "reference_shape_published_ms": 13.05,
    }))


if __name__ == "__main__":
    main()
