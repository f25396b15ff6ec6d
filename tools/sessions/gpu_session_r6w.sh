#!/bin/bash
# round 4: where the host CPU of a proof goes: syscall counts (strace -c -f) and a per-thread CPU breakdown
set -u
OUT=gpurun_out/r6w
mkdir -p $OUT
which strace perf ltrace 2>&1 | head -3
if which strace > /dev/null 2>&1; then
  timeout 300 strace -c -f -o $OUT/strace_summary.txt python tools/host_cpu_per_proof.py 8 192 > $OUT/under_strace.json 2> $OUT/strace.err
  head -25 $OUT/strace_summary.txt
fi
python - <<'PY'
import json, os, sys, time, threading, subprocess
sys.path.insert(0, os.getcwd())
import luminair_amd
from luminair_amd import synthetic as syn
import bench
tabs = syn.config2_add_only(1 << 20, 42)
provers = [luminair_amd.Prover(0) for _ in range(8)]
bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]
for p, b in zip(provers, bufs): p.ctx.prove_tables(b)
def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
            name = open("/proc/self/task/%s/comm" % t).read().strip()
            out[t] = (name, int(f[11]), int(f[12]))   # utime, stime in ticks
        except Exception:
            pass
    return out
bench.throughput(provers, bufs, 32, 8)
a = snap(); t0 = time.perf_counter()
r = bench.throughput(provers, bufs, 384, 0)
b = snap(); dt = time.perf_counter() - t0
tick = os.sysconf("SC_CLK_TCK")
rows = []
for t, (name, u, s) in b.items():
    u0, s0 = (a[t][1], a[t][2]) if t in a else (0, 0)
    rows.append((name, (u - u0) / tick, (s - s0) / tick))
rows.sort(key=lambda x: -(x[1] + x[2]))
print("proofs/s %.1f, wall %.2f s" % (r["value"], dt))
for name, u, s in rows[:16]: print("%-18s user %.2f s  sys %.2f s" % (name, u, s))
PY
