#!/usr/bin/env python3
"""CPU baseline probe: the oracle's proof of the reference's 32x32 Add shape and of a 2^20-row Add trace on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luminair_amd import synthetic as syn
from oracle.cbackend import CKernels
from oracle.prover import prove
from oracle.channel import ProtocolVariant
K = CKernels()
small = syn.config2_graph_faithful(1024, 42)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); prove(small, kernels=K, variant=ProtocolVariant.PINNED); ts.append(1e3 * (time.perf_counter() - t0))
print("OMP_NUM_THREADS=%s OMP_WAIT_POLICY=%s anchor ms:" % (os.environ.get("OMP_NUM_THREADS"), os.environ.get("OMP_WAIT_POLICY")), " ".join("%.1f" % t for t in ts))
if len(sys.argv) > 1:
    tabs = syn.config2_add_only(1 << 20, 42)
    prove(tabs, kernels=K)
    t0 = time.perf_counter(); prove(tabs, kernels=K); print("   2^20: %.2f s" % (time.perf_counter() - t0))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); prove(small, kernels=K, variant=ProtocolVariant.PINNED); ts.append(1e3 * (time.perf_counter() - t0))
    print("   anchor after the big proof:", " ".join("%.1f" % t for t in ts))
