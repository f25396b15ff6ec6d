import sys, hashlib, threading, time
sys.path.insert(0, ".")
import numpy as np, luminair_amd
from luminair_amd import synthetic as syn, backend
tabs = syn.config2_add_only(1 << 20, 42)
N_CTX = int(sys.argv[1]) if len(sys.argv) > 1 else 4          # contexts = proofs in flight
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 2400 // N_CTX
provers = [luminair_amd.Prover(0) for _ in range(N_CTX)]
bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]
ref = hashlib.sha256(provers[0].ctx.prove_tables(bufs[0])).hexdigest()
bad = []
def work(i):
    for it in range(ITERS):
        h = hashlib.sha256(provers[i].ctx.prove_tables(bufs[i])).hexdigest()
        if h != ref: bad.append((i, it, h))
t0 = time.time()
ths = [threading.Thread(target=work, args=(i,)) for i in range(N_CTX)]
[t.start() for t in ths]; [t.join() for t in ths]
print("%d concurrent proofs on %d contexts in %.1f s (%.0f proofs/s), mismatches: %d" % (N_CTX * ITERS, N_CTX, time.time() - t0, N_CTX * ITERS / (time.time() - t0), len(bad)), ref[:16])
# mixed workloads concurrently: LUT graph (PINNED) + chain + linear layer, each checked against its own first result
import itertools
jobs = [("chain", syn.chain_graph(30000, 3), None, backend.VARIANT_KAT), ("linear", syn.linear_layer(200, 300, 4, True), None, backend.VARIANT_KAT)]
t, l = syn.activation_graph(20000, 5); jobs.append(("lut", t, l, backend.VARIANT_PINNED))
t, l = syn.config4_black_scholes_shape(); jobs.append(("cfg4", t, l, backend.VARIANT_PINNED))
bad2 = []
def work2(name, tabs, luts, variant):
    p = luminair_amd.Prover(0, protocol_variant=variant)
    b = [(k, r, len(r)) for k, r in tabs]
    first = p.ctx.prove_tables(b, luts)
    for it in range(150):
        if p.ctx.prove_tables(b, luts) != first: bad2.append((name, it))
ths = [threading.Thread(target=work2, args=j) for j in jobs]
[t.start() for t in ths]; [t.join() for t in ths]
print("mixed concurrent workloads: mismatches", len(bad2))
