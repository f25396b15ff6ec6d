#!/usr/bin/env python3
"""Solo duration of the big Merkle launches: commit of C columns x 2^L rows on device handles, timed per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from luminair_amd import backend
ctx = backend.Context(0)
rng = np.random.default_rng(1)
for ncols, log in ((15, 21), (12, 21), (4, 22)):
    h = ctx.col_from_cpu(rng.integers(0, (1 << 31) - 1, size=(ncols, 1 << log), dtype=np.uint64).astype(np.uint32))
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); t = ctx.commit([h]); ts.append(1e6 * (time.perf_counter() - t0)); t.free()
    ncomp = (2 << log)
    print("LMN_MERKLE_SUB=%s  %2d cols x 2^%d: commit %.1f us (median, incl. tree top + copies)  -> %.1f G compressions/s" %
          (os.environ.get("LMN_MERKLE_SUB"), ncols, log, sorted(ts)[6], ncomp / sorted(ts)[6] / 1e3))
    h.free()
