#!/usr/bin/env python3
"""Host-side wall-clock marks of solo proofs (LMN_HOST_PROFILE=1): where the CPU time between GPU syncs goes."""
import os, sys
os.environ["LMN_HOST_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
p = luminair_amd.Prover(0)
tabs = syn.config2_add_only(1 << 20, 42)
bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
for i in range(4):
    sys.stderr.write("---- proof %d\n" % i)
    p.ctx.prove_tables(bufs)
