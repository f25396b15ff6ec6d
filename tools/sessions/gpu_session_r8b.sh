#!/bin/bash
# round 5, session b: protocol flags (ABI 6) on the GPU: whole GPU suite (PINNED now includes the prefixed proof of work),
# the flag-search tests through the HIP library, tools/pin_variant.py with --tables on the device
set -u
OUT=gpurun_out/r8b
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
python - <<'PY' > gpurun_out/r8b/pin_variant_gpu.txt 2>&1
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from luminair_amd import backend as B, synthetic as syn
import luminair_amd
flags = B.PV_CLAIM17 | B.PV_MIX_U64_HASHED | B.PV_POW_PREFIXED | B.PV_SQRT_NEG | B.PV_REM_TWO_SLOTS
tabs = syn.sqrt_rem_graph(5000, 4)
p = luminair_amd.Prover(0, protocol_variant=flags)
proof = p.prove(luminair_amd.LuminairPie.from_tables(tabs)).to_bincode()
d = tempfile.mkdtemp()
open(os.path.join(d, "proof.bin"), "wb").write(proof)
for k, rows in tabs:
    rows.astype("<u4").tofile(os.path.join(d, "table_%d.bin" % k))
r = subprocess.run([sys.executable, "tools/pin_variant.py", os.path.join(d, "proof.bin"), "--tables", d], capture_output=True, text=True)
print(r.stdout, r.stderr, "exit", r.returncode)
PY
tail -12 $OUT/pin_variant_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; python -c "
import json; d=json.loads(open('$OUT/driver_cmd.json').read().strip().splitlines()[-1]); print(d['value'], d['prove_latency_ms'])"
