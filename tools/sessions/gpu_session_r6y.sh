#!/bin/bash
# round 4: twiddle tables built on the device and shared by the contexts of a process: parity, time to the first proof
# (previous library = tools/bin/variants/prev.so), throughput unchanged
set -u
OUT=gpurun_out/r6y
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size or op_level or ragged or proof_equals_oracle or thresholds or maximum or config5 or 2_24" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
cp luminair_amd/csrc/libluminair_hip.so /tmp/new.so
for v in new prev; do
  if [ $v = prev ]; then cp tools/bin/variants/prev.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so; fi
  for lg in 12 20 24; do timeout 300 python tools/first_proof.py $lg 2>> $OUT/err.log | sed "s/^{/{\"lib\": \"$v\", /" | tee -a $OUT/first_proof.jsonl; done
done
cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so
timeout 600 python tools/max_size.py 25 2>> $OUT/err.log | tee $OUT/max_size_25.json
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', round(d['value'],1), 'short', round(d['short_region']['value'],1), 'solo', round(d['prove_latency_ms'],3))"
