#!/bin/bash
# round 4: fixed-shape FFT kernels (fft_fixed.hip) - selftests at every instantiated size, parity subset, throughput,
# solo latency and stage times; LMN_NO_FFT_FIXED=1 = the generic kernels on the same box for comparison
set -u
OUT=gpurun_out/r5a
mkdir -p $OUT
python - <<'PY' 2>&1 | tail -20
import luminair_amd, time
p = luminair_amd.Prover(0)
for log in (13, 14, 16, 18, 19, 20, 21, 22, 23, 24):
    t = time.time(); p.ctx.fft_selftest(log, 3); print("selftest", log, "ok", round(time.time() - t, 2), flush=True)
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${KEXPR:-kat or full or op_level or sizes or ragged or config or random}" > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | tail -3
for v in fixed generic fixed generic; do
  if [ $v = generic ]; then export LMN_NO_FFT_FIXED=1; else unset LMN_NO_FFT_FIXED; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("quotients_ms","oods_ms","composition_ms","logup_ms","fft_ms","merkle_fused_ms","transpose_ms")}, d.get("roofline_other",{}).get("avg_launch_ms"))
PY
done
unset LMN_NO_FFT_FIXED
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; python -c "
import json; d=json.loads(open('$OUT/driver_cmd.json').read().strip().splitlines()[-1]); print('driver cmd', round(d['value'],1), d.get('errors'))"
