#!/bin/bash
# round-4 measurement set: full GPU suite, smoke, driver command x6, default bench, round profile (kernel stats / PMC /
# VALU / timeline / overlap), config latencies, small-proof batches
set -u
OUT=gpurun_out/r5h
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver rc=$?"
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json, glob
runs = []
for f in sorted(glob.glob("gpurun_out/r5h/driver_cmd_*.json")) + ["gpurun_out/r5h/bench_default.json"]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    runs.append(d)
    sub = {k: round(d[k]["value"], 1) for k in ("host_rows", "host_rows_pinned", "config_2b", "mul_only", "config_3", "short_region") if isinstance(d.get(k), dict) and "value" in d[k]}
    sp = {k: round(v["value"]) for k, v in (d.get("small_proofs") or {}).items() if isinstance(v, dict) and "value" in v}
    print(f.split("/")[-1], round(d["value"], 1), d["steps"], d.get("errors"), sub, sp, "solo", round(d.get("prove_latency_ms"), 3), "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None, "bound", d["roofline"]["bound"])
json.dump(runs[:-1], open("gpurun_out/r5h/driver_cmd_runs.json", "w"))
PY
bash tools/profile_round.sh r4 > $OUT/profile_round.log 2>&1; tail -30 $OUT/profile_round.log
timeout 900 python tools/config_latency.py > $OUT/config_latency.jsonl 2> $OUT/config_latency.err; echo "config rc=$?"; cut -c1-260 $OUT/config_latency.jsonl
timeout 600 python tools/small_proof_batch.py 1 16 64 128 > $OUT/small_proof_batch.jsonl 2> $OUT/small_proof_batch.err; cut -c1-230 $OUT/small_proof_batch.jsonl
