#!/bin/bash
# round-3 numbers on the final code: driver command x5, default bench (full line), config latencies, RCCL probe
set -u
OUT=gpurun_out/r3n
mkdir -p $OUT
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver rc=$?"
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
timeout 600 python tools/config_latency.py > $OUT/config_latency.jsonl 2> $OUT/config_latency.err; echo "config rc=$?"
cat $OUT/config_latency.jsonl | cut -c1-400
# can two RCCL ranks share one GPU?  (weak #9: RCCL with more than one rank has never executed here)
cat > /tmp/rccl_dup.py <<'PY'
import os, sys, torch, torch.distributed as dist
r = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=r, world_size=2)
t = torch.ones(4, device="cuda") * (r + 1)
dist.all_reduce(t); torch.cuda.synchronize()
print("rank", r, "all_reduce ->", t.tolist(), flush=True)
dist.destroy_process_group()
PY
NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 /tmp/rccl_dup.py > $OUT/rccl_dup.log 2>&1; echo "rccl dup rc=$?"; grep -iE "duplicate|all_reduce|error|invalid" $OUT/rccl_dup.log | head -8
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3n/driver_cmd_*.json"))+["gpurun_out/r3n/bench_default.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), "lat", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "errors", d["errors"])
        for k in ("host_rows","config_2b","mul_only","config_3"):
            if k in d: print("   ", k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in d[k].items() if a in ("value","prove_latency_ms","error")})
        if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"), {t:round(v["value"],2) for t,v in d["cpu_baseline"].get("by_threads",{}).items()}, d["cpu_baseline"].get("reference_shape_32x32_add_ms"))
        r=d["roofline"]; print("    roofline", r["kernel"], round(r["frac"],3), "traffic", r["traffic"], "alu", round(r["alu_ceiling"]["frac"],3), [ (o["kernel"], round(o["frac"],3), round(o["alu_ceiling"]["frac"],3), o["traffic"]) for o in d["roofline_other"]])
    except Exception as e:
        print(f, "ERR", e)
PY
