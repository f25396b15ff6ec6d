#!/bin/bash
set -u
mkdir -p gpurun_out/r3u
MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 600 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3u/force_dist.json 2> gpurun_out/r3u/force_dist.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3u/force_dist.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["n_gpus"], d["config"]["ranks_in_process_group"], d["config"]["collective_backend"], d["errors"], d.get("sharded_proof"))
PY
tail -3 gpurun_out/r3u/force_dist.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 1 --steps 8 --warmup 2 --shard-proof --shard-workload config2a > gpurun_out/r3u/shard_proof_n1.json 2> gpurun_out/r3u/shard_proof_n1.err; echo "shard-proof rc=$?"; tail -1 gpurun_out/r3u/shard_proof_n1.json | cut -c1-600
