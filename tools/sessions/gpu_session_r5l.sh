#!/bin/bash
# round 4: row-parallel front end of sharded proofs: multi-rank parity on one GPU, replay estimate for config 5 / 2a at G = 8
set -u
OUT=gpurun_out/r5l
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_sharded_prove.py -m gpu -x -q > $OUT/sharded_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/sharded_tests.log | tail -2
LMN_REPLAY_A2A=1 timeout 1500 python tools/shard_replay.py config5 8 > $OUT/replay_5.json 2> $OUT/replay_5.err; tail -2 $OUT/replay_5.err | grep -v amdgpu
LMN_REPLAY_A2A=1 LMN_SHARD_ROWS_MIN_LOG=20 timeout 900 python tools/shard_replay.py config2a 8 > $OUT/replay_2a_rows.json 2> $OUT/replay_2a_rows.err
LMN_REPLAY_A2A=1 timeout 1500 python tools/shard_replay.py config5 2 4 > $OUT/replay_5_g24.json 2> $OUT/replay_5_g24.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5l/replay_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "no result", e); continue
    for r in d["replay"]:
        print(f.split("/")[-1], "unsharded", round(d["unsharded_ms"],2), "G", r["world"], "ideal", round(r["rank0_ms_ideal"],2), "pcie", round(r["rank0_ms_pcie"],2),
              "ring", round(r["estimated_latency_ms"],2), "direct", round(r["estimated_latency_direct_links_ms"],2), "recvMB", round(r["exchanged_bytes_received_per_rank"]/1e6,1),
              "a2a", r["all_to_alls_per_proof"], "ag", r["all_gathers_per_proof"], {k:v for k,v in r["rank0_stage_ms"].items() if k in ("transpose_ms","logup_ms","main_commit_ms","interaction_commit_ms","fft_ms","fri_ms","composition_ms")})
PY
