#!/bin/bash
# round 4: single-proof sharding with the all-to-all (SURVEY 8e stages A/B): multi-rank parity on one GPU, then the
# record / replay estimate of rank 0's critical path with and without the all-to-all
set -u
OUT=gpurun_out/r5d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_sharded_prove.py -m gpu -x -q > $OUT/sharded_tests.log 2>&1; tail -3 $OUT/sharded_tests.log
for a2a in 1 0; do
  LMN_REPLAY_A2A=$a2a timeout 900 python tools/shard_replay.py config2a 2 4 8 > $OUT/replay_2a_a2a$a2a.json 2> $OUT/replay_2a_a2a$a2a.err; tail -2 $OUT/replay_2a_a2a$a2a.err
  LMN_REPLAY_A2A=$a2a timeout 1500 python tools/shard_replay.py config5 8 > $OUT/replay_5_a2a$a2a.json 2> $OUT/replay_5_a2a$a2a.err; tail -2 $OUT/replay_5_a2a$a2a.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5d/replay_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "no result", e); continue
    for r in d["replay"]:
        print(f.split("/")[-1], "unsharded", round(d["unsharded_ms"],2), "G", r["world"], "ideal", round(r["rank0_ms_ideal"],2), "pcie", round(r["rank0_ms_pcie"],2),
              "ring", round(r["estimated_latency_ms"],2), "direct", round(r["estimated_latency_direct_links_ms"],2), "recvMB", round(r["exchanged_bytes_received_per_rank"]/1e6,1),
              "a2a", r["all_to_alls_per_proof"], "ag", r["all_gathers_per_proof"], {k:v for k,v in r["rank0_stage_ms"].items() if k in ("fft_ms","logup_ms","main_commit_ms","interaction_commit_ms","fri_ms")})
PY
