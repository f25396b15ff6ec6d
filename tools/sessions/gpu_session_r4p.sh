#!/bin/bash
# sharded-proof replay (one proof over G ranks) with the round's final kernels
set -u
OUT=gpurun_out
timeout 900 python tools/shard_replay.py config2a 2 4 8 > $OUT/r3_shard_replay_config2a.json 2> $OUT/r3_shard_replay_config2a.err; echo "replay 2a rc=$?"
timeout 1500 python tools/shard_replay.py config5 8 > $OUT/r3_shard_replay_config5.json 2> $OUT/r3_shard_replay_config5.err; echo "replay 5 rc=$?"
python - <<'PY'
import json
for n in ("config2a","config5"):
    try:
        txt=[l for l in open("gpurun_out/r3_shard_replay_%s.json"%n).read().splitlines() if l.startswith("{")][-1]
        d=json.loads(txt)
        print(n, "unsharded", round(d["unsharded_ms"],3))
        for r in d["replay"]:
            print("  G=%d ideal %.3f pcie %.3f est(ring) %.3f est(direct) %.3f  calls %d groups %d MB %.1f" % (r["world"], r["rank0_ms_ideal"], r["rank0_ms_pcie"], r["estimated_latency_ms"], r["estimated_latency_direct_links_ms"], r["all_gathers_per_proof"], r["all_gather_groups_per_proof"], r["gathered_bytes_per_proof"]/1e6))
            print("     ", r["rank0_stage_ms"])
    except Exception as e:
        print(n, "ERR", e)
PY
