#!/bin/bash
# round 6, session c: (1) whole GPU suite on the tree with the device-resident transcript in sharded proofs and sharded
# blow-ups 4 / 8; (2) traffic experiments of VERDICT r5 item 6 as measured upper bounds (ablation build, garbage proofs):
# mask 64 = no transpose launch at all (what feeding the first inverse pass from the transpose's LDS tile could save at
# most), mask 128 = the composition tree's leaf loads served from a 16 KB stand-in (what feeding them from the last forward
# pass's LDS tile could save at most), alternating with mask 0; (3) record / replay of one proof sharded over 2 / 4 / 8 ranks
# on the round-6 tree (tools/shard_replay.py), with the device transcript and with LMN_HOST_FS=1.
set -u
OUT=gpurun_out/r10c
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
cp tools/bin/variants/ablate.so $LIB
for rep in 1 2 3 4; do
for m in 0 64 128 192; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate_traffic.jsonl
done
done
cp /tmp/new.so $LIB
LMN_REPLAY_A2A=1 timeout 900 python tools/shard_replay.py config2a 2 4 8 > $OUT/replay_2a.json 2> $OUT/replay_2a.err; tail -2 $OUT/replay_2a.err
LMN_HOST_FS=1 LMN_REPLAY_A2A=1 timeout 900 python tools/shard_replay.py config2a 8 > $OUT/replay_2a_hostfs.json 2> $OUT/replay_2a_hostfs.err; tail -2 $OUT/replay_2a_hostfs.err
LMN_REPLAY_A2A=1 timeout 1500 python tools/shard_replay.py config5 2 4 8 > $OUT/replay_5.json 2> $OUT/replay_5.err; tail -2 $OUT/replay_5.err
python - <<PY
import json
for n in ("replay_2a","replay_2a_hostfs","replay_5"):
    try:
        txt=[l for l in open("$OUT/%s.json"%n).read().splitlines() if l.startswith("{")]
        for l in txt: print(n, l[:600])
    except Exception as e: print(n,"ERR",e)
PY
