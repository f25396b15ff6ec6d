#!/bin/bash
# round 4: profile set of the final tree (tools/profile_round.sh)
set -u
mkdir -p gpurun_out/r7a
bash tools/profile_round.sh r7a > gpurun_out/r7a/profile_round.log 2>&1; tail -12 gpurun_out/r7a/profile_round.log
