#!/bin/bash
# round 4: the declared maximum table size (2^25 rows) end to end, verified by the product verifier
set -u
OUT=gpurun_out/r6d
mkdir -p $OUT
timeout 600 python tools/max_size.py 25 > $OUT/max_size_25.json 2> $OUT/max_size_25.err; echo "rc=$?"; cat $OUT/max_size_25.json; tail -3 $OUT/max_size_25.err
