#!/bin/bash
set -u
mkdir -p gpurun_out/r3t
free -g | head -2
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_table_2_24 or non_default_pcs" ) > gpurun_out/r3t/pytest.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r3t/pytest.log
