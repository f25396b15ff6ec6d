#!/bin/bash
python tools/small_proof_throughput.py 1 4 8 16 32 2>/dev/null
GPU_MAX_HW_QUEUES=16 python tools/small_proof_throughput.py 8 16 32 2>/dev/null
