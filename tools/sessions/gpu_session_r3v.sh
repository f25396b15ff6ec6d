#!/bin/bash
for v in 0 1 2; do LMN_FFT_SPLIT=$v python tools/config5_latency.py 2>/dev/null; done
