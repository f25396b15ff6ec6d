#!/bin/bash
python tools/config4_ablation.py
LMN_LOGUP_SCAN_V1=1 python tools/config4_ablation.py
LMN_NO_FFT_FUSION=1 python tools/config4_ablation.py
LMN_LOGUP_SCAN_V1=1 LMN_NO_FFT_FUSION=1 python tools/config4_ablation.py
