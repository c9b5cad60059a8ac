#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 tools/_diag/mfma_rate > gpurun_out/r03_mfma_rate.txt 2>&1
timeout -k 10 400 python tools/wino16_diag.py 0 --shapes > gpurun_out/r03_wino16_diag.txt 2>&1 && \
timeout -k 10 400 python tools/wino16_diag.py 1 3 5 7 10 >> gpurun_out/r03_wino16_diag.txt 2>&1 && \
timeout -k 10 100 python tools/_diag/clock_probe.py > gpurun_out/r03_clock_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_wino16_diag.txt | tail -40
grep "per launch" gpurun_out/r03_clock_probe.txt
