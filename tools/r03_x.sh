#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in v1 v2 v3; do echo "== wgrad $v"; C2S_KBENCH_LIB=tools/_diag/libs/libc2s_wgrad_$v.so timeout -k 10 200 python tools/kbench.py --reps 7 --only "3x3 @" 2>&1 | grep -v amdgpu.ids | cut -c1-37,62- ; done
