#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in 0 1; do echo "== S2WINO=$v"; C2S_S2WINO=$v timeout -k 10 200 python tools/kbench.py --reps 9 --only "4x4s2" 2>&1 | grep -v amdgpu.ids | cut -c1-62; done
for i in 1 2 3; do
C2S_S2WINO=0 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s2wino=0', d['ms_per_step'], d['value'])" && \
C2S_S2WINO=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s2wino=1', d['ms_per_step'], d['value'])" || exit 1
done
