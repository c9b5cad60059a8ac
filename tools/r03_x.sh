#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
C2S_WINO16=0 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino16=0', d['ms_per_step'], d['value'])" && \
C2S_WINO16=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino16=1', d['ms_per_step'], d['value'])" || exit 1
done
C2S_WINO16=0 timeout -k 10 300 python bench.py --model timeunet --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timeunet wino16=0', d['ms_per_step'], d['value'])"
C2S_WINO16=1 timeout -k 10 300 python bench.py --model timeunet --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timeunet wino16=1', d['ms_per_step'], d['value'])"
