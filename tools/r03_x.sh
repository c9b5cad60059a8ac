#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/s2_suite.log 2>&1
echo "test rc=$?" >> gpurun_out/s2_suite.log
tail -4 gpurun_out/s2_suite.log
grep -q "test rc=0" gpurun_out/s2_suite.log || exit 1
timeout -k 10 300 python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timeunet', d['ms_per_step'], d['value'])"
timeout -k 10 300 python bench.py --model wtae --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wtae', d['ms_per_step'], d['value'])"
timeout -k 10 300 python bench.py --batch 8 --T 48 --size 256 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d['value'])"
