#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/s2_suite.log 2>&1
echo "test rc=$?" >> gpurun_out/s2_suite.log
tail -4 gpurun_out/s2_suite.log
grep -q "test rc=0" gpurun_out/s2_suite.log || exit 1
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], d['loss'])" || exit 1
done
C2S_S2WINO=0 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench s2wino=0', d['ms_per_step'], d['value'], d['loss'])"
