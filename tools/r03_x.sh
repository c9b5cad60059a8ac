#!/bin/bash
# scratch: wide Winograd kernel A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wide_winograd" > gpurun_out/w16_test.log 2>&1
echo "test rc=$?" >> gpurun_out/w16_test.log
tail -5 gpurun_out/w16_test.log
grep -q "test rc=0" gpurun_out/w16_test.log || exit 1
timeout -k 10 300 python tools/wino16_diag.py 0 1 2 > gpurun_out/w16_diag.log 2>&1
cat gpurun_out/w16_diag.log
