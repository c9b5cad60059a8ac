set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "squeeze" > $O/se_tests.log 2>&1 || { tail -40 $O/se_tests.log; exit 1; }
tail -3 $O/se_tests.log
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_abi.py -x -q -m gpu > $O/model_tests.log 2>&1 || { tail -60 $O/model_tests.log; exit 1; }
tail -3 $O/model_tests.log
