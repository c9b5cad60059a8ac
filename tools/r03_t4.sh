set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_abi.py -x -q -m gpu -k "mbconv or abi" > $O/mb_tests.log 2>&1 || { tail -60 $O/mb_tests.log; exit 1; }
tail -3 $O/mb_tests.log
