set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 C2S_BENCH_FORCE_DIST=1 python tests/dist_worker.py 2>&1 | grep -E "DIST_|Error|assert" | head
