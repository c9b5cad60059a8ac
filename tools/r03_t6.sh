set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "norm" > $O/norm_tests.log 2>&1 || { tail -30 $O/norm_tests.log; exit 1; }
tail -2 $O/norm_tests.log
timeout -k 10 400 python tools/bf16_deviation.py > $O/bf16_deviation.txt 2>&1 || { tail -20 $O/bf16_deviation.txt; exit 1; }
cat $O/bf16_deviation.txt
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae.json 2> $O/bench_utae.err
cut -c1-330 $O/bench_utae.json
