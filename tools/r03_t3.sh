set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "norm" > $O/norm_tests.log 2>&1 || { tail -30 $O/norm_tests.log; exit 1; }
tail -3 $O/norm_tests.log
timeout -k 10 200 python tools/norm_bench.py > $O/norm_bench.txt 2>&1 || { tail -20 $O/norm_bench.txt; exit 1; }
cat $O/norm_bench.txt
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae.json 2> $O/bench_utae.err
cut -c1-330 $O/bench_utae.json
timeout -k 10 120 python bench.py --model timeunet --batch 8 --T 61 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/bench_tu.err
cut -c1-300 $O/bench_tu.json
timeout -k 10 300 python tools/tile_bench.py > $O/tile_bench.txt 2>&1 || { tail -20 $O/tile_bench.txt; exit 1; }
cat $O/tile_bench.txt
