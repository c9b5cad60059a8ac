set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_base; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/tu.err
cut -c1-300 $O/bench_tu.json
timeout -k 10 200 python tools/ltae_bench.py > $O/ltae_bench.txt 2>&1
cat $O/ltae_bench.txt
