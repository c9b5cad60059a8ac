set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t1; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "norm" > $O/norm_tests.log 2>&1 || { tail -30 $O/norm_tests.log; exit 1; }
tail -3 $O/norm_tests.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae.json 2> $O/bench_utae.err
cut -c1-330 $O/bench_utae.json
C2S_NORM_ONEPASS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae_2pass.json 2> $O/bench_utae_2pass.err
cut -c1-330 $O/bench_utae_2pass.json
timeout -k 10 120 python bench.py --model timeunet --batch 8 --T 61 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/bench_tu.err
cut -c1-300 $O/bench_tu.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_1s.json 2> $O/prof_1s.err
find $O -name '*kernel_trace.csv' -size +20M -delete
