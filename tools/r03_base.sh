set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_base; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae.json 2> $O/bench_utae.err
cat $O/bench_utae.json | cut -c1-400
python bench.py --model timeunet --batch 8 --T 61 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/bench_tu.err
cat $O/bench_tu.json | cut -c1-300
C2S_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae_1s.json 2> $O/bench_utae_1s.err
cat $O/bench_utae_1s.json | cut -c1-300
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_1s.json 2> $O/prof_1s.err
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tu_1s -- python bench.py --model timeunet --batch 8 --T 61 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_tu_1s.json 2> $O/prof_tu_1s.err
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O $O/prof_1s/* | head -30
