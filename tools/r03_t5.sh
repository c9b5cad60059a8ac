set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_t5; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "norm" > $O/norm_tests.log 2>&1 || { tail -30 $O/norm_tests.log; exit 1; }
tail -3 $O/norm_tests.log
timeout -k 10 200 python tools/norm_bench.py > $O/norm_bench_ticket.txt 2>&1 || { tail -20 $O/norm_bench_ticket.txt; exit 1; }
grep "onepass=1" $O/norm_bench_ticket.txt
C2S_NORM_BWD_TICKET=0 timeout -k 10 200 python tools/norm_bench.py > $O/norm_bench_persist.txt 2>&1
grep "onepass=1" $O/norm_bench_persist.txt | head -2
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae.json 2> $O/bench_utae.err
cut -c1-330 $O/bench_utae.json
C2S_NORM_BWD_TICKET=0 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_utae_persist.json 2> $O/bench_utae_persist.err
cut -c1-330 $O/bench_utae_persist.json
timeout -k 10 120 python bench.py --model timeunet --batch 8 --T 61 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/bench_tu.err
cut -c1-300 $O/bench_tu.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_2s.json 2> $O/prof_2s.err
find $O -name '*kernel_trace.csv' -delete
