set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/gx; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_ltae_paths_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "ltae or timeunet or wtae" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -3 $O/t.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -- python tools/ltae_bench.py --no-attn --reps 3 > $O/b1.txt 2>&1
for f in $O/s1/*/*kernel_stats.csv; do echo $f; grep -i "ltae\|reduce\|gwc" $f | cut -d, -f1-4 | cut -c1-150; done
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/tu.err
python -c "import json;print('timeunet', json.load(open('$O/bench_tu.json'))['ms_per_step'])"
