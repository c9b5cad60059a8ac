set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "ltae or norm or wide_winograd or train_step or absrel or timeunet" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python tools/ltae_bench.py > $O/ltae_bench.txt 2>&1
timeout -k 10 200 python tools/ltae_bench.py --no-attn > $O/ltae_bench_noattn.txt 2>&1
cat $O/ltae_bench.txt $O/ltae_bench_noattn.txt
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/tu.err
cut -c1-300 $O/bench_tu.json
