set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "conv4x4s2 or slice_sums or winograd_vs_direct or both_conv or train_step or hipgraph" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/tu.err; cut -c1-300 $O/bench_tu.json
python bench.py --model wtae --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_wtae.json 2> $O/wt.err; cut -c1-300 $O/bench_wtae.json
