set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_ltae_paths_gpu.py -x -q -m gpu -k "ltae or utae or wtae or pixel_gn" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.json 2> $O/prof.err
cut -c1-200 $O/prof.json
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4_b/prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ltae' in r['Name'] or 'copyBuffer' in r['Name'] or 'reduce_partials' in r['Name']:
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
find $O -name '*kernel_trace.csv' -delete
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-330 $O/bench.json
