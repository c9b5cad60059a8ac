set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_j; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/tr.json 2> $O/tr.err
python tools/trace_gaps.py $O/tr 4
find $O -name '*kernel_trace.csv' -delete
