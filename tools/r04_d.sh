set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_d; mkdir -p $O
show() { python - "$1" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if ('ltae' in r['Name'] and 'fold' not in r['Name']) or 'gwc' in r['Name']: print("  ", r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
}
echo "== no-attn"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python tools/ltae_bench.py --no-attn --reps 3 > $O/p0.log 2>&1; show $O/p0
echo "== attn"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python tools/ltae_bench.py --reps 3 > $O/p1.log 2>&1; show $O/p1
find $O -name '*kernel_trace.csv' -delete
