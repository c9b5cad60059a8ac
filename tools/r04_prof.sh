set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_prof; mkdir -p $O
C2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_1s.json 2> $O/prof_1s.err
find $O -name '*kernel_trace.csv' -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4_prof/prof_1s/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows: tot+=float(r['TotalDurationNs'])
print("total kernel ms/step", tot/7/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls'])/7:6.1f} {float(r['TotalDurationNs'])/7/1e6:7.3f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
