set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/tu; mkdir -p $O
python bench.py --model timeunet --batch 8 --T 61 --steps 15 --warmup 3 --no-cpu-baseline > $O/bench_tu.json 2> $O/tu.err
python -c "import json;print('timeunet', json.load(open('$O/bench_tu.json'))['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tu -- python bench.py --model timeunet --batch 8 --T 61 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_tu.json 2> $O/prof_tu.err
python tools/trace_gaps.py $O/prof_tu 4 > $O/trace_gaps_tu.txt
cat $O/trace_gaps_tu.txt
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/tu/prof_tu/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
adam=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
sel=rows[adam[-2]+1:adam[-1]+1]
t0=int(sel[0]['Start_Timestamp'])
with open('gpurun_out/tu/one_step.txt','w') as o:
    for r in sel:
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} q{r['Queue_Id']} g{r['Grid_Size_X']}x{r['Grid_Size_Y']} {r['Kernel_Name'].replace('(anonymous namespace)::','')[:70]}\n")
PY
find $O -name '*kernel_trace.csv' -delete
