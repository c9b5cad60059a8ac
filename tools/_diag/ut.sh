set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ut; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.json 2> $O/prof.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ut/prof/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
adam=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
sel=rows[adam[-2]+1:adam[-1]+1]
t0=int(sel[0]['Start_Timestamp'])
with open('gpurun_out/ut/one_step.txt','w') as o:
    for r in sel:
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} q{r['Queue_Id']} g{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} w{r['Workgroup_Size_X']} {r['Kernel_Name'].replace('(anonymous namespace)::','')[:80]}\n")
PY
find $O -name '*kernel_trace.csv' -delete
