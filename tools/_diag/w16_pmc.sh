# PMC stall counters of the 8-wave Winograd kernel's diagnostic builds (as built / no raw stream / no U stream / raw tiles from the L2):
# python tools/wino16_diag.py --build 0 8 9 10 first.  ~50 s per pass.  (A pass with the TA_* / TD_* stall counters hung rocprofv3
# twice on this pool -- the run was killed after seven silent minutes -- and is not in the list.)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w16pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for v in 0 8 9 10; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g${i}_v$v -- python tools/wino16_diag.py --child tools/_diag/libc2s_w16diag$v.so > $O/g${i}_v$v.log 2>&1 || echo "pass g$i v$v failed: $(tail -2 $O/g${i}_v$v.log)"
    echo "pass g$i variant $v done"          # (a run that prints nothing for 7 minutes is taken to be hung)
  done
done
python - <<'PY'
import csv,glob,re,collections
res=collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/w16pmc/g*_v*')):
    if not d.rsplit('/',1)[1].startswith('g') or d.endswith('.log'): continue
    v=d.rsplit('_v',1)[1]
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            if 'conv_winograd16_kernel<false' not in r['Kernel_Name']: continue
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for c,(s,n) in acc.items(): res[c][v]=s/max(n,1)
with open('gpurun_out/w16pmc/summary.txt','w') as o:
    o.write(f"{'counter (average per forward launch)':48s} {'as built':>14s} {'8: no raw':>14s} {'9: no U':>14s} {'10: raw in L2':>14s}\n")
    for c in sorted(res):
        o.write(f"{c:48s} "+" ".join(f"{res[c].get(v,float('nan')):14.4g}" for v in ('0','8','9','10'))+"\n")
print(open('gpurun_out/w16pmc/summary.txt').read())
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
