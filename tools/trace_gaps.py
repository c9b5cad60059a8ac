#!/usr/bin/env python
"""Diagnostic: from a rocprofv3 --kernel-trace CSV of bench.py, per HIP stream (queue): busy time, idle gaps between consecutive
kernels, and the kernels that follow the largest gaps, for the timed steps.  Usage: python tools/trace_gaps.py <dir> [steps]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: from the last-but-N adam kernel onward
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo, hi = adam[-nsteps - 1] + 1, adam[-1] + 1
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print(f"{nsteps} steps: {len(sel)} launches, span {(t1 - t0) / nsteps / 1e6:.3f} ms per step")
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = []
    for a, b in zip(rs, rs[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
    tot_gap = sum(g for g, _, _ in gaps)
    print(f"queue {q}: {len(rs) / nsteps:.0f} launches/step, busy {busy / nsteps / 1e6:.3f} ms/step, gaps {tot_gap / nsteps / 1e6:.3f} ms/step "
          f"({sum(1 for g, _, _ in gaps if g > 3000) / nsteps:.0f} gaps > 3 us per step)")
    for g, a, b in sorted(gaps, reverse=True)[:8]:
        print(f"      {g / 1e3:8.1f} us  after {a}  before {b}")
# union busy time over all queues (any kernel running)
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
run, last, idle = 0, t0, 0
for t, k in ev:
    if run == 0:
        idle += t - last
    run += k
    last = t
print(f"no kernel running at all: {idle / nsteps / 1e6:.3f} ms per step")
