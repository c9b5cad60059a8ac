#!/usr/bin/env python
"""Per-launch extract of the dominant kernel from a rocprofv3 --kernel-trace of bench.py: every launch of
conv_winograd16_kernel<false> with its grid, duration and the executed-MFMA rate if it is the 64->64 3x3 @128x128, N=128 launch of
the in_conv block (the longest forward launch of a step).  Usage: python tools/dominant_extract.py <trace dir> <out.csv>"""
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_winograd16_kernel<false" in r["Kernel_Name"]]      # (<false> in round 3, <false, 4, 16> since the shape template)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
big = max(durs)
EXEC_GF = 2.0 * 128 * 64 * 64 * 9 * 128 * 128 * 16.0 / 36.0 / 1e9          # executed MFMA GFLOP of the dominant launch
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["launch", "kernel", "grid", "workgroup", "duration_us", "dominant_64to64_3x3_128px_N128", "executed_TFLOPs", "frac_of_157.3"])
    n_dom, sum_dom = 0, 0.0
    for i, (r, us) in enumerate(zip(rows, durs)):
        dom = us > 0.8 * big
        if dom:
            n_dom += 1
            sum_dom += us
        w.writerow([i, "conv_winograd16_kernel<false>", r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), f"{us:.1f}",
                    int(dom), f"{EXEC_GF * 1e3 / us:.1f}" if dom else "", f"{EXEC_GF * 1e3 / us / 157.3:.3f}" if dom else ""])
    avg = sum_dom / max(n_dom, 1)
    w.writerow(["# dominant launches", n_dom, "average_us", f"{avg:.1f}", "executed_TFLOPs", f"{EXEC_GF * 1e3 / avg:.1f}", "frac_of_157.3", f"{EXEC_GF * 1e3 / avg / 157.3:.3f}"])
print(f"dominant launch: {n_dom} launches, average {avg:.1f} us = {EXEC_GF * 1e3 / avg:.1f} executed TFLOP/s = {EXEC_GF * 1e3 / avg / 157.3:.3f} of 157.3")
