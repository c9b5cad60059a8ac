#!/usr/bin/env python
"""HBM bytes of one train step by kernel, from two rocprofv3 PMC passes over bench.py (FETCH_SIZE and WRITE_SIZE in
SEPARATE passes, MI355X_MICROARCH.md HBM section; counter unit KiB per dispatch; FETCH_SIZE x2.000 on gfx950 as calibrated
by tools/pmc_traffic.py on kernels with known byte counts):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/st_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/st_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  python tools/step_traffic.py gpurun_out/st_fetch gpurun_out/st_write profiles/r03_step_traffic.csv
"""
import csv
import glob
import os
import re
import sys

FETCH_FACTOR = 2.0


def totals(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            k = k[:k.index("(")] if "(" in k else k
            s = acc.setdefault(k, [0.0, 0])
            s[0] += float(r["Counter_Value"]) * 1024.0
            s[1] += 1
    return acc


def main(df, dw, out, steps=None):
    fe, wr = totals(df, "FETCH_SIZE"), totals(dw, "WRITE_SIZE")
    steps = int(steps) if steps else fe.get("adam_kernel", [0, 1])[1]       # (a trace without optimiser steps: pass the repeat count)
    rows = []
    for k in sorted(set(fe) | set(wr)):
        f, n = fe.get(k, [0.0, 0])
        w, _ = wr.get(k, [0.0, 0])
        rows.append((k, n / steps, f * FETCH_FACTOR / steps / 1e9, w / steps / 1e9))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    tot = sum(r[2] + r[3] for r in rows)
    norm = sum(r[2] + r[3] for r in rows if r[0].startswith(("norm_", "row_stats")))
    with open(out, "w") as f:
        f.write(f"# HBM traffic of one train step by kernel: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -- "
                f"python bench.py --steps 3 --warmup 1 --no-cpu-baseline; bytes per step ({steps} steps in the trace); FETCH_SIZE x{FETCH_FACTOR} "
                f"(calibration: tools/pmc_traffic.py)\n# total {tot:.2f} GB per step; normalisation kernels {norm:.2f} GB of it\n")
        f.write("kernel,calls_per_step,read_GB,write_GB\n")
        for k, n, r, w in rows:
            if r + w >= 0.001:
                f.write(f"\"{k}\",{n:.1f},{r:.3f},{w:.3f}\n")
    print(f"total {tot:.2f} GB per step, normalisation {norm:.2f} GB -> {out}")


if __name__ == "__main__":
    if len(sys.argv) not in (4, 5):
        raise SystemExit(__doc__)
    main(*sys.argv[1:])
