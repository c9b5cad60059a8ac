#!/usr/bin/env python
"""MFMA-pipe utilisation by kernel from a rocprofv3 PMC pass (its own run, kernel trace only):

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/mfma \\
      -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  python tools/mfma_util.py gpurun_out/mfma profiles/r04_mfma_util.csv

Per kernel: launches, MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs) as rocprofv3's derived counter defines it
(1,024 SIMDs on MI355X), and the fp32 MFMA FLOPs (MOPS x 512).  Counters are sampled with the kernels serialised by the profiler:
the utilisation is the kernel's own, not the step's."""
import csv
import glob
import os
import re
import sys

d, out = sys.argv[1], sys.argv[2]
SIMDS = 256 * 4
XCDS = 8                # GRBM_GUI_ACTIVE comes back summed over the eight XCDs (13.4 M cycles for a 768 us launch = 8 x 1.68 M)
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k)
        k = k[:k.index("(")] if "(" in k else k
        key = (k, r["Dispatch_Id"])
        c, v = r["Counter_Name"], float(r["Counter_Value"])
        slot = acc.setdefault(key, {})
        # one row per XCD and counter: busy cycles and operations add up, the active-cycle count is a clock (rocprofv3's own
        # MfmaUtil takes reduce(GRBM_GUI_ACTIVE, max))
        slot[c] = max(slot.get(c, 0.0), v) if c == "GRBM_GUI_ACTIVE" else slot.get(c, 0.0) + v
per = {}
for (k, _), c in acc.items():
    s = per.setdefault(k, [0, 0.0, 0.0, 0.0])
    s[0] += 1
    s[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    s[2] += c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    s[3] += c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512
rows = sorted(per.items(), key=lambda kv: -kv[1][2])
with open(out, "w") as f:
    f.write("# MFMA-pipe utilisation by kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32, own pass):\n"
            "# util = MFMA busy cycles / (GPU active cycles per XCD x 1024 SIMDs); tflops = fp32 MFMA FLOPs / active cycles at 2.4 GHz\n")
    f.write("kernel,launches,mfma_util_percent,mfma_gflop_per_launch,mfma_tflops_at_2.4GHz\n")
    for k, (n, busy, act, fl) in rows:
        if busy <= 0:
            continue
        f.write(f"\"{k}\",{n},{100.0 * busy / (act * SIMDS):.1f},{fl / n / 1e9:.2f},{fl / (act / 2.4e9) / 1e12:.1f}\n")
print(open(out).read())
