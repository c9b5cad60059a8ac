#!/usr/bin/env python
"""HBM traffic of the dominant kernel (bench.py `roofline.traffic`) from rocprofv3 PMC counters.

Two steps, both on the GPU box (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes,
counter unit KiB per dispatch; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x and other access widths
have to be calibrated on a known byte count in the kernel's own access pattern):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/pmc_traffic.py run
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/pmc_traffic.py run
  python tools/pmc_traffic.py parse gpurun_out/pmc_fetch gpurun_out/pmc_write        # -> profiles/dominant_kernel_traffic.json

`run` launches, on buffers far beyond the 256 MiB Infinity Cache,
  * calibration kernels with known byte counts and the two access widths of the library:
      add_kernel        (c2s_add_inplace)  dword per lane:  reads 2 x 1 GiB, writes 1 GiB
      row_stats_kernel  (c2s_norm_fwd)     16 B per lane:   reads the 512 MiB activation tensor once
  * the dominant layer of the default bench workload: 64->64 3x3 reflect @128x128, N = 128 frames, forward
    (conv_winograd16_kernel<false>, or conv_winograd_kernel<4,false> under C2S_WINO16=0; the input gather is dword-per-lane
    buffer loads, the stores are float2 / float4).
`parse` averages the raw counters per kernel, derives the read correction factor of each access width from the calibration
kernels and writes bytes_per_launch = fetch_raw x factor(dword) + write_raw x factor(store)."""
import csv
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_FR, CH, HW_ = 128, 64, 128
ADD_FLOATS = 1 << 28                     # 1 GiB per operand


def run():
    import torch
    from crop2seg_amd import _lib, engine as E
    dev = torch.device("cuda")
    L = E.lib()
    a = torch.zeros(ADD_FLOATS, device=dev)
    b = torch.ones(ADD_FLOATS, device=dev)
    for _ in range(3):
        E.check(L.c2s_add_inplace(a.data_ptr(), b.data_ptr(), ADD_FLOATS, None), "add")
    torch.cuda.synchronize()
    del a, b
    x = torch.randn(N_FR, CH, HW_, HW_, device=dev)
    w = torch.randn(CH, CH, 3, 3, device=dev) * 0.05
    bias = torch.randn(CH, device=dev)
    params = {"w": w, "b": bias, "n.weight": torch.ones(CH, device=dev), "n.bias": torch.zeros(CH, device=dev)}
    ws = E.Workspace(dev)
    E.ONEPASS_NORM = False               # the calibration kernel is the two-pass statistics kernel (one 16-B read of the tensor)
    for _ in range(3):
        ctx = E.Ctx(params, {}, {}, ws, True, None)
        E.norm_act(ctx, x, "n", _lib.NORM_GROUP, 4, True, None, None)
    torch.cuda.synchronize()
    for _ in range(5):
        ctx = E.Ctx(params, {}, {}, ws, True, None)
        E.conv2d(ctx, [x], "w", "b", 3, 1, 1, _lib.PAD_REFLECT, None)
    torch.cuda.synchronize()


def averages(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            s = acc.setdefault(k, [0.0, 0])
            s[0] += float(r["Counter_Value"])
            s[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def pick(avgs, needle):
    hits = [(k, v) for k, v in avgs.items() if needle in k]
    if not hits:
        raise SystemExit(f"kernel {needle!r} not in the counter file")
    return max(hits, key=lambda kv: kv[1][0])


def parse(dir_fetch, dir_write):
    fetch = averages(dir_fetch, "FETCH_SIZE")
    write = averages(dir_write, "WRITE_SIZE")
    KiB = 1024.0
    add_f, add_w = pick(fetch, "add_kernel")[1][0] * KiB, pick(write, "add_kernel")[1][0] * KiB
    rs_f = pick(fetch, "row_stats_kernel")[1][0] * KiB
    act = float(N_FR * CH * HW_ * HW_ * 4)
    f_dword = (2.0 * ADD_FLOATS * 4) / add_f            # known bytes / reported bytes, dword-per-lane reads
    f_b128 = act / rs_f                                 # 16-B-per-lane reads
    f_store = (ADD_FLOATS * 4.0) / add_w                # dword-per-lane stores
    wide = any("conv_winograd16_kernel<false" in k for k in fetch)      # the default for planes >= 32 wide (C2S_WINO16)
    needle, key = ("conv_winograd16_kernel<false", "conv_winograd16_kernel<false>") if wide else \
                  ("conv_winograd_kernel<4, false>", "conv_winograd_kernel<4,false>")
    wk, (wf, nf) = pick(fetch, needle)
    _, (ww, nw) = pick(write, needle)
    raw_f, raw_w = wf * KiB, ww * KiB
    # the kernel's stores are float4 / float2 (exact per the guide); its input gather is dword per lane
    bytes_per_launch = raw_f * f_dword + raw_w
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "unknown"
    out = {
        key: {
            "N": N_FR, "H": HW_, "bytes_per_launch": bytes_per_launch,
            "fetch_raw_bytes": raw_f, "write_raw_bytes": raw_w, "dispatches": [nf, nw],
            "fetch_factor_dword_reads": f_dword, "fetch_factor_16B_reads": f_b128, "write_factor_dword_stores": f_store,
            "algorithmic_bytes": 2.0 * act + CH * CH * 9 * 4,
            "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/pmc_traffic.py run; "
                      f"commit {commit}, {time.strftime('%Y-%m-%d')}; reads corrected by the dword-read factor measured on add_kernel "
                      f"in the same pass",
        }
    }
    path = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) == 4 and sys.argv[1] == "parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
