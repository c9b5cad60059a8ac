#!/usr/bin/env python
"""Diagnostic (GPU box): build a stamped copy of the library (-DC2S_WN_STAMP) and print per-workgroup phase times of
the Winograd kernel at the dominant layer (64->64 3x3 @128x128, N=128).  Not part of the product."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crop2seg_amd import build as B  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libc2s_stamp.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call([B.hipcc(), *B.FLAGS, "-DC2S_WN_STAMP", *extra, "-shared", "-fPIC", *srcs, "-o", out])
from crop2seg_amd import _lib  # noqa: E402
_lib.LIB_PATH = out          # diagnostic build replaces the product library for this process only
from crop2seg_amd import engine as E  # noqa: E402

L = _lib
lib = E.lib()
assert hasattr(lib, "c2s_debug_winograd_stamps"), "stamped library not loaded"
dev = torch.device("cuda")
N, Cc, H = 128, 64, 128
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
b = torch.randn(Cc, device=dev)
x = torch.randn(N, Cc, H, H, device=dev)
ctx = E.Ctx({"w": w, "b": b}, {}, {"w": torch.empty_like(w), "b": torch.empty_like(b)}, E.Workspace(dev), True, None)
for _ in range(3):
    E.conv2d(ctx, [x], "w", "b", 3, 1, 1, L.PAD_REFLECT, None)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
lib.c2s_debug_winograd_stamps.argtypes = [C.c_void_p]
assert lib.c2s_debug_winograd_stamps(buf.ctypes.data) == 0
s = buf.reshape(-1, 8).astype(np.int64)
s = s[s[:, 0] > 0]
tiles = s[:, 0].astype(np.float64)
print("workgroups:", len(s), " tiles per workgroup (median):", np.median(tiles))
names = {1: "first begin_tile (once)", 2: "commit chunk 0 + barrier", 3: "K loop", 4: "next tile: begin_tile + requests",
         5: "exchange half 0 (write+barrier)", 7: "reads+stores half 0, exchange half 1", 6: "reads+stores half 1"}
tot = 0.0
for k in range(1, 8):
    v = s[:, k] / (1.0 if k == 1 else tiles)
    tot += np.median(v) if k > 1 else 0
    print(f"{names[k]:42s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles/tile")
print(f"sum of per-tile medians: {tot:.0f} cycles")
