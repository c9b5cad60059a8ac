#!/usr/bin/env python
"""Diagnostic (GPU box): where conv_wgrad_winograd_kernel<2> spends wave 0's cycles per tile (stamped build, -DC2S_WW_STAMP) at
the 64 -> 64 3x3 @128x128, N = 128 layer: commit + barrier, the MFMA loop, the barrier behind it.  `--build` compiles the stamped
library here (no GPU needed); without it the library must exist.  Not part of the product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crop2seg_amd import build as B  # noqa: E402

out = os.path.join(ROOT, "tools", "_diag", "libs", "libc2s_wwstamp.so")
if "--build" in sys.argv:
    B.build(verbose=False)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "_diag", "libs", "conv_wgrad_stamp.o")
    subprocess.check_call([B.hipcc(), *B.FLAGS, "-DC2S_WW_STAMP", "-c", os.path.join(B.CSRC, "conv_wgrad.hip"), "-o", obj])
    objs = [obj if s == "conv_wgrad.hip" else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    print(out)
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from crop2seg_amd import _lib  # noqa: E402
_lib.LIB_PATH = out
from crop2seg_amd import engine as E  # noqa: E402

dev = torch.device("cuda")
N, Cc, H = 128, 64, 128
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
params, grads = {"w": w, "b": torch.zeros(Cc, device=dev)}, {"w": torch.empty_like(w), "b": torch.zeros(Cc, device=dev)}
x = torch.randn(N, Cc, H, H, device=dev)
gy = torch.randn(N, Cc, H, H, device=dev)
ctx = E.Ctx(params, {}, grads, E.Workspace(dev), True, E.Tape())
vflags = torch.ones(N, dtype=torch.int32, device=dev)


def wgrad():
    E._wgrad_launch(ctx, [x], gy, Cc, H, H, 3, 1, 1, _lib.PAD_REFLECT, grads["w"], Cc * 9, 9, list(range(9)), 0, vflags)


import time  # noqa: E402
t_end = time.time() + 0.5
while time.time() < t_end:
    for _ in range(20):
        wgrad()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    wgrad()
e1.record()
torch.cuda.synchronize()
print(f"wgrad launch (kernel + slice sum), back to back: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
lib = E.lib()
buf = np.zeros(1024 * 4, dtype=np.uint64)
lib.c2s_debug_ww_stamps.argtypes = [C.c_void_p]
assert lib.c2s_debug_ww_stamps(buf.ctypes.data) == 0
st = buf.reshape(-1, 4).astype(np.int64)
st = st[st[:, 3] > 0]
tiles = N * (H // 4) * (H // 32) / len(st)
for i, name in enumerate(["commit + barrier", "MFMA loop (16 k-steps x 8 MFMAs of 64 cycles = 8192 issue cycles per wave)", "barrier behind the loop", "total"]):
    m = np.median(st[:, i])
    print(f"  {name:85s} {m:10.0f} cycles = {100 * m / np.median(st[:, 3]):5.1f} %   ({m / tiles:8.0f} per tile, {tiles:.0f} tiles per workgroup, {len(st)} workgroups)")
