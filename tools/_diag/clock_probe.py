#!/usr/bin/env python
"""Diagnostic (GPU box): sample rocm-smi (sclk, power) while a kernel loops: (a) the pure fp32-MFMA micro-benchmark,
(b) the 8-wave and 4-wave Winograd launches at 64->64 @128^2 N=128, (c) an HBM copy.  Not part of the product."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from crop2seg_amd import _lib, engine as E  # noqa: E402

L = _lib
lib = E.lib()
dev = torch.device("cuda")


def sample(tag, stop):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        keep = [ln.strip() for ln in r.stdout.splitlines() if ("sclk" in ln or "Power" in ln or "mclk" in ln) and "GPU[0]" in ln]
        print(f"[{tag}] " + " | ".join(keep), flush=True)
        time.sleep(0.5)


def run(tag, fn, seconds=3.0):
    stop = threading.Event()
    th = threading.Thread(target=sample, args=(tag, stop))
    fn()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    print(f"[{tag}] {dt / n * 1e6:.1f} us per launch", flush=True)


N, Cc, H = 128, 64, 128
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
b = torch.randn(Cc, device=dev)
x = torch.randn(N, Cc, H, H, device=dev)
out = torch.empty_like(x)
taps = (C.c_int * 9)(*range(9))
d = L.ConvDesc(N, Cc, 0, H, H, Cc, 64, H, H, H, H, 3, 3, 1, 1, 1, L.PAD_REFLECT, 1, 1, 0, 0, 0, 0)
u16 = torch.empty(lib.c2s_winograd16_packed_floats(Cc, 64), device=dev)
E.check(lib.c2s_pack_weights_winograd16(w.data_ptr(), u16.data_ptr(), Cc, Cc, 64, Cc * 9, 9, taps, None), "pack")
u4 = torch.empty(lib.c2s_winograd_packed_floats(Cc, 64), device=dev)
E.check(lib.c2s_pack_weights_winograd(w.data_ptr(), u4.data_ptr(), Cc, Cc, 64, Cc * 9, 9, taps, None), "pack")
run("idle", lambda: None, 1.5)
run("hbm copy 537 MB", lambda: out.copy_(x))
run("winograd 8-wave", lambda: E.check(lib.c2s_conv3x3_winograd16(C.byref(d), x.data_ptr(), None, u16.data_ptr(), b.data_ptr(), out.data_ptr(), None, None), "c"))
run("winograd 4-wave", lambda: E.check(lib.c2s_conv3x3_winograd(C.byref(d), x.data_ptr(), None, u4.data_ptr(), b.data_ptr(), out.data_ptr(), None, None), "c"))
wp = torch.empty(9 * Cc * 64, device=dev)
E.check(lib.c2s_pack_weights(w.data_ptr(), wp.data_ptr(), Cc, Cc, 64, 9, Cc * 9, 9, taps, None), "pack")
run("direct igemm", lambda: E.check(lib.c2s_conv_igemm(C.byref(d), x.data_ptr(), None, wp.data_ptr(), b.data_ptr(), out.data_ptr(), None, None), "c"))
