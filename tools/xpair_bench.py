"""conv_xpair_kernel in isolation at the U-TAE shapes: data gradient of the 4x4 stride-2 down convolution (64 -> 64, 64x64 ->
128x128, N = 128 frames), per output-row-parity launch; variants: with / without accumulation into an existing gradient,
with / without the reflect adjoint.  HIP-event timed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from crop2seg_amd import _lib
from crop2seg_amd import engine as E
from crop2seg_amd._lib import ConvDesc, check, lib

dev = torch.device("cuda")
E.Workspace(dev)


def bench(N, Cin, Cout, Ho, accumulate, radj, iters=20):
    Hin = 2 * Ho
    g = torch.randn(N, Cin, Ho, Ho, device=dev)
    out = torch.randn(N, Cout, Hin, Hin, device=dev)
    wpk = torch.randn(8 * Cin * ((Cout + 31) // 32 * 32), device=dev) * 0.05
    ts = []
    for py in range(2):
        d = ConvDesc(N, Cin, 0, Ho, Ho, Cout, (Cout + 31) // 32 * 32, Ho, Ho, Hin, Hin, 2, 2, 1, 1 - py, 0, _lib.PAD_ZEROS, 2, 2, py, 0,
                     accumulate, radj)
        for it in range(iters + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().c2s_conv_xpair(C.byref(d), g.data_ptr(), wpk.data_ptr(), None, out.data_ptr(), None, E._stream()), "xpair")
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1))
    t = 2 * sum(ts) / len(ts)
    fl = 2.0 * N * Cin * Cout * 16 * Ho * Ho
    print(f"N={N} {Cin}->{Cout} {Ho}^2->{2 * Ho}^2 accumulate={accumulate} adjoint={radj}: {t * 1e3:7.1f} us for both parities = "
          f"{fl / t / 1e9:6.1f} TFLOP/s ({fl / t / 1e9 / 157.3 * 100:4.1f} % of the fp32 MFMA peak)", flush=True)


for acc, adj in ((1, 1), (0, 1), (0, 0), (1, 0)):
    bench(128, 64, 64, 64, acc, adj)
bench(128, 64, 64, 32, 1, 1)
bench(4, 128, 64, 16, 0, 0)
