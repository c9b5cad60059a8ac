"""conv_s2dgrad_kernel in isolation at the U-TAE shapes (data gradient of the 4x4 stride-2 down convolutions, N = 128 frames):
64 -> 64 gy 64x64 -> 128x128 and 128 <- 64 gy 32x32 -> 64x64, with / without accumulation into an existing gradient and the
reflect adjoint.  HIP-event timed; C2S_DIAG_LIB selects a diagnostic build of the library."""
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


def bench(N, Kc, Cs, Ho, accumulate, radj, iters=20):
    Hin = 2 * Ho
    CsP = (Cs + 63) // 64 * 64
    g = torch.randn(N, Kc, Ho, Ho, device=dev)
    out = torch.randn(N, Cs, Hin, Hin, device=dev)
    W = torch.randn(Kc, Cs, 4, 4, device=dev) * 0.05            # forward weight [Cout = Kc][Cin = Cs][4][4]
    d = ConvDesc(N, Kc, 0, Ho, Ho, Cs, CsP, Hin, Hin, Hin, Hin, 4, 4, 2, 1, 1, _lib.PAD_ZEROS, 1, 1, 0, 0, accumulate, radj)
    assert lib().c2s_conv4x4s2_dgrad_winograd_supported(C.byref(d))
    upk = torch.empty(lib().c2s_s2dgrad_packed_floats(Kc, CsP), device=dev)
    taps = (C.c_int * 16)(*range(16))
    check(lib().c2s_pack_weights_s2dgrad(W.data_ptr(), upk.data_ptr(), Kc, Cs, CsP, 16, Cs * 16, taps, E._stream()), "pack")
    ts = []
    for it in range(iters + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().c2s_conv4x4s2_dgrad_winograd(C.byref(d), g.data_ptr(), upk.data_ptr(), out.data_ptr(), None, E._stream()), "s2dgrad")
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    t = ts[len(ts) // 2]
    fl = 2.0 * N * Kc * Cs * 16 * Ho * Ho
    print(f"N={N} gy {Kc} -> gx {Cs} {Ho}^2->{Hin}^2 accumulate={accumulate} adjoint={radj}: median {t * 1e3:7.1f} us (min {ts[0] * 1e3:7.1f}) = "
          f"{fl / t / 1e9:6.1f} TFLOP/s algorithmic, {fl * 36 / 64 / t / 1e9 / 157.3 * 100:4.1f} % of the fp32 MFMA peak executed", flush=True)


for acc, adj in ((1, 1), (0, 1), (0, 0), (1, 0)):
    bench(128, 64, 64, 64, acc, adj)
bench(128, 64, 128, 32, 1, 1)
bench(128, 64, 128, 32, 0, 1)
