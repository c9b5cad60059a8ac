#!/usr/bin/env python
"""Diagnostic (GPU box): time the 8-wave Winograd kernel (conv_winograd16.hip) at the dominant layer (64->64 3x3 @128x128,
N=128) as built, and with one part removed per diagnostic build (-DC2S_W16_DIAG=n: 1 no MFMA, 2 no staging, 3 no output
stores, 4 no LDS operand reads).  Each variant runs in a child process on its own library.  Not part of the product."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(libpath, shapes=((128, 64, 128),)):
    import ctypes as C
    import torch
    from crop2seg_amd import _lib
    _lib.LIB_PATH = libpath
    from crop2seg_amd import engine as E
    L = _lib
    lib = E.lib()
    dev = torch.device("cuda")
    for N, Cc, H in shapes:
        child_shape(lib, L, E, C, torch, dev, N, Cc, H)


def child_shape(lib, L, E, C, torch, dev, N, Cc, H):
    w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
    b = torch.randn(Cc, device=dev)
    x = torch.randn(N, Cc, H, H, device=dev)
    out = torch.empty_like(x)
    CP = (Cc + 63) // 64 * 64
    taps = (C.c_int * 9)(*range(9))
    for wide in (1, 0):
        for adj in (0, 1):
            if wide:
                upk = torch.empty(lib.c2s_winograd16_packed_floats(Cc, CP), device=dev)
                E.check(lib.c2s_pack_weights_winograd16(w.data_ptr(), upk.data_ptr(), Cc, Cc, CP, Cc * 9, 9, taps, None), "pack")
                fn = lib.c2s_conv3x3_winograd16
            else:
                upk = torch.empty(lib.c2s_winograd_packed_floats(Cc, CP), device=dev)
                E.check(lib.c2s_pack_weights_winograd(w.data_ptr(), upk.data_ptr(), Cc, Cc, CP, Cc * 9, 9, taps, None), "pack")
                fn = lib.c2s_conv3x3_winograd
            d = L.ConvDesc(N, Cc, 0, H, H, Cc, CP, H, H, H, H, 3, 3, 1, 1, 1, L.PAD_ZEROS if adj else L.PAD_REFLECT, 1, 1, 0, 0, 0, adj)
            # batches of 40 launches back to back after half a second of the same kernel: a launch timed alone after an idle gap
            # runs at ~2.1 GHz (the clock ramps with load; rocm-smi shows 2.39 GHz / 1.25 kW in the loop) and reads 10 % slower
            import time as _time
            t_end = _time.time() + 0.5
            while _time.time() < t_end:
                for _ in range(40):
                    fn(C.byref(d), x.data_ptr(), None, upk.data_ptr(), None if adj else b.data_ptr(), out.data_ptr(), None, None)
                torch.cuda.synchronize()
            ts = []
            for i in range(13):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    E.check(fn(C.byref(d), x.data_ptr(), None, upk.data_ptr(), None if adj else b.data_ptr(), out.data_ptr(), None, None), "conv")
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 40)
            print(f"   N={N} {Cc}->{Cc} @{H}  {'8-wave' if wide else '4-wave'} {'data gradient' if adj else 'forward      '}: "
                  f"min {min(ts[1:]) * 1e3:7.1f} us   median {sorted(ts[1:])[6] * 1e3:7.1f} us", flush=True)
            if wide and hasattr(lib, "c2s_debug_w16_stamps"):
                import numpy as np
                buf = np.zeros(1024 * 4, dtype=np.uint64)
                lib.c2s_debug_w16_stamps.argtypes = [C.c_void_p]
                if lib.c2s_debug_w16_stamps(buf.ctypes.data) == 0:
                    st = buf.reshape(-1, 4).astype(np.int64)
                    st = st[st[:, 1] > 0]
                    first = np.median(st[:, 2] >> 20)
                    st[:, 2] &= (1 << 20) - 1
                    wt, tot, nch, epi = (np.median(st[:, i]) for i in range(4))
                    print(f"      wave 0 of {len(st)} workgroups (last launch): {tot:.0f} cycles in all; {wt:.0f} = {100 * wt / tot:.1f} % between the "
                          f"s_waitcnt and the end of the chunk barrier ({wt / max(nch, 1):.0f} per chunk, {nch:.0f} chunks); epilogues {epi:.0f} = "
                          f"{100 * epi / tot:.1f} %; of the waits, the first chunks of the tiles: {first:.0f} ({first / max(nch / 8, 1):.0f} per tile)", flush=True)


def build_variant(v):
    """Recompile conv_winograd16.hip with -DC2S_W16_DIAG=v and link it with the product's other objects (tools/_diag/)."""
    from crop2seg_amd import build as B
    B.build(verbose=False)
    ddir = os.path.join(ROOT, "tools", "_diag")
    os.makedirs(ddir, exist_ok=True)
    obj = os.path.join(ddir, f"conv_winograd16_diag{v}.o")
    out = os.path.join(ddir, f"libc2s_w16diag{v}.so")
    defs = [f"-DC2S_W16_DIAG={v}"]
    if 4000 <= v < 5000:                           # 4000 + n: variant n with the wait / epilogue cycle counters (C2S_W16_STAMP)
        defs = [f"-DC2S_W16_DIAG={v - 4000}", "-DC2S_W16_STAMP"]
    subprocess.check_call([B.hipcc(), *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, "conv_winograd16.hip"), "-o", obj])
    objs = [obj if s == "conv_winograd16.hip" else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
    subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


NAMES = {4000: "as built, with wait counters", 4010: "variant 10 (raw tiles from the L2), with wait counters", 4008: "variant 8 (no raw staging), with wait counters", 4009: "variant 9 (no U staging), with wait counters", 15: "no left halo column", 16: "no right halo column", 17: "no halo columns (whole lines only)", 13: "raw rows shifted onto a 128-byte boundary (2 lines per row instead of 3)", 14: "raw rows of exactly one 128-byte line", 900: "round-3 kernel (A/B reference, built by hand)", 102: "raw tiles from frames 0-1 (8 MB)", 108: "raw tiles from frames 0-7 (32 MB)", 132: "raw tiles from frames 0-31 (128 MB)",
         0: "as built", 1: "no MFMA", 2: "no staging after the first two chunks", 3: "no output stores",
         4: "no LDS operand reads in the K loop", 5: "no staging and no LDS operand reads (MFMA + transform + epilogue)",
         6: "as 5, without the barrier per chunk", 7: "as 5, without the input transforms",
         8: "no raw-tile staging after the first two chunks", 9: "no U staging after the first two chunks", 10: "raw tiles all read from frame 0 (L2 hits)",
         11: "as 7, without the epilogue (MFMA loop + barriers + tile walk)", 12: "as 11, without the barrier per chunk"}

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], ((128, 64, 128), (128, 64, 64), (128, 64, 32), (128, 128, 32), (16, 64, 128)) if "--shapes" in sys.argv else ((128, 64, 128),))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--build":          # here (no GPU): the libraries travel with the snapshot
        for v in [int(a) for a in sys.argv[2:]] or sorted(NAMES):
            print(build_variant(v))
        sys.exit(0)
    extra = [a for a in sys.argv[1:] if a.startswith("--")]
    for v in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or sorted(NAMES):
        out = os.path.join(ROOT, "tools", "_diag", f"libc2s_w16diag{v}.so")
        if not os.path.exists(out):
            out = build_variant(v)
        print(f"variant {v}: {NAMES[v]}", flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, *extra])
