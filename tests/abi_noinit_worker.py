"""Child process of tests/test_abi.py::test_wgrad_winograd_launches_without_c2s_init: a C-ABI caller that never calls c2s_init
(the INTEGRATION.md ctypes route) launches the 8-wave Winograd weight-gradient kernel, which needs 84 KB of dynamic LDS --
the raise of the limit lives in the per-device init hook, which the entry point must run itself."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from crop2seg_amd._lib import LIB_PATH, WgradDesc
    lib = C.CDLL(LIB_PATH)                                   # raw handle: nothing of crop2seg_amd.engine (which calls c2s_init) is used
    lib.c2s_wgrad_workspace_floats.restype = C.c_size_t
    lib.c2s_wgrad_workspace_floats.argtypes = [C.POINTER(WgradDesc)]
    lib.c2s_conv_wgrad.restype = C.c_int
    lib.c2s_conv_wgrad.argtypes = [C.POINTER(WgradDesc)] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p, C.c_void_p]
    lib.c2s_last_error.restype = C.c_char_p
    N, Cc, H = 2, 64, 128
    x = torch.randn(N, Cc, H, H, device="cuda")
    g = torch.randn(N, Cc, H, H, device="cuda")
    d = WgradDesc(N, Cc, 0, H, H, Cc, H, H, 3, 3, 1, 1, 1, 1, 64)
    nfl = lib.c2s_wgrad_workspace_floats(C.byref(d))
    slabs = torch.full((nfl,), float("nan"), device="cuda")
    rc = lib.c2s_conv_wgrad(C.byref(d), x.data_ptr(), None, g.data_ptr(), slabs.data_ptr(), nfl, None, None)
    torch.cuda.synchronize()
    assert rc == 0, lib.c2s_last_error()
    assert bool(torch.isfinite(slabs[: 64 * 9 * 64 * 64]).all()), "the launch did not run (dynamic LDS limit not raised?)"
    print("ABI_NOINIT_OK")


if __name__ == "__main__":
    main()
