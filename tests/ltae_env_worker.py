"""Child process of tests/test_ltae_paths_gpu.py: runs L-TAE op-parity cases (tests/test_ops_gpu.py::test_ltae_attention_fwd_bwd,
same oracle, same bars) in a process whose environment selects a kernel path -- the dispatch switches of csrc/ltae.hip are
read once per process (function-local statics), so each path needs a fresh process.  Prints LTAE_ENV_OK when all pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import ctypes as C
    import test_ops_gpu as T
    from crop2seg_amd import _lib
    cases = [tuple(int(v) if i < 4 else v == "1" for i, v in enumerate(c.split(","))) for c in sys.argv[1:]]
    for case in cases:
        B, Tn, Cc, h, with_emb, pad, drop = case
        d = _lib.LtaeDesc(B, Tn, Cc, h * h, 16, 256, 1e-5, 0.1 if drop else 0.0, 0, None, None)
        print("case", case, "uses_streaming", _lib.lib().c2s_ltae_uses_streaming(C.byref(d)), flush=True)
        T.test_ltae_attention_fwd_bwd(*case)
    print("LTAE_ENV_OK", len(cases))


if __name__ == "__main__":
    main()
