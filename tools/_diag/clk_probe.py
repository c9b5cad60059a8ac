"""Loop one kernel for ~3 s and sample the GPU's clocks and power from rocm-smi meanwhile."""
import ctypes as C, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from crop2seg_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
from crop2seg_amd import engine as E
L = _lib; lib = E.lib(); dev = torch.device("cuda")
N, Cc, H = 128, 64, 128
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05; b = torch.randn(Cc, device=dev)
x = torch.randn(N, Cc, H, H, device=dev); out = torch.empty_like(x)
CP = 64; taps = (C.c_int * 9)(*range(9))
upk = torch.empty(lib.c2s_winograd16_packed_floats(Cc, CP), device=dev)
E.check(lib.c2s_pack_weights_winograd16(w.data_ptr(), upk.data_ptr(), Cc, Cc, CP, Cc * 9, 9, taps, None), "pack")
d = L.ConvDesc(N, Cc, 0, H, H, Cc, CP, H, H, H, H, 3, 3, 1, 1, 1, L.PAD_REFLECT, 1, 1, 0, 0, 0, 0)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            samples.append(r.stdout.strip().replace("\n", " | "))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().replace("\n", " | ")[:600])
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 4.0:
    for _ in range(50):
        E.check(lib.c2s_conv3x3_winograd16(C.byref(d), x.data_ptr(), None, upk.data_ptr(), b.data_ptr(), out.data_ptr(), None, None), "conv")
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print(f"{n} launches, {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch back to back")
for s_ in samples[:12]: print(s_[:600])
