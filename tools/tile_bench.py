"""N3 end to end: the web app's prediction loop on one Sentinel-2 tile -- 100 patches of T = 60 x 10 x 128 x 128
through TimeUNet_v1 in eval mode (prediction.py:194-202 forces model='timeunet'), softmax + top-1 + stitch to 1098 x 1098.
Patches/s of `predict_tile` for the reference's batch size (1) and for batched inference; inputs resident in HBM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import crop2seg_amd as C2S
from crop2seg_amd.inference import predict_tile
from crop2seg_amd.learning.utils import default_config, get_model

torch.manual_seed(1)
net = get_model(default_config("timeunet")).cuda()
net.apply(C2S.weight_init)
net.eval()
g = torch.Generator(device="cuda").manual_seed(2)
T = 60
x = torch.randn(100, T, 10, 128, 128, device="cuda", generator=g)
lengths = torch.randint(27, T + 1, (100,), generator=torch.Generator().manual_seed(3)).tolist()
dates = (5 * torch.arange(T, device="cuda"))[None].repeat(100, 1)
for b, tb in enumerate(lengths):
    x[b, tb:] = 0
    dates[b, tb:] = 0
for bs in (1, 2, 5, 10, 20):
    predict_tile(net, x, dates, batch_size=bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        proba, top1 = predict_tile(net, x, dates, batch_size=bs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"predict_tile TimeUNet_v1 100 patches T={T} (irregular lengths 27..{T}) 128x128 -> {tuple(proba.shape)}: batch_size {bs:2d}: "
          f"{dt * 1e3:8.1f} ms per tile = {100 / dt:7.1f} patches/s", flush=True)
