"""Deviation of reduced-precision convolution arithmetic from the fp32 path at BASELINE.json configs[1]'s shape (SURVEY.md 8c.5:
"bf16 has no reference counterpart -- report the deviation from the fp32 oracle, do not claim 1e-3").

Modes of the 3x3 stride-1 convolutions (forward + data gradient; every other kernel stays fp32):
    f32     exact fp32 MFMA (default; the parity-tested path)
    bf16x3  split precision: hi*hi + hi*lo + lo*hi on the bf16 MFMA (opt-in, C2S_CONV_MODE=bf16x3)
    bf16    plain bf16 inputs: the lo parts dropped (c2s_bf16x3_set_single_product(1)) -- what "bf16" in configs[1] would mean
Reported: logits (max abs / max |logit|), arg-max agreement, loss, flat parameter-gradient relative L2 -- eval mode (BatchNorm
frozen: well-conditioned) and train mode (batch statistics), U-TAE B=4 T=32 128x128, weight_init weights, seeded inputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import crop2seg_amd as C2S
from crop2seg_amd import engine as E
from crop2seg_amd._lib import lib
from crop2seg_amd.learning.synthetic import synthetic_batch
from crop2seg_amd.learning.utils import default_config, get_model


def run(mode, training, x, dates, y):
    E.CONV_MODE = "f32" if mode == "f32" else "bf16x3"
    lib().c2s_bf16x3_set_single_product(1 if mode == "bf16" else 0)
    torch.manual_seed(1)
    net = get_model(default_config("utae")).cuda()
    net.apply(C2S.weight_init)
    net.train(training)
    net.spec.attn_dropout = net.spec.mlp_dropout = 0.0
    logits = net(x, batch_positions=dates)
    w = torch.ones(15, device="cuda")
    w[-1] = 0
    loss = torch.nn.functional.cross_entropy(logits, y, weight=w)
    loss.backward()
    g = torch.cat([p.grad.flatten() for p in net.parameters()])
    lib().c2s_bf16x3_set_single_product(0)
    E.CONV_MODE = "f32"
    return logits.detach(), float(loss), g


def main():
    x, dates, y, _ = synthetic_batch(4, 32, 128, 128, 1, "cuda", irregular=False)
    for training in (False, True):
        ref = run("f32", training, x, dates, y)
        for mode in ("bf16x3", "bf16"):
            lg, loss, g = run(mode, training, x, dates, y)
            e = float((lg - ref[0]).abs().max() / ref[0].abs().max())
            agree = float((lg.argmax(1) == ref[0].argmax(1)).float().mean())
            ge = float((g - ref[2]).norm() / ref[2].norm())
            print(f"U-TAE B=4 T=32 128x128 {'train' if training else 'eval '} mode {mode:6s}: logits {e:.2e} of max|logit|, arg-max agreement "
                  f"{100 * agree:.3f} %, loss {loss:.6f} (fp32 {ref[1]:.6f}, rel {abs(loss - ref[1]) / ref[1]:.1e}), gradient rel L2 {ge:.2e}", flush=True)


if __name__ == "__main__":
    main()
