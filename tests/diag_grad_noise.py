"""Diagnostic (not a test): per-tensor gradient error of the HIP path and of the fp32 CPU oracle, both measured
against the fp64 CPU oracle, on one golden case.  Usage: python tests/diag_grad_noise.py <golden name>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Golden  # noqa: E402
from oracle import crop2seg_oracle as O  # noqa: E402


def main(name):
    import crop2seg_amd as C2S
    from crop2seg_amd.backbones import functional as Fn
    g = Golden(name)
    kw = g.dropout_kwargs()
    if g.cfg.model == "wtae":
        kw.pop("mlp_keep", None)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in g.sd.items()}
    kw64 = {k: v.double() for k, v in kw.items()}
    _, l64, g64, _ = O.loss_and_grads(sd64, g.x.double(), g.dates, g.y, g.cfg, g.training, **kw64)
    _, l32, g32, _ = O.loss_and_grads(g.sd, g.x, g.dates, g.y, g.cfg, g.training, **kw)
    cls = {"utae": C2S.UTAE, "timeunet": C2S.TimeUNet_v1, "wtae": C2S.WTAE}[g.cfg.model]
    net = cls(input_dim=10, out_conv=[32, 15])
    net.load_state_dict(g.sd)
    net = net.cuda().train(g.training)
    drop = Fn.DropoutState()
    if g.attn_keep is not None:
        drop.attn_keep = g.attn_keep.cuda()
    if g.mlp_keep is not None:
        drop.mlp_keep = g.mlp_keep.cuda()
    if g.training and g.attn_keep is None:
        net.spec.attn_dropout = net.spec.mlp_dropout = 0.0
    logits = net(g.x.cuda(), batch_positions=g.dates.cuda(), dropout_state=drop)
    w = torch.ones(15, device="cuda")
    w[-1] = 0
    loss = torch.nn.functional.cross_entropy(logits, g.y.cuda(), weight=w)
    loss.backward()
    print(f"loss fp64 {float(l64):.8f} fp32 {float(l32):.8f} hip {float(loss):.8f}")
    params = dict(net.named_parameters())
    gmax = max(float(v.norm()) for v in g64.values())
    print(f"{'tensor':58s} {'|g|/gmax':>9s} {'hip/64':>9s} {'cpu32/64':>9s} ratio")
    for n in g64:
        ref = g64[n]
        nrm = float(ref.norm()) + 1e-300
        eh = float((params[n].grad.cpu().double() - ref).norm()) / nrm
        ec = float((g32[n].double() - ref).norm()) / nrm
        print(f"{n:58s} {nrm / gmax:9.2e} {eh:9.2e} {ec:9.2e} {eh / max(ec, 1e-30):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "utae_train_p0_tame")
