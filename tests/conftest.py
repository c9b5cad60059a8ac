import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class Golden:
    """One fixture written by oracle/make_golden.py (outputs of the imported reference)."""

    def __init__(self, name):
        from oracle import seeded
        from oracle.crop2seg_oracle import BackboneConfig
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"]))
        self.key_shapes = [(str(k), tuple(json.loads(str(s)))) for k, s in zip(self.z["keys"], self.z["shapes"])]
        self.sd = seeded.make_state(self.key_shapes, self.meta["wseed"], self.meta["flavour"])
        drift = (seeded.checksum(self.sd).numpy() - self.z["wsum"])
        assert np.abs(drift).max() < 1e-6 * max(1.0, np.abs(self.z["wsum"]).max()), "seeded weights drifted (torch RNG changed?)"
        self.cfg = BackboneConfig(model=self.meta["model"])
        if self.meta["widths"]:
            self.cfg.encoder_widths, self.cfg.decoder_widths = self.meta["widths"]
        self.x = torch.from_numpy(self.z["x"])
        self.dates = torch.from_numpy(self.z["dates"])
        self.y = torch.from_numpy(self.z["y"])
        self.training = self.meta["mode"] == "train"
        self.attn_keep = torch.from_numpy(self.z["attn_keep"]) if "attn_keep" in self.z else None
        self.mlp_keep = torch.from_numpy(self.z["mlp_keep"]) if "mlp_keep" in self.z else None

    def dropout_kwargs(self):
        kw = {}
        if self.attn_keep is not None:
            kw["attn_keep"] = self.attn_keep
        if self.mlp_keep is not None:
            kw["mlp_keep"] = self.mlp_keep
        return kw

    def grad_names(self):
        return sorted({k.split("/")[1] for k in self.z.files if k.startswith("grad/")})

    def check_grad(self, name, g, rtol, atol_frac=1e-6):
        """Compare a gradient tensor with the stored summary. Returns (rel_err, ref_norm)."""
        flat = g.detach().double().flatten().cpu()
        ref_norm = float(self.z[f"grad/{name}/norm"])
        if f"grad/{name}/whole" in self.z:
            ref = torch.from_numpy(self.z[f"grad/{name}/whole"]).double()
            got = flat
        else:
            idx = torch.from_numpy(self.z[f"grad/{name}/idx"])
            ref = torch.from_numpy(self.z[f"grad/{name}/val"]).double()
            got = flat[idx]
        err = float((got - ref).norm())
        scale = float(ref.norm())
        return err, scale, ref_norm, float(flat.norm())


@pytest.fixture(scope="session")
def goldens():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
