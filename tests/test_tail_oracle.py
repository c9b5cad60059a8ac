"""Pin oracle/tail_oracle.py (metrics tail, focal loss) against vectors produced by the imported reference
(oracle/make_golden_tail.py -> tests/golden/tail_*.npz), and check the unpinned restatements against independent
formulations.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tail_oracle as TO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["tail_miou_k15", "tail_miou_k4"])
def test_confusion_and_miou_match_reference(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    K = int(z["K"])
    conf = np.zeros((K, K), dtype=np.int64)
    for logits, y in zip(z["logits"], z["y"]):
        pred = torch.from_numpy(logits).argmax(dim=1).numpy()
        conf += TO.confusion_matrix(pred, y, K)
    assert np.array_equal(conf, z["conf"])
    miou, acc = TO.miou_acc(conf, int(z["ignore_index"]))
    assert miou == float(z["miou"]) and acc == float(z["acc"])


FOCAL = ["tail_focal_g2", "tail_focal_g1_ignore", "tail_focal_g2_weighted", "tail_focal_g1_sum", "tail_focal_g2_weighted_sum"]


def focal_args(z):
    """(class weights | None, size_average) of a focal fixture (the round-2 fixtures carry neither: the defaults)."""
    w = torch.from_numpy(z["weight"]) if "weight" in z.files and z["weight"].size else None
    return w, (bool(z["size_average"]) if "size_average" in z.files else True)


@pytest.mark.parametrize("name", FOCAL)
def test_focal_matches_reference(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    logits = torch.from_numpy(z["logits"]).requires_grad_(True)
    w, sa = focal_args(z)
    loss = TO.focal_ce(logits, torch.from_numpy(z["y"]), float(z["gamma"]), weight=w, size_average=sa)
    loss.backward()
    # (the reference sums its N x N product tensor in fp32: the product-of-sums restatement differs in the last bits)
    assert abs(float(loss) - float(z["loss"])) <= (1e-7 if w is None else 2e-6) * abs(float(z["loss"]))
    ref = torch.from_numpy(z["grad"])
    assert float((logits.grad - ref).abs().max()) <= (1e-9 if w is None and sa else 2e-6 * float(ref.abs().max()))


def test_top2_rule_equals_topk_where_defined():
    g = torch.Generator().manual_seed(5)
    logits = torch.relu(torch.randn(2, 15, 24, 24, generator=g)) * (torch.rand(2, 15, 24, 24, generator=g) > 0.4)
    st = TO.stable_top2(logits)
    tk = logits.topk(2, dim=1).indices
    ok = TO.top2_defined(logits)
    assert 0.05 < float(ok.float().mean()) < 0.99                     # the fixture has both tied and untied pixels
    assert torch.equal(st[:, 0][ok], tk[:, 0][ok]) and torch.equal(st[:, 1][ok], tk[:, 1][ok])
    assert torch.equal(st[:, 0], logits.argmax(dim=1))                # first maximum, as torch.argmax


def test_boundary_target_equals_neighbour_rule():
    g = torch.Generator().manual_seed(6)
    y = torch.randint(0, 5, (2, 12, 17), generator=g)
    yb = TO.boundary_target(y, 5)
    ref = torch.zeros_like(y)
    for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        sh = torch.roll(y, shifts=(dy, dx), dims=(1, 2))
        inside = torch.ones_like(y, dtype=torch.bool)
        if dy == 1: inside[:, 0] = False
        if dy == -1: inside[:, -1] = False
        if dx == 1: inside[:, :, 0] = False
        if dx == -1: inside[:, :, -1] = False
        ref |= ((sh != y) & inside).long()
    assert torch.equal(yb, ref)


def test_collate_restatement_pads_with_zero_frames():
    rng = np.random.default_rng(0)
    series = [rng.integers(0, 4000, (t, 10, 8, 8)).astype(np.int16) for t in (3, 5)]
    dates = [np.arange(t) * 5 + 1 for t in (3, 5)]
    mean = rng.normal(1000, 100, 10)
    std = rng.uniform(200, 400, 10)
    x, dd = TO.collate_series(series, dates, TO.CHANNELS_LIKE_PASTIS, mean, std)
    assert x.shape == (2, 5, 10, 8, 8) and dd.shape == (2, 5)
    assert float(x[0, 3:].abs().max()) == 0.0 and dd[0].tolist() == [1, 6, 11, 0, 0]
    c = 3
    want = (series[1][2, TO.CHANNELS_LIKE_PASTIS[c]].astype(np.float32) - np.float32(mean[c])) / np.float32(std[c])
    assert np.array_equal(x[1, 2, c].numpy(), want)


def test_softmax_stitch_restatement_tiling():
    g = torch.Generator().manual_seed(7)
    patches = [torch.randn(1, 3, 4, 4, generator=g) for _ in range(4)]
    proba, t1 = TO.softmax_stitch(patches, grid=2, crop=7)
    assert proba.shape == (3, 7, 7) and t1.shape == (7, 7)
    p3 = torch.softmax(patches[3], 1)[0]                 # patch (h=1, w=1) -> rows 4.., cols 4..
    assert torch.equal(proba[:, 4:7, 4:7], p3[:, :3, :3])
    assert torch.equal(t1[0:4, 4:7], torch.softmax(patches[1], 1)[0].argmax(0)[:, :3])


def test_smooth_cross_entropy_restatement_against_a_naive_loop():
    """SmoothCrossEntropy2D restatement (smooth_loss.py:58-84) against a per-pixel loop over the definition."""
    import math
    g = torch.Generator().manual_seed(2)
    B, K, H, W = 2, 15, 6, 7
    y = torch.randint(0, 4, (B, H, W), generator=g)
    logits = torch.randn(B, K, H, W, generator=g)
    got = float(TO.smooth_cross_entropy_2d(logits, y, None, 0.1, background_treatment=True))
    bd = [0.6] + [0.4 * p for p in TO.DEFAULT_CLASS_PROPORTIONS]
    eps, tot = 0.1 / K, 0.0
    for b in range(B):
        for r in range(H):
            for c in range(W):
                present = {int(y[b, r, c])}
                for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < H and 0 <= cc < W:
                        present.add(int(y[b, rr, cc]))
                large = (1 - eps * (K - len(present))) / len(present)
                t = [large if k in present else eps for k in range(K)]
                if int(y[b, r, c]) == 0:
                    t = bd
                z = logits[b, :, r, c].double()
                lse = float(torch.logsumexp(z, 0))
                tot += -sum(t[k] * (float(z[k]) - lse) for k in range(K))
    assert math.isclose(got, tot / (B * H * W), rel_tol=1e-5)
