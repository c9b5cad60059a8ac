"""GPU parity of the steps on either side of the backbone through the C ABI, against
oracle/tail_oracle.py and the vectors of the imported reference (tests/golden/tail_*.npz).  Integer / index results are
held to bit-exactness; the fp32 collate arithmetic too (IEEE subtract and divide); softmax / focal values to 1e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tail_oracle as TO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["tail_miou_k15", "tail_miou_k4"])
def test_metrics_kernel_matches_reference_fixture(name):
    from crop2seg_amd.learning.metrics import IoU, StepMeters
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    K, ign = int(z["K"]), int(z["ignore_index"])
    iou = IoU(K, ignore_index=ign)
    iou_idx = IoU(K, ignore_index=ign)
    meters = StepMeters(K, ignore_index=ign)
    conf2 = np.zeros((K, K), dtype=np.int64)
    for logits, y in zip(z["logits"], z["y"]):
        lg, yy = torch.from_numpy(logits).cuda(), torch.from_numpy(y).cuda()
        iou.add(lg, yy)                                          # (N,K,H,W) scores: arg-maxed inside the kernel
        iou_idx.add(lg.argmax(dim=1), yy)                        # (N,H,W) class indices
        pred, pred2 = meters.update(lg, yy, torch.tensor([1.5], device="cuda"), want_pred=True)
        rp, rp2, _, rc2 = TO.metrics_tail(torch.from_numpy(logits), torch.from_numpy(y), K)
        assert torch.equal(pred.cpu(), rp) and torch.equal(pred2.cpu(), rp2)
        conf2 += rc2
    for m in (iou, iou_idx, meters.iou):
        assert np.array_equal(m.conf_metric.conf.cpu().numpy(), z["conf"])
        miou, acc = m.get_miou_acc()
        assert miou == float(z["miou"]) and acc == float(z["acc"])
    assert np.array_equal(meters.iou_top2.conf_metric.conf.cpu().numpy(), conf2)
    assert meters.get_miou_acc_top2() == TO.miou_acc(conf2, ign)
    assert meters.loss_mean() == 1.5


def test_metrics_at_full_size_properties():
    """B=4, 128x128, K=15 (BASELINE configs[1] logits): row sums of the confusion matrix = class histogram of the target;
    top-2 accuracy >= top-1 accuracy; >= 40 % exactly-zero logits resolve to the lowest class index."""
    from crop2seg_amd.learning.metrics import StepMeters
    g = torch.Generator().manual_seed(3)
    logits = torch.relu(torch.randn(4, 15, 128, 128, generator=g)) * (torch.rand(4, 15, 128, 128, generator=g) > 0.45)
    y = torch.randint(0, 15, (4, 128, 128), generator=g)
    m = StepMeters(15, ignore_index=-1)
    pred, pred2 = m.update(logits.cuda(), y.cuda(), want_pred=True)
    conf = m.iou.conf_metric.conf.cpu().numpy()
    assert np.array_equal(conf.sum(1), np.bincount(y.reshape(-1).numpy(), minlength=15))
    assert np.array_equal(conf, TO.confusion_matrix(logits.argmax(1).numpy(), y.numpy(), 15))
    assert torch.equal(pred.cpu(), logits.argmax(1))
    c2 = m.iou_top2.conf_metric.conf.cpu().numpy()
    assert np.trace(c2) >= np.trace(conf) and c2.sum() == conf.sum() == 4 * 128 * 128
    allzero = (logits.abs().sum(1) == 0)
    assert bool((pred.cpu()[allzero] == 0).all())


@pytest.mark.parametrize("shape", [(2, 12, 17), (4, 128, 128), (1, 1, 5)])
def test_boundary_target_bit_exact(shape):
    from crop2seg_amd.learning.losses import boundary_target
    g = torch.Generator().manual_seed(11)
    # blobs of classes, like a crop map
    y = torch.randint(0, 15, (shape[0], (shape[1] + 3) // 4, (shape[2] + 3) // 4), generator=g)
    y = y.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :shape[1], :shape[2]].contiguous()
    assert torch.equal(boundary_target(y.cuda()).cpu(), TO.boundary_target(y, 15))


@pytest.mark.parametrize("name", ["tail_focal_g2", "tail_focal_g1_ignore", "tail_focal_g2_weighted", "tail_focal_g1_sum",
                                  "tail_focal_g2_weighted_sum"])
def test_focal_loss_matches_reference_fixture(name):
    """Fixtures from the imported src/learning/focal_loss.py; the *_weighted / *_sum ones (round 4) cover the module's class
    weights -- the reference's [N,1] x [N] broadcast: product of the weight mean / sum and the focal mean / sum -- and
    size_average=False."""
    from crop2seg_amd.learning.losses import FocalCELoss, focal_ce
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    w = torch.from_numpy(z["weight"]) if "weight" in z.files and z["weight"].size else None
    sa = bool(z["size_average"]) if "size_average" in z.files else True
    logits, y = torch.from_numpy(z["logits"]).cuda(), torch.from_numpy(z["y"]).cuda()
    loss, grad = focal_ce(logits, y, float(z["gamma"]), want_grad=True, class_w=w, size_average=sa)
    assert abs(float(loss) - float(z["loss"])) <= 4e-6 * abs(float(z["loss"]))
    ref = torch.from_numpy(z["grad"])
    assert float((grad.cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    base = torch.full((1,), 0.25, device="cuda")
    focal_ce(logits, y, float(z["gamma"]), loss_out=base, class_w=w, size_average=sa)
    assert abs(float(base) - 0.25 - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    crit = FocalCELoss(gamma=float(z["gamma"]), size_average=sa, weight=w)           # the module-shaped wrapper
    assert abs(float(crit(logits, y, want_grad=True)) - float(z["loss"])) <= 4e-6 * abs(float(z["loss"]))
    assert torch.equal(crit.grad(), grad)
    flat = crit(logits.permute(0, 2, 3, 1).reshape(-1, logits.shape[1]), y.reshape(-1))     # (N, C) logits, (N,) targets
    assert abs(float(flat) - float(z["loss"])) <= 4e-6 * abs(float(z["loss"]))


@pytest.mark.parametrize("eps,ignore", [(0.0, False), (0.1, False), (0.2, True)])
def test_cross_entropy_label_smoothing_and_ignore(eps, ignore):
    """nn.CrossEntropyLoss(weight, label_smoothing) with a zero class weight (train.py:463-468) and torch's ignore_index."""
    from crop2seg_amd import engine as E
    g = torch.Generator().manual_seed(29)
    logits = (2 * torch.randn(3, 15, 32, 32, generator=g)).requires_grad_(True)
    y = torch.randint(0, 15, (3, 32, 32), generator=g)
    if ignore:
        y[torch.rand(3, 32, 32, generator=g) < 0.1] = -100
    cw = torch.ones(15)
    cw[-1] = 0
    ref = torch.nn.functional.cross_entropy(logits, y, weight=cw, label_smoothing=eps)
    ref.backward()
    loss, gl = E.cross_entropy(logits.detach().cuda(), y.cuda(), cw.cuda(), E.Workspace(torch.device("cuda")), True,
                               label_smoothing=eps)
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert float((gl.cpu() - logits.grad).norm() / logits.grad.norm()) <= 1e-5
    ws = E.Workspace(torch.device("cuda"))
    ybad = y.clone()
    ybad[0, 0, :3] = 15                       # torch raises on these; the kernel skips and counts them
    ybad[1, 1, 1] = -7
    keep = (ybad >= 0) & (ybad < 15) | (ybad == -100)
    loss2, _ = E.cross_entropy(logits.detach().cuda(), ybad.cuda(), cw.cuda(), ws, False, label_smoothing=eps)
    ref2 = torch.nn.functional.cross_entropy(logits.detach(), torch.where(keep, ybad, torch.full_like(ybad, -100)), weight=cw,
                                             label_smoothing=eps)
    assert E.bad_target_count(ws) == 4 and abs(float(loss2) - float(ref2)) <= 1e-5 * abs(float(ref2))


@pytest.mark.parametrize("dtype", [np.int16, np.uint16, np.float32])
@pytest.mark.parametrize("mode", ["zero_copy", "staged"])
def test_collate_series_bit_exact(dtype, mode):
    from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, SeriesCollator
    rng = np.random.default_rng(1)
    lengths = [5, 3, 7, 1]
    if dtype == np.float32:
        series = [rng.normal(1500, 900, (t, 10, 32, 32)).astype(np.float32) for t in lengths]
    else:
        series = [rng.integers(0, 12000, (t, 10, 32, 32)).astype(dtype) for t in lengths]
    dates = [np.sort(rng.integers(1, 400, t)).astype(np.int64) for t in lengths]
    mean = rng.normal(1200, 300, 10)
    std = rng.uniform(300, 900, 10)
    coll = SeriesCollator(CHANNELS_LIKE_PASTIS, mean, std, mode=mode)
    x, dd, valid = coll(series, dates)
    rx, rd = TO.collate_series(series, dates, CHANNELS_LIKE_PASTIS, mean, std)
    assert x.shape == rx.shape
    assert torch.equal(x.cpu(), rx), "collate arithmetic must be bit-exact (IEEE fp32 subtract + divide)"
    assert torch.equal(dd.cpu(), rd)
    assert valid.view(4, 7).cpu().tolist() == [[1] * t + [0] * (7 - t) for t in lengths]
    # the model's own frame detection agrees with the flags the collator emits
    from crop2seg_amd import engine as E
    assert torch.equal(E.frame_flags(x, 0.0), valid)
    # second call reuses the pinned staging buffers; fixed T (pad_collate max_size); no normalisation
    c2 = SeriesCollator(None, None, None, max_size=9, mode=mode)
    x2, d2, v2 = c2(series[:2], dates[:2])
    r2, _ = TO.collate_series(series[:2], dates[:2], list(range(10)), None, None)
    assert x2.shape == (2, 9, 10, 32, 32) and torch.equal(x2[:, :5].cpu(), r2) and float(x2[:, 5:].abs().max()) == 0.0


def test_softmax_stitch_matches_restatement():
    from crop2seg_amd import engine as E
    from crop2seg_amd._lib import check, lib
    g = torch.Generator().manual_seed(13)
    grid, h1, K, crop = 3, 16, 15, 41
    logits = torch.relu(2 * torch.randn(grid * grid, K, h1, h1, generator=g)) * (torch.rand(grid * grid, K, h1, h1, generator=g) > 0.3)
    rp, rt = TO.softmax_stitch([l[None] for l in logits], grid=grid, crop=crop)
    proba = torch.full((K, crop, crop), -1.0, device="cuda")
    top1 = torch.full((crop, crop), -1, device="cuda", dtype=torch.int64)
    ld = logits.cuda()
    for first, n in ((0, 4), (4, 5)):                             # two batches of patches
        check(lib().c2s_softmax_stitch(ld[first:first + n].contiguous().data_ptr(), proba.data_ptr(), top1.data_ptr(), first, n, K,
                                       h1, h1, grid, crop, crop, E._stream()), "softmax_stitch")
    assert float((proba.cpu() - rp).abs().max()) <= 1e-6
    agree = (top1.cpu() == rt)
    # top-1 = first maximum of the probabilities; a disagreement needs two probabilities within one ulp of each other
    top2 = rp.topk(2, dim=0).values
    assert bool(agree[(top2[0] - top2[1]) > 1e-6].all()) and float(agree.float().mean()) > 0.99


def test_predict_tile_equals_patch_by_patch():
    """N3: batched tile inference == the reference's B=1 loop (same model, eval mode), bit for bit."""
    import crop2seg_amd as C2S
    from crop2seg_amd.inference import predict_tile
    from oracle import seeded
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(seeded.make_state(ks, 3, "tame"))
    net = net.cuda().eval()
    grid, h1, T = 3, 32, 4
    g = torch.Generator().manual_seed(17)
    x = torch.randn(grid * grid, T, 10, h1, h1, generator=g).cuda()
    dates = (5 * torch.arange(T))[None].repeat(grid * grid, 1).cuda()
    proba, top1 = predict_tile(net, x, dates, grid=grid, crop=90, batch_size=4)
    with torch.no_grad():
        singles = [net(x[i:i + 1].contiguous(), batch_positions=dates[i:i + 1].contiguous()).cpu() for i in range(grid * grid)]
    rp, rt = TO.softmax_stitch(singles, grid=grid, crop=90)
    assert proba.shape == (15, 90, 90) and float((proba.cpu() - rp).abs().max()) <= 1e-6
    top2 = rp.topk(2, dim=0).values
    assert bool((top1.cpu() == rt)[(top2[0] - top2[1]) > 1e-6].all())
    p1, t1 = predict_tile(net, x, dates, grid=grid, crop=90, batch_size=1)
    assert torch.equal(p1, proba) and torch.equal(t1, top1), "batched inference must equal the B=1 loop bit for bit"


@pytest.mark.parametrize("mode", ["zero_copy", "staged"])
@pytest.mark.parametrize("slots", [1, 2])
def test_collator_back_to_back_calls_do_not_overwrite_each_other(mode, slots):
    """The collate kernel / H2D copies read pinned memory asynchronously: a second call must not rewrite the staging set
    of the first before the GPU has consumed it (the stream is kept busy so that the host runs ahead)."""
    from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, SeriesCollator
    rng = np.random.default_rng(5)
    lengths = [9, 12, 7, 11]
    batches = []
    for k in range(3):
        series = [rng.integers(0, 12000, (t, 10, 64, 64)).astype(np.int16) for t in lengths]
        dates = [np.sort(rng.integers(1, 400, t)).astype(np.int64) for t in lengths]
        batches.append((series, dates))
    mean, std = rng.normal(1200, 300, 10), rng.uniform(300, 900, 10)
    coll = SeriesCollator(CHANNELS_LIKE_PASTIS, mean, std, mode=mode, slots=slots)
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(40):                       # ~ tens of ms of queued GPU work in front of the first collate launch
        busy = busy @ busy * 1e-3
    outs = [coll(s, d) for s, d in batches]   # no synchronisation in between
    torch.cuda.synchronize()
    for (s, d), (x, dd, _) in zip(batches, outs):
        rx, rd = TO.collate_series(s, d, CHANNELS_LIKE_PASTIS, mean, std)
        assert torch.equal(x.cpu(), rx) and torch.equal(dd.cpu(), rd)


def test_prefetch_loader_yields_the_same_batches():
    from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, PrefetchLoader, SeriesCollator
    rng = np.random.default_rng(9)
    batches = []
    for k in range(5):
        lengths = [int(t) for t in rng.integers(3, 9, 3)]
        series = [rng.integers(0, 12000, (t, 10, 32, 32)).astype(np.uint16) for t in lengths]
        dates = [np.sort(rng.integers(1, 400, t)).astype(np.int64) for t in lengths]
        batches.append((series, dates, rng.integers(0, 15, (3, 32, 32)).astype(np.int64)))
    mean, std = rng.normal(1200, 300, 10), rng.uniform(300, 900, 10)
    loader = PrefetchLoader(batches, SeriesCollator(CHANNELS_LIKE_PASTIS, mean, std))
    n = 0
    for (s, d, t), (x, dd, valid, y) in zip(batches, loader):
        rx, rd = TO.collate_series(s, d, CHANNELS_LIKE_PASTIS, mean, std)
        assert torch.equal(x.cpu(), rx) and torch.equal(dd.cpu(), rd) and np.array_equal(y.cpu().numpy(), t)
        n += 1
    assert n == 5
    with pytest.raises(StopIteration):
        next(loader)


def test_prefetch_loader_can_be_abandoned_and_surfaces_worker_errors():
    """A consumer that leaves the loop early must not strand the worker on a full queue (close() / __del__), and an exception
    in the worker (a bad batch) reaches the consumer at its next request, not after the queue has drained."""
    from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, PrefetchLoader, SeriesCollator
    rng = np.random.default_rng(3)

    def gen(n, bad_at=None):
        for k in range(n):
            if k == bad_at:
                raise ValueError("bad batch")
            series = [rng.integers(0, 12000, (4, 10, 32, 32)).astype(np.uint16) for _ in range(2)]
            dates = [np.arange(1, 5, dtype=np.int64) for _ in range(2)]
            yield series, dates, None

    loader = PrefetchLoader(gen(50), SeriesCollator(CHANNELS_LIKE_PASTIS))
    next(loader)
    loader.close()
    assert not loader._thread.is_alive(), "the worker must exit once the consumer has gone away"
    loader = PrefetchLoader(gen(5, bad_at=1), SeriesCollator(CHANNELS_LIKE_PASTIS))
    next(loader)
    with pytest.raises(ValueError, match="bad batch"):
        next(loader)


def test_predict_tile_timeunet_equals_patch_by_patch():
    """N3 on the model the web app runs (prediction.py:194-202 forces model='timeunet', B=1, T ~ 60): a 2x2 tile of
    128x128 patches with T=60 -- at B=1 the full-resolution L-TAE has exactly 4 register tiles per CU, the dispatch
    threshold; batched inference must equal the B=1 loop bit for bit and the restatement of the loop to 1e-6."""
    import crop2seg_amd as C2S
    from crop2seg_amd.inference import predict_tile
    from oracle import seeded
    net = C2S.TimeUNet_v1(input_dim=10, out_conv=[32, 15])
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(seeded.make_state(ks, 5, "tame"))
    net = net.cuda().eval()
    grid, h1, T = 2, 128, 60
    g = torch.Generator().manual_seed(23)
    x = torch.randn(grid * grid, T, 10, h1, h1, generator=g)
    lengths = [60, 41, 27, 55]
    for b, tb in enumerate(lengths):
        x[b, tb:] = 0
    dates = (5 * torch.arange(T))[None].repeat(grid * grid, 1)
    for b, tb in enumerate(lengths):
        dates[b, tb:] = 0
    x, dates = x.cuda(), dates.cuda()
    proba, top1 = predict_tile(net, x, dates, grid=grid, crop=250, batch_size=4)
    p1, t1 = predict_tile(net, x, dates, grid=grid, crop=250, batch_size=1)
    assert torch.equal(p1, proba) and torch.equal(t1, top1), "batched inference must equal the B=1 loop bit for bit"
    with torch.no_grad():
        singles = [net(x[i:i + 1].contiguous(), batch_positions=dates[i:i + 1].contiguous()).cpu() for i in range(grid * grid)]
    rp, rt = TO.softmax_stitch(singles, grid=grid, crop=250)
    assert proba.shape == (15, 250, 250) and float((proba.cpu() - rp).abs().max()) <= 1e-6
    top2 = rp.topk(2, dim=0).values
    assert bool((top1.cpu() == rt)[(top2[0] - top2[1]) > 1e-6].all())


@pytest.mark.parametrize("bg,weighted,reduction", [(True, False, "mean"), (False, True, "mean"), (True, True, "mean"),
                                                  (True, True, "sum"), (False, True, "none")])
def test_smooth_cross_entropy_2d(bg, weighted, reduction):
    """N4: SmoothCrossEntropy2D (smooth_loss.py:18-84) -- loss and dL/dlogits against the restatement (dilation soft targets +
    torch's own CrossEntropyLoss with probability targets) differentiated by autograd.  Parity unpinned: the reference
    module cannot be imported here (torchnet)."""
    from crop2seg_amd.learning.losses import SmoothCrossEntropy2D
    g = torch.Generator().manual_seed(31)
    B, K, H, W = 3, 15, 40, 36
    y = torch.randint(0, K, (B, (H + 3) // 4, (W + 3) // 4), generator=g)
    y = y.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :H, :W].contiguous()       # fields with borders
    logits = (2 * torch.randn(B, K, H, W, generator=g)).requires_grad_(True)
    cw = None
    if weighted:
        cw = torch.rand(K, generator=g) + 0.5
        cw[-1] = 0
    ref = TO.smooth_cross_entropy_2d(logits, y, cw, 0.1, background_treatment=bg, reduction=reduction)
    ref.sum().backward()                          # 'none': [B,H,W] terms; grad() is the gradient of their sum
    crit = SmoothCrossEntropy2D(weight=cw, label_smoothing=0.1, background_treatment=bg, reduction=reduction)
    loss = crit(logits.detach().cuda(), y.cuda(), want_grad=True)
    if reduction == "none":
        assert loss.shape == ref.shape and float((loss.cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
        legacy = SmoothCrossEntropy2D(weight=cw, reduce=False, label_smoothing=0.1, background_treatment=bg)
        assert legacy.reduction == "none" and SmoothCrossEntropy2D(size_average=False).reduction == "sum"
    else:
        assert abs(float(loss) - float(ref)) <= 4e-6 * abs(float(ref))
    gl = crit.grad().cpu()
    assert float((gl - logits.grad).abs().max()) <= 2e-6 * float(logits.grad.abs().max())
    crit.check_targets()
    ybad = y.clone()
    ybad[0, 0, 0] = K
    crit(logits.detach().cuda(), ybad.cuda())
    with pytest.raises(ValueError, match="outside"):
        crit.check_targets()


@pytest.mark.parametrize("region", ["boundary", "interior"])
def test_step_meters_test_region_and_boundary_meter(region):
    """iterate()'s test_region relabelling (utils.py:362-373) and the boundary-head meter (utils.py:342,383-384): confusion
    matrices bit-exact against the restatement fed through the reference-pinned confusion-matrix restatement."""
    from crop2seg_amd.learning.losses import boundary_target
    from crop2seg_amd.learning.metrics import StepMeters
    g = torch.Generator().manual_seed(41)
    B, K, H, W = 3, 15, 48, 40
    y = torch.randint(0, K, (B, H // 4, W // 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2).contiguous()
    out = torch.randn(B, K, H, W, generator=g)
    out_b = torch.randn(B, 2, H, W, generator=g)
    m = StepMeters(K, ignore_index=-1, add_boundary_loss=True, test_region=region)
    yd = y.cuda()
    y_b = boundary_target(yd)
    for _ in range(2):
        m.update(out.cuda(), yd, torch.tensor([0.5], device="cuda"))
        m.update_boundary(out_b.cuda(), y_b)
    yr = TO.region_target(y, K, region, -1)
    assert torch.equal(m.region_target(yd).cpu(), yr)
    rp, rp2, rc, rc2 = TO.metrics_tail(out, yr, K)
    assert np.array_equal(m.iou.conf_metric.conf.cpu().numpy(), 2 * rc)
    assert np.array_equal(m.iou_top2.conf_metric.conf.cpu().numpy(), 2 * rc2)
    cb = TO.confusion_matrix(out_b.argmax(1).numpy(), TO.boundary_target(y, K).numpy(), 2)
    assert np.array_equal(m.iou_boundary.conf_metric.conf.cpu().numpy(), 2 * cb)
    assert m.get_miou_acc() == TO.miou_acc(2 * rc, -1) and m.get_miou_acc_boundary() == TO.miou_acc(2 * cb, None)


@pytest.mark.parametrize("dtype", [np.int16, np.float32])
def test_collate_series_with_ndvi_channel(dtype):
    """add_ndvi (s2_ts_cz_crop.py:376-391,401-402): the NDVI of the raw bands as an eleventh, un-normalised channel -- bit-exact
    against the restatement, including pixels whose band sum is 0 and quotients outside [-1, 1] (negative reflectances)."""
    from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, SeriesCollator
    rng = np.random.default_rng(3)
    lengths = [4, 6, 2]
    series = [rng.integers(-300, 9000, (t, 10, 32, 32)).astype(dtype) for t in lengths]
    for s in series:
        s[0, :, :4, :4] = 0                                  # no-data corner: band sum 0 -> NDVI 0
        s[-1, 3, 5, 5], s[-1, 2, 5, 5] = 10, -30             # NIR + red < 0, quotient -2 -> clipped to 0
    dates = [np.sort(rng.integers(1, 400, t)).astype(np.int64) for t in lengths]
    mean, std = rng.normal(1200, 300, 10), rng.uniform(300, 900, 10)
    coll = SeriesCollator(CHANNELS_LIKE_PASTIS, mean, std, add_ndvi=True)
    x, dd, valid = coll(series, dates)
    rx, rd = TO.collate_series(series, dates, CHANNELS_LIKE_PASTIS, mean, std, add_ndvi=True)
    assert x.shape == (3, 6, 11, 32, 32) and torch.equal(x.cpu(), rx) and torch.equal(dd.cpu(), rd)
    assert float(x[:, :, 10].abs().max()) <= 1.0 and float(x[0, 0, 10, :4, :4].abs().max()) == 0.0
