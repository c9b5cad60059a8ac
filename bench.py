#!/usr/bin/env python
"""bench.py -- train patches/s of the U-TAE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one optimiser step of the reference's training loop on one synthetic batch already resident in HBM:
zero_grad -> forward -> CrossEntropy -> backward -> (gradient all-reduce over RCCL when N > 1) -> Adam, in train mode
(BatchNorm batch statistics, dropout on).  N = 1 runs BASELINE.json configs[1]'s shape (U-TAE, B = 4 patches of
T = 32 x 10 x 128 x 128) in fp32 -- the arithmetic type of the reference; the bf16 named there has no reference
counterpart and is not claimed.  N > 1: one process per GPU (torch.distributed.run), B = 4 per GPU
(weak scaling), one flat 4.3 MB gradient all-reduce per step.

Prints ONE JSON line on rank 0 with the roofline of the dominant kernel (the 64->64 3x3 convolution of the in_conv block at
128x128 -- the 8-wave Winograd F(2x2,3x3) kernel conv_winograd16_kernel<false> --, timed with HIP events on the launch
stream inside the timed region) and the CPU baseline (the CPU oracle --
a restatement of the reference validated against it -- timed on this host's cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The step uses two HIP streams (crop2seg_amd/engine.py: weight gradients next to the data-gradient chain).  RCCL
# creates streams of its own; with the runtime's default of 4 hardware queues the side stream then shares a queue with
# the main stream and the overlap turns into a 0.6 ms loss per step (measured with a world-size-1 process group).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X dense fp32 MFMA/vector peak (MI355X_MICROARCH.md, chip-level parameters)
# HBM bytes of ONE launch of the dominant kernel come from a committed rocprofv3 --pmc run (FETCH_SIZE and WRITE_SIZE in
# separate passes, tools/pmc_traffic.py) -- never from a constant in this file.  The file names the commit / date /
# command it was measured with; without it `traffic` is null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")

# algorithmic FLOPs per patch, forward: (a*T + b) * H*W; a train step is 3x (forward, data and weight gradients)
STEP_FLOPS = {"utae": (182962, 161216), "timeunet": (151424, 288960), "wtae": (96936, 256192)}


def dominant_traffic(kernel_key, N, H):
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        e = t.get(kernel_key)
        if e and e.get("N") == N and e.get("H") == H:
            return e["bytes_per_launch"], f"profiles/{os.path.basename(TRAFFIC_FILE)}: {e.get('source', '')}"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def ltae_kernel_name(B, T, H):
    """The forward kernel c2s_ltae_attn_fwd_ws launches at the TimeUNet shape (C = 64, full resolution), asked of the library."""
    import ctypes as C
    from crop2seg_amd import _lib
    d = _lib.LtaeDesc(B, T, 64, H * H, 16, 256, 1e-5, 0.1, 0, None, None)
    path = _lib.lib().c2s_ltae_fwd_path(C.byref(d))
    return {0: "ltae_fwd_kernel (16-pixel LDS kernel)", 1: "ltae_prep_kernel + ltae_stream_fwd_kernel (three-pass streaming forward)",
            2: "ltae_reg_fwd_kernel (register-resident forward: x read once)"}.get(path, f"unknown path {path}")


def _cpu_steps(O, sd, cfg, B, T, H, W, n_warm, n_timed):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, 10, H, W, generator=g)
    dates = (5 * torch.arange(T))[None, :].repeat(B, 1).to(torch.int64)
    y = torch.randint(0, 15, (B, H, W), generator=g)
    P = B * (H // 8) * (W // 8)
    ak = (torch.rand(16, P, T, generator=g) >= 0.1).float()
    mk = (torch.rand(P, 128, generator=g) >= 0.2).float()
    names = O.parameter_names(sd)
    params = {n: sd[n].clone() for n in names}
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    times = []
    for it in range(n_warm + n_timed):
        t0 = time.perf_counter()
        cur = dict(sd)
        cur.update(params)
        _, loss, grads, bn = O.loss_and_grads(cur, x, dates, y, cfg, True, attn_keep=ak, mlp_keep=mk)
        O.adam_step(params, grads, m, v, it + 1)
        if it >= n_warm:
            times.append(time.perf_counter() - t0)
    return sum(times) / len(times)


def cpu_baseline(B, T, H, W, threads):
    """Train steps of the CPU oracle (fwd + CE + bwd + Adam, train mode) on this host's cores: the workload of the
    bench line (1 warm-up + 3 timed steps) and BASELINE.json configs[0] (U-TAE B=2, T=16)."""
    from oracle import crop2seg_oracle as O
    import crop2seg_amd as C2S
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    net.apply(C2S.weight_init)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cfg = O.BackboneConfig()
    dt = _cpu_steps(O, sd, cfg, B, T, H, W, 1, 3)
    dt_c1 = _cpu_steps(O, sd, cfg, 2, 16, 128, 128, 1, 3)
    return {"value": B / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle (torch CPU restatement of the reference, validated against it), U-TAE train step "
                      f"(fwd+CE+bwd+Adam, train mode) B={B} T={T} {H}x{W} fp32: 1 warm-up + 3 timed steps, {dt:.2f} s/step on "
                      f"{threads} threads",
            "configs0_value": 2 / dt_c1,
            "configs0_sample": f"BASELINE.json configs[0]: U-TAE B=2 T=16 128x128, 1 warm-up + 3 timed steps, {dt_c1:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="utae", choices=["utae", "timeunet", "wtae"])
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU")
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--irregular", dest="irregular", action="store_true", default=None,
                    help="pad_collate-style batch: series lengths T_b ~ U{27..T}, frames t >= T_b zero in x and dates "
                         "(default for --model timeunet: BASELINE.json configs[2] is defined on it)")
    ap.add_argument("--regular", dest="irregular", action="store_false")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--conv-mode", default=None, choices=["f32", "bf16x3"],
                    help="convolution arithmetic: exact fp32 MFMA (default) or opt-in split-precision bf16x3")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured hipGraph (single-stream) instead of launching eagerly; eager is the "
                         "default because the weight gradients then overlap the data-gradient chain on a side stream")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    args = ap.parse_args()
    # stdout carries exactly one line (the JSON): anything libraries write to file descriptor 1 while the bench runs --
    # RCCL prints a five-line version banner there when its communicator is created -- goes to stderr instead
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("C2S_BENCH_FORCE_DIST") == "1"   # the override rehearses the RCCL path on one GPU
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import crop2seg_amd as C2S
    from crop2seg_amd import engine as E
    from crop2seg_amd.learning.utils import TrainStep, default_config, get_model
    from crop2seg_amd.learning.synthetic import synthetic_batch

    if args.conv_mode is not None:
        E.CONV_MODE = args.conv_mode
    torch.manual_seed(1)                                   # --rdm_seed 1 (reference train.py:113,616-618)
    net = get_model(default_config(args.model)).to(device)
    net.apply(C2S.weight_init)                             # after .to(device), as in train.py:447-450
    net.train()
    step = TrainStep(net, num_classes=15, distributed=distributed)
    B, T, H = args.batch, args.T, args.size
    irregular = args.irregular if args.irregular is not None else args.model == "timeunet"
    x, dates, y, lengths = synthetic_batch(B, T, H, H, 1 + rank, device, irregular=irregular)

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()

    use_graph = args.graph and not args.no_graph
    if use_graph:
        step(x, dates, y)                                  # one eager step (allocator warm-up), then capture
        step.capture(x, dates, y)
        run = step.replay
    else:
        run = lambda: step(x, dates, y)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # HIP-event timing of the dominant kernel inside the timed region
    # (eager launches only: a captured graph cannot carry the timing events; bench_roofline() below measures the
    # same launches eagerly right after the timed region in graph mode)
    E.PROFILE = {"match": dict(KH=3, S=1, C0=64, C1=0, Cout=64, Hin=H, N=B * T, reflect_adjoint=0), "events": [], "ltae_events": []} if not use_graph else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = run()
    t_issue = time.perf_counter() - t0                     # host time to issue the launches (eager mode: must stay < dt)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = E.PROFILE
    E.PROFILE = None
    if use_graph:
        # roofline of the dominant kernel: the same launches, timed with HIP events on the launch stream during
        # extra eager steps (identical kernels, shapes and data as inside the graph)
        prof = {"match": dict(KH=3, S=1, C0=64, C1=0, Cout=64, Hin=H, N=B * T, reflect_adjoint=0), "events": [], "ltae_events": []}
        E.PROFILE = prof
        for _ in range(3):
            step(x, dates, y)
        torch.cuda.synchronize()
        E.PROFILE = None
    if distributed:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(loss)
    assert loss_val == loss_val, "loss is NaN"
    assert step.ws.sync_error() == 0, "a one-pass normalisation wait gave up (sync area error word set)"

    if rank == 0:
        ms = [a.elapsed_time(b) for a, b in prof["events"]]
        kernel_ms = sum(ms) / max(len(ms), 1)
        N = B * T
        n_real = sum(lengths)                                          # padded frames are skipped by the per-frame kernels
        flops_algo = 2.0 * n_real * 64 * 64 * 9 * H * H               # direct-convolution FLOPs of one launch
        wino = E.WINOGRAD and E.CONV_MODE == "f32"
        # FLOPs the kernel actually issues on the MFMA pipe: Winograd F(2x2,3x3) needs 16 multiplies per 2x2 output block
        # where the direct form needs 36 -- the roofline fraction is priced on the EXECUTED count (<= 1 by construction)
        flops_exec = flops_algo * (16.0 / 36.0 if wino else 1.0)
        achieved = flops_exec / (kernel_ms * 1e-3) / 1e12 if ms else 0.0
        kname = ("conv_winograd16_kernel<false>" if E._wide_winograd(H, H, [64]) else "conv_winograd_kernel<4,false>") if wino \
            else "conv_igemm_kernel<3,1,2,false>"
        traffic, traffic_src = dominant_traffic(kname, N, H)
        roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F32_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": 2.0 * n_real * 64 * H * H * 4 + 64 * 64 * 9 * 4,
                    "kernel": kname + f" 64->64 3x3 reflect @{H}x{H}, N={N} frames of which {n_real} real (forward launches of the in_conv block; the "
                              "data-gradient launches of the same kernel share the GPU with the weight gradients of the side "
                              "stream, so their event-to-event time is not the kernel's own)",
                    "launches_timed": len(ms), "avg_launch_ms": kernel_ms,
                    "executed_flops_per_launch": flops_exec,
                    "algorithmic_flops_per_launch": flops_algo,
                    "algorithmic_tflops": flops_algo / (kernel_ms * 1e-3) / 1e12 if ms else 0.0,
                    "note": ("achieved/frac = MFMA FLOPs the kernel executes (fp32 Winograd F(2x2,3x3): 16/36 of the direct-"
                             "convolution count) / launch time / fp32 MFMA peak; algorithmic_tflops prices the same time with the "
                             "direct-convolution FLOPs of SURVEY.md 8d" if wino else
                             "direct implicit GEMM: executed = algorithmic FLOPs")}
        lt = prof.get("ltae_events") or []
        ltae_roofline = None
        if lt and args.model == "timeunet":
            lms = [a.elapsed_time(b) for a, b in lt]
            lavg = sum(lms) / len(lms)
            Ppix = B * H * H
            # x read; emb and attn_pre written, + the keep flags of the dropout as bits (the train step never reads the post-dropout
            # weights of TimeUNet: they are not stored, engine.ltae_attention(need_attn=False))
            lbytes = 4.0 * (Ppix * T * 64 + Ppix * 256 + 16 * Ppix * T) + 8.0 * 16 * Ppix
            ltae_roofline = {"bound": "hbm", "achieved": lbytes / (lavg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": lbytes / (lavg * 1e-3) / 8e12, "traffic": None,
                             "kernel": ltae_kernel_name(B, T, H), "avg_launch_ms": lavg,
                             "launches_timed": len(lms), "algorithmic_bytes_per_launch": lbytes}
        a_, b_ = STEP_FLOPS[args.model]
        step_flops = (a_ * (n_real / B) + b_) * H * H * 3.0           # per-frame terms scale with the real frames of a patch
        cfg_id = {("utae", 4, 32, 128): "BASELINE.json configs[1] shape (fp32 instead of the bf16 named there: the reference is fp32-only)",
                  ("timeunet", 8, 61, 128): "BASELINE.json configs[2] shape",
                  ("wtae", 4, 32, 128): "BASELINE.json configs[3] per-GPU shape",
                  ("utae", 8, 48, 256): "BASELINE.json configs[4] per-GPU shape"}.get((args.model, B, T, H), "off-BASELINE shape")
        pad_note = (f"irregular series lengths T_b={lengths} (pad_collate zero padding of x and dates)" if irregular
                    else "regular series (no temporal padding)")
        out = {
            "metric": "train patches/sec (Tx10x128x128) U-TAE" if args.model == "utae" else f"train patches/sec {args.model}",
            "value": world * B * args.steps / dt,
            "unit": "patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "host_issue_ms_per_step": 1e3 * t_issue / args.steps,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if E.CONV_MODE == "f32" else "bf16x3 (split-fp32 operands on bf16 MFMA, fp32 accumulate) + f32",
            "data": "synthetic",
            "config": {"workload": f"{args.model} train step (fwd+CE+bwd+Adam, train mode), B={B}/GPU T={T} 10x{H}x{H}, {pad_note}, "
                                   f"weight_init random weights; {cfg_id}",
                       "global_batch": world * B, "T": T, "parallelism": f"dp{world}", "hipgraph": use_graph},
            "roofline": roofline,
            "loss": loss_val,
            "step_mfma_frac": (world * B * args.steps / dt / world) * step_flops / (PEAK_F32_TFLOPS * 1e12),
        }
        if ltae_roofline:
            out["roofline_ltae"] = ltae_roofline
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            try:
                threads = len(os.sched_getaffinity(0))
            except Exception:
                pass
            threads = min(threads, args.cpu_threads)        # a 1-GPU box owns a 16-core share of the host
            # BASELINE.json's metric is U-TAE's: the CPU leg always times the U-TAE step, at the bench shape when that is
            # no larger than configs[1] (B=4, T=32, 128x128: ~25 s of CPU work), else at configs[1]
            same = args.model == "utae" and B * T * H * H <= 4 * 32 * 128 * 128
            out["cpu_baseline"] = cpu_baseline(*((B, T, H, H) if same else (4, 32, 128, 128)), threads)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
