#!/usr/bin/env python
"""bench.py -- train patches/s of the U-TAE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one optimiser step of the reference's training loop on one synthetic batch already resident in HBM:
zero_grad -> forward -> CrossEntropy -> backward -> (gradient all-reduce over RCCL when N > 1) -> Adam, in train mode
(BatchNorm batch statistics, dropout on).  N = 1 runs BASELINE.json configs[1]'s shape (U-TAE, B = 4 patches of
T = 32 x 10 x 128 x 128) in fp32 -- the arithmetic type of the reference; the bf16 named there has no reference
counterpart and is not claimed.  N > 1: one process per GPU (torch.distributed.run), B = 4 per GPU
(weak scaling), one flat 4.3 MB gradient all-reduce per step.

Prints ONE JSON line on rank 0 with the roofline of the dominant kernel (the 64->64 3x3 implicit-GEMM convolution at
128x128, timed with HIP events on the launch stream inside the timed region) and the CPU baseline (the CPU oracle --
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
# HBM bytes of ONE launch of the dominant kernel (64->64 3x3 @128x128, N=128): rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate passes (profiles/r01_pmc_winograd_64x64_128px.csv): 995,960 KiB + 524,288 KiB.
# FETCH_SIZE is quoted uncorrected: the x2 correction of the micro-architecture guide is calibrated for 16-B/lane
# streaming reads, this kernel gathers 4 B/lane.  Algorithmic bytes: 537 MB in + 537 MB out + 0.15 MB weights
# (the fetch excess is the halo of the 4x32-pixel tiles: 6x34 / 4x32 = 1.6x).
DOMINANT_KERNEL_TRAFFIC_BYTES = (995960.4 + 524288.0) * 1024
DOMINANT_KERNEL_TRAFFIC_BYTES_DIRECT = (822876.8 + 524288.0) * 1024    # C2S_WINOGRAD=0 / bf16x3: conv_igemm_kernel


def synthetic_batch(B, T, H, W, seed, device, n_classes=15):
    """x ~ N(0,1) f32 [B,T,10,H,W]; dates = 5*t; y ~ U{0..14}; generator seed = 1 + rank."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 10, H, W, generator=g)
    dates = (5 * torch.arange(T))[None, :].repeat(B, 1).to(torch.int64)
    y = torch.randint(0, n_classes, (B, H, W), generator=g)
    return x.to(device), dates.to(device), y.to(device)


def cpu_baseline(T, H, W, threads):
    """One train step of the CPU oracle (fwd + CE + bwd + Adam) on a bounded sample of the same workload."""
    from oracle import crop2seg_oracle as O
    import crop2seg_amd as C2S
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    net.apply(C2S.weight_init)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cfg = O.BackboneConfig()
    B = 2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, 10, H, W, generator=g)
    dates = (5 * torch.arange(T))[None, :].repeat(B, 1).to(torch.int64)
    y = torch.randint(0, 15, (B, H, W), generator=g)
    P = B * (H // 8) * (W // 8)
    ak = (torch.rand(16, P, T, generator=g) >= 0.1).float()
    mk = (torch.rand(P, 128, generator=g) >= 0.2).float()
    names = O.parameter_names(sd)
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    t0 = time.perf_counter()
    _, loss, grads, bn = O.loss_and_grads(sd, x, dates, y, cfg, True, attn_keep=ak, mlp_keep=mk)
    params = {n: sd[n] for n in names}
    O.adam_step(params, grads, m, v, 1)
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"1 train step (fwd+CE+bwd+Adam, train mode) of the CPU oracle, U-TAE B={B} T={T} {H}x{W} fp32, "
                      f"{dt:.1f} s on {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="utae", choices=["utae", "timeunet", "wtae"])
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU")
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
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

    if args.conv_mode is not None:
        E.CONV_MODE = args.conv_mode
    torch.manual_seed(1)                                   # --rdm_seed 1 (reference train.py:113,616-618)
    net = get_model(default_config(args.model)).to(device)
    net.apply(C2S.weight_init)                             # after .to(device), as in train.py:447-450
    net.train()
    step = TrainStep(net, num_classes=15, distributed=distributed)
    B, T, H = args.batch, args.T, args.size
    x, dates, y = synthetic_batch(B, T, H, H, 1 + rank, device)

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
    E.PROFILE = {"match": dict(KH=3, S=1, C0=64, C1=0, Cout=64, Hin=H, N=B * T, reflect_adjoint=0), "events": []} if not use_graph else None
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
        prof = {"match": dict(KH=3, S=1, C0=64, C1=0, Cout=64, Hin=H, N=B * T, reflect_adjoint=0), "events": []}
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

    if rank == 0:
        ms = [a.elapsed_time(b) for a, b in prof["events"]]
        kernel_ms = sum(ms) / max(len(ms), 1)
        flops_launch = 2.0 * (B * T) * 64 * 64 * 9 * H * H           # algorithmic FLOPs of one launch
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12 if ms else 0.0
        wino = E.WINOGRAD and E.CONV_MODE == "f32"
        full = B * T == 128 and H == 128
        roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F32_TFLOPS,
                    "traffic": (DOMINANT_KERNEL_TRAFFIC_BYTES if wino else DOMINANT_KERNEL_TRAFFIC_BYTES_DIRECT) if full else None,
                    "kernel": ("conv_winograd_kernel<4,*>" if wino else "conv_igemm_kernel<3,1,2,*>") +
                              " 64->64 3x3 reflect @128x128 (forward launches; the data-gradient launches of the same kernel share the GPU "
                              "with the weight gradients of the side stream, so their event-to-event time is not the kernel's own)",
                    "launches_timed": len(ms), "avg_launch_ms": kernel_ms,
                    "algorithmic_flops_per_launch": flops_launch}
        if wino:
            # `achieved` is the ALGORITHMIC (direct-convolution) FLOP count of SURVEY.md 8d over the launch time; the
            # Winograd F(2x2,3x3) kernel executes 16/36 of those multiplies on the MFMA pipe, so frac may exceed 1.
            roofline["executed_flops_per_launch"] = flops_launch * 16.0 / 36.0
            roofline["executed_tflops"] = achieved * 16.0 / 36.0
            roofline["executed_frac_of_peak"] = achieved * 16.0 / 36.0 / PEAK_F32_TFLOPS
            roofline["note"] = ("fp32 Winograd F(2x2,3x3): achieved = algorithmic direct-convolution FLOPs / time; "
                                "executed_* = MFMA FLOPs actually issued (2.25x fewer)")
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
            "config": {"workload": f"{args.model} train step (fwd+CE+bwd+Adam, train mode), B={B}/GPU T={T} 10x{H}x{H}, "
                                   f"random-init weight_init weights, BASELINE.json configs[1] shape in fp32",
                       "global_batch": world * B, "T": T, "parallelism": f"dp{world}", "hipgraph": use_graph},
            "roofline": roofline,
            "loss": loss_val,
        }
        step_flops = {"utae": (182962 * T + 161216) * H * H * 3.0}.get(args.model)
        if step_flops:
            out["step_mfma_frac"] = (out["value"] / world) * step_flops / (PEAK_F32_TFLOPS * 1e12)
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            try:
                threads = len(os.sched_getaffinity(0))
            except Exception:
                pass
            threads = min(threads, args.cpu_threads)        # a 1-GPU box owns a 16-core share of the host
            out["cpu_baseline"] = cpu_baseline(T, H, H, threads)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
