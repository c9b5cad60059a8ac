"""Child process of tests/test_dist_gpu.py: one rank of a torch.distributed job with backend "nccl" (RCCL).

The process group is created before anything else touches the GPU; then `TrainStep(distributed=True)` -- the object
bench.py --gpus N runs on every rank -- takes three optimiser steps next to a plain `TrainStep` from the same initial
state.  With one rank the gradient all-reduce (forced with C2S_BENCH_FORCE_DIST=1 so that the RCCL kernel really is
launched between backward and Adam) is the identity, so the two must agree bit for bit.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    import crop2seg_amd as C2S
    from crop2seg_amd.learning.utils import TrainStep
    from crop2seg_amd.learning.synthetic import synthetic_batch
    from oracle import seeded

    def fresh(distributed):
        torch.manual_seed(1)
        net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
        ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        net.load_state_dict(seeded.make_state(ks, 7 + (rank if not distributed else 0), "tame"))
        net = net.cuda().train()
        return net, TrainStep(net, num_classes=15, distributed=distributed)

    x, dates, y, _ = synthetic_batch(2, 6, 32, 32, 1 + rank, "cuda", lengths=[6, 4])
    net_d, step_d = fresh(True)
    net_p, step_p = fresh(False)
    assert step_d.dp is not None and step_d.dp.world == world
    for it in range(3):
        ld, _ = step_d(x, dates, y)
        lp, _ = step_p(x, dates, y)
    torch.cuda.synchronize()
    if world == 1:
        assert float(ld) == float(lp), (float(ld), float(lp))
        assert torch.equal(step_d.flat_param, step_p.flat_param), "distributed step differs from the plain step"
        for (k, a), (_, b) in zip(net_d.state_dict().items(), net_p.state_dict().items()):
            assert torch.equal(a, b), k
    else:
        # every rank must hold the same parameters after three steps on different shards
        ref = step_d.flat_param.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, step_d.flat_param), "ranks diverged"
    # hipGraph path with the collective between the two graphs (dropout off: a replay mixes a device-side counter into
    # the mask seed, so masks differ from the eager step by design)
    for net in (net_d, net_p):
        net.spec.attn_dropout = 0.0
        net.spec.mlp_dropout = 0.0
    step_d.capture(x, dates, y)
    lg, _ = step_d.replay()
    lq, _ = step_p(x, dates, y)
    torch.cuda.synchronize()
    if world == 1:
        assert float(lg) == float(lq) and torch.equal(step_d.flat_param, step_p.flat_param), "graph replay with the all-reduce differs"
    # the exchange must not cost the step anything worth noticing: at the bench shape (U-TAE, B=4, T=32, 128x128) the eager step
    # with the two-bucket exchange inside a world-size-1 RCCL group against the plain step, same process, interleaved
    if world == 1 and os.environ.get("C2S_DIST_TIMING", "1") != "0":
        import time
        xb, db, yb, _ = synthetic_batch(4, 32, 128, 128, 3, "cuda")
        net_d2, step_d2 = fresh(True)
        net_p2, step_p2 = fresh(False)
        def timed(step, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step(xb, db, yb)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        for st in (step_d2, step_p2):
            timed(st, 3)
        td = min(timed(step_d2, 8) for _ in range(3))
        tp_ = min(timed(step_p2, 8) for _ in range(3))
        print(f"DIST_TIMING distributed {td:.3f} ms plain {tp_:.3f} ms", flush=True)
        assert step_d2._early_off > 0, "the early bucket was never started"
        assert td <= tp_ + 0.15, f"the world-size-1 exchange costs {td - tp_:.3f} ms per step"
    dist.barrier()
    dist.destroy_process_group()
    print(f"DIST_OK rank {rank} world {world} loss {float(ld):.6f}")


if __name__ == "__main__":
    main()
