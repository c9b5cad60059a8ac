"""World-size-2 test of the data-parallel exchange (gloo on CPU).  Per-rank gradients come from the CPU oracle
(test infrastructure) on each rank's shard; the exchange code under test is crop2seg_amd.learning.ddp -- the same
object TrainStep uses on the GPU with backend "nccl" (RCCL).

Parity definition under DDP: the reduced gradient equals the mean of the per-shard oracle gradients;
parameters stay identical on all ranks after the optimiser step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import crop2seg_oracle as O, seeded
    from crop2seg_amd.learning.ddp import FlatDataParallel
    import crop2seg_amd as C2S
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    # every rank starts from DIFFERENT weights; sync_parameters must make them rank 0's
    sd = seeded.make_state(ks, 100 + rank, "tame")
    names = O.parameter_names(sd)
    sizes = [sd[n].numel() for n in names]
    flat_p = torch.cat([sd[n].flatten() for n in names])
    dp = FlatDataParallel()
    bufs = [sd[k] for k in sd if k not in names]     # BatchNorm running statistics + step counters (differ per rank too)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k].fill_(10 + rank)
    dp.sync_parameters(flat_p, bufs)
    off = 0
    for n, s in zip(names, sizes):
        sd[n] = flat_p[off:off + s].view_as(sd[n]).clone()
        off += s
    # shard: rank r owns patches [r]
    x, dates, y = seeded.make_inputs(2, 3, 10, 16, 16, 7, [3, 2])
    cfg = O.BackboneConfig()
    _, loss, grads, _ = O.loss_and_grads(sd, x[rank:rank + 1], dates[rank:rank + 1], y[rank:rank + 1], cfg, False)
    flat_g = torch.cat([grads[n].flatten() for n in names])
    local = flat_g.clone()
    scale = dp.reduce_gradients(flat_g)
    # the two-bucket form TrainStep uses to overlap the exchange with the encoder's backward pass: the suffix (decoder, temporal
    # encoder, head) starts early as an asynchronous collective, the prefix follows; together they must equal the one-bucket sum
    two = local.clone()
    cut = two.numel() // 3
    work = dp.reduce_async(two[cut:])
    assert work is not None
    dp.reduce_gradients(two[:cut])
    work.wait()
    assert torch.equal(two, flat_g), "two-bucket exchange differs from the single all-reduce"
    torch.save({"flat_p": flat_p, "local": local, "reduced": flat_g, "scale": scale,
                "bufs": torch.cat([b.double().flatten() for b in bufs])}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat_p"], r1["flat_p"]), "parameters differ after the initial broadcast"
    assert torch.equal(r0["bufs"], r1["bufs"]) and float(r0["bufs"].abs().sum()) > 0, "BatchNorm buffers differ after the broadcast"
    assert float(r0["bufs"].max()) == 10.0, "num_batches_tracked must be rank 0's"
    assert r0["scale"] == 0.5 and r1["scale"] == 0.5
    assert torch.equal(r0["reduced"], r1["reduced"]), "ranks hold different reduced gradients"
    expect = r0["local"] + r1["local"]
    assert torch.allclose(r0["reduced"], expect, rtol=0, atol=0) or float((r0["reduced"] - expect).abs().max()) < 1e-6
    # mean of per-shard gradients == scale * reduced
    mean = 0.5 * (r0["local"] + r1["local"])
    assert float((r0["reduced"] * r0["scale"] - mean).abs().max()) < 1e-6
    assert float(r0["local"].abs().sum()) > 0 and not torch.equal(r0["local"], r1["local"])
