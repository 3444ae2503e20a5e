"""N > 1 data-parallel path on CPU: two gloo ranks, flat single all-reduce of all gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shapeclipper_amd.parallel import FlatGradAllReduce
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    shared = torch.nn.Linear(4, 4)
    net = torch.nn.ModuleDict(dict(a=shared, b=torch.nn.Sequential(shared, torch.nn.Linear(4, 2)),
                                   bn=torch.nn.BatchNorm1d(2), unused=torch.nn.Linear(3, 3)))
    red = FlatGradAllReduce(net, world)
    w0 = [p.detach().clone() for p in net.parameters()]
    x = torch.full((5, 4), float(rank + 1))
    red.zero_grad()
    y = net["bn"](net["b"](x))
    y.sum().backward()
    ok = all(p.grad is None for p in net["unused"].parameters())                        # autograd produced nothing there
    local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in red.params])
    red.all_reduce()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok &= torch.allclose(red.flat, sum(gathered) / world, atol=1e-6)
    ok &= all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters())      # grads are views of the flat buffer
    ok &= bool(torch.all(net["unused"].weight.grad == 0))                               # unused params contribute zeros
    net["bn"].running_mean.fill_(float(rank))
    red.broadcast_buffers()
    ok &= float(net["bn"].running_mean[0]) == 0.0
    ws = [torch.zeros_like(w0[0]) for _ in range(world)]
    dist.all_gather(ws, w0[0])
    ok &= torch.equal(ws[0], ws[1])                                                     # parameters start identical
    # a second step starts from dropped grads (first accumulation = pointer move) and ends in the flat views again,
    # with no residue of the first step
    red.zero_grad()
    ok &= all(p.grad is None for p in net.parameters())
    net["bn"](net["b"](x)).sum().backward()
    local2 = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in red.params])
    red.all_reduce()
    gathered2 = [torch.zeros_like(local2) for _ in range(world)]
    dist.all_gather(gathered2, local2)
    ok &= torch.allclose(red.flat, sum(gathered2) / world, atol=1e-6)
    ok &= all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters())
    # sharded-eval record gather: uneven shards, result sorted by sample index on every rank
    from shapeclipper_amd.parallel import gather_eval_records
    mine = torch.tensor([[float(i)] + [float(i) * 0.5] * 9 for i in range(7) if i % world == rank])
    allr = gather_eval_records(mine, world)
    ok &= allr.shape == (7, 10) and torch.equal(allr[:, 0], torch.arange(7.0)) and torch.allclose(allr[:, 1], torch.arange(7.0) * 0.5)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_allreduce_two_ranks():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]
