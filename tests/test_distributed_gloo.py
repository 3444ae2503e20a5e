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


def _worker_graph(rank, world, port, out):
    """The product Graph's REAL parameter layout (36.8 M parameters, 147 MB of gradients) through the reducer calls of
    Runner.train_iteration -- zero_grad, broadcast_buffers, backward, all_reduce, optimizer step -- with a stub forward (the HIP kernels
    need a GPU): schedule of the exchange (one collective, or the early / late halves with the early one issued from inside backward),
    its result, and the persistent buffer home."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from shapeclipper_amd.model.graph import Graph
    from shapeclipper_amd.parallel import FlatGradAllReduce, _is_late
    from shapeclipper_amd.utils import options
    opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_gloo", "--output_root=/tmp/sc_pytest",
                                               "--tb!", "--arch.enc_pretrained!", "--cpu"]), verbose=False)
    torch.manual_seed(7 + rank)
    g = Graph(opt)
    keys_before = list(g.state_dict().keys())
    failed = []

    def chk(k, cond):
        if not bool(cond):
            failed.append(k)
    calls = []
    real_ar, real_bc = dist.all_reduce, dist.broadcast
    dist.all_reduce = lambda t, *a, **k: (calls.append(("all_reduce", t.numel())), real_ar(t, *a, **k))[1]
    dist.broadcast = lambda t, *a, **k: (calls.append(("broadcast", t.numel())), real_bc(t, *a, **k))[1]
    try:
        results = {}
        for overlap in (False, True):
            red = FlatGradAllReduce(g, world, overlap=overlap)
            chk(1, red.flat.numel() == sum(p.numel() for p in g.parameters()) and red.nbytes > 140e6)
            chk(2, list(g.state_dict().keys()) == keys_before)                          # buffers were re-homed, not replaced
            optim = torch.optim.SGD(g.parameters(), lr=0.1)
            names = {id(p): n for n, p in g.named_parameters()}
            if overlap:
                late = [names[id(p)] for p in red.params[red.n_early:]]
                chk(3, len(late) > 0 and all(_is_late(n) for n in late) and not any(_is_late(names[id(p)]) for p in red.params[:red.n_early]))
                chk(4, 0.03 < 1 - red.early_numel / red.flat.numel() < 0.2)         # the late segment is a small part of the bytes (5.5 %)
            calls.clear()
            # one "training iteration" as Runner.train_iteration issues it
            red.zero_grad()
            g.encoder.bn1.running_mean.fill_(float(rank + 1))
            red.broadcast_buffers()
            chk(5, calls == [("broadcast", red.buf_flat.numel())])                      # ONE collective for all BN statistics
            chk(6, float(g.encoder.bn1.running_mean[0]) == 1.0)                          # rank 0's value everywhere
            chk(7, g.encoder.bn1.running_mean.data_ptr() >= red.buf_flat.data_ptr())     # still a view of the flat home
            calls.clear()
            coef = float(rank + 1)
            # stub forward: parameters in module order, so autograd finishes the trunks' first layers last
            loss = sum((p * p).sum() * coef for p in g.parameters())
            loss.backward()
            in_backward = list(calls)
            red.all_reduce()
            optim.step()
            n_ar = [c for c in calls if c[0] == "all_reduce"]
            if overlap:
                chk(8, in_backward == [("all_reduce", red.early_numel)])                 # issued from the hook, inside backward
                chk(9, n_ar == [("all_reduce", red.early_numel), ("all_reduce", red.flat.numel() - red.early_numel)])
                chk(10, red.collectives == 2)
            else:
                chk(11, in_backward == [] and n_ar == [("all_reduce", red.flat.numel())] and red.collectives == 1)
            # mean over ranks of d/dp [coef * sum p^2] = 2 p * mean(coef); parameters were broadcast from rank 0 at construction
            mean_coef = sum(range(1, world + 1)) / world
            flat_params = torch.cat([(p.detach() + 0.1 * p.grad).reshape(-1) for p in red.params])      # undo the SGD step
            chk(12, torch.allclose(red.flat, 2 * mean_coef * flat_params, rtol=1e-5, atol=1e-7))
            chk(13, all(p.grad.data_ptr() >= red.flat.data_ptr() for p in g.parameters()))
            results[overlap] = torch.cat([p.detach().reshape(-1) for p in g.parameters()]).clone()
            if overlap:
                # an early parameter WITHOUT a gradient on one rank only.  The schedule is static: both ranks still issue
                # [early, late] (rank 1 starts its early collective from all_reduce() instead of from the hook) and the sum is right.
                calls.clear()
                red.zero_grad()
                skip = red.params[0] if rank == 1 else None
                loss = sum((p * p).sum() * coef for p in g.parameters() if p is not skip)
                loss.backward()
                chk(14, len(calls) == (0 if rank == 1 else 1))                          # rank 1's hook count never completes
                red.all_reduce()
                chk(15, [c for c in calls if c[0] == "all_reduce"] ==
                    [("all_reduce", red.early_numel), ("all_reduce", red.flat.numel() - red.early_numel)])
                want0 = 2 * red.params[0].detach() * (1.0 / world)                       # only rank 0 contributed (coef 1), mean over 2 ranks
                chk(16, torch.allclose(red.views[0], want0, rtol=1e-5, atol=1e-7))
                want1 = 2 * mean_coef * red.params[1].detach()
                chk(17, torch.allclose(red.views[1], want1, rtol=1e-5, atol=1e-7))
                # a backward() without zero_grad() in between must not launch duplicate early collectives or leave a stale handle
                calls.clear()
                sum((p * p).sum() for p in g.parameters()).backward()
                red.all_reduce()
                chk(18, len([c for c in calls if c[0] == "all_reduce"]) == 2)
                # NaN/Inf flag: MAX over the ranks -- every rank sees rank 1's bad step
                f = red.reduce_flag(torch.tensor(rank == 1))
                chk(19, float(f) > 0)
                f = red.reduce_flag(torch.tensor(False))
                chk(20, float(f) == 0)
                # hooks are removed by close(): a closed reducer launches nothing from a later backward pass
                red.close()
                calls.clear()
                sum((p * p).sum() for p in g.parameters()).backward()
                chk(21, calls == [])
            del red, optim
        out[rank] = list(failed)
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
        dist.destroy_process_group()


def test_reducer_schedule_on_the_real_parameter_layout():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_graph, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == [] and out[1] == [], "failed checks (numbered in source order): rank 0 %s, rank 1 %s" % (out[0], out[1])


def test_flat_allreduce_two_ranks():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]
