"""world_size-2 gloo test (CPU) of the N>1 path: env sharding + the sum-of-grads / global-count
gradient exchange must reproduce the single-process mean-loss gradients exactly."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from com_marl_amd.dist import allreduce_sum_grads, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    pol = torch.nn.Linear(6, 3)
    cri = torch.nn.Linear(6, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(10, 6, generator=g)
    mask = (torch.rand(10, generator=g) < 0.7).float()
    mask[0] = 1.0
    lo, hi = shard_range(10, rank, world)            # ragged shards: valid counts differ per rank
    xs, ms = x[lo:hi], mask[lo:hi]
    (pol(xs).sum(-1) * ms).sum().backward()           # SUM losses per rank
    (cri(xs).squeeze(-1) ** 2).sum().backward()
    nv, nc = allreduce_sum_grads(list(pol.parameters()), list(cri.parameters()), ms.sum(), float(hi - lo))
    q.put((rank, nv, nc, [p.grad.clone().numpy() for p in list(pol.parameters()) + list(cri.parameters())]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_exchange_equals_single_process_mean():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: mean over valid / mean over all
    torch.manual_seed(0)
    pol = torch.nn.Linear(6, 3)
    cri = torch.nn.Linear(6, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(10, 6, generator=g)
    mask = (torch.rand(10, generator=g) < 0.7).float()
    mask[0] = 1.0
    ((pol(x).sum(-1) * mask).sum() / mask.sum()).backward()
    (cri(x).squeeze(-1) ** 2).mean().backward()
    want = [p.grad.numpy() for p in list(pol.parameters()) + list(cri.parameters())]
    for rank, nv, nc, grads in res:
        assert nv == float(mask.sum()) and nc == 10.0
        for a, b in zip(grads, want):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)


def test_shard_range_partitions():
    from com_marl_amd.dist import shard_range
    for total, world in ((4096, 8), (10, 3), (7, 8)):
        r = [shard_range(total, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
