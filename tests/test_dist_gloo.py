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


def _worker_bucket(rank, world, port, q):
    import torch.distributed as dist
    from com_marl_amd.dist import GradBucket
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    pol, cri = torch.nn.Linear(5, 2), torch.nn.Linear(5, 1)
    unused = torch.nn.Parameter(torch.zeros(3))                      # a parameter no loss reaches: grad stays None
    bucket = GradBucket(list(pol.parameters()) + [unused], list(cri.parameters()))
    flat_ptr = bucket.flat.data_ptr()
    out = []
    counts = [(20_000_001 + rank, 30_000_003 + 5 * rank), (7 + rank, 11)]        # first pair: beyond f32's 2^24 integers
    for step, (nv, nc) in enumerate(counts):
        for p in list(pol.parameters()) + list(cri.parameters()):
            p.grad = None
        x = torch.full((4, 5), float(rank + 1 + step))
        pol(x).sum().backward()
        cri(x).sum().backward()
        tot = bucket.allreduce(torch.tensor(nv), float(nc))
        assert bucket.flat.data_ptr() == flat_ptr and unused.grad is None
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.pol[:2] + bucket.cri, bucket.views[:2] + bucket.views[3:]))
        out.append((tot.tolist(), [p.grad.clone().numpy() for p in list(pol.parameters()) + list(cri.parameters())]))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_persistent_bucket_counts_are_exact_and_grads_are_views():
    """dist.GradBucket: the same flat buffer every step, p.grad re-pointed at its slices, counts exact beyond 2^24
    (they travel as two small words each), a parameter without gradient stays without."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bucket, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for step, (nv, nc) in enumerate([(40_000_003, 60_000_011), (15, 22)]):
        for rank in range(world):
            tot, grads = res[rank][step]
            assert tot == [float(nv), float(nc)]
        # d(sum of outputs)/dW = column sums of x; both ranks' x summed, divided by the global count
        xsum = sum(4.0 * (r + 1 + step) for r in range(world))
        np.testing.assert_allclose(res[0][step][1][0], np.full((2, 5), xsum / np.float32(nv)), rtol=1e-6)
        np.testing.assert_allclose(res[0][step][1][2], np.full((1, 5), xsum / np.float32(nc)), rtol=1e-6)
        for a, b in zip(res[0][step][1], res[1][step][1]):
            np.testing.assert_array_equal(a, b)
