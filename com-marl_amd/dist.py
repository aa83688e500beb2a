"""Multi-GPU plumbing: one process per GPU, envs sharded by global id, ONE
exchange step per optimiser step - an RCCL (backend "nccl" on ROCm) all-reduce over xGMI of the flat
fp32 bucket [policy grads ‖ critic grads ‖ n_valid ‖ n_crit].  The payload is ~280-330 KB, i.e.
latency-bound, so a single bucket is used rather than per-tensor or overlapped buckets.
Pure torch.distributed: covered on CPU with the gloo backend (tests/test_dist_gloo.py)."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(total, rank, world):
    """Rank r of k owns envs [r*B/k, (r+1)*B/k) (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_grads(policy_params, critic_params, n_valid, n_crit, group=None):
    """Gradients were produced from SUM losses on every rank.  After one all-reduce(sum) of the flat
    bucket, dividing the policy part by the global number of valid steps and the critic part by the
    global number of padded steps gives exactly the gradient of the single-process mean losses
    (centralized_ma_ppo.py:437-438, comm_base_critic.py:88-89) - not a mean of per-rank means."""
    pol = [p for p in policy_params if p.grad is not None]
    cri = [p for p in critic_params if p.grad is not None]
    dev = (pol + cri)[0].grad.device
    counts = torch.stack([torch.as_tensor(n_valid, dtype=torch.float32, device=dev).reshape(()),
                          torch.as_tensor(n_crit, dtype=torch.float32, device=dev).reshape(())])
    flat = torch.cat([p.grad.reshape(-1) for p in pol + cri] + [counts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    nv, nc = flat[-2], flat[-1]
    off = 0
    for i, p in enumerate(pol + cri):
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p) / (nv if i < len(pol) else nc))
        off += n
    return float(nv), float(nc)
