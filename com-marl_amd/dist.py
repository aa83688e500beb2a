"""Multi-GPU plumbing: one process per GPU, envs sharded by global id, ONE
exchange step per optimiser step - an RCCL (backend "nccl" on ROCm) all-reduce over xGMI of the flat
fp32 bucket [policy grads ‖ critic grads ‖ n_valid (2 words) ‖ n_crit (2 words)].  The payload is
~280-330 KB, i.e. latency-bound, so a single bucket is used rather than per-tensor or overlapped buckets.
Pure torch.distributed: covered on CPU with the gloo backend (tests/test_dist_gloo.py)."""
import torch
import torch.distributed as dist

_SPLIT = 12                      # a count travels as (n >> 12, n & 4095): both words and their sums over the ranks are
_LOW = (1 << _SPLIT) - 1         # integers far below 2^24, so the f32 all-reduce adds them exactly up to 2^36 samples


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(total, rank, world):
    """Rank r of k owns envs [r*B/k, (r+1)*B/k) (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """The persistent flat exchange buffer of one (policy, critic) pair.

    ``allreduce`` gathers the freshly produced gradients into the bucket with one multi-tensor copy, reduces the bucket
    in place, divides the two parts by the GLOBAL counts on the device and re-points every ``p.grad`` at its slice of
    the bucket - the optimiser then reads the reduced gradients where they are.  Per optimiser step: no allocation, no
    ``cat``, no copy back, no host synchronisation (round 2 did a ``cat`` of 32 tensors, 32 ``copy_`` and a
    ``float(count)``)."""

    def __init__(self, policy_params, critic_params):
        self.pol = [p for p in policy_params if p.requires_grad]
        self.cri = [p for p in critic_params if p.requires_grad]
        ps = self.pol + self.cri
        if not ps:
            raise ValueError("GradBucket: no trainable parameters")
        dev = ps[0].device
        self.n_pol = sum(p.numel() for p in self.pol)
        self.n_cri = sum(p.numel() for p in self.cri)
        self.flat = torch.zeros(self.n_pol + self.n_cri + 4, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in ps:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.counts = None                         # (n_valid, n_crit) of the last reduce: float64 device tensor [2]

    def matches(self, policy_params, critic_params):
        a = [p for p in policy_params if p.requires_grad] + [p for p in critic_params if p.requires_grad]
        b = self.pol + self.cri
        return len(a) == len(b) and all(x is y for x, y in zip(a, b))

    def allreduce(self, n_valid, n_crit, group=None):
        ps = self.pol + self.cri
        dev = self.flat.device
        src, dst, absent = [], [], set()
        for k, (p, v) in enumerate(zip(ps, self.views)):
            if p.grad is None:                     # a parameter the loss does not reach (the same on every rank: replicas of
                v.zero_()                          # one model) contributes zeros and keeps grad = None for the optimiser
                absent.add(k)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad.reshape(v.shape))
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        as64 = lambda x: (x.to(device=dev, dtype=torch.float64) if torch.is_tensor(x)       # noqa: E731  (a Python number would
                          else torch.tensor(float(x), dtype=torch.float64, device=dev)).reshape(())   # become f32 through as_tensor)
        c = torch.stack([as64(n_valid), as64(n_crit)]).round().to(torch.int64)
        self.flat[-4:] = torch.stack([c >> _SPLIT, c & _LOW], dim=1).reshape(4).to(torch.float32)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        w = self.flat[-4:].to(torch.float64).reshape(2, 2)
        tot = w[:, 0] * float(1 << _SPLIT) + w[:, 1]                      # exact global counts
        self.counts = tot
        d = tot.to(torch.float32)
        self.flat[:self.n_pol].div_(d[0])
        self.flat[self.n_pol:self.n_pol + self.n_cri].div_(d[1])
        for k, (p, v) in enumerate(zip(ps, self.views)):
            if k not in absent:
                p.grad = v
        return tot


def allreduce_sum_grads(policy_params, critic_params, n_valid, n_crit, group=None, bucket=None):
    """Gradients were produced from SUM losses on every rank.  After one all-reduce(sum) of the flat
    bucket, dividing the policy part by the global number of valid steps and the critic part by the
    global number of padded steps gives exactly the gradient of the single-process mean losses
    (centralized_ma_ppo.py:437-438, comm_base_critic.py:88-89) - not a mean of per-rank means.
    One-shot form (builds a bucket, returns the global counts as Python floats = one host read); the training loop
    keeps a GradBucket and calls its ``allreduce``, which stays on the device."""
    b = bucket if bucket is not None else GradBucket(policy_params, critic_params)
    tot = b.allreduce(n_valid, n_crit, group=group)
    nv, nc = tot.tolist()
    return float(nv), float(nc)
