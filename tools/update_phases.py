"""Where one PPO update (CentralizedMAPPO.train_once, headline config) spends its time: synchronising timers around the
phases of train_once (so the sum is a little above the un-instrumented update)."""
import collections, os, sys, time
sys.path.insert(0, '.')
import torch
import bench
from com_marl_amd import algos

def main():
    acc = collections.OrderedDict()
    def wrap(obj, name, label=None):
        f = getattr(obj, name)
        def g(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize(); acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, g)
    C = algos.CentralizedMAPPO
    for n in ("process_samples", "_advantages", "_old_log_likelihood", "_diagnostics", "_log_performance", "_baseline_loss"):
        wrap(C, n)
    orig_cl = C._compute_loss
    def cl(self, *a, reduce=True, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig_cl(self, *a, reduce=reduce, **k)
        torch.cuda.synchronize()
        key = "_compute_loss(full batch)" if reduce else "_compute_loss(minibatch fwd)"
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
        return r
    C._compute_loss = cl
    orig_bw = torch.Tensor.backward
    def bw(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig_bw(self, *a, **k)
        torch.cuda.synchronize(); acc["backward"] = acc.get("backward", 0.0) + time.perf_counter() - t0
        return r
    torch.Tensor.backward = bw
    from com_marl_amd import optim
    wrap(optim.Adam, "step", "optimizer.step")
    orig_to = C.train_once
    calls = [0]
    def to(self, *a, **k):
        r = orig_to(self, *a, **k)
        calls[0] += 1
        if calls[0] == 1:
            acc.clear()                                   # epoch 0 = warm-up (allocations, first-use set-up)
        return r
    C.train_once = to
    sys.argv = [sys.argv[0], "--steps", "100", "--warmup", "20", "--no-cpu-baseline"]
    orig_tl = None
    from com_marl_amd import train_bench
    f = train_bench.train_loop_measurement
    def tl(*a, **k):
        k["epochs"] = 2
        r = f(*a, **k)
        return r
    train_bench.train_loop_measurement = tl
    bench.main()
    tot = sum(acc.values())
    print("phase totals over the 2 timed epochs, seconds:")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {k:34s} {v:8.4f}  {100 * v / tot:5.1f} %")
main()
