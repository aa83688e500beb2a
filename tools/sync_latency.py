"""Where the fixed ~150 us of a short timed region goes: host-visible latency of launch + synchronize on this box."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
def timeit(f, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6
print("synchronize on an idle device: %.1f us" % timeit(lambda: torch.cuda.synchronize()))
def k_sync():
    x.add_(1.0); torch.cuda.synchronize()
print("1 small kernel + synchronize: %.1f us" % timeit(k_sync))
def k_poll():
    x.add_(1.0); ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
print("1 small kernel + event-poll + synchronize: %.1f us" % timeit(k_poll))
def k20_sync():
    for _ in range(20): x.add_(1.0)
    torch.cuda.synchronize()
print("20 small kernels + synchronize: %.1f us" % timeit(k20_sync))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): x.add_(1.0)
def g_sync():
    g.replay(); torch.cuda.synchronize()
print("graph(20 small kernels) + synchronize: %.1f us" % timeit(g_sync))
def g_poll():
    g.replay(); ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
print("graph(20 small kernels) + event-poll + synchronize: %.1f us" % timeit(g_poll))
g80 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g80):
    for _ in range(80): x.add_(1.0)
def g80_poll():
    g80.replay(); ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
print("graph(80 small kernels) + event-poll + synchronize: %.1f us" % timeit(g80_poll))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); ev0.record(); g80.replay(); ev1.record(); torch.cuda.synchronize()
print("graph(80) device time by events: %.1f us" % (ev0.elapsed_time(ev1) * 1e3))
