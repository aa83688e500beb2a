import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from com_marl_amd import envs as E
import bench
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cfgname in ("pp_map10", "co_map20", "pp_map30"):
    c = dict(bench.CONFIGS[cfgname])
    for B in (64, 256, 1024, 4096, 16384):
        if cfgname != "pp_map10" and B > 4096: continue
        env = E.GridEnvBatch(c["scenario"], bench.env_params(c), B, device="cuda:0", seed=1)
        env.reset_all()
        act = torch.randint(0, 5, (B, env.N), dtype=torch.int32, device="cuda:0")
        t = timeit(lambda: env.step_device(act))
        env.check_status()
        print(f"{cfgname} B={B}: env step {t:.1f} us", flush=True)
