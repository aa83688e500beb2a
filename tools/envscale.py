"""Env-step kernel time vs batch size / lanes-per-env (graph-replayed launches, HIP events)."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np, torch
    from com_marl_amd import envs as E
    import bench
    def timeit(fn, inner=20, reps=10):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / (inner * reps) * 1e3
    cfgname = sys.argv[2]
    c = dict(bench.CONFIGS[cfgname])
    for B in [int(x) for x in sys.argv[3].split(",")]:
        env = E.GridEnvBatch(c["scenario"], bench.env_params(c), B, device="cuda:0", seed=1)
        env.reset_all()
        act = torch.randint(0, 5, (B, env.N), dtype=torch.int32, device="cuda:0")
        t = timeit(lambda: env.step_device(act))
        env.check_status()
        print(f"{cfgname} LPE={os.environ.get('COMMARL_ENV_LPE','auto')} B={B}: env step {t:.1f} us", flush=True)
elif os.environ.get("PHASES"):
    for k in (1, 2, 3, 4, 5, 6, 7, 0):
        print("stop", k, flush=True)
        subprocess.run([sys.executable, __file__, "child", os.environ.get("CFG", "pp_map10"), os.environ.get("BS", "4096")],
                       env=dict(os.environ, COMMARL_ENV_STOP=str(k)))
else:
    for cfg, Bs, lpes in (("pp_map10", "1024,4096,16384", ("16", "32", "64")), ("co_map20", "512,2048,8192", ("32", "64")),
                          ("pp_map30", "256,1024,4096", ("64",)), ("co_map30", "256,1024,4096", ("64",))):
        for lpe in lpes:
            subprocess.run([sys.executable, __file__, "child", cfg, Bs], env=dict(os.environ, COMMARL_ENV_LPE=lpe))
