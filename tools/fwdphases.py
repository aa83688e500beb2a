"""Cumulative phase timing of the fused policy kernel: re-runs it with COMMARL_FWD_STOP=k."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np, torch
    from com_marl_amd import envs as E, nets
    import bench
    c = dict(bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "pp_map10"])
    B = int(os.environ.get("ENVS", c["envs"]))
    env = E.GridEnvBatch(c["scenario"], bench.env_params(c), B, device="cuda:0", seed=1)
    spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=env.N, device="cuda:0")
    env.reset_all()
    adj = None if env.adj_const else env.dist_adj
    ch = None if env.ch_const else env.channels
    fn = lambda: pol.act_device(env.obs.view(B, -1), None, adj, ch, policy_step=0)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); e1.synchronize()
    print(f"stop={os.environ.get('COMMARL_FWD_STOP','0')}: {e0.elapsed_time(e1)/200*1e3:.1f} us")
else:
    for cfg in os.environ.get("CFGS", "pp_map10").split(","):
        for rows in os.environ.get("ROWS", "0").split(","):
            print("rows per workgroup", rows, flush=True)
            for k in [int(x) for x in os.environ.get("STOPS", "1,2,3,4,5,6,7,0").split(",")]:
                subprocess.run([sys.executable, __file__, "child", cfg],
                               env=dict(os.environ, COMMARL_FWD_STOP=str(k), COMMARL_FWD_ROWS=rows))
