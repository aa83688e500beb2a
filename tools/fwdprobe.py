"""In-kernel phase clocks of the fused policy forward (COMMARL_FWD_PROBE=1): one eager launch, stamps of thread 0 of
every workgroup averaged on the host and printed to stderr by the library."""
import os, sys
os.environ["COMMARL_FWD_PROBE"] = "1"
sys.path.insert(0, '.')
import numpy as np, torch
from com_marl_amd import envs as E, nets
import bench
for name in (sys.argv[1:] or ["pp_map10"]):
    c = dict(bench.CONFIGS[name])
    B = c["envs"]
    env = E.GridEnvBatch(c["scenario"], bench.env_params(c), B, device="cuda:0", seed=1)
    spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=env.N, device="cuda:0")
    env.reset_all()
    adj = None if env.adj_const else env.dist_adj
    ch = None if env.ch_const else env.channels
    print(name, flush=True)
    pol.act_device(env.obs.view(B, -1), None, adj, ch, policy_step=0)
    torch.cuda.synchronize()
