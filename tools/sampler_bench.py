"""Time CentralizedMAOnPolicyVectorizedSampler.obtain_samples at the headline config (4096 envs, one full horizon per env):
    python tools/sampler_bench.py [epochs]      (COMMARL_SAMPLER_GRAPH=0 / COMMARL_PERSISTENT=0 for the A/B forms)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from com_marl_amd import envs as E, nets
from com_marl_amd.algos import CentralizedMAPPO
from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler, tabular

c = dict(bench.CONFIGS["pp_map10"])
B = int(os.environ.get("ENVS", c["envs"]))
env = E.GridEnvBatch("pp", bench.env_params(c), B, device="cuda:0", seed=1)
spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
pol = nets.CommCategoricalMLPPolicy(spec, n_agents=env.N, device="cuda:0")
crit = nets.CommBaseCritic(spec, n_agents=env.N, device="cuda:0")
algo = CentralizedMAPPO(env_spec=spec, policy=pol, baseline=crit, max_path_length=200, discount=0.99, gae_lambda=0.97, device="cuda:0")


class Shell:
    def __init__(self): self.batch, self.spec, self.bound_return = env, spec, 0.0


smp = CentralizedMAOnPolicyVectorizedSampler(algo, Shell(), n_envs=B)
smp.start_worker()
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    paths = smp.obtain_samples(ep, batch_size=B * env.N * 200)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"epoch {ep}: {dt * 1e3:.2f} ms, {smp.last_steps} steps ({smp.last_steps_run} run), {dt / smp.last_steps_run * 1e6:.1f} us/step, "
          f"gpu {1e3 * (tabular.rows['PolicyExecTime'] + tabular.rows['EnvExecTime']):.2f} ms, graphs {len(smp.engine._graphs)}", flush=True)
