"""Where the host time of one PPO epoch goes at the reference's batch (64 envs, 30 000 agent-steps): stage times from
algo.stats and a cProfile of one train_once.   python tools/update_hostprof.py [envs] [batch_size]"""
import cProfile
import importlib
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

bench = importlib.import_module("bench")
E = importlib.import_module("com_marl_amd.envs")
nets = importlib.import_module("com_marl_amd.nets")
algos = importlib.import_module("com_marl_amd.algos")
sampler = importlib.import_module("com_marl_amd.sampler")

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
dev = torch.device("cuda:0")
c = bench.CONFIGS["pp_map10"]
env = E.GridEnvBatch(c["scenario"], bench.env_params(c), envs, device=dev, seed=1)
spec = E.EnvSpec(E._Box([0.0] * (env.N * env.d), [1.0] * (env.N * env.d)), E._Discrete(5))
torch.manual_seed(1)
pol = nets.CommCategoricalMLPPolicy(spec, n_agents=env.N, device=dev)
crit = nets.CommBaseCritic(spec, n_agents=env.N, device=dev)
pol.set_rng(1)
algo = algos.CentralizedMAPPO(env_spec=spec, policy=pol, baseline=crit, max_path_length=c["max_env_steps"], discount=0.99, center_adv=True,
                              positive_adv=False, gae_lambda=0.97, policy_ent_coeff=0.1, entropy_method="regularized", clip_grad_norm=7,
                              optimization_n_minibatches=3, optimization_mini_epochs=10, device=dev)


class Shell:
    def __init__(self, batch, spec):
        self.batch, self.spec, self.bound_return = batch, spec, 0.0


smp = sampler.CentralizedMAOnPolicyVectorizedSampler(algo, Shell(env, spec), n_envs=envs)
smp.start_worker()
for ep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    paths = smp.obtain_samples(ep, batch_size=bs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if ep == 3:
        pr = cProfile.Profile()
        pr.enable()
    algo.train_once(itr=ep, paths=paths)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if ep == 3:
        pr.disable()
    s = algo.stats
    print(f"epoch {ep}: rollout {1e3*(t1-t0):.1f} ms, train_once {1e3*(t2-t1):.1f} ms (optimiser loop {1e3*s['EpochTime']:.1f} ms), "
          f"paths {s['NumTrajs']//env.N}", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
