"""End-to-end sanity: a few dozen PPO epochs on the headline env (device sampler + CentralizedMAPPO.train_once with every fused
path on) - the average return and capture count should climb.  python tools/learn_check.py [epochs] [envs] [config]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from com_marl_amd import envs as E, nets
from com_marl_amd.algos import CentralizedMAPPO
from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = dict(bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "pp_map10"])
dev = torch.device("cuda:0")
env = E.GridEnvBatch(cfg["scenario"], bench.env_params(cfg), B, device=dev, seed=1)
spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
torch.manual_seed(1)
policy = nets.CommCategoricalMLPPolicy(spec, n_agents=env.N, device=dev)
policy.set_rng(1, env_id_offset=0)
critic = nets.CommBaseCritic(spec, n_agents=env.N, device=dev)
mpl = cfg["max_env_steps"]
algo = CentralizedMAPPO(env_spec=spec, policy=policy, baseline=critic, max_path_length=mpl, discount=0.99, center_adv=True,
                        positive_adv=False, gae_lambda=0.97, policy_ent_coeff=0.1, entropy_method="regularized", clip_grad_norm=7,
                        optimization_n_minibatches=3, optimization_mini_epochs=10, device=dev)

class Shell:
    def __init__(self, batch, spec):
        self.batch, self.spec, self.bound_return = batch, spec, 0.0

smp = CentralizedMAOnPolicyVectorizedSampler(algo, Shell(env, spec), n_envs=B)
smp.start_worker()
t0 = time.time()
for ep in range(epochs):
    paths = smp.obtain_samples(ep, batch_size=B * env.N * mpl)
    algo.train_once(itr=ep, paths=paths)
    s = algo.stats
    print(f"epoch {ep:3d}  return {s['AverageReturn']:8.2f}  captures {s['AverageCaptureCount']:6.2f}  success {s['SuccessRate']:.3f}  "
          f"steps {s['AverageStepCount']:6.1f}  kl {s['KL']:.2e}  entropy {s['Entropy']:.3f}  ({time.time() - t0:.0f} s)", flush=True)
