import os, sys; sys.path.insert(0, '.')
import numpy as np, torch
from tests import dist_train_child as D
from com_marl_amd import envs as E, nets
from com_marl_amd.algos import CentralizedMAPPO
from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler, PathBatch
params = dict(load=2, max_env_steps=D.MPL, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=10, Rsen=1,
              n_agents=4, n_preys=4, n_gcn_layers=2, mode="train", trRcom=2, trpl=0.3, seed=D.SEED)
env = E.PredatorPreyWrapper(centralized=True, params=params, n_envs=D.B_UNION, device="cuda:0")
torch.manual_seed(D.SEED)
pol = nets.CommCategoricalMLPPolicy(env.spec, n_agents=4, device="cuda:0"); crit = nets.CommBaseCritic(env.spec, n_agents=4, device="cuda:0")
pol.set_rng(D.SEED)
algo = CentralizedMAPPO(env_spec=env.spec, policy=pol, baseline=crit, max_path_length=D.MPL, discount=0.99, center_adv=True, positive_adv=False,
                        gae_lambda=0.97, policy_ent_coeff=0.1, entropy_method="regularized", clip_grad_norm=0.05,
                        optimization_n_minibatches=1, optimization_mini_epochs=3, device="cuda:0")
smp = CentralizedMAOnPolicyVectorizedSampler(algo, env, n_envs=D.B_UNION); smp.start_worker()
paths = smp.obtain_samples(0, batch_size=D.B_UNION * 4 * D.MPL)
obs, avail, actions, rewards, valids, baselines, returns, da, ch = algo.process_samples(0, paths)
adv = algo._advantages(rewards, baselines, valids)
with torch.no_grad(): oll = algo._old_log_likelihood(obs, actions, da, ch)
def grads(idx, reduce):
    pol.zero_grad()
    sl = lambda x: None if x is None else x[idx]
    out = algo._compute_loss(0, obs[idx], None, actions[idx], rewards[idx], valids[idx], baselines[idx], sl(da), sl(ch), adv[idx], oll[idx], reduce=reduce)
    if reduce: out.backward(); return {n: p.grad.clone() for n, p in pol.named_parameters()}, None
    tot, cnt = out; tot.backward(); return {n: p.grad.clone() for n, p in pol.named_parameters()}, float(cnt)
P = obs.shape[0]; allidx = torch.arange(P, device="cuda"); sel = paths.env_idx < D.SPLIT
for fused in ("1", "0"):
    os.environ["COMMARL_FUSED_LINEAR"] = fused
    gu, _ = grads(allidx, True)
    ga, ca = grads(allidx[sel], False); gb, cb = grads(allidx[~sel], False)
    print("fused =", fused)
    for n in gu:
        s = (ga[n] + gb[n]) / (ca + cb)
        print(f"  {n:60s} |g|max {float(gu[n].abs().max()):.3e}  shards-vs-union {float((s - gu[n]).abs().max() / gu[n].abs().max()):.2e}")
    if fused == "1": g1 = gu
    else:
        for n in gu: print(f"  fused-vs-unfused {n:50s} {float((g1[n]-gu[n]).abs().max()/gu[n].abs().max()):.2e}")
