"""Training-size forward of the teams-of-4 nets (cm_policy_forward_saved / cm_critic_forward_saved, 274 k envs): time per call;
COMMARL_FWD_OCC3_MIN=0 selects the two-workgroups-per-CU build for comparison."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from com_marl_amd import nets
from com_marl_amd.envs import EnvSpec, _Box, _Discrete
N, d, S = 4, 21, int(os.environ.get("S", 274000))
spec = EnvSpec(_Box(np.zeros(N * d), np.ones(N * d)), _Discrete(5))
torch.manual_seed(0)
pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, device="cuda:0")
crit = nets.CommBaseCritic(spec, n_agents=N, device="cuda:0")
obs = torch.rand(S, N * d, device="cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
tp = t(lambda: pol._probs(obs, None, None, None))
tc = t(lambda: crit._values_grad(obs, None, None))
with torch.no_grad():
    tn = t(lambda: pol.evaluate_nograd(obs, None, None))
print(f"occ3_min={os.environ.get('COMMARL_FWD_OCC3_MIN', 'default')}: policy training forward {tp:.3f} ms, critic {tc:.3f} ms, policy no-save forward {tn:.3f} ms")
