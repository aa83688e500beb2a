import sys
sys.path.insert(0, '.')
import numpy as np, torch
from com_marl_amd import envs as E, nets
for N, d in ((96, 77), (128, 53), (200, 53)):
    spec = E.EnvSpec(E._Box(np.zeros(d * N), np.ones(d * N)), E._Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=N, device="cuda:0")
    obs = torch.rand(3, N * d, device="cuda:0")
    adj = (torch.rand(3, N, N, device="cuda:0") < 0.5).float()
    try:
        a, p, at = pol.act_device(obs, None, adj, None, policy_step=0)
        print(N, "act ok", p.shape)
    except Exception as e:
        print(N, "act fail:", str(e)[:200])
    try:
        with torch.no_grad():
            v = crit.forward(obs, None, adj, None)
        print(N, "critic ok", v.shape)
    except Exception as e:
        print(N, "critic fail:", str(e)[:200])
    try:
        pr, _ = pol._probs(obs, None, adj, torch.ones(3, 2, N, N, device="cuda:0"))
        pr.sum().backward()
        print(N, "train path ok")
    except Exception as e:
        print(N, "train fail:", str(e)[:200])
