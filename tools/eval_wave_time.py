"""Time of policy.evaluate_nograd at the PPO batch of the headline config, wave-owned against workgroup-tiled kernel."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

nets = importlib.import_module("com_marl_amd.nets")
E = importlib.import_module("com_marl_amd.envs")
N, d, dev = 4, 21, torch.device("cuda:0")
spec = E.EnvSpec(E._Box(np.zeros(N * d), np.ones(N * d)), E._Discrete(5))
pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, device=dev)
P, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4113, 200
obs = torch.rand(P, T, N * d, device=dev)
for mn in ("16384", "1000000000"):
    os.environ["COMMARL_TRAIN_FWD_WAVE_MIN"] = mn
    for _ in range(2):
        pol.evaluate_nograd(obs, None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lg, pr = pol.evaluate_nograd(obs, None, None)
    torch.cuda.synchronize()
    print(f"wave_min={mn}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per evaluate_nograd of {P * T} envs; probs sum {float(pr.sum()):.3f}", flush=True)
