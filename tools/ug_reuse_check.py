"""Deviation of the graph-replayed PPO update from the eager one over 3 epochs (third epoch = second one's batch again), with and
without taking over earlier epochs' graphs:  python tools/ug_reuse_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402
import test_hip_ppo_parity as T  # noqa: E402

runs = {}
for tag, mode, reuse in (("eager", "0", "1"), ("eager2", "0", "1"), ("graph_reuse", "1", "1"), ("graph_fresh", "1", "0")):
    os.environ["COMMARL_UPDATE_GRAPH"] = mode
    os.environ["COMMARL_UPDATE_GRAPH_REUSE"] = reuse
    env, pol, crit, algo, smp = T._small_setup(torch, B=64, mpl=15, scenario="pp")
    per_epoch = []
    for itr in range(3):
        if itr < 2:
            paths = smp.obtain_samples(itr, batch_size=64 * env.n_agents * 15)
        np.random.seed(11 + min(itr, 1))
        algo.train_once(itr=itr, paths=paths)
        per_epoch.append({k: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()})
    runs[tag] = per_epoch
for tag in ("eager2", "graph_reuse", "graph_fresh"):
    print(tag, ["%.2e" % max(float(np.abs(a[k] - b[k]).max()) for k in a) for a, b in zip(runs["eager"], runs[tag])], flush=True)
