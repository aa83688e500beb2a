"""Times the reference's own Python sampler in the build container (the reference cannot travel to the GPU box):
CentralizedMAOnPolicyVectorizedSampler.obtain_samples, n_envs=1, single process, one torch thread.
Usage: python tools/time_reference.py   (needs /root/reference)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import random
import numpy as np
import torch
import ref_loader
import gen_golden as G

torch.set_num_threads(1)
ns = ref_loader.load_reference_ppo(ref_loader.load_reference())
for name, scen, params in (("PP map10 N=4 (config 1/2)", "pp", G.pp_params(10, 1, 0.04, 2)),
                           ("CO map20 N=24 (config 3)", "co", G.co_params(20, 2, 0.06)),
                           ("PP map30 N=72 (config 4)", "pp", G.pp_params(30, 2, 0.08, 4)),
                           ("CO map30 N=54 IID (config 5)", "co", G.co_params(30, 2, 0.06, loss=0.3))):
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    cls = ns.PredatorPreyWrapper if scen == "pp" else ns.CoverageWrapper
    env = cls(centralized=True, params=dict(params))
    o = env.reset()
    spec = ref_loader.make_env_spec(len(o))

    class Shell:
        def __init__(s, e): s.__dict__['_e'] = e; s.__dict__['spec'] = spec
        def __getattr__(s, k): return getattr(s.__dict__['_e'], k)
    n = params["n_agents"]
    pol = ns.CommCategoricalMLPPolicy(spec, n_agents=n)
    crit = ns.CommBaseCritic(spec, n_agents=n)
    algo = ns.CentralizedMAPPO(env_spec=spec, policy=pol, baseline=crit, max_path_length=params["max_env_steps"], device="cpu")
    smp = ns.ReferenceSampler(algo, Shell(env), n_envs=1)
    smp.start_worker()
    T = params["max_env_steps"]
    target = 3 * T * n                      # ~3 episodes
    t0 = time.perf_counter()
    paths = smp.obtain_samples(0, batch_size=target)
    dt = time.perf_counter() - t0
    steps = sum(len(p["rewards"]) for p in paths)
    print(f"{name}: {steps} env-steps in {dt:.2f} s = {steps / dt:.0f} env-steps/s (reference Python sampler, 1 core)", flush=True)
