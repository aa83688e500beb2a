"""Train-loop leg of bench.py: one PPO epoch = rollout of one full horizon
on every env + CentralizedMAPPO.train_once at the reference schedule (3 minibatches x 10
mini-epochs, one RCCL gradient all-reduce per optimiser step when world > 1)."""
import time

import torch


def train_loop_measurement(env, policy, cfg, spec, world, rank, dev, seed, epochs=2, kind="commdp", batch_size=None):
    from . import nets
    from .algos import CentralizedMAPPO
    from .sampler import CentralizedMAOnPolicyVectorizedSampler

    class _Shell:                                   # what the sampler needs from the env object
        def __init__(self, batch, spec):
            self.batch, self.spec, self.bound_return = batch, spec, 0.0
    mpl = cfg["max_env_steps"]
    torch.manual_seed(seed + 1)
    if kind == "cent":                               # runner_pp_cent.py:61-63
        critic = nets.GaussianMLPBaseline(env_spec=spec, hidden_sizes=(64, 64, 64), device=dev)
    else:
        critic = nets.CommBaseCritic(spec, n_agents=env.N, device=dev)
    algo = CentralizedMAPPO(env_spec=spec, policy=policy, baseline=critic, max_path_length=mpl, discount=0.99,
                            center_adv=True, positive_adv=False, gae_lambda=0.97, policy_ent_coeff=0.1,
                            entropy_method="regularized", clip_grad_norm=7, optimization_n_minibatches=3,
                            optimization_mini_epochs=10, device=dev)
    smp = CentralizedMAOnPolicyVectorizedSampler(algo, _Shell(env, spec), n_envs=env.B)
    smp.start_worker()
    bs = batch_size or env.B * env.N * mpl            # default: every env contributes at least one full path
    t_roll = t_upd = 0.0
    steps = env_steps = 0
    stats = {}
    for ep in range(epochs + 1):                      # epoch 0 = warm-up (allocations, autotune)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        paths = smp.obtain_samples(ep, batch_size=bs)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        algo.train_once(itr=ep, paths=paths)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        t2 = time.perf_counter()
        if ep:
            t_roll += t1 - t0
            t_upd += t2 - t1
            steps += smp.last_steps
            env_steps += smp.last_steps * env.B
            stats = dict(algo.stats)
    tot = torch.tensor([t_roll + t_upd], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.MAX)
    total = float(tot.item())
    return dict(value=world * env_steps / total, unit="env-steps/s", epochs=epochs,
                rollout_s_per_epoch=t_roll / epochs, update_s_per_epoch=t_upd / epochs,
                env_steps_per_epoch_per_gpu=env_steps // epochs, paths_per_epoch=stats.get("NumTrajs", 0) // env.N,
                envs_per_gpu=env.B, batch_size_agent_steps=bs,
                schedule="3 minibatches x 10 mini-epochs, Adam lr 3e-4, clip 0.1, grad all-reduce per step" if world > 1
                else "3 minibatches x 10 mini-epochs, Adam lr 3e-4, clip 0.1",
                loss_before=stats.get("LossBefore"), loss_after=stats.get("LossAfter"), kl=stats.get("KL"))
