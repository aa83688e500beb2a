"""Two-rank rehearsal of the distributed branch of CentralizedMAPPO.train_once (com-marl_amd/algos.py: SUM losses,
one all-reduce of the flat gradient bucket per optimiser step, division by the GLOBAL counts, clip after the reduce,
equal optimiser-step counts) - SURVEY.md §8(e).  Both ranks are fresh processes on GPU 0 talking gloo (the GPU box
has one card; RCCL needs one card per rank).

Every process rolls out the SAME union batch (2B envs, global env ids -> identical trajectories), rank r then trains on
the paths of ITS env shard only; the parent trains one process on the union.  One minibatch per mini-epoch, so the
union's optimiser steps see exactly the ranks' paths together; the parameters after 3 steps must agree to float
summation order (the reference's loss is a mean over valid steps / padded steps of the whole batch,
centralized_ma_ppo.py:437-438, comm_base_critic.py:88-89 - so it must be sum-of-grads / global counts, not a mean of
per-rank means: the shards below are ragged on purpose).

    python -m tests.dist_train_child <rank> <world> <port> <outdir>      (one rank)
"""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_UNION, MPL, SEED = 96, 9, 11
SPLIT = 37                                  # rank 0 owns global envs [0, 37), rank 1 [37, 96): ragged on purpose


def _train(rank, world, port):
    import torch
    from com_marl_amd import envs as E, nets
    from com_marl_amd.algos import CentralizedMAPPO
    from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler, PathBatch
    torch.cuda.set_device(0)
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    params = dict(load=2, max_env_steps=MPL, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=10, Rsen=1,
                  n_agents=4, n_preys=4, n_gcn_layers=2, mode="train", trRcom=2, trpl=0.3, seed=SEED)   # range adj + IID
    env = E.PredatorPreyWrapper(centralized=True, params=params, n_envs=B_UNION, device="cuda:0")
    torch.manual_seed(SEED)
    pol = nets.CommCategoricalMLPPolicy(env.spec, n_agents=4, device="cuda:0")
    crit = nets.CommBaseCritic(env.spec, n_agents=4, device="cuda:0")
    pol.set_rng(SEED)
    algo = CentralizedMAPPO(env_spec=env.spec, policy=pol, baseline=crit, max_path_length=MPL, discount=0.99,
                            center_adv=True, positive_adv=False, gae_lambda=0.97, policy_ent_coeff=0.1,
                            entropy_method="regularized", clip_grad_norm=0.05, optimization_n_minibatches=1,
                            optimization_mini_epochs=int(os.environ.get("COMMARL_DIST_STEPS", "3")), device="cuda:0")
    if os.environ.get("COMMARL_DIST_NOCLIP"):
        algo._lr_clip_range = 1e9                     # diagnostic: the surrogate without its clip (no gradient discontinuity)
    smp = CentralizedMAOnPolicyVectorizedSampler(algo, env, n_envs=B_UNION)
    smp.start_worker()
    paths = smp.obtain_samples(0, batch_size=B_UNION * 4 * MPL)
    n_union = len(paths)
    if world > 1:
        lo, hi = (0, SPLIT) if rank == 0 else (SPLIT, B_UNION)
        sel = (paths.env_idx >= lo) & (paths.env_idx < hi)
        paths = PathBatch(paths.engine, paths.env_idx[sel], paths.start[sel], paths.length[sel], 4)
    np.random.seed(3)
    algo.train_once(itr=0, paths=paths)
    out = {"pol." + k: v.detach().cpu().numpy() for k, v in pol.state_dict().items()}
    out.update({"crit." + k: v.detach().cpu().numpy() for k, v in crit.state_dict().items()})
    out["n_paths"], out["n_union"] = len(paths), n_union
    out["grad_norm"] = algo.stats["GradNorm"]
    out["loss_after"] = algo.stats["LossAfter"]
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return out


def _compare(steps, noclip):
    """One comparison: two fresh ranks vs the union trained in a fresh single process (all three are children)."""
    import tempfile
    tmp = tempfile.mkdtemp()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, COMMARL_DIST_STEPS=str(steps))
    if noclip:
        env["COMMARL_DIST_NOCLIP"] = "1"
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dist_train_child", str(r), "2", str(port), tmp], cwd=ROOT, env=env)
             for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], f"rank exit codes {rcs}"
    rc = subprocess.run([sys.executable, "-m", "tests.dist_train_child", "9", "1", "0", tmp], cwd=ROOT, env=env, timeout=600).returncode
    assert rc == 0, f"union run exit code {rc}"
    ranks = [dict(np.load(os.path.join(tmp, f"rank{r}.npz"))) for r in range(2)]
    ref = dict(np.load(os.path.join(tmp, "rank9.npz")))
    assert int(ranks[0]["n_paths"]) + int(ranks[1]["n_paths"]) == int(ref["n_paths"]) == int(ranks[0]["n_union"])
    assert int(ranks[0]["n_paths"]) != int(ranks[1]["n_paths"])          # ragged shards: mean-of-means would differ
    worst = 0.0
    for k, v in ref.items():
        if not k.startswith(("pol.", "crit.")):
            continue
        if os.environ.get("COMMARL_DIST_DEBUG"):
            print(f"{k:70s} rank-vs-union {np.abs(ranks[0][k] - v).max():.3e}")
        np.testing.assert_array_equal(ranks[0][k], ranks[1][k], err_msg=f"replicas diverged: {k}")
        # Adam steps of lr 3e-4: a parameter moved by ~1e-3; agreement to summation order of the f32 gradient sums
        np.testing.assert_allclose(ranks[0][k], v, rtol=0, atol=2e-6, err_msg=f"{k} (steps={steps}, noclip={noclip})")
        worst = max(worst, float(np.abs(ranks[0][k] - v).max()))
    # clip-after-reduce: the reported norm is that of the GLOBAL gradient on every rank
    np.testing.assert_allclose(float(ranks[0]["grad_norm"]), float(ref["grad_norm"]), rtol=1e-4)
    np.testing.assert_allclose(float(ranks[1]["grad_norm"]), float(ref["grad_norm"]), rtol=1e-4)
    # GradNorm is recorded after the clip (centralized_ma_ppo.py:253-256): ~0.05 means the clip was active on every step,
    # so clipping before instead of after the reduce would have changed the parameters compared above
    assert abs(float(ref["grad_norm"]) - 0.05) < 1e-3, ref["grad_norm"]
    return dict(n_paths=[int(ranks[0]["n_paths"]), int(ranks[1]["n_paths"])], max_param_diff=worst, grad_norm=float(ref["grad_norm"]))


def run_two_rank_case():
    """Two comparisons.  (a) the update as configured, two optimiser steps.  (b) five optimiser steps with PPO's ratio clip
    switched off: min(r.A, clip(r).A) has a gradient discontinuity where a sample's ratio crosses 1 +- 0.1, so from the
    third step on a 1e-8 parameter difference (float summation order) can flip one borderline sample in or out of the
    gradient and move parameters by ~1e-5 - measured: 2.3e-5 after three clipped steps, 3e-8 after five unclipped ones.
    That sensitivity is PPO's, on one GPU as on two; (b) shows the exchange step itself stays exact over a longer chain."""
    a = _compare(2, False)
    b = _compare(5, True)
    return dict(n_paths=a["n_paths"], max_param_diff=max(a["max_param_diff"], b["max_param_diff"]), grad_norm=a["grad_norm"],
                clipped_2_steps=a["max_param_diff"], unclipped_5_steps=b["max_param_diff"])


if __name__ == "__main__":
    r, w, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    res = _train(r, w, port)
    np.savez(os.path.join(outdir, f"rank{r}.npz"), **res)
