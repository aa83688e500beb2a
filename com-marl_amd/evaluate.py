"""Greedy evaluation: ``eval_model`` with the reference's signature and return contract
(exp_runners/predatorprey/eval_pp.py:9-104; the coverage twin is the same loop), run as ONE batched
device rollout instead of a Python loop over episodes.

The reference plays ``n_eval_episodes`` episodes one after the other on one env; here the B envs of
the wrapper each play their *first* episode after a reset, all at once (policy forward with
``greedy=True`` + env step kernel, captured in the same RolloutEngine the sampler uses), and as many
such rounds are played as it takes to collect ``n_eval_episodes``.  Per episode the function returns
exactly what the reference returns: the per-step ``success`` list, the per-step ``VECTORS`` lists, the
per-episode sums (mean for ``nodeDeg``) and ``env.bound_return``.

Differences a maintainer should know (documented, not silent):
  * episodes are independent Philox streams (global env id, round) rather than one sequential
    generator re-seeded by ``fix_randomness(seed)``; ``seed`` re-keys nothing here - the wrapper's own
    ``seed=`` decides the streams;
  * ``render`` / ``inspect_steps`` are not supported (UI is out of scope) and raise;
  * with a range-limited adjacency, ``nodeDeg`` of an episode's terminal step is the degree *before*
    that step (the env auto-resets on done, vec_env_executor.py:36-43, so the post-step graph of a
    finished episode is never materialised).
"""
import numpy as np
import torch

from .rollout import RolloutEngine

VECTORS = ['reward', 'capture_cnt', 'step_cnt', 'move_cnt', 'penalty_cnt', 'nodeDeg', 'variable', 'vars2']   # testing.py:209


def _first_episodes(eng, base, T, greedy):
    """reset + T steps; returns host arrays for every env's first episode."""
    eng.reset()
    eng.fork()
    for t in range(T):
        eng.step(t, greedy=greedy)
    eng.join()
    eng.bump(T)
    base.batch.check_status()
    pl = eng.path_len[:T]                                                  # [T,B]
    ended = pl > 0
    # first t at which each env finished (T-1 when the loop limit cut it, eval_pp.py:73)
    first = torch.where(ended.any(0), ended.to(torch.int32).argmax(0), torch.full_like(pl[0], T - 1)).long()
    h = dict(first=first.cpu().numpy(), reward=eng.reward64[:T].cpu().numpy(), details=eng.details[:T].cpu().numpy(),
             success=eng.success[:T].cpu().numpy(), done=eng.done[:T].cpu().numpy())
    if eng.dist_adj is not None:
        h["deg"] = eng.dist_adj[:T + 1].sum(-1).mean(-1).cpu().numpy()   # [T+1,B] ave_deg (env_communication.py:232)
    return h


def eval_model(env, policy, itr, n_eval_episodes=100, max_env_steps=200, eval_greedy=True, render=False,
               inspect_steps=False, seed=1, flag=None):
    """eval_pp.py:9.  -> (episode_data, epi_success, epi_rewards, bound_return)."""
    if render or inspect_steps:
        raise NotImplementedError("rendering is outside the MI355X path")
    if flag is not None and flag[0]:
        return None, None, None, None
    base = getattr(env, "env", env)
    batch = base.batch
    B, N = batch.B, batch.N
    pp = batch.scenario == "pp"
    T = int(max_env_steps)
    if T > batch.cfg.max_path_length:
        raise ValueError(f"max_env_steps={T} exceeds the env's max_path_length={batch.cfg.max_path_length}")
    env.eval_n_epi = 0
    policy.sync_weights()
    policy.reset([True] * B)
    eng = RolloutEngine(batch, policy, T, store_attn=False, store_probs=False)
    episode_data, epi_success = [], []
    epi_rewards = {vec: [] for vec in VECTORS}
    eval_rewards = []
    nA = float(N)
    while len(episode_data) < n_eval_episodes:
        h = _first_episodes(eng, base, T, bool(eval_greedy))
        for b in range(min(B, n_eval_episodes - len(episode_data))):
            n = int(h["first"][b]) + 1
            det = h["details"][:n, b].astype(np.float64)
            rew = h["reward"][:n, b]
            if "deg" in h:
                deg = np.concatenate([h["deg"][1:n, b], h["deg"][n - 1:n, b]])
            else:
                deg = np.full(n, N)
            if pp:                                                          # predator_prey.py:440-448
                cols = dict(capture_cnt=det[:, 0], move_cnt=det[:, 1] / nA, penalty_cnt=det[:, 2],
                            variable=det[:, 4] / nA, vars2=np.zeros(n))
            else:                                                           # coverage.py:308-315
                cols = dict(capture_cnt=det[:, 0] / nA, move_cnt=det[:, 1] / nA, penalty_cnt=det[:, 2] / nA,
                            variable=det[:, 4] / nA, vars2=det[:, 3] / nA)
            cols.update(reward=rew, step_cnt=np.ones(n), nodeDeg=deg)
            step_data = {vec: cols[vec].tolist() for vec in VECTORS}
            step_success = h["success"][:n, b].tolist()
            episode_data.append((step_success, step_data))
            epi_success.append(int(h["success"][n - 1, b]))
            for vec in VECTORS:
                epi_rewards[vec].append(float(np.mean(cols[vec]) if vec == 'nodeDeg' else np.sum(cols[vec])))
            eval_rewards.append(float(rew.sum()))
    env.eval_n_epi = len(episode_data)
    env.last_eval_average_reward = (sum(eval_rewards) / len(eval_rewards)) / base.bound_return    # eval_pp.py:95
    return episode_data, epi_success, epi_rewards, base.bound_return


def eval_model_co(env, policy, itr, **kwargs):
    """exp_runners/coverage/eval_co.py:9-101: same loop; the fourth return value is the per-episode
    list of ``env.bound_return`` (``epi_optRew``) instead of the scalar."""
    out = eval_model(env, policy, itr, **kwargs)
    if out[0] is None:
        return out
    return out[0], out[1], out[2], [out[3]] * len(out[0])


def eval_simple(args, env, algo, _eval=eval_model):
    """exp_runners/predatorprey/eval_pp.py:107-157 (`--mode eval`): ``args.n_eval_episodes`` episodes of at most
    ``args.max_env_steps`` steps with the algo's policy, printing the average length of the episodes that ended
    before the step limit.  The reference's loop exists to drive ``env.my_render`` (UI, out of scope): with
    ``args.render`` set this raises, otherwise the episodes are played as one batched device rollout (eval_model)."""
    import time
    start = time.time()
    data, _succ, _rew, _bound = _eval(env, algo.policy, 0, n_eval_episodes=args.n_eval_episodes,
                                      max_env_steps=args.max_env_steps, eval_greedy=bool(args.eval_greedy),
                                      render=bool(getattr(args, "render", False)),
                                      inspect_steps=bool(getattr(args, "inspect_steps", False)))
    traj_len = [len(step_success) for step_success, _ in data if len(step_success) < args.max_env_steps]
    print('Average trajectory length = {}'.format(np.mean(traj_len) if traj_len else float('nan')))
    print(f'test_time: {time.time() - start}')
    return traj_len


def eval_simple_co(args, env, algo):
    """exp_runners/coverage/eval_co.py:104-150: the same loop on the coverage env."""
    return eval_simple(args, env, algo, _eval=eval_model_co)
