#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE; runs only in the build container).

Imports the Python reference from /root/reference through ``ref_loader`` (no reference
source is copied), drives its PredatorPrey / Coverage envs through the reference's own
``VecEnvExecutor`` (one env per executor - the reference is effectively n_envs=1,
SURVEY.md App. B-1), and records

  * the inputs      : actions, and an RNG *tape* of every realised random draw per call
                      site: ``random.randint`` spawn candidates,
                      ``np.random.choice`` prey-move outcomes, ``torch.rand`` IID / GE uniforms
  * the outputs     : positions, alive flags, visited map, obs (as ``torch.Tensor`` would
                      cast them: f32), reward (f64), done, reward details, dist_adj, channels

into small ``tests/golden/*.npz`` files.  It also emits policy / critic / PPO-math
fixtures at a fixed seeded ``state_dict``.

Usage:  python oracle/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

def spawn_cap(n_agents, n_preys):
    """Max (row, col) candidates recorded per reset."""
    return 4 * (n_agents + n_preys) + 64


# --------------------------------------------------------------------------------------
# RNG tape recorder
# --------------------------------------------------------------------------------------
class Tape:
    """Wraps the three global generators the reference draws from and logs the draws."""

    def __init__(self):
        self.randints = []      # python ``random.randint`` results, in call order
        self.choices = []       # (prey_index, outcome) of np.random.choice prey draws
        self.rands = []         # torch.rand tensors in call order
        self.cur_prey = -1
        self._orig_randint = random.randint
        self._orig_choice = np.random.choice
        self._orig_rand = torch.rand

    def install(self):
        tape = self

        def randint(a, b):
            v = tape._orig_randint(a, b)
            tape.randints.append(v)
            return v

        def choice(a, size=None, replace=True, p=None):
            out = tape._orig_choice(a, size, replace, p)
            if p is not None and size == 1:   # prey move draw (predator_prey.py:401)
                tape.choices.append((tape.cur_prey, int(out[0])))
            return out

        def rand(*args, **kw):
            t = tape._orig_rand(*args, **kw)
            tape.rands.append(t.clone())
            return t

        random.randint = randint
        np.random.choice = choice
        torch.rand = rand

    def uninstall(self):
        random.randint = self._orig_randint
        np.random.choice = self._orig_choice
        torch.rand = self._orig_rand

    def clear(self):
        self.randints, self.choices, self.rands = [], [], []


def hook_prey_index(env, tape):
    """Note which prey a np.random.choice draw belongs to (tape is indexed [prey, trial])."""
    orig = env.prey_random_move

    def wrapped(prey_i):
        tape.cur_prey = prey_i
        return orig(prey_i)

    env.prey_random_move = wrapped


# --------------------------------------------------------------------------------------
# configs (defaults exp_runners/*/utils_*.py)
# --------------------------------------------------------------------------------------
def pp_params(map_, sen, den, cap, loss=0.0, max_env_steps=200, rcom=9):
    n = int(int(den * 100) * (map_ / 10) ** 2)
    return dict(load=cap, max_env_steps=max_env_steps, capture_reward=10, step_cost=0.1, rm=0, penalty=0,
                grid_size=map_, Rsen=sen, env_param_print=0, n_groups=1, n_nodes=1, n_agents=n, n_preys=n,
                n_gcn_layers=2, curriculum_learning=0, channelType='FC', loss_apply=1, mode='train',
                trRcom=rcom, teRcom=rcom, trpl=loss, calc_diameter=False, Pgb=0.0196, Pbg=0.282, GE_INIT=1)


def co_params(map_, sen, den, loss=0.0, max_env_steps=400, rcom=9, obst='Easy'):
    n = int(int(den * 100) * (map_ / 10) ** 2)
    return dict(load=2, max_env_steps=max_env_steps, capture_reward=2, step_cost=0, rm=0, penalty=1,
                revisit_penalty=0.5, lazy_penalty=1, grid_size=map_, Rsen=sen, env_param_print=0, n_groups=3,
                n_nodes=1, n_agents=n, n_preys=0, n_gcn_layers=2, curriculum_learning=0, channelType='FC',
                loss_apply=1, mode='train', trRcom=rcom, teRcom=rcom, trpl=loss, calc_diameter=False,
                obstComplex=obst, add_clock=0, Pgb=0.0196, Pbg=0.282, GE_INIT=1)


def cfg_json(scenario, p, channel, max_path_length):
    c = dict(scenario=scenario, n_agents=int(p['n_agents']), n_preys=int(p.get('n_preys', 0)),
             grid=int(p['grid_size']), rsen=int(p['Rsen']), load=int(p['load']),
             max_steps=int(p['max_env_steps']), max_path_length=int(max_path_length), n_hops=int(p['n_gcn_layers']), rcom=int(p['trRcom']),
             channel=channel, ploss=float(p['trpl']), pgb=float(p['Pgb']), pbg=float(p['Pbg']),
             capture_reward=float(abs(p['capture_reward'])), step_cost=float(p['step_cost']),
             rm=float(p['rm']), penalty=float(p['penalty']),
             lazy_penalty=float(p.get('lazy_penalty', 0)), revisit_penalty=float(p.get('revisit_penalty', 0)),
             obst=p.get('obstComplex', 'Easy'), add_clock=int(p.get('add_clock', 0)),
             ge_init=int(p.get('GE_INIT', 1)), loss_apply=int(p.get('loss_apply', 1)))
    return json.dumps(c)


# --------------------------------------------------------------------------------------
# action scripts: mostly random, partly chasing so that captures / blocks / coverage happen
# --------------------------------------------------------------------------------------
def chase_actions(env, rng, p_random):
    n = env.n_agents
    acts = rng.randint(0, 5, size=n)
    if hasattr(env, 'prey_pos') and env._prey_alive is not None:
        live = [env.prey_pos[j] for j in range(env.n_preys) if env._prey_alive[j]]
        for i in range(n):
            if rng.rand() < p_random or not live:
                continue
            r, c = env.agent_pos[i]
            d = [abs(r - pr) + abs(c - pc) for pr, pc in live]
            pr, pc = live[int(np.argmin(d))]
            if abs(r - pr) + abs(c - pc) <= 1:
                acts[i] = 4
            elif abs(r - pr) >= abs(c - pc):
                acts[i] = 0 if pr > r else 2
            else:
                acts[i] = 3 if pc > c else 1
    elif hasattr(env, '_visited'):
        for i in range(n):
            if rng.rand() >= p_random:
                a = bfs_to_unvisited(env, i)
                if a is not None:
                    acts[i] = a
    return acts


def bfs_to_unvisited(env, i):
    """First action of a shortest vacant-cell path from agent i to the nearest unvisited cell."""
    from collections import deque
    G = env._grid_shape[0]
    start = tuple(env.agent_pos[i])
    moves = [(1, 0, 0), (0, -1, 1), (-1, 0, 2), (0, 1, 3)]
    seen = {start}
    q = deque()
    for dr, dc, a in moves:
        nxt = (start[0] + dr, start[1] + dc)
        if 0 <= nxt[0] < G and 0 <= nxt[1] < G and env._full_obs[nxt[0]][nxt[1]] == '0':
            q.append((nxt, a))
            seen.add(nxt)
    while q:
        (r, c), a = q.popleft()
        if env._visited[r][c] == 0:
            return a
        for dr, dc, _ in moves:
            nxt = (r + dr, c + dc)
            if nxt not in seen and 0 <= nxt[0] < G and 0 <= nxt[1] < G and env._full_obs[nxt[0]][nxt[1]] == '0':
                seen.add(nxt)
                q.append((nxt, a))
    return None


# --------------------------------------------------------------------------------------
# env roll-out recording
# --------------------------------------------------------------------------------------
def snapshot(env, scenario):
    n = env.n_agents
    s = dict(agent_pos=np.array([env.agent_pos[i] for i in range(n)], dtype=np.int32),
             step_count=np.int32(env._step_count), success=np.int32(env.success))
    if scenario == 'pp':
        s['prey_pos'] = np.array([env.prey_pos[j] for j in range(env.n_preys)], dtype=np.int32)
        s['prey_alive'] = np.array(env._prey_alive, dtype=np.uint8)
    else:
        s['visited'] = np.array(env._visited, dtype=np.uint8)
        s['total_capture'] = np.int32(env.total_capture_cnt)
    s['dist_adj'] = np.asarray(env.dist_adj, dtype=np.float32)
    s['channels'] = np.asarray(env.channels, dtype=np.float32)
    return s


def details_vec(scenario, d, n_agents):
    """Integer view of the reward_details dict (means are sum/N exactly)."""
    if scenario == 'pp':
        return np.array([d['capture_cnt'], round(float(d['move_cnt']) * n_agents), d['penalty_cnt'], 0,
                         round(float(d['variable']) * n_agents), 0], dtype=np.int32)
    return np.array([round(d['capture_cnt'] * n_agents), round(d['move_cnt'] * n_agents),
                     round(d['penalty_cnt'] * n_agents), round(d['vars2'] * n_agents),
                     round(d['variable'] * n_agents), 0], dtype=np.int32)


def split_tape(tape, scenario, env, L, reset_happened, channel):
    """Attribute this call's draws to tape arrays."""
    n, M = env.n_agents, getattr(env, 'n_preys', 0)
    out = {}
    prey = np.full((max(M, 1), 5), 255, dtype=np.uint8)
    cnt = [0] * max(M, 1)
    for j, o in tape.choices:
        prey[j, cnt[j]] = o
        cnt[j] += 1
    out['prey_tape'] = prey
    SPAWN_CAP = spawn_cap(n, M)
    sp = np.full((SPAWN_CAP, 2), -1, dtype=np.int32)
    ri = tape.randints
    assert len(ri) % 2 == 0 and len(ri) // 2 <= SPAWN_CAP, len(ri)
    if ri:
        sp[:len(ri) // 2] = np.array(ri, dtype=np.int32).reshape(-1, 2)
    out['spawn_tape'] = sp
    out['spawn_n'] = np.int32(len(ri) // 2)
    # torch.rand draws: IID -> one [L,n,n] per comm update; GE -> 2 per hop transition.
    iid = np.zeros((2, L, n, n), dtype=np.float32)
    ge = np.zeros((2, L, 2, n, n), dtype=np.float32)
    ge_init = np.zeros((n, n), dtype=np.float32)
    r = [t.numpy() for t in tape.rands]
    if channel == 'IID':
        if tape.first_call:            # initial reset: only the reset's comm update draws
            assert len(r) == 1, len(r)
            iid[1] = r[0]
        else:
            assert len(r) == (2 if reset_happened else 1), len(r)
            iid[0] = r[0]
            if reset_happened:
                iid[1] = r[1]
    elif channel == 'GE':
        # step update: L transitions (2 rands each) with loss_apply=1, ONE with loss_apply=0; reset update: [one (n,n)
        # rand of get_init_state when GE_INIT is neither 0 nor 1] then L-1 transitions (loss_apply=1) or none
        per_step = getattr(env, 'loss_apply', 1) == 0
        rand_init = getattr(env, 'GE_INIT', 1) not in (0, 1)
        k = 0
        if not tape.first_call:
            for l in range(1 if per_step else L):
                ge[0, l, 0], ge[0, l, 1] = r[k], r[k + 1]
                k += 2
        if reset_happened or tape.first_call:
            slot = 1
            if rand_init:
                ge_init[:] = r[k]
                k += 1
            for l in range(1, 1 if per_step else L):
                ge[slot, l, 0], ge[slot, l, 1] = r[k], r[k + 1]
                k += 2
        assert k == len(r), (k, len(r))
    else:
        assert len(r) == 0
    out['iid_u'] = iid
    out['ge_u'] = ge
    out['ge_init_u'] = ge_init
    return out


def record_env(ns, scenario, params, B, T, seed, channel='FC', max_path_length=None, p_random=0.5, agent_condition=None):
    """Roll B independent reference envs for T steps through the reference VecEnvExecutor.
    agent_condition [B,N] (PP): written into env.agent_condition after the initial reset - the gate of
    predator_prey.py:258; the env itself sets it back to ones at every reset (:152), which the recording shows."""
    import importlib
    VecEnvExecutor = importlib.import_module('garage.sampler.vec_env_executor').VecEnvExecutor
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    arng = np.random.RandomState(seed + 1000)
    L = params['n_gcn_layers']
    mpl = max_path_length or params['max_env_steps']
    tape = Tape()
    tape.first_call = False
    envs, vecs = [], []
    for b in range(B):
        cls = ns.PredatorPreyWrapper if scenario == 'pp' else ns.CoverageWrapper
        if scenario == 'pp':
            e = cls(centralized=True, params=dict(params))
        else:
            e = cls(centralized=True, max_steps=params['max_env_steps'], params=dict(params))
        if channel == 'GE':   # GE is unreachable from params (App. B-4): switch it on directly
            e.channelType = 'GE'
            e.Pgb, e.Pbg, e.GE_INIT = params['Pgb'], params['Pbg'], params['GE_INIT']
            e.loss_apply = params['loss_apply']
        if scenario == 'pp':
            hook_prey_index(e, tape)
        envs.append(e)
        vecs.append(VecEnvExecutor(envs=[e], max_path_length=mpl))

    rec = {k: [] for k in ('actions', 'prey_tape', 'spawn_tape', 'spawn_n', 'iid_u', 'ge_u', 'ge_init_u', 'reward', 'done',
                           'details', 'obs', 'agent_pos', 'prey_pos', 'prey_alive', 'prey_alive_info', 'visited',
                           'total_capture', 'step_count', 'success', 'dist_adj', 'channels', 'agent_cond')}

    def push_state(lst_keys, snaps, obs):
        for k in lst_keys:
            if k in snaps[0]:
                rec[k].append(np.stack([s[k] for s in snaps]))
        rec['obs'].append(np.stack([torch.Tensor(np.asarray(o)).numpy() for o in obs]))

    state_keys = ('agent_pos', 'prey_pos', 'prey_alive', 'visited', 'total_capture', 'step_count', 'success',
                  'dist_adj', 'channels')
    tape.install()
    try:
        # ---- initial reset (t = -1) ----
        init, snaps, obs0 = [], [], []
        for b in range(B):
            tape.clear()
            tape.first_call = True
            o = vecs[b].reset(0)[0]
            init.append(split_tape(tape, scenario, envs[b], L, True, channel))
            tape.first_call = False
            snaps.append(snapshot(envs[b], scenario))
            obs0.append(o)
        if agent_condition is not None:
            for b in range(B):
                envs[b].agent_condition = np.asarray(agent_condition[b], dtype=np.float64).copy()
            rec['agent_cond'].append(np.stack([np.asarray(e.agent_condition, dtype=np.uint8) for e in envs]))
        push_state(state_keys, snaps, obs0)
        init_tape = {k: np.stack([d[k] for d in init]) for k in init[0]}

        for t in range(T):
            acts, tapes, snaps, obs_t, rew, dn, det, pinfo = [], [], [], [], [], [], [], []
            for b in range(B):
                a = chase_actions(envs[b], arng, p_random)
                tape.clear()
                o, (r, d), done, info = vecs[b].step(np.array([a]), 0)
                tapes.append(split_tape(tape, scenario, envs[b], L, bool(done[0]), channel))
                acts.append(a.astype(np.int32))
                snaps.append(snapshot(envs[b], scenario))
                obs_t.append(o[0])
                rew.append(np.float64(r[0]))
                dn.append(np.uint8(done[0]))
                det.append(details_vec(scenario, d, envs[b].n_agents))
                if scenario == 'pp':
                    pinfo.append(np.asarray(info['prey_alive'][0], dtype=np.uint8))
            rec['actions'].append(np.stack(acts))
            if agent_condition is not None:
                rec['agent_cond'].append(np.stack([np.asarray(e.agent_condition, dtype=np.uint8) for e in envs]))
            for k in ('prey_tape', 'spawn_tape', 'spawn_n', 'iid_u', 'ge_u', 'ge_init_u'):
                rec[k].append(np.stack([d[k] for d in tapes]))
            rec['reward'].append(np.array(rew))
            rec['done'].append(np.array(dn))
            rec['details'].append(np.stack(det))
            if pinfo:
                rec['prey_alive_info'].append(np.stack(pinfo))
            push_state(state_keys, snaps, obs_t)
    finally:
        tape.uninstall()

    out = {k: np.stack(v) for k, v in rec.items() if v}
    if channel not in ('IID',):
        out.pop('iid_u', None)
    if channel != 'GE':
        out.pop('ge_u', None)
    if channel != 'GE' or params.get('GE_INIT', 1) in (0, 1):
        out.pop('ge_init_u', None)
    if scenario != 'pp':
        out.pop('prey_tape', None)
    for k, v in init_tape.items():
        if k in out or k in ('spawn_tape', 'spawn_n'):
            out['init_' + k] = v
    out['cfg'] = np.array(cfg_json(scenario, params, channel, mpl))
    if scenario == 'co':
        out['n_empty_cells'] = np.int32(envs[0].n_empty_cells)
        out['bound_return'] = np.float64(envs[0].bound_return)
    return out


# --------------------------------------------------------------------------------------
# direct-call fixtures: GE matrix evolution, adjacency ties
# --------------------------------------------------------------------------------------
def record_ge(ns, n=6, hops=20, seed=7):
    torch.manual_seed(seed)
    tape = Tape()
    tape.install()
    try:
        state = torch.ones(n, n).bool()
        seq = ns.ge.get_next_state_matrix(n_sequence=hops, state=state, Pgb=0.0196 * 8, Pbg=0.282)
    finally:
        tape.uninstall()
    u = np.stack([t.numpy() for t in tape.rands]).reshape(hops, 2, n, n)
    return dict(u=u.astype(np.float32), states=seq.numpy().astype(np.uint8), pgb=np.float32(0.0196 * 8),
                pbg=np.float32(0.282))


def record_faults_direct(ns):
    """The dormant fault / delay helpers (env_communication.py:270-301) called directly.  np.random.choice draws
    `random_sample(size)` from the legacy global generator, so re-seeding and calling random_sample gives the very
    uniforms each call consumed."""
    ec = ns.env_communication
    out = {}
    iid_u, iid_c, iid_p = [], [], []
    for seed, n, p_fault in ((1, 4, 0.3), (2, 24, 0.1), (3, 72, 0.5), (4, 9, 0.0), (5, 9, 1.0)):
        np.random.seed(seed)
        u = np.random.random_sample(n)
        np.random.seed(seed)
        c = ec.iid_fault(n, p_fault)
        iid_u.append(np.pad(u, (0, 72 - n))); iid_c.append(np.pad(c, (0, 72 - n), constant_values=-1)); iid_p.append((n, p_fault))
    out['iid_u'], out['iid_cond'], out['iid_np'] = np.array(iid_u), np.array(iid_c, dtype=np.int64), np.array(iid_p, dtype=np.float64)
    ge_in, ge_out, ge_u, ge_pr = [], [], [], []
    rng = np.random.RandomState(11)
    for seed, n, p, r in ((6, 8, 0.2, 0.3), (7, 8, 0.9, 0.05), (8, 24, 0.5, 0.5), (9, 24, 0.0196, 0.282), (10, 5, 0.5, 0.9)):
        cond = (rng.rand(n) < 0.6).astype(np.int64)
        np.random.seed(seed)
        ug = np.random.random_sample(1)[0]
        ub = np.random.random_sample(1)[0]
        np.random.seed(seed)
        new = ec.GE_fault(cond.copy(), p, r)
        ge_in.append(np.pad(cond, (0, 24 - n), constant_values=-1)); ge_out.append(np.pad(new, (0, 24 - n), constant_values=-1))
        ge_u.append((ug, ub)); ge_pr.append((n, p, r))
    out['ge_in'], out['ge_out'] = np.array(ge_in, dtype=np.int64), np.array(ge_out, dtype=np.int64)
    out['ge_u'], out['ge_npr'] = np.array(ge_u, dtype=np.float64), np.array(ge_pr, dtype=np.float64)
    L, N, th = 2, 6, 7
    adj = (rng.rand(N, N) < 0.6).astype(np.float32)
    np.fill_diagonal(adj, 1)
    links = [(rng.rand(L, N, N) < 0.7).astype(np.float32) for _ in range(4)]
    d0 = ec.delays_init(adj, links[0], th)
    seq = [d0]
    for lk in links[1:]:
        seq.append(ec.calc_delays(adj, lk, seq[-1][-1]))
    out['delay_adj'], out['delay_links'], out['delay_th'] = adj, np.array(links), np.int64(th)
    out['delays'] = np.array(seq, dtype=np.int64)                     # [4 calls, L, N, N]
    return out


def record_adj_ties(ns, seed=3):
    """Adjacency with many exact ties dx^2+dy^2 == 2*Rcom^2."""
    rng = np.random.RandomState(seed)
    out = {}
    for n in (4, 24, 26, 54, 72):
        pos = rng.randint(0, 31, size=(n, 2))
        pos[1] = pos[0] + np.array([9, 9])      # exact tie
        pos[2] = pos[0] + np.array([9, -9]) if pos[0][1] >= 9 else pos[0] + np.array([-9, 9])
        x = torch.FloatTensor([[0, 0], [9, 9]])
        th = torch.cdist(x, x)[0][-1]
        adj, deg, _ = ns.env_communication.get_graph(9, th, n, {i: list(map(int, pos[i])) for i in range(n)})
        out[f'pos_{n}'] = pos.astype(np.int32)
        out[f'adj_{n}'] = adj.astype(np.float32)
        out[f'deg_{n}'] = np.float64(deg)
    return out


def record_adj_ties_grid32(ns, seed=17):
    """As record_adj_ties, with every position inside a 32 x 32 grid (the device env's largest side) so that the
    recorded adjacency can be driven through the oracle AND the HIP env step: distinct cells, a diamond of exact ties
    (p, p+(9,9), p+(9,-9), p+(18,0): four pairs at dx^2+dy^2 == 162) plus near-ties on both sides (161, 164)."""
    rng = np.random.RandomState(seed)
    out = {}
    x = torch.FloatTensor([[0, 0], [9, 9]])
    th = torch.cdist(x, x)[0][-1]
    for n in (4, 24, 26, 54, 72):
        while True:
            pos = rng.randint(0, 32, size=(n, 2))
            p = np.array([rng.randint(0, 13), rng.randint(9, 22)])
            pos[0], pos[1], pos[2], pos[3] = p, p + [9, 9], p + [9, -9], p + [18, 0]
            if n > 6:
                pos[4] = p + [10, 8]             # 164 > 162 from p: not adjacent
                pos[5] = p + [12, 4] if n > 5 else pos[5]     # 160 <= 162: adjacent
                pos[6] = p + [1, 0]              # from p+(9,9): 64+81 = 145; from p+(10,8): 81+64
            if len({tuple(q) for q in pos}) == n and pos.min() >= 0 and pos.max() <= 31:
                break
        adj, deg, _ = ns.env_communication.get_graph(9, th, n, {i: list(map(int, pos[i])) for i in range(n)})
        out[f'pos_{n}'] = pos.astype(np.int32)
        out[f'adj_{n}'] = np.asarray(adj).astype(np.float32)
        out[f'deg_{n}'] = np.float64(deg)
    return out


# --------------------------------------------------------------------------------------
# policy / critic / PPO-math fixtures
# --------------------------------------------------------------------------------------
def record_policy(ns, env_fix, n_agents, seed=1, take=6, n_hops=None, residual=True):
    """Policy + critic forward at a seeded state_dict on observations from an env fixture.  n_hops / residual:
    non-default GCN depth (0 = no message passing: embeddings[-1] is the encoder output itself) and skip connection."""
    torch.manual_seed(seed)
    T1, B = env_fix['obs'].shape[:2]
    d_total = env_fix['obs'].shape[2]
    spec = ref_loader.make_env_spec(d_total)
    kw = {} if n_hops is None else dict(n_gcn_layers=n_hops)
    pol = ns.CommCategoricalMLPPolicy(spec, n_agents=n_agents, residual=residual, **kw)
    crit = ns.CommBaseCritic(spec, n_agents=n_agents, residual=residual, **kw)
    # make biases non-zero so that bias handling is actually pinned
    with torch.no_grad():
        for net in (pol, crit):
            for name, p in net.named_parameters():
                if name.endswith('bias') and 'gcn' not in name:
                    p.uniform_(-0.1, 0.1)
    idx = np.linspace(0, T1 - 1, take).astype(int)
    obs = env_fix['obs'][idx].reshape(-1, d_total)                       # [S, N*d]
    adj = env_fix['dist_adj'][idx].reshape(-1, n_agents, n_agents)
    ch = env_fix['channels'][idx].reshape(-1, env_fix['channels'].shape[2], n_agents, n_agents)
    S = obs.shape[0]
    avail = np.ones((S, n_agents * 5), dtype=np.float32)
    rng = np.random.RandomState(seed)
    if n_hops is not None and n_hops != ch.shape[1]:                     # other depth: seeded random link masks
        ch = (rng.rand(S, n_hops, n_agents, n_agents) < 0.8).astype(np.float32)
        ch[:, :, np.arange(n_agents), np.arange(n_agents)] = 1.0
    avail_masked = avail.copy().reshape(S, n_agents, 5)
    avail_masked[rng.rand(S, n_agents) < 0.3, 1] = 0                     # forbid action 1 sometimes
    avail_masked = avail_masked.reshape(S, -1)
    out = {}
    with torch.no_grad():
        for tag, av in (('', avail), ('_masked', avail_masked)):
            dist, attn = pol.forward(obs, av, adj, ch, get_actions=True)
            out['probs' + tag] = dist.probs.numpy()
            out['attn' + tag] = attn.numpy()
        emb, _ = ns.CommBaseNet.forward(pol, torch.Tensor(obs).reshape(S, n_agents, -1), torch.Tensor(adj),
                                        torch.Tensor(ch), True)
        for i, e in enumerate(emb):
            out[f'emb{i}'] = e.numpy()
        acts = rng.randint(0, 5, size=(S, n_agents))
        tobs, tav, tadj, tch = (torch.Tensor(obs), torch.Tensor(avail), torch.Tensor(adj.reshape(S, -1)),
                                torch.Tensor(ch.reshape(S, -1, n_agents)))
        out['entropy'] = pol.entropy(tobs, tav, tadj, tch).numpy()
        out['loglik'] = pol.log_likelihood(tobs, tav, tadj, tch, torch.Tensor(acts)).numpy()
        out['values'] = crit.forward(tobs, tav, tadj, tch).numpy()
        returns = torch.Tensor(rng.randn(S).astype(np.float32) * 3)
        out['critic_loss'] = crit.compute_loss(tobs, returns, tadj, tch).numpy()
        out['returns'] = returns.numpy()
    out.update(obs=obs.astype(np.float32), adj=adj, channels=ch, avail_masked=avail_masked, actions=acts,
               residual=np.int32(residual))
    for name, p in pol.state_dict().items():
        out['pol.' + name] = p.numpy()
    for name, p in crit.state_dict().items():
        out['crit.' + name] = p.numpy()
    return out


def record_variants(env_fix, n_agents, seed=21, take=6, with_grads=True):
    """Obs-DP / CENT policies and the Gaussian baseline at seeded state_dicts on observations
    of an env fixture: probabilities (all-ones and masked avail), greedy actions, entropy, log-likelihood,
    values, loss and the gradients of a PPO-shaped scalar through each net."""
    import contextlib
    import io
    ns = ref_loader.load_reference_variants(ref_loader.load_reference())
    torch.manual_seed(seed)
    T1 = env_fix['obs'].shape[0]
    d_total = env_fix['obs'].shape[2]
    spec = ref_loader.make_env_spec(d_total)
    nets = dict(dec=ns.DecCategoricalMLPPolicy(spec, n_agents, hidden_sizes=[128, 64, 32]),
                cent=ns.CentralizedCategoricalMLPPolicy(spec, n_agents=n_agents, hidden_sizes=[128, 64, 32]),
                gb=ns.GaussianMLPBaseline(env_spec=spec, hidden_sizes=(64, 64, 64)))
    with torch.no_grad():
        for net in nets.values():
            for name, p in net.named_parameters():
                if name.endswith('bias'):
                    p.uniform_(-0.1, 0.1)
    # big matrices are not stored: they are drawn from a seeded numpy stream (xavier-uniform range) that the
    # tests redraw (tests/test_hip_variants_parity.py::_state_dict) - keeps the fixtures small
    regen = {}
    with torch.no_grad():
        for tag, net in nets.items():
            for k, (name, p) in enumerate(net.state_dict().items()):
                if p.numel() > 20000:
                    lim = float(np.sqrt(6.0 / (p.shape[0] + p.shape[1])))
                    sd_seed = 1000 * seed + 37 * k + len(tag)
                    p.copy_(torch.from_numpy(np.random.RandomState(sd_seed).uniform(-lim, lim, tuple(p.shape))
                                             .astype(np.float32)))
                    regen[f'{tag}.sd.{name}'] = [sd_seed, lim, list(p.shape)]
    idx = np.linspace(0, T1 - 1, take).astype(int)
    obs = env_fix['obs'][idx].reshape(-1, d_total).astype(np.float32)    # [S, N*d]
    S = obs.shape[0]
    rng = np.random.RandomState(seed)
    avail = np.ones((S, n_agents * 5), dtype=np.float32)
    avail_masked = avail.copy().reshape(S, n_agents, 5)
    avail_masked[rng.rand(S, n_agents) < 0.3, 1] = 0
    avail_masked = avail_masked.reshape(S, -1)
    acts = rng.randint(0, 5, size=(S, n_agents))
    wts = rng.randn(S).astype(np.float32)
    out = dict(obs=obs, avail_masked=avail_masked, actions=acts, weights=wts, regen=np.array(json.dumps(regen)))
    tobs, tav = torch.Tensor(obs), torch.Tensor(avail)
    for tag in ('dec', 'cent'):
        pol = nets[tag]
        with torch.no_grad():
            for suffix, av in (('', avail), ('_masked', avail_masked)):
                out[f'{tag}.probs{suffix}'] = pol.forward(obs, av, get_actions=True).probs.numpy()
            a, info = pol.get_actions(obs, avail_masked, greedy=True)
            out[f'{tag}.greedy_masked'] = np.asarray(a)
            out[f'{tag}.entropy'] = pol.entropy(tobs, tav).numpy()
            out[f'{tag}.loglik'] = pol.log_likelihood(tobs, tav, torch.Tensor(acts)).numpy()
        scalar = -(pol.log_likelihood(tobs, tav, torch.Tensor(acts)) * torch.Tensor(wts)).mean() \
            - 0.1 * pol.entropy(tobs, tav).mean()
        pol.zero_grad()
        scalar.backward()
        out[f'{tag}.scalar'] = scalar.detach().numpy()
        if with_grads:
            for name, p in pol.named_parameters():
                if p.numel() <= 20000:
                    out[f'{tag}.grad.{name}'] = p.grad.clone().numpy()
                else:                                            # big: keep a strided sample + the norm
                    out[f'{tag}.gradnorm.{name}'] = p.grad.norm().numpy()
                    out[f'{tag}.gradrows.{name}'] = p.grad[::16].clone().numpy()
        for name, p in pol.state_dict().items():
            if f'{tag}.sd.{name}' not in regen:
                out[f'{tag}.sd.{name}'] = p.numpy()
    gb = nets['gb']
    returns = torch.Tensor(rng.randn(S).astype(np.float32) * 3)
    with contextlib.redirect_stdout(io.StringIO()):
        with torch.no_grad():
            out['gb.values'] = gb.forward(tobs.reshape(1, S, -1)).numpy()[0]
        loss = gb.compute_loss(tobs.reshape(1, S, -1), returns.reshape(1, S))
    gb.zero_grad()
    loss.backward()
    out['gb.loss'] = loss.detach().numpy()
    out['gb.returns'] = returns.numpy()
    if with_grads:
        for name, p in gb.named_parameters():
            if p.numel() <= 20000:
                out[f'gb.grad.{name}'] = p.grad.clone().numpy()
            else:
                out[f'gb.gradnorm.{name}'] = p.grad.norm().numpy()
                out[f'gb.gradrows.{name}'] = p.grad[::16].clone().numpy()
    for name, p in gb.state_dict().items():
        if f'gb.sd.{name}' not in regen:
            out[f'gb.sd.{name}'] = p.numpy()
    return out


def record_net_options(env_fix, n_agents, seed=31, take=4):
    """The two non-default net options the reference CLI reaches (env_uitils.py:85,88): ``attention_type='dot'``
    (attention_module.py:38-41: no linear_in, scores = E.E^T) for policy AND critic, and the critic's
    ``aggregator_type='direct'`` (comm_base_critic.py:48-49,84-87,115-118: one MLP over the concatenated embeddings).
    At seeded state_dicts on observations of an env fixture: forward outputs and the gradients of a PPO-shaped scalar
    (policy) / the Gaussian NLL (critic) with respect to every parameter."""
    ns = ref_loader.load_reference()
    T1 = env_fix['obs'].shape[0]
    d_total = env_fix['obs'].shape[2]
    spec = ref_loader.make_env_spec(d_total)
    idx = np.linspace(0, T1 - 1, take).astype(int)
    obs = env_fix['obs'][idx].reshape(-1, d_total).astype(np.float32)
    adj = env_fix['dist_adj'][idx].reshape(-1, n_agents, n_agents).astype(np.float32)
    ch = env_fix['channels'][idx].reshape(-1, env_fix['channels'].shape[2], n_agents, n_agents).astype(np.float32)
    S = obs.shape[0]
    rng = np.random.RandomState(seed)
    acts = rng.randint(0, 5, size=(S, n_agents))
    wts = rng.randn(S).astype(np.float32)
    returns = (rng.randn(S) * 3).astype(np.float32)
    avail = np.ones((S, n_agents * 5), dtype=np.float32)
    out = dict(obs=obs, adj=adj, channels=ch, actions=acts, weights=wts, returns=returns)
    tobs, tav, tadj, tch = (torch.Tensor(obs), torch.Tensor(avail), torch.Tensor(adj.reshape(S, -1)),
                            torch.Tensor(ch.reshape(S, -1, n_agents)))
    for tag, att, agg in (('dot', 'dot', 'sum'), ('direct', 'general', 'direct')):
        torch.manual_seed(seed + len(tag))
        pol = ns.CommCategoricalMLPPolicy(spec, n_agents=n_agents, attention_type=att)
        crit = ns.CommBaseCritic(spec, n_agents=n_agents, attention_type=att, aggregator_type=agg)
        with torch.no_grad():
            for net in (pol, crit):
                for name, p in net.named_parameters():
                    if name.endswith('bias') and 'gcn' not in name:
                        p.uniform_(-0.1, 0.1)
        with torch.no_grad():
            dist, attn = pol.forward(obs, avail, adj, ch, get_actions=True)
            out[f'{tag}.probs'], out[f'{tag}.attn'] = dist.probs.numpy(), attn.numpy()
            out[f'{tag}.values'] = crit.forward(tobs, tav, tadj, tch).numpy()
        scalar = -(pol.log_likelihood(tobs, tav, tadj, tch, torch.Tensor(acts)) * torch.Tensor(wts)).mean() \
            - 0.1 * pol.entropy(tobs, tav, tadj, tch).mean()
        pol.zero_grad()
        scalar.backward()
        loss = crit.compute_loss(tobs, torch.Tensor(returns), tadj, tch)
        crit.zero_grad()
        loss.backward()
        out[f'{tag}.scalar'], out[f'{tag}.critic_loss'] = scalar.detach().numpy(), loss.detach().numpy()
        for pre, net in (('pol', pol), ('crit', crit)):
            for name, p in net.state_dict().items():
                out[f'{tag}.{pre}.{name}'] = p.numpy()
            for name, p in net.named_parameters():
                out[f'{tag}.g{pre}.{name}'] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone().numpy()
    return out


def record_ppo_math(ns, seed=5):
    """GAE / returns / per-path normalisation on a ragged 3-path batch."""
    rng = np.random.RandomState(seed)
    lens = [7, 12, 4]
    Tm = max(lens)
    rewards = [rng.randn(l) for l in lens]                     # f64 as the sampler yields
    gamma, lam = 0.99, 0.97
    returns = np.stack([ns.pad_to_last(ns.tensor_utils.discount_cumsum(r, gamma).copy(), total_length=Tm).numpy()
                        for r in rewards])
    rew_pad = torch.stack([ns.pad_to_last(r, total_length=Tm) for r in rewards])
    baselines = torch.Tensor(rng.randn(3, Tm).astype(np.float32))   # critic output incl. padded steps (A-5)
    adv = ns.compute_advantages(gamma, lam, Tm, baselines, rew_pad, 'cpu')
    valids = torch.Tensor(lens).int()
    import torch.nn.functional as F
    mv = [(v.mean(), v.var(unbiased=False)) for v in ns.filter_valids(adv, valids)]
    means, variances = zip(*mv)
    adv_n = F.batch_norm(adv.t(), torch.Tensor(means), torch.Tensor(variances), eps=1e-8).t()
    # clipped surrogate + entropy bonus, mean over valid (centralized_ma_ppo.py:431-438,540-589)
    new_ll = torch.Tensor(rng.randn(3, Tm).astype(np.float32) * 0.1)
    old_ll = torch.Tensor(rng.randn(3, Tm).astype(np.float32) * 0.1)
    ent = torch.Tensor(rng.rand(3, Tm).astype(np.float32))
    ratio = (new_ll - old_ll).exp()
    obj = torch.min(ratio * adv_n, torch.clamp(ratio, 0.9, 1.1) * adv_n) + 0.1 * ent
    loss = -torch.cat(ns.filter_valids(obj, valids)).mean()
    return dict(rewards_pad=np.stack([np.pad(r, (0, Tm - len(r))) for r in rewards]), lens=np.array(lens),
                returns=returns, baselines=baselines.numpy(), adv=adv.numpy(), adv_norm=adv_n.numpy(),
                new_ll=new_ll.numpy(), old_ll=old_ll.numpy(), ent=ent.numpy(), loss=loss.numpy(),
                gamma=np.float64(gamma), lam=np.float64(lam))


def record_adam(ns, seed=9, steps=3):
    """The reference's vendored torch-1.9 Adam (my_optimizer/adam.py) on a small tensor."""
    torch.manual_seed(seed)
    p = torch.nn.Parameter(torch.randn(37))
    opt = ns.Adam([p], lr=3e-4, eps=1e-5, device='cpu')
    grads, params = [], [p.detach().clone().numpy()]
    for _ in range(steps):
        g = torch.randn(37)
        p.grad = g.clone()
        opt.step()
        grads.append(g.numpy())
        params.append(p.detach().clone().numpy())
    return dict(grads=np.stack(grads), params=np.stack(params))


def record_ppo_step(seed=11, kind='comm'):
    """Two optimiser steps of the reference CentralizedMAPPO on paths from the reference sampler
    (PP map10, Tmax 12): process_samples tensors, losses, gradients, clipped grad-norm, parameters
    after each Adam step.  kind: 'comm' (Comm-DP, runner_pp_commDP.py:49-74),
    'obsdp' (runner_pp_obsDP.py:52-72) or 'cent' (runner_pp_cent.py:51-63)."""
    import contextlib
    import io
    ns = ref_loader.load_reference_ppo(ref_loader.load_reference())
    if kind != 'comm':
        ref_loader.load_reference_variants(ns)
    params = pp_params(10, 1, 0.04, 2, max_env_steps=12)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    env = ns.PredatorPreyWrapper(centralized=True, params=dict(params))
    spec = ref_loader.make_env_spec(84)

    class Shell:                      # what GarageEnv adds for the sampler: .spec, attribute passthrough
        def __init__(self, e):
            self.__dict__['_e'] = e
            self.__dict__['spec'] = spec

        def __getattr__(self, k):
            return getattr(self.__dict__['_e'], k)
    if kind == 'comm':
        policy = ns.CommCategoricalMLPPolicy(spec, n_agents=4)
    elif kind == 'obsdp':
        policy = ns.DecCategoricalMLPPolicy(spec, 4, hidden_sizes=[128, 64, 32], name='dec_categorical_mlp_policy')
    else:
        policy = ns.CentralizedCategoricalMLPPolicy(spec, n_agents=4, hidden_sizes=[128, 64, 32], name='centralized')
    critic = (ns.GaussianMLPBaseline(env_spec=spec, hidden_sizes=(64, 64, 64)) if kind == 'cent'
              else ns.CommBaseCritic(spec, n_agents=4))
    gnn_critic = kind != 'cent'
    with torch.no_grad():
        for net in (policy, critic):
            for name, p in net.named_parameters():
                if name.endswith('bias') and 'gcn' not in name:
                    p.uniform_(-0.1, 0.1)
    algo = ns.CentralizedMAPPO(env_spec=spec, policy=policy, baseline=critic, max_path_length=12, discount=0.99,
                               center_adv=True, positive_adv=False, gae_lambda=0.97, policy_ent_coeff=0.1,
                               entropy_method='regularized', stop_entropy_gradient=False, clip_grad_norm=7,
                               optimization_n_minibatches=3, optimization_mini_epochs=10, device='cpu')
    sampler = ns.ReferenceSampler(algo, Shell(env), n_envs=1)
    sampler.start_worker()
    paths = sampler.obtain_samples(0, batch_size=9 * 12 * 4)
    # the sampler's path-dict contract: key -> (shape, dtype) of one reference path
    def _desc(v):
        if isinstance(v, dict):
            return {k: _desc(x) for k, x in v.items()}
        a = np.asarray(v)
        return [list(a.shape), str(a.dtype)]
    path_contract = json.dumps({k: _desc(v) for k, v in paths[0].items()})
    # make the batch ragged (random-policy episodes all run to the time limit): keep a prefix of two paths
    for i, n in ((1, 5), (4, 9)):
        for k, v in list(paths[i].items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 12 and k != 'success':
                paths[i][k] = v[:n]
    out = {'path_contract': np.array(path_contract)}
    for name, p in policy.state_dict().items():
        out['pol0.' + name] = p.clone().numpy()
    for name, p in critic.state_dict().items():
        out['crit0.' + name] = p.clone().numpy()
    with contextlib.redirect_stdout(io.StringIO()):          # GaussianMLPBaseline.forward prints shapes
        obs, avail, actions, rewards, valids, baselines, returns, dist_adjs, channels = algo.process_samples(0, paths)
    P, T = rewards.shape
    out.update(obs=obs.numpy(), actions=actions.numpy().astype(np.int32), rewards=rewards.numpy(),
               valids=valids.numpy(), baselines=baselines.numpy(), returns=returns.numpy(),
               dist_adjs=dist_adjs.numpy(), channels=channels.numpy(),
               rewards64=np.stack([np.pad(np.asarray(p['rewards'], np.float64), (0, T - len(p['rewards'])))
                                   for p in paths]))
    for step in (1, 2):
        loss = algo._compute_loss(0, obs, avail, actions, rewards, valids, baselines, dist_adjs, channels)
        bl = critic.compute_loss(obs, returns, dist_adjs, channels) if gnn_critic else critic.compute_loss(obs, returns)
        algo._baseline_optimizer.zero_grad()
        bl.backward()
        algo._optimizer.zero_grad()
        loss.backward()
        out[f'loss{step}'] = loss.detach().numpy()
        out[f'critic_loss{step}'] = bl.detach().numpy()
        for name, p in policy.named_parameters():
            out[f'gpol{step}.' + name] = p.grad.clone().numpy()
        for name, p in critic.named_parameters():
            out[f'gcrit{step}.' + name] = p.grad.clone().numpy()
        torch.nn.utils.clip_grad_norm_(policy.parameters(), 7)
        out[f'grad_norm{step}'] = np.float64(policy.grad_norm())
        algo._optimizer.step()
        algo._baseline_optimizer.step()
        for name, p in policy.state_dict().items():
            out[f'pol{step}.' + name] = p.clone().numpy()
        for name, p in critic.state_dict().items():
            out[f'crit{step}.' + name] = p.clone().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(HERE, '..', 'tests', 'golden'))
    ap.add_argument('--only', default=None, help='regenerate a single fixture (e.g. ppo_step)')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    ns = ref_loader.load_reference()

    def save(name, d):
        if args.only and name != args.only:
            return
        if callable(d):
            d = d()
        path = os.path.join(args.out, name + '.npz')
        np.savez_compressed(path, **d)
        print(f'{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB')

    fx = {}
    if args.only and args.only.startswith(('ppo_step', 'variants_', 'ppo_math', 'adam', 'adj_ties_grid32', 'faults_direct', 'env_pp_map10_cond', 'net_options_', 'env_pp_map40', 'env_co_map40')):
        return late(save, args)
    # config 1/2: PP map10 sen1 den.04 cap2 (full 200-step horizon, chasing so captures happen)
    fx['pp_map10_cap2'] = record_env(ns, 'pp', pp_params(10, 1, 0.04, 2), B=3, T=230, seed=1, p_random=0.35)
    # short horizon: time-limit dones + auto-resets every 9 steps
    fx['pp_map10_cap2_T9'] = record_env(ns, 'pp', pp_params(10, 1, 0.04, 2, max_env_steps=9), B=4, T=40, seed=2)
    # load 3 (capv=3 -> reward_individual with interior adj = load = 3)
    fx['pp_map10_cap3'] = record_env(ns, 'pp', pp_params(10, 1, 0.08, 3, max_env_steps=60), B=2, T=70, seed=3,
                                     p_random=0.3)
    # config 4: PP map30 sen2 den.08 cap4 (N=M=72, Euclid adjacency)
    fx['pp_map30_cap4'] = record_env(ns, 'pp', pp_params(30, 2, 0.08, 4, max_env_steps=14), B=2, T=18, seed=4,
                                     p_random=0.3)
    # dense small map: many blocked moves, prey retry exhaustion; IID loss; Euclid adjacency (map20, Rcom 9)
    fx['pp_map20_cap2_iid'] = record_env(ns, 'pp', pp_params(20, 1, 0.06, 2, loss=0.3, max_env_steps=25), B=2,
                                         T=30, seed=5, channel='IID', p_random=0.3)
    # config 3: CO map20 sen2 den.06
    fx['co_map20'] = record_env(ns, 'co', co_params(20, 2, 0.06, max_env_steps=30), B=2, T=35, seed=6,
                                p_random=1.0)
    # CO map10 N=3 long enough to finish coverage? use small horizon + full-coverage unlikely; keep 400 default
    fx['co_map10_T400'] = record_env(ns, 'co', co_params(10, 1, 0.03), B=2, T=420, seed=7, p_random=0.25)
    # VecEnv truncation shorter than the env's own 400-step limit (what the CO runner does, coverage.py:39)
    fx['co_map10_trunc'] = record_env(ns, 'co', co_params(10, 1, 0.03), B=2, T=30, seed=11, p_random=0.6,
                                      max_path_length=12)
    # config 5: CO map30 sen2 den.06 loss .3 (IID), N=54
    fx['co_map30_iid'] = record_env(ns, 'co', co_params(30, 2, 0.06, loss=0.3, max_env_steps=8), B=2, T=11,
                                    seed=8, channel='IID', p_random=1.0)
    # GE channel through the env (direct switch-on) on CO map20
    fx['co_map20_ge'] = record_env(ns, 'co', co_params(20, 2, 0.06, max_env_steps=10), B=2, T=14, seed=9,
                                   channel='GE', p_random=1.0)
    # GE variants that exist in the code but not on the CLI: one transition per env step
    # (loss_apply=0), all-bad and random (stationary) initial states
    fx['co_map20_ge_step'] = record_env(ns, 'co', dict(co_params(20, 2, 0.06, max_env_steps=7), loss_apply=0), B=2, T=12,
                                        seed=12, channel='GE', p_random=1.0)
    fx['co_map20_ge_bad'] = record_env(ns, 'co', dict(co_params(20, 2, 0.06, max_env_steps=7), GE_INIT=0), B=2, T=12,
                                       seed=13, channel='GE', p_random=1.0)
    fx['pp_map10_ge_bad_step'] = record_env(ns, 'pp', dict(pp_params(10, 1, 0.04, 2, max_env_steps=9), GE_INIT=0,
                                                           loss_apply=0), B=3, T=25, seed=14, channel='GE')
    # (random init together with loss_apply=0 is shape-inconsistent in the reference itself - the state gets an extra
    # leading axis, env_communication.py:121 - so that combination has no fixture and is refused by cm_env_create)
    fx['pp_map10_ge_rand'] = record_env(ns, 'pp', dict(pp_params(10, 1, 0.04, 2, max_env_steps=9), GE_INIT=2), B=3, T=25,
                                        seed=15, channel='GE')
    # Hard obstacles
    fx['co_map10_hard'] = record_env(ns, 'co', co_params(10, 1, 0.06, max_env_steps=40, obst='Hard'), B=2, T=45,
                                     seed=10, p_random=1.0)
    for k, v in fx.items():
        save('env_' + k, v)

    save('ge_direct', record_ge(ns))
    save('adj_ties', record_adj_ties(ns))
    save('policy_pp_map10', record_policy(ns, fx['pp_map10_cap2'], 4))
    save('policy_co_map20', record_policy(ns, fx['co_map20'], 24, take=3))
    save('policy_pp_map30', record_policy(ns, fx['pp_map30_cap4'], 72, take=2))
    save('policy_co_map30_iid', record_policy(ns, fx['co_map30_iid'], 54, take=2))
    save('policy_pp_map10_hops0', record_policy(ns, fx['pp_map10_cap2'], 4, seed=2, take=4, n_hops=0))
    save('policy_pp_map10_hops1_nores', record_policy(ns, fx['pp_map10_cap2'], 4, seed=3, take=4, n_hops=1, residual=False))
    save('policy_co_map20_hops3', record_policy(ns, fx['co_map20'], 24, seed=4, take=2, n_hops=3))
    late(save, args, ns)


def late(save, args, ns=None):
    """Fixtures that do not need the env recordings of this run."""
    ns = ns or ref_loader.load_reference()
    save('adj_ties_grid32', lambda: record_adj_ties_grid32(ns))
    save('faults_direct', lambda: record_faults_direct(ns))
    save('env_pp_map10_cond', lambda: record_env(ns, 'pp', pp_params(10, 1, 0.04, 2, max_env_steps=12), B=4, T=30, seed=21,
                                                 p_random=0.6, agent_condition=np.array([[1, 0, 1, 0], [0, 0, 0, 0],
                                                                                          [1, 1, 1, 1], [0, 1, 1, 1]])))
    # maps beyond 32 cells a side (README.md:50,76 "10, 20, 30, or multiples of 10"; utils_pp.py:55-65 has a >= 40 branch):
    # PP map 40 den .08 -> N = M = 128, CO map 40 den .06 -> N = 96 on a 42 x 42 grid (visited rows take two 32-bit words)
    save('env_pp_map40_cap4', lambda: record_env(ns, 'pp', pp_params(40, 2, 0.08, 4, max_env_steps=9), B=2, T=12, seed=41,
                                                 p_random=0.3))
    save('env_co_map40', lambda: record_env(ns, 'co', co_params(40, 2, 0.06, max_env_steps=8), B=2, T=11, seed=42, p_random=1.0))
    save('ppo_math', lambda: record_ppo_math(ns))
    save('adam', lambda: record_adam(ns))
    save('ppo_step', lambda: record_ppo_step())
    save('ppo_step_obsdp', lambda: record_ppo_step(seed=12, kind='obsdp'))
    save('ppo_step_cent', lambda: record_ppo_step(seed=13, kind='cent'))
    save('variants_pp_map10', lambda: record_variants(np.load(os.path.join(args.out, 'env_pp_map10_cap2.npz')), 4))
    save('variants_co_map20', lambda: record_variants(np.load(os.path.join(args.out, 'env_co_map20.npz')), 24, take=3))
    save('variants_pp_map30', lambda: record_variants(np.load(os.path.join(args.out, 'env_pp_map30_cap4.npz')), 72, take=2))
    save('net_options_pp_map10', lambda: record_net_options(np.load(os.path.join(args.out, 'env_pp_map10_cap2.npz')), 4))
    save('net_options_co_map20', lambda: record_net_options(np.load(os.path.join(args.out, 'env_co_map20.npz')), 24, seed=33, take=2))


if __name__ == '__main__':
    main()
