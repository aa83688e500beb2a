"""CentralizedMAOnPolicyVectorizedSampler - same name, ctor and ``obtain_samples`` contract as
com_marl/sampler/centralized_ma_on_policy_vectorized_sampler.py:20-245, but the B envs really
are B independent envs (the reference's n_envs > 1 shares one env object) and
the whole loop stays on the GPU (rollout.RolloutEngine).

``obtain_samples`` returns a ``PathBatch``: a lazy ``Sequence`` of the reference's path dicts
(materialised to numpy only when indexed) that also exposes the device-resident trajectory and
the (env, start, length) index, which CentralizedMAPPO consumes directly without a host trip.
"""
import os
import time
from collections.abc import Sequence

import numpy as np
import torch

from .rollout import RolloutEngine


class _Tabular:
    """The reference records its timers and progress columns into the global ``dowel.tabular``
    (sampler.py:236-239, centralized_ma_ppo.py:345-385), which the runner dumps to progress.csv.  ``record`` keeps the
    row here (``rows``: what tests and bench.py read) AND forwards it to ``dowel.tabular`` when dowel is importable,
    so the reference's logger sees every column this package produces."""

    def __init__(self):
        self.rows = {}
        self._dowel = None                     # None = not looked up yet, False = not importable

    def _sink(self):
        if self._dowel is None:
            try:
                import dowel
                self._dowel = dowel.tabular
            except Exception:
                self._dowel = False
        return self._dowel

    def record(self, k, v):
        self.rows[k] = v
        sink = self._sink()
        if sink:
            sink.record(k, v)


tabular = _Tabular()

# which engine buffers a path key is built from (host copies are made per buffer, on first use)
_PATH_KEYS = ("observations", "actions", "avail_actions", "rewards", "rewards_details", "dones", "dist_adjs", "channels",
              "attentions", "ave_degs", "diameters", "ave_trputs", "success", "agent_infos", "env_infos")


class _LazyPath(dict):
    """One path dict of the reference's format whose values are built on first access: the runner's
    ``sum(len(p['rewards']) for p in paths)`` (local_runner_wrapper.py:50-52) then costs one host copy of the reward
    buffer instead of the whole trajectory.  It IS a dict (isinstance, ==, pickling after ``materialize()``); keys a
    caller adds (the reference's ``if 'returns' not in path: path['returns'] = ...``, centralized_ma_ppo.py:635-636) live
    next to the lazy ones and are seen by ``in`` / ``get`` / ``keys`` / ``items`` / ``len`` / pickling."""

    def __init__(self, batch, i):
        super().__init__()
        self._batch, self._i = batch, i

    def __missing__(self, k):
        if k not in _PATH_KEYS:
            raise KeyError(k)
        v = self._batch._build(self._i, k)
        dict.__setitem__(self, k, v)
        return v

    def _all_keys(self):
        return list(_PATH_KEYS) + [k for k in dict.keys(self) if k not in _PATH_KEYS]

    def materialize(self):
        for k in _PATH_KEYS:
            self[k]
        return self

    def __contains__(self, k):
        return k in _PATH_KEYS or dict.__contains__(self, k)

    def __iter__(self):
        return iter(self._all_keys())

    def __len__(self):
        return len(self._all_keys())

    def keys(self):
        return self._all_keys()

    def values(self):
        return [self[k] for k in self._all_keys()]

    def items(self):
        return [(k, self[k]) for k in self._all_keys()]

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __eq__(self, other):
        return dict(self.items()) == (dict(other.items()) if isinstance(other, _LazyPath) else other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (dict(self.items()),))

    def __repr__(self):
        return f"<path {self._i}: {len(self['rewards'])} steps>"


class PathBatch(Sequence):
    """Completed paths of one obtain_samples call.

    Device side: time-major trajectory tensors of the engine (valid until the next call) plus
    ``env_idx`` / ``start`` / ``length`` [P] (int64, device).  ``batch[i]`` and iteration give the i-th
    reference path dict (keys and shapes of SURVEY.md §3.2) as a lazily filled dict: each value comes to the
    host the first time it is read, one engine buffer at a time."""

    def __init__(self, engine, env_idx, start, length, n_agents):
        self.engine, self.env_idx, self.start, self.length = engine, env_idx, start, length
        self.n_agents = n_agents
        self._bufs = {}                       # engine buffer name -> numpy (host copies made so far)
        self._idx = None
        self._T = None
        self._paths = {}                      # index -> the ONE _LazyPath of that path (keys a caller sets on it persist)
        self._generation = getattr(engine, "generation", None)

    def __len__(self):
        return int(self.length.numel())

    @property
    def n_samples(self):
        return int(self.length.sum().item()) * self.n_agents

    @property
    def host_buffers(self):
        """Names of the engine buffers copied to the host so far (tests pin the cheap paths with this)."""
        return sorted(self._bufs)

    def _index(self):
        if self._idx is None:
            self._idx = (self.env_idx.cpu().numpy(), self.start.cpu().numpy(), self.length.cpu().numpy())
            self._T = int((self._idx[1] + self._idx[2]).max()) if len(self._idx[0]) else 0
        return self._idx

    def _buf(self, name):
        if name not in self._bufs:
            if self._generation != getattr(self.engine, "generation", None):
                raise RuntimeError("this PathBatch belongs to an earlier rollout: the engine's trajectory buffers were "
                                   "overwritten by a later obtain_samples / reset (materialize() the paths you keep)")
            self._index()
            t = getattr(self.engine, name)
            extra = 1 if name in ("obs", "dist_adj", "channels") else 0
            self._bufs[name] = None if t is None else t[:self._T + extra].cpu().numpy()
        return self._bufs[name]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        p = self._paths.get(i)
        if p is None:
            p = self._paths[i] = _LazyPath(self, i)
        return p

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def _build(self, i, key):
        env_idx, start, length = self._index()
        e = self.engine
        env = e.env
        b, s, n = int(env_idx[i]), int(start[i]), int(length[i])
        N, Lh = env.N, env.Lh
        sl = slice(s, s + n)
        pp = env.scenario == "pp"
        if key == "observations":
            return self._buf("obs")[sl, b].reshape(n, -1).astype(np.float64)
        if key == "actions":
            return self._buf("actions")[sl, b].astype(np.int64)
        if key == "avail_actions":
            return np.ones((n, N * 5), dtype=np.int64)
        if key == "rewards":
            return self._buf("reward64")[sl, b].copy()
        if key == "dones":
            return self._buf("done")[sl, b].astype(bool)
        if key == "dist_adjs":
            h = self._buf("dist_adj")
            return h[sl, b].reshape(n, N * N) if h is not None else np.ones((n, N * N), np.float64)   # get_graph Rcom == 0 (:219-223)
        if key == "channels":
            h = self._buf("channels")
            if h is not None:
                return h[sl, b].reshape(n, Lh * N, N)
            if "_const_ch" not in self._bufs:
                self._bufs["_const_ch"] = env.channels[0].cpu().numpy().reshape(Lh * N, N)
            return np.broadcast_to(self._bufs["_const_ch"], (n, Lh * N, N)).copy()
        if key == "attentions":
            h = self._buf("attn")
            return None if h is None else h[sl, b]
        if key == "ave_degs":
            h = self._buf("dist_adj")
            return h[sl, b].reshape(n, N, N).sum(-1).mean(-1) if h is not None else np.full(n, N)
        if key == "diameters":
            return np.full(n, N if e.dist_adj is None else 0)
        if key == "ave_trputs":
            return np.full(n, env.n_empty_cells if not pp else 0)
        if key == "success":
            return self._buf("success")[s + n - 1].astype(np.int64)       # [n_envs], read when the path ended (:194)
        if key == "agent_infos":
            pr, at = self._buf("probs"), self._buf("attn")
            return dict(action_probs=None if pr is None else pr[sl, b], attention_weights=None if at is None else at[sl, b])
        if key == "env_infos":
            h = self._buf("prey_alive")
            return dict(prey_alive=h[sl, b].astype(bool)) if h is not None else {}
        if key == "rewards_details":
            det, rew = self._buf("details")[sl, b], self._buf("reward64")
            nA = float(N)
            details = []
            for k in range(n):
                dd = det[k]
                if pp:
                    details.append(dict(reward=float(rew[s + k, b]), capture_cnt=int(dd[0]), step_cnt=1,
                                        move_cnt=dd[1] / nA, penalty_cnt=int(dd[2]), variable=dd[4] / nA, vars2=0))
                else:
                    details.append(dict(reward=float(rew[s + k, b]), capture_cnt=dd[0] / nA, step_cnt=1,
                                        move_cnt=dd[1] / nA, penalty_cnt=dd[2] / nA, variable=dd[4] / nA, vars2=dd[3] / nA))
            return np.asarray(details)
        raise KeyError(key)


class CentralizedMAOnPolicyVectorizedSampler:
    """sampler.py:20.  ``env`` is a com_marl_amd.envs wrapper (or anything exposing ``.batch``)."""

    def __init__(self, algo, env, n_envs=None):
        self.algo = algo
        self.env = env
        base = getattr(env, "env", env)                  # tolerate a GarageEnv-style shell
        self._base = base
        self.batch = base.batch
        self._n_envs = self.batch.B if n_envs is None else n_envs
        if self._n_envs != self.batch.B:
            raise ValueError(f"n_envs={n_envs} but the env batch holds {self.batch.B} envs: build the env with n_envs")
        self._n_agents = self.batch.N
        self._vec_env = None
        self._env_spec = getattr(env, "spec", None)
        self.engine = None
        self._capacity = 0

    def start_worker(self):
        """Reference builds the VecEnvExecutor here (:41-56); we size the device trajectory lazily."""
        self._vec_env = self

    def shutdown_worker(self):
        self.engine = None

    # VecEnvExecutor facade used by callers that reach through sampler._vec_env (ma_batch_polopt.py:113-118)
    @property
    def envs(self):
        return [self._base]

    @property
    def num_envs(self):
        return self._n_envs

    def _ensure_engine(self, horizon):
        if self.engine is None or self._capacity < horizon:
            self.engine = RolloutEngine(self.batch, self.algo.policy, horizon)
            self._capacity = horizon
        return self.engine

    def _kernel_split(self, eng):
        """Share of a step's GPU time spent in the policy kernel, measured once per engine with HIP events on a
        scratch step (policy forward x3, env step x1; the env state is snapshotted and restored, and the slots the
        scratch step writes are rewritten by the rollout).  The stepping loop runs policy + env as ONE fused launch
        per step, so the two timers the reference reports separately (sampler.py:236-237) are the loop's measured
        GPU time split by this ratio."""
        if getattr(eng, "_pol_share", None) is not None:
            return eng._pol_share
        part, (lo, hi) = eng.parts[0], eng.bounds[0]
        nb = hi - lo
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        saved = part.get_state()
        kw = dict(out_actions=eng.actions[0][lo:hi], out_probs=None if eng.probs is None else eng.probs[0][lo:hi],
                  out_attn=None if eng.attn is None else eng.attn[0][lo:hi], policy_step=0, step_base=eng.step_base,
                  env_id_offset=eng.id0 + lo)
        args = (eng.obs[0][lo:hi].view(nb, -1), None, None if eng.dist_adj is None else eng.dist_adj[0][lo:hi],
                None if eng.channels is None else eng.channels[0][lo:hi])
        self.algo.policy.act_device(*args, **kw)                       # untimed first call
        ev[0].record()
        for _ in range(3):
            self.algo.policy.act_device(*args, **kw)
        ev[1].record()
        part.step_device(eng.actions[0][lo:hi], out=eng._out(0, lo, hi))
        ev[2].record()
        ev[2].synchronize()
        t_pol, t_env = ev[0].elapsed_time(ev[1]) / 3.0, ev[1].elapsed_time(ev[2])
        part.set_state(**saved)
        eng._pol_share = t_pol / max(t_pol + t_env, 1e-9)
        return eng._pol_share

    def obtain_samples(self, itr, batch_size=None, whole_paths=True, chunk=16, use_graph=None):
        """Roll until the completed paths hold >= batch_size agent-steps (:119), checking the stop rule once per `chunk`
        steps on the device; returns the completed paths up to the exact step at which the reference loop would have
        stopped.  The stepping loop replays one captured hipGraph per `chunk`-step span of trajectory slots
        (RolloutEngine.run_span; captured on the first rollout of an engine, reused by every later one) - eager
        stepping (``use_graph=False`` / COMMARL_SAMPLER_GRAPH=0) gives the same PathBatch, 2.3x slower at the headline
        config.  No check happens before step batch_size / (B N): the completed paths cannot hold the batch earlier."""
        mpl = self.algo.max_path_length
        B, N = self._n_envs, self._n_agents
        if not batch_size:
            batch_size = mpl * B
        if use_graph is None:
            use_graph = os.environ.get("COMMARL_SAMPLER_GRAPH", "1") != "0"
        chunk = max(1, int(chunk))
        # after t steps the completed paths hold >= B*N*(t - mpl) samples  =>  t <= batch/(B*N) + mpl
        t_min = int(np.ceil(batch_size / (B * N)))
        bound = t_min + mpl
        horizon = -(-bound // chunk) * chunk                       # whole spans: a rollout may run past its stop step
        eng = self._ensure_engine(horizon)
        policy = self.algo.policy
        policy.sync_weights()
        t_all = time.time()
        eng.reset()
        policy.reset([True] * B)
        pol_share = self._kernel_split(eng)
        if use_graph and not getattr(eng, "_spans_ready", False):
            # every span graph of the horizon once, up front: a later rollout that runs one span longer must not pay a
            # capture (capturing executes nothing: the rollout below starts from the state the reset left)
            for t0 in range(0, eng.H, chunk):
                eng.prepare_graph(min(chunk, eng.H - t0), t0, tail=False)
            eng._spans_ready = True
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        t, stop_t = 0, None
        counted, base = 0, 0                                       # steps whose completed samples are in `base` already
        while stop_t is None:
            t1 = min(t + chunk, eng.H)
            if use_graph:
                eng.run_span(t, t1 - t)
            else:
                for k in range(t, t1):
                    eng.step(k)
                eng.join()
            if t1 >= t_min:
                # stop rule on device: first step at which cumulative completed samples >= batch_size (one host read)
                cs = eng.path_len[counted:t1].sum(dim=1, dtype=torch.int64).cumsum(0) * N + base
                hit = cs >= batch_size
                first, last = torch.stack([torch.where(hit.any(), hit.to(torch.int64).argmax(), hit.new_full((), -1, dtype=torch.int64)),
                                           cs[-1]]).tolist()
                if first >= 0:
                    stop_t = counted + first + 1
                elif t1 >= eng.H:
                    raise RuntimeError("sampler horizon exhausted before batch_size was reached (bug in the bound)")
                counted, base = t1, last
            t = t1
        ev1.record()
        # fresh action-sampling draws for the next rollout (the reference's generator simply keeps advancing): every draw
        # a returned path used sits at a policy step < stop_t <= bound, whatever the span length was
        eng.bump(bound)
        self.batch.check_status()
        T = stop_t
        # completed paths within [0, T): every (t, b) with path_len > 0
        pl = eng.path_len[:T]
        idx = torch.nonzero(pl)                                    # [P, 2] (t, b), ordered by t then b
        length = pl[idx[:, 0], idx[:, 1]].to(torch.int64)
        start = idx[:, 0] + 1 - length
        paths = PathBatch(eng, idx[:, 1], start, length, N)
        torch.cuda.synchronize(self.batch.device)
        total = time.time() - t_all
        gpu = ev0.elapsed_time(ev1) * 1e-3                         # seconds of the stepping loop on the device
        tabular.record('PolicyExecTime', gpu * pol_share)          # sampler.py:236-239
        tabular.record('EnvExecTime', gpu * (1.0 - pol_share))
        tabular.record('ProcessExecTime', max(total - gpu, 0.0))   # host-side bookkeeping around the device loop
        tabular.record('BoundReturn', float(getattr(self._base, "bound_return", 0.0)))
        self.last_steps = T
        self.last_steps_run = t
        return paths if whole_paths else truncate_paths(paths, batch_size)


def truncate_paths(paths, max_samples):
    """What ``whole_paths=False`` does in the reference (sampler :245 -> garage/sampler/utils.py:91-140): keep the shortest
    prefix of paths holding >= max_samples steps, then cut the last kept path to the exact count - but only the keys
    observations / actions / rewards / env_infos / agent_infos are accepted there, and the Com-MARL path dicts carry more
    (:206-221), so the reference raises ValueError at the first other key ('avail_actions', third in the dict).  Same
    outcome here, key order included; no runner passes whole_paths=False."""
    accepted_arrays, accepted_dicts = ('observations', 'actions', 'rewards'), ('env_infos', 'agent_infos')
    lens = [len(p['rewards']) for p in paths]
    keep = len(lens)
    while keep > 0 and sum(lens[:keep - 1]) >= max_samples:       # paths beyond the prefix that already holds the batch
        keep -= 1
    out = [paths[i] for i in range(keep)]
    if not out:
        return out
    cut = max_samples - sum(lens[:keep - 1])                      # steps the last kept path may contribute
    last, short = out[-1], {}
    for key in last.keys():
        if key in accepted_arrays:
            short[key] = last[key][:cut]
        elif key in accepted_dicts:
            short[key] = {k: (None if v is None else v[:cut]) for k, v in last[key].items()}
        else:
            raise ValueError('Unexpected key {} found in path. Valid keys: {}'.format(
                key, set(accepted_arrays + accepted_dicts)))
    out[-1] = short
    return out
