"""CentralizedMAOnPolicyVectorizedSampler - same name, ctor and ``obtain_samples`` contract as
com_marl/sampler/centralized_ma_on_policy_vectorized_sampler.py:20-245, but the B envs really
are B independent envs (the reference's n_envs > 1 shares one env object) and
the whole loop stays on the GPU (rollout.RolloutEngine).

``obtain_samples`` returns a ``PathBatch``: a lazy ``Sequence`` of the reference's path dicts
(materialised to numpy only when indexed) that also exposes the device-resident trajectory and
the (env, start, length) index, which CentralizedMAPPO consumes directly without a host trip.
"""
import time
from collections.abc import Sequence

import numpy as np
import torch

from .rollout import RolloutEngine


class _Tabular:
    """dowel.tabular stand-in: the reference records timers into a global table (sampler.py:236-239)."""

    def __init__(self):
        self.rows = {}

    def record(self, k, v):
        self.rows[k] = v


tabular = _Tabular()


class PathBatch(Sequence):
    """Completed paths of one obtain_samples call.

    Device side: time-major trajectory tensors of the engine (valid until the next call) plus
    ``env_idx`` / ``start`` / ``length`` [P] (int64, device).  ``batch[i]`` builds the i-th
    reference path dict (keys and shapes of SURVEY.md §3.2)."""

    def __init__(self, engine, env_idx, start, length, n_agents):
        self.engine, self.env_idx, self.start, self.length = engine, env_idx, start, length
        self.n_agents = n_agents
        self._host = None

    def __len__(self):
        return int(self.length.numel())

    @property
    def n_samples(self):
        return int(self.length.sum().item()) * self.n_agents

    def _to_host(self):
        if self._host is None:
            e = self.engine
            T = int((self.start + self.length).max().item()) if len(self) else 0
            h = dict(env_idx=self.env_idx.cpu().numpy(), start=self.start.cpu().numpy(), length=self.length.cpu().numpy())
            for k in ("obs", "actions", "probs", "attn", "reward64", "done", "details", "prey_alive", "success",
                      "dist_adj", "channels"):
                t = getattr(e, k)
                h[k] = None if t is None else t[:T + (1 if k in ("obs", "dist_adj", "channels") else 0)].cpu().numpy()
            self._host = h
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        h = self._to_host()
        e = self.engine
        b, s, n = int(h["env_idx"][i]), int(h["start"][i]), int(h["length"][i])
        N, Lh = e.env.N, e.env.Lh
        sl = slice(s, s + n)
        env = e.env
        obs = h["obs"][sl, b].reshape(n, -1).astype(np.float64)
        if h["dist_adj"] is not None:
            adj = h["dist_adj"][sl, b].reshape(n, N * N)
        else:
            adj = np.ones((n, N * N), np.float64)                          # get_graph Rcom == 0 (:219-223)
        if h["channels"] is not None:
            ch = h["channels"][sl, b].reshape(n, Lh * N, N)
        else:
            c1 = env.channels[0].cpu().numpy().reshape(Lh * N, N)
            ch = np.broadcast_to(c1, (n, Lh * N, N)).copy()
        det = h["details"][sl, b]
        pp = env.scenario == "pp"
        nA = float(N)
        details = []
        for k in range(n):
            dd = det[k]
            if pp:
                details.append(dict(reward=float(h["reward64"][s + k, b]), capture_cnt=int(dd[0]), step_cnt=1,
                                    move_cnt=dd[1] / nA, penalty_cnt=int(dd[2]), variable=dd[4] / nA, vars2=0))
            else:
                details.append(dict(reward=float(h["reward64"][s + k, b]), capture_cnt=dd[0] / nA, step_cnt=1,
                                    move_cnt=dd[1] / nA, penalty_cnt=dd[2] / nA, variable=dd[4] / nA, vars2=dd[3] / nA))
        attn = h["attn"][sl, b] if h["attn"] is not None else None
        probs = h["probs"][sl, b] if h["probs"] is not None else None
        ave_deg = adj.reshape(n, N, N).sum(-1).mean(-1) if h["dist_adj"] is not None else np.full(n, N)
        path = dict(
            observations=obs, actions=h["actions"][sl, b].astype(np.int64),
            avail_actions=np.ones((n, N * 5), dtype=np.int64), rewards=h["reward64"][sl, b].copy(),
            rewards_details=np.asarray(details), dones=h["done"][sl, b].astype(bool),
            dist_adjs=adj, channels=ch, attentions=attn,
            ave_degs=ave_deg, diameters=np.full(n, N if h["dist_adj"] is None else 0),
            ave_trputs=np.full(n, env.n_empty_cells if not pp else 0),
            success=h["success"][s + n - 1].astype(np.int64),             # [n_envs], read when the path ended (:194)
            agent_infos=dict(action_probs=probs, attention_weights=attn),
            env_infos=dict(prey_alive=h["prey_alive"][sl, b].astype(bool)) if h["prey_alive"] is not None else {})
        return path


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

    def obtain_samples(self, itr, batch_size=None, whole_paths=True, chunk=32):
        """Roll until the completed paths hold >= batch_size agent-steps (:119), checking the stop
        rule once per `chunk` steps on the device; returns the completed paths up to the exact step
        at which the reference loop would have stopped."""
        mpl = self.algo.max_path_length
        B, N = self._n_envs, self._n_agents
        if not batch_size:
            batch_size = mpl * B
        # after t steps the completed paths hold >= B*N*(t - mpl) samples  =>  t <= batch/(B*N) + mpl
        horizon = int(np.ceil(batch_size / (B * N))) + mpl
        eng = self._ensure_engine(horizon)
        policy = self.algo.policy
        policy.sync_weights()
        t_pol = t_env = 0.0
        t_all = time.time()
        eng.reset()
        policy.reset([True] * B)
        t, stop_t = 0, None
        while stop_t is None:
            t1 = min(t + chunk, horizon)
            for k in range(t, t1):
                eng.step(k)
            eng.join()
            # stop rule on device: first step at which cumulative completed samples >= batch_size
            done_samples = eng.path_len[:t1].sum(dim=1, dtype=torch.int64).cumsum(0) * N
            hit = torch.nonzero(done_samples >= batch_size)
            if hit.numel():
                stop_t = int(hit[0].item()) + 1
            elif t1 >= horizon:
                raise RuntimeError("sampler horizon exhausted before batch_size was reached (bug in the bound)")
            t = t1
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
        tabular.record('PolicyExecTime', t_pol)                   # not separable without per-step syncs
        tabular.record('EnvExecTime', t_env)
        tabular.record('ProcessExecTime', total)
        tabular.record('BoundReturn', float(getattr(self._base, "bound_return", 0.0)))
        self.last_steps = T
        if not whole_paths:
            raise NotImplementedError("whole_paths=False (truncate_paths) is unused by the runners")
        return paths
