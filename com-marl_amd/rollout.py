"""Device-resident rollout engine: the inner loop of
CentralizedMAOnPolicyVectorizedSampler.obtain_samples (com_marl/sampler/
centralized_ma_on_policy_vectorized_sampler.py:119-232) for B envs at once.

One step = one fused launch (cm_rollout_step) where the library has a fused kernel for the shape, else two kernel
launches; either way no host synchronisation and no copies:
    policy forward + sample  (cm_policy_forward)  reads  obs[t], dist_adj[t], channels[t]
                                                   writes actions[t], probs[t], attn[t]
    env step + auto-reset    (cm_env_step)        reads  actions[t]
                                                   writes obs[t+1], reward[t], done[t], ... , dist_adj[t+1], channels[t+1]
The trajectory buffers are time-major [H(+1), B, ...] in HBM, so every step reads and writes
contiguous [B, ...] slabs and the kernels write straight into them (zero-copy).  A chunk of
steps can be captured into a hipGraph and replayed (the launch-bound regime at 4096 x N=4).
"""
import contextlib
import os

import torch

from . import _lib as L

_null = contextlib.nullcontext


class _Parts:
    """Facade over several GridEnvBatch shards that together form one batch (global env ids are
    contiguous: shard k starts at the end of shard k-1)."""

    def __init__(self, parts):
        self.parts = parts
        e0 = parts[0]
        self.device, self.N, self.M, self.d, self.Lh = e0.device, e0.N, e0.M, e0.d, e0.Lh
        self.adj_const, self.ch_const, self.scenario = e0.adj_const, e0.ch_const, e0.scenario
        self.n_empty_cells, self.channels = e0.n_empty_cells, e0.channels
        self.B = sum(p.B for p in parts)
        self.bounds, lo = [], 0
        for p in parts:
            assert (p.N, p.d, p.Lh, p.scenario) == (e0.N, e0.d, e0.Lh, e0.scenario) and p.device == e0.device
            assert p.cfg.env_id_offset == e0.cfg.env_id_offset + lo, "shards must cover contiguous global env ids"
            self.bounds.append((lo, lo + p.B))
            lo += p.B

    def check_status(self):
        for p in self.parts:
            p.check_status()


class RolloutEngine:
    def __init__(self, env, policy, horizon, store_attn=True, store_probs=True, fused="auto", persistent="auto",
                 graph_fused=True):
        """env: envs.GridEnvBatch, or a list of shards of one batch (then every shard runs its own
        policy -> env chain on its own HIP stream: the chains are independent, so kernels of different
        shards overlap and their phases drift apart instead of contending in lockstep).
        policy: nets.CommCategoricalMLPPolicy (or the Obs-DP / CENT policy) on the same device."""
        if isinstance(env, (list, tuple)):
            env = _Parts(list(env)) if len(env) > 1 else env[0]
        self.env, self.policy, self.H = env, policy, int(horizon)
        self.parts = env.parts if isinstance(env, _Parts) else [env]
        self.bounds = env.bounds if isinstance(env, _Parts) else [(0, env.B)]
        self.streams = [torch.cuda.Stream(device=env.device) for _ in self.parts] if len(self.parts) > 1 else [None]
        self.id0 = self.parts[0].cfg.env_id_offset
        dev, B, N, M, d, Lh = env.device, env.B, env.N, max(env.M, 1), env.d, env.Lh
        A = policy._action_dim
        store_attn = store_attn and hasattr(policy, "comm")     # Obs-DP / CENT policies have no attention output
        H = self.H
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        z = lambda *shape, dtype=f32: torch.zeros(*shape, dtype=dtype, device=dev)   # noqa: E731
        self.obs = z(H + 1, B, N, d)
        self.actions = z(H, B, N, dtype=i32)
        self.probs = z(H, B, N, A) if store_probs else None
        self.attn = z(H, B, N, N) if store_attn else None
        self.reward = z(H, B)
        self.reward64 = z(H, B, dtype=torch.float64)
        self.done = z(H, B, dtype=u8)
        self.details = z(H, B, 6, dtype=i32)
        self.prey_alive = z(H, B, M, dtype=u8) if env.M else None
        self.success = z(H, B, dtype=i32)
        self.path_len = z(H, B, dtype=i32)
        # constant masks are never stored per step (4N^2 [adj != const] rule of SURVEY.md §8d)
        self.dist_adj = None if env.adj_const else z(H + 1, B, N, N)
        self.channels = None if env.ch_const else z(H + 1, B, Lh, N, N)
        # device-side Philox counter base of the action sampler (uint32 bits), ONE PER SHARD: every shard bumps its own
        # copy on its own stream (all copies always hold the same value), so that the shards' chains share nothing
        self.step_bases = [torch.zeros(1, dtype=i32, device=dev) for _ in self.parts]
        self._graphs = {}
        # fused: step() uses cm_rollout_step (policy forward + sample + env step in one launch) where the library has a fused
        # kernel for the shape.  fused="auto": teams of 4 only - for larger teams the fused kernel runs the env step of ONE env on
        # a 256-thread workgroup that is mostly idle (measured per step: N = 24 127 vs 95 us, N = 72 424 vs 176 us), so they take
        # the two-kernel form.  True forces it (parity tests), False disables it.
        # persistent: a chunk / span of steps as ONE cm_rollout_chunk launch per shard.  "auto": teams of 4 on the wave-owned
        # kernel (csrc/cm_rollout_w.hip: a wave keeps its four envs and the weights stay in LDS / registers for the whole
        # chunk - 17.0 us per step at the headline config against 24.7 for one launch per step, which re-stages 145 KB of
        # weights per workgroup every step); the older workgroup-tiled kernels (COMMARL_POLICY_KERNEL=h / f32) step faster
        # with one launch per step under a hipGraph and keep that form.  Both forms are bit-identical
        # (tests/test_hip_fused_parity.py).
        if fused == "auto":
            fused = getattr(policy, "_n_agents", 0) == 4
        if persistent == "auto":
            pk = os.environ.get("COMMARL_POLICY_KERNEL", "w")[:1]
            persistent = bool(fused) and getattr(policy, "_n_agents", 0) == 4 and pk not in ("h", "f", "v") \
                and os.environ.get("COMMARL_PERSISTENT", "1") != "0"
        self._fused = None if fused else False              # None = try the fused step, False = two launches per step
        self._persistent = bool(persistent and fused)
        self._capturing = False
        # captured chunks: one fused launch per step (cm_rollout_step) or the two-kernel form - measured per config in
        # bench.py; COMMARL_GRAPH_FUSED=0/1 forces either for A/B runs
        gf_env = os.environ.get("COMMARL_GRAPH_FUSED")
        self._fused_in_graph = bool(fused) and (gf_env == "1" if gf_env is not None else bool(graph_fused))
        self.t = 0
        self.generation = 0                      # bumped by reset(): a PathBatch of an earlier rollout refuses to read the buffers

    @property
    def step_base(self):
        return self.step_bases[0]

    def bump(self, n):
        """Advance the sampler's Philox counter base by n policy steps (what a chunk does at its end)."""
        for k, st in enumerate(self.streams):
            with torch.cuda.stream(st) if st is not None else _null():
                self.step_bases[k].add_(n)

    # ------------------------------------------------------------------------------------------
    def _out(self, t, lo, hi):
        o = dict(obs=self.obs[t + 1][lo:hi], reward=self.reward[t][lo:hi], reward_f64=self.reward64[t][lo:hi],
                 done=self.done[t][lo:hi], details=self.details[t][lo:hi], success=self.success[t][lo:hi],
                 path_len=self.path_len[t][lo:hi])
        if self.prey_alive is not None:
            o["prey_alive"] = self.prey_alive[t][lo:hi]
        if self.dist_adj is not None:
            o["dist_adj"] = self.dist_adj[t + 1][lo:hi]
        if self.channels is not None:
            o["channels"] = self.channels[t + 1][lo:hi]
        return o

    def reset(self):
        """VecEnvExecutor.reset: every env restarts; slot 0 receives the first observation."""
        for part, (lo, hi) in zip(self.parts, self.bounds):
            o = dict(obs=self.obs[0][lo:hi])
            if self.dist_adj is not None:
                o["dist_adj"] = self.dist_adj[0][lo:hi]
            if self.channels is not None:
                o["channels"] = self.channels[0][lo:hi]
            part.reset_all(out=o)
        self.t = 0
        self.generation += 1

    def _step_part(self, k, t, greedy):
        part, (lo, hi) = self.parts[k], self.bounds[k]
        nb = hi - lo
        if self._fused is not False and (not self._capturing or self._fused_in_graph) and hasattr(self.policy, "step_fused"):
            # policy forward + sample + env step of this shard in one launch (cm_rollout_step); shapes without a fused
            # kernel report "not available" once and the two-launch path below is used from then on
            ok = self.policy.step_fused(
                part, self.obs[t][lo:hi].view(nb, -1),
                None if self.dist_adj is None else self.dist_adj[t][lo:hi],
                None if self.channels is None else self.channels[t][lo:hi],
                part._out(self._out(t, lo, hi)), greedy=greedy, out_actions=self.actions[t][lo:hi],
                out_probs=None if self.probs is None else self.probs[t][lo:hi],
                out_attn=None if self.attn is None else self.attn[t][lo:hi],
                policy_step=t, step_base=self.step_bases[k], env_id_offset=self.id0 + lo)
            self._fused = ok
            if ok:
                return
        self.policy.act_device(
            self.obs[t][lo:hi].view(nb, -1), None,
            None if self.dist_adj is None else self.dist_adj[t][lo:hi],
            None if self.channels is None else self.channels[t][lo:hi],
            greedy=greedy, out_actions=self.actions[t][lo:hi],
            out_probs=None if self.probs is None else self.probs[t][lo:hi],
            out_attn=None if self.attn is None else self.attn[t][lo:hi],
            want_probs=self.probs is not None, want_attn=self.attn is not None,
            policy_step=t, step_base=self.step_bases[k], env_id_offset=self.id0 + lo)
        part.step_device(self.actions[t][lo:hi], out=self._out(t, lo, hi))

    def step(self, t, greedy=False):
        """Slot t -> t+1 (asynchronous).  With several shards each chain goes to its own stream; call
        join() (or run_chunk) before the host or another stream consumes the slot."""
        if len(self.parts) == 1:
            self._step_part(0, t, greedy)
            return
        for k, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                self._step_part(k, t, greedy)

    def fork(self):
        cur = torch.cuda.current_stream(self.env.device)
        for st in self.streams:
            if st is not None:
                st.wait_stream(cur)

    def join(self):
        cur = torch.cuda.current_stream(self.env.device)
        for st in self.streams:
            if st is not None:
                cur.wait_stream(st)

    def _wrap_part(self, k, n):
        """Carry the last slot written by an n-step chunk into slot 0 for shard k (what `obses = next_obses` does)."""
        lo, hi = self.bounds[k]
        self.obs[0][lo:hi].copy_(self.obs[n][lo:hi])
        if self.dist_adj is not None:
            self.dist_adj[0][lo:hi].copy_(self.dist_adj[n][lo:hi])
        if self.channels is not None:
            self.channels[0][lo:hi].copy_(self.channels[n][lo:hi])

    def _chunk_tail(self, k, n):
        """Shard k: counter bump + slot n -> slot 0 in one launch (cm_chunk_tail) on the current stream."""
        lo, hi = self.bounds[k]
        pairs = [(b[n][lo:hi], b[0][lo:hi]) for b in (self.obs, self.dist_adj, self.channels) if b is not None]
        args = []
        for src, dst in pairs + [(None, None)] * (3 - len(pairs)):
            args += [L.ptr(src), L.ptr(dst), 0 if src is None else src.numel() * src.element_size()]
        with torch.cuda.device(self.env.device):
            L.check(L.lib().cm_chunk_tail(L.ptr(self.step_bases[k]), n, *args, L.current_stream()), "cm_chunk_tail")

    def _strides(self):
        e = self.env
        B, N, M, d, Lh, A = e.B, e.N, max(e.M, 1), e.d, e.Lh, self.policy._action_dim
        return L.ChunkStrides(obs=B * N * d, actions=B * N, probs=B * N * A, attn=B * N * N, reward=B, reward_f64=B,
                              done=B, details=B * 6, dist_adj=B * N * N, channels=B * Lh * N * N, prey_alive=B * M,
                              success=B, path_len=B)

    def steps_fused(self, t0, n, greedy=False, tail=False):
        """Slots t0 .. t0+n-1 in one persistent launch per shard (cm_rollout_chunk); tail: followed by the chunk's tail - slot
        t0+n into slot 0, Philox base += n (cm_rollout_chunk_tail: the same launch where the library can).  False when the
        library has no fused kernel for this shape (nothing was launched)."""
        if self._fused is False or not hasattr(self.policy, "chunk_fused") or getattr(self.parts[0].cfg, "rng_mode", 0) != L.RNG_PHILOX:
            return False
        st = self._strides()
        self.fork()
        for k, stream in enumerate(self.streams):
            part, (lo, hi) = self.parts[k], self.bounds[k]
            nb = hi - lo
            with torch.cuda.stream(stream) if stream is not None else _null():
                ok = self.policy.chunk_fused(
                    part, n, st, self.obs[t0][lo:hi].view(nb, -1),
                    None if self.dist_adj is None else self.dist_adj[t0][lo:hi],
                    None if self.channels is None else self.channels[t0][lo:hi],
                    part._out(self._out(t0, lo, hi)), greedy=greedy, out_actions=self.actions[t0][lo:hi],
                    out_probs=None if self.probs is None else self.probs[t0][lo:hi],
                    out_attn=None if self.attn is None else self.attn[t0][lo:hi],
                    policy_step=t0, step_base=self.step_bases[k], env_id_offset=self.id0 + lo,
                    tail_next=None if not tail else (self.obs[0][lo:hi], None if self.dist_adj is None else self.dist_adj[0][lo:hi],
                                                     None if self.channels is None else self.channels[0][lo:hi]))
            if not ok:
                assert k == 0, "fused chunk availability must not differ between shards"
                self._fused = False
                self.join()
                return False
        self.join()
        self._fused = True
        return True

    def _chunk(self, n, t0=0, tail=True):
        """All shards: fork, every shard's n-step chain over slots t0 .. t0+n-1 (+ its tail) on its own stream, join.  The
        launches are issued step by step across the shards (t outer, shard inner): a captured graph submits its nodes in
        capture order, so issuing one shard's whole chain first would start the other shard's chain only after it
        (measured: ~100 us stagger per replay, profiles/r02_trace_short.txt).  tail=False (the sampler's spans): no carry
        of slot t0+n into slot 0 and no bump of the sampler's Philox base - the caller bumps once per rollout."""
        spl = int(os.environ.get("COMMARL_STEPS_PER_LAUNCH", "1"))
        if spl > 1 and self._fused is not False and self._fused_in_graph:
            # experiment: several steps per launch (cm_rollout_chunk with a short trip count) - one grid drain per spl steps
            t = 0
            while t < n:
                m = min(spl, n - t)
                if not self.steps_fused(t0 + t, m):
                    raise L.CommarlError("COMMARL_STEPS_PER_LAUNCH needs the fused chunk kernel")
                t += m
            if tail:
                for k, st in enumerate(self.streams):
                    with torch.cuda.stream(st) if st is not None else _null():
                        self._chunk_tail(k, t0 + n)
                self.join()
            return
        if self._persistent and self.steps_fused(t0, n, tail=tail):    # one launch per shard for the whole span, its tail included
            return
        self.fork()
        for t in range(t0, t0 + n):
            for k, st in enumerate(self.streams):
                with torch.cuda.stream(st) if st is not None else _null():
                    self._step_part(k, t, False)
        if tail:
            for k, st in enumerate(self.streams):
                with torch.cuda.stream(st) if st is not None else _null():
                    self._chunk_tail(k, t0 + n)
        self.join()

    def prepare_graph(self, n=None, t0=0, tail=True):
        """Capture + instantiate the hipGraph of an n-step chunk over slots t0 .. t0+n-1 (t0 + n <= H, default the whole
        horizon) WITHOUT advancing the rollout.
        One graph holds every shard's chain as a parallel branch (fork ... join).  Measured alternatives
        (profiles/r02_steps_sweep.txt, config 2): one single-stream graph per shard replayed on the shards' own
        streams runs 37.7 us/step against 36.4 for the joint graph - independent streams start in phase, so policy
        kernels meet policy kernels, whereas the joint graph's second branch starts ~100 us (three steps) after the
        first and the shards stay out of phase; that same stagger is why a SHORT run (bench.py --steps 20) is better
        off with a single shard.
        One-time host-side setup that must not happen inside a capture (the weight pack, the kernels'
        hipFuncSetAttribute calls) is triggered by one scratch step whose effects are undone: the env state is
        snapshotted before and restored after it, and every trajectory slot it wrote is rewritten by the chunk itself.
        Call it before a timed region; run_chunk() / run_span() call it on first use otherwise."""
        n = self.H if n is None else int(n)
        t0 = int(t0)
        assert n >= 1 and t0 >= 0 and t0 + n <= self.H
        key = (t0, n, bool(tail))
        g = self._graphs.get(key)
        if g is not None:
            return g
        self.policy.sync_weights()
        if not self._graphs:                                # first capture of this engine: the scratch step
            saved = [part.get_state() for part in self.parts]
            saved_slot = [b[t0:t0 + 2].clone() for b in (self.obs, self.dist_adj, self.channels) if b is not None]
            self._capturing = True                          # the launch form the capture will issue
            try:
                self.fork()
                self.step(t0)
                self.join()
            finally:
                self._capturing = False
            torch.cuda.synchronize(self.env.device)
            for part, st in zip(self.parts, saved):
                part.set_state(**st)
            # slot t0 is the chunk's INPUT when the sampler captures mid-rollout (the scratch step only wrote slot t0 + 1
            # and the per-step outputs, all rewritten by the chunk; restoring both slots keeps the argument simple)
            for b, keep in zip([b for b in (self.obs, self.dist_adj, self.channels) if b is not None], saved_slot):
                b[t0:t0 + 2].copy_(keep)
        torch.cuda.synchronize(self.env.device)
        g = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            # "thread_local": the capturing thread may not make a call a capture rejects, other threads (RCCL's watchdog)
            # are not this capture's business.  The one such call this package could make - cm_env_destroy -> hipFree of a
            # garbage-collected env handle, the cause of round 2's invalidated captures - is kept out by L.capture_guard():
            # garbage is collected before the capture opens and handle destruction is deferred until it has closed.
            with L.capture_guard():
                with torch.cuda.graph(g, capture_error_mode=os.environ.get("COMMARL_CAPTURE_MODE", "thread_local")):
                    self._chunk(n, t0, tail)
        finally:
            self._capturing = False
        self._graphs[key] = g
        return g

    def run_chunk(self, use_graph=True, n=None, weights_synced=False):
        """n steps (default: the whole horizon H) from slot 0, filling slots 0..n-1 (+ slot n of obs / masks), then
        carrying slot n into slot 0 and advancing the sampler's Philox base by n (cm_chunk_tail, per shard): one
        persistent launch per shard where the library has a fused kernel for the shape and the engine was built with
        persistent=True; otherwise n x (policy, env) launches per shard - replayed from one hipGraph per chunk length
        with use_graph.  The graph path and the eager path produce the same trajectory slot by slot
        (tests/test_hip_ppo_parity.py)."""
        n = self.H if n is None else int(n)
        assert 1 <= n <= self.H
        if not use_graph:
            if not weights_synced:
                self.policy.sync_weights()
            self._chunk(n)
            return
        g = self._graphs.get((0, n, True)) or self.prepare_graph(n)
        if not weights_synced:              # in-place refresh of the weight pack the graph points at; a caller that steps
            self.policy.sync_weights()      # many chunks between optimiser steps syncs once itself (~15 us of host time)
        g.replay()

    def run_span(self, t0, n, use_graph=True, weights_synced=True):
        """Slots t0 .. t0+n-1 -> t0+1 .. t0+n, nothing else (no carry into slot 0, no Philox bump): what obtain_samples
        strings together.  One hipGraph per (t0, n), captured on first use and reused by every later rollout of this
        engine (the trajectory slots, the weight pack and the Philox base are fixed device addresses)."""
        if not use_graph:
            if not weights_synced:
                self.policy.sync_weights()
            self._chunk(n, t0, tail=False)
            return
        g = self._graphs.get((t0, n, False)) or self.prepare_graph(n, t0, tail=False)
        if not weights_synced:
            self.policy.sync_weights()
        g.replay()

    def invalidate_graphs(self):
        """Call after the policy weights were re-packed at a new address (never needed when
        parameters are updated in place: the pack buffer is rewritten, see nets._packed)."""
        self._graphs.clear()
