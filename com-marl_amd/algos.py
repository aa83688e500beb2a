"""CentralizedMAPPO - the PPO update of com_marl/torch/algos/centralized_ma_ppo.py:39-659 with
the trajectory resident in HBM.

Same ctor kwargs and the same numbers:
returns f64 recurrence -> f32, GAE over the padded length with V of padded steps as the critic
outputs them, per-path advantage normalisation (biased variance, eps 1e-8), clip range fixed at
0.1, entropy bonus, mean over valid steps, Gaussian-NLL critic loss over padded steps, grad-clip
on the policy only, two Adam optimisers (lr 3e-4, eps 1e-5), minibatches over shuffled paths.

What changed is where it runs: the padded [P,T,...] batch is gathered on the device from the
sampler's time-major buffers; returns / GAE / normalisation are HIP scan kernels
(cm_discount_returns, cm_gae); critic baselines, old-policy log-likelihoods and the KL/entropy
diagnostics are fused no-grad launches; the training forward shares ONE trunk evaluation for the
entropy and the new log-likelihood (the reference runs it twice); multi-GPU adds one RCCL
all-reduce of the flat (policy ‖ critic ‖ counts) gradient bucket per optimiser step.
"""
import collections
import copy
import os
import time

import numpy as np
import torch
from torch.distributions import Categorical

from . import _lib as L
from .sampler import CentralizedMAOnPolicyVectorizedSampler, PathBatch, tabular


class _SurrogateFn(torch.autograd.Function):
    """-(sum over valid steps of the clipped surrogate + entropy bonus) from the policy logits in one launch
    (cm_ppo_surrogate: the ~30 elementwise / reduction launches of :390-438 + :540-589 and as many again in their
    autograd); the gradient wrt the logits is produced by the same launch.  Returns (total f32, count int64)."""

    @staticmethod
    def forward(ctx, logits, actions, old_ll, adv, valids, clip, ent_coeff, add_entropy):
        P, T, N, A = logits.shape
        dev = logits.device
        lg = logits.contiguous()
        need = logits.requires_grad
        dl = torch.empty_like(lg) if need else None
        total = torch.empty((), dtype=torch.float64, device=dev)
        count = torch.empty((), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().cm_ppo_surrogate(P, T, N, A, L.ptr(lg), L.ptr(actions.contiguous()), L.ptr(old_ll.contiguous()),
                                             L.ptr(adv.contiguous()), L.ptr(valids), float(clip), float(ent_coeff), int(add_entropy),
                                             L.ptr(total), L.ptr(count), L.ptr(dl), L.current_stream()), "cm_ppo_surrogate")
        ctx.dl = dl
        ctx.mark_non_differentiable(count)
        return total.to(torch.float32), count

    @staticmethod
    def backward(ctx, g_total, _g_count):
        dl, ctx.dl = ctx.dl, None
        return (dl.mul_(g_total),) + (None,) * 7


def _fused_loss_ok(policy, obs, avail_actions, actions, valids):
    return (obs.is_cuda and avail_actions is None and hasattr(policy, "_logits") and policy._action_dim <= 8
            and actions.dtype == torch.int32 and valids.dtype == torch.int32 and os.environ.get("COMMARL_FUSED_LOSS", "1") != "0")


def _dist_ready():
    from .dist import is_distributed
    return is_distributed()


class _UpdateGraphs:
    """hipGraphs of the optimiser steps of train_once: the reference walks the SAME minibatches in each of its mini-epochs
    (centralized_ma_ppo.py:209-268: one permutation per epoch), so a minibatch's step is the same ~70 launches on the same
    buffers every time - only the weights, the Adam moments and the step number differ, and those live in device memory
    (optim.Adam.begin_device_steps).  The first mini-epoch of a process goes eagerly (first-use set-up inside the library,
    Adam's moments); after that a minibatch's step is captured the first time it comes up and every later step is one graph
    launch.  The graphs outlive the epoch: a later epoch whose minibatch i has the same shapes (the usual case at a fixed
    number of envs and path length) copies its tensors into the captured step's input buffers and replays - no capture at
    all (a capture costs as much host time as ~5 replayed steps).  For small batches the step is launch-bound (~3 ms of
    host work per eager step at the reference's 30 000 agent-steps per epoch); large batches keep the eager path
    (COMMARL_UPDATE_GRAPH=1 forces the graphs, =0 disables them; the default threshold is in agent rows per minibatch)."""
    MAX_ROWS = 1 << 18
    MAX_CACHED = 12

    @classmethod
    def maybe(cls, algo, minibatches, T, distributed):
        mode = os.environ.get("COMMARL_UPDATE_GRAPH", "auto")
        E = algo._optimization_mini_epochs
        obs = minibatches[0][0]
        if (mode == "0" or distributed or not obs.is_cuda or E < 3
                or not all(hasattr(o, "begin_device_steps") for o in (algo._optimizer, algo._baseline_optimizer))
                or E * len(minibatches) > algo._optimizer.DEV_STEP_CAPACITY
                or not _fused_loss_ok(algo.policy, obs, None, minibatches[0][1], minibatches[0][3])    # (the framework's Categorical
                or not all(getattr(n, "_graph_capturable_update", False) for n in (algo.policy, algo.baseline))):   # validates on the host)
            return None                                      # only the nets whose whole step is this library's launches (CommBaseNet)
        rows = max(mb[0].shape[0] for mb in minibatches) * T * getattr(algo.policy, "_n_agents", 1)
        if mode != "1" and rows > cls.MAX_ROWS:
            return None
        self = getattr(algo, "_update_graphs", None)
        owner = (id(algo.policy), id(algo.baseline), id(algo._optimizer), id(algo._baseline_optimizer),
                 next(algo.policy.parameters()).data_ptr(), next(algo.baseline.parameters()).data_ptr(), obs.device)
        if self is None or self.owner != owner:
            self = algo._update_graphs = cls(algo, owner)
        # the first optimiser steps of a process go eagerly (first-use set-up inside the library, Adam's moment buffers and norm
        # workspace); from then on a step is replayed (or captured) from mini-epoch 0
        first = 0 if getattr(algo, "_eager_stepped", False) else 1
        self._begin(len(minibatches), (E - first) * len(minibatches), first)
        # what a captured step has baked in besides its tensors: part of the cache key
        grp = lambda o: tuple((g["lr"], tuple(g["betas"]), g["eps"]) for g in o.param_groups)   # noqa: E731
        self.hyper = (T, grp(algo._optimizer), grp(algo._baseline_optimizer), algo._lr_clip_range, algo._policy_ent_coeff,
                      algo._entropy_regularzied, algo._clip_grad_norm, os.environ.get("COMMARL_CRITIC_STREAM", "1"))
        return self

    def __init__(self, algo, owner):
        self.algo, self.owner = algo, owner
        self.cache = collections.OrderedDict()               # (minibatch index, shapes) -> [graph, output, input tensors]
        self.pool, self.stream = None, None
        self.cur, self.done, self.n_steps, self.first_epoch = [], 0, 0, 1
        self.armed, self.broken, self.hyper = False, False, None
        self.captured = self.reused = 0                       # (counters: captures taken / earlier epochs' graphs taken over)

    def _begin(self, n_mb, n_steps, first_epoch):
        self.cur, self.done, self.n_steps, self.first_epoch = [None] * n_mb, 0, n_steps, first_epoch
        self.armed, self.broken = False, False

    def _key(self, i, inputs):
        return (i, self.hyper) + tuple(None if t is None else (tuple(t.shape), t.dtype) for t in inputs)

    def step(self, i, inputs, fn):
        """Step of minibatch i on `inputs` (its tensors: fn(*inputs) issues the step): replayed from its graph -> the step's
        gradient-norm output.  First time this epoch: the graph of an earlier epoch with the same shapes gets the tensors copied
        into its input buffers, else the step is captured with these tensors as its input buffers.  A capture that fails
        switches the rest of the epoch back to eager steps."""
        if self.broken:
            return fn(*inputs)
        if not self.armed:                                   # the moments exist, the step counts are known
            self.algo._optimizer.begin_device_steps(self.n_steps)
            self.algo._baseline_optimizer.begin_device_steps(self.n_steps)
            self.armed = True
        if self.cur[i] is None:
            key = self._key(i, inputs)
            ent = self.cache.get(key)
            if ent is not None and os.environ.get("COMMARL_UPDATE_GRAPH_REUSE", "1") != "0":
                with torch.no_grad():
                    for dst, src in zip(ent[2], inputs):
                        if dst is not None and dst.data_ptr() != src.data_ptr():
                            dst.copy_(src)
                self.cache.move_to_end(key)
                self.reused += 1
            else:
                try:
                    g, out = self._capture(lambda: fn(*inputs))
                except Exception as e:                       # noqa: BLE001 - whatever the capture tripped over, the step itself is fine
                    import warnings
                    warnings.warn(f"PPO update: hipGraph capture of the optimiser step failed ({type(e).__name__}: {e}); eager steps from here")
                    torch.cuda.synchronize()
                    self.close()
                    self.broken = True
                    return fn(*inputs)
                ent = self.cache[key] = [g, out, list(inputs)]
                self.captured += 1
                while len(self.cache) > self.MAX_CACHED:
                    self.cache.popitem(last=False)
            self.cur[i] = ent
        g, out = self.cur[i][0], self.cur[i][1]
        g.replay()
        self.done += 1
        return out

    def _capture(self, fn):
        # capture_begin / capture_end directly: torch.cuda.graph() would synchronise the device and empty the allocator's cache
        # in front of every capture
        g = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=cur.device)
        self.stream.wait_stream(cur)
        # a capture freezes the host's decisions: the weight-pack cache must not answer "fresh" (it would, right after the
        # epoch's no-grad forward) - the replays run after optimiser steps the host-side version counters never saw
        for net in (self.algo.policy, self.algo.baseline):
            if hasattr(net, "_pack_sig"):
                net._pack_sig = None
        with L.capture_guard(), torch.cuda.stream(self.stream):
            kw = {} if self.pool is None else dict(pool=self.pool)
            g.capture_begin(capture_error_mode=os.environ.get("COMMARL_CAPTURE_MODE", "thread_local"), **kw)
            try:
                out = fn()
            finally:
                g.capture_end()
        cur.wait_stream(self.stream)
        if self.pool is None:
            self.pool = g.pool()
        return g, out

    def close(self):
        """Book the epoch's replayed steps into the optimisers' host state (idempotent); the graphs stay for later epochs."""
        if self.armed:
            self.algo._optimizer.end_device_steps(self.done)
            self.algo._baseline_optimizer.end_device_steps(self.done)
            self.armed, self.done = False, 0
        self.cur = [None] * len(self.cur)


class CentralizedMAPPO:
    def __init__(self, env_spec, policy, baseline, optimizer=None, baseline_optimizer=None,
                 optimization_n_minibatches=1, optimization_mini_epochs=1, policy_lr=3e-4, lr_clip_range=2e-1,
                 max_path_length=500, num_train_per_epoch=1, discount=0.99, gae_lambda=1, center_adv=True,
                 positive_adv=False, policy_ent_coeff=0.0, use_softplus_entropy=False, stop_entropy_gradient=False,
                 entropy_method='no_entropy', clip_grad_norm=None, device='cpu'):
        self.device = device
        self.env_spec, self.policy, self.baseline = env_spec, policy, baseline
        self.discount, self.max_path_length, self.n_samples = discount, max_path_length, num_train_per_epoch
        self._gae_lambda, self._center_adv, self._positive_adv = gae_lambda, center_adv, positive_adv
        self._policy_ent_coeff = policy_ent_coeff
        self._use_softplus_entropy, self._stop_entropy_gradient = use_softplus_entropy, stop_entropy_gradient
        self._entropy_method = entropy_method
        self._lr_clip_range = 0.1                       # hard-coded in the reference (:115); ctor arg ignored
        self._eps = 1e-8
        self._maximum_entropy = entropy_method == 'max'
        self._entropy_regularzied = entropy_method == 'regularized'
        self._check_entropy_configuration(entropy_method, center_adv, stop_entropy_gradient, policy_ent_coeff)
        if use_softplus_entropy or self._maximum_entropy or stop_entropy_gradient:
            raise NotImplementedError("only the runners' entropy settings ('regularized' / 'no_entropy') are built")
        # default: the multi-tensor Adam of optim.py (two launches per step, clip folded in); it and torch.optim.Adam are
        # the same update as the vendored torch-1.9 Adam (my_optimizer/_functional.py:72-98): pinned by tests against the
        # reference's optimiser.
        from .optim import Adam as FusedAdam
        on_gpu = next(policy.parameters()).is_cuda
        opt = optimizer or (FusedAdam if on_gpu else torch.optim.Adam)
        bopt = baseline_optimizer or (FusedAdam if on_gpu else torch.optim.Adam)
        self._optimizer = opt(policy.parameters(), lr=policy_lr, eps=1e-5)
        self._baseline_optimizer = bopt(baseline.parameters(), lr=policy_lr, eps=1e-5)
        self._optimization_n_minibatches = optimization_n_minibatches
        self._optimization_mini_epochs = optimization_mini_epochs
        self._clip_grad_norm = clip_grad_norm
        self._old_policy = copy.deepcopy(self.policy)
        self._old_policy._pack_sig, self._old_policy._pack = None, None
        self.sampler_cls = CentralizedMAOnPolicyVectorizedSampler
        self.episode_reward_mean = collections.deque(maxlen=100)
        self.stats = {}

    def __getstate__(self):
        """Snapshots pickle the algo (snapshotter.py:102-104); a HIP stream is not picklable and is re-made on first use."""
        st = dict(self.__dict__)
        st.pop("_side_stream", None)
        st.pop("_bucket", None)
        st.pop("_eager_stepped", None)                   # per process: the first optimiser steps of a process run eagerly
        st.pop("_update_graphs", None)                   # hipGraphs of optimiser steps (rebuilt on demand)
        return st

    @staticmethod
    def _check_entropy_configuration(entropy_method, center_adv, stop_entropy_gradient, policy_ent_coeff):
        if entropy_method not in ('max', 'regularized', 'no_entropy'):
            raise ValueError('Invalid entropy_method')
        if entropy_method == 'max':
            if center_adv:
                raise ValueError('center_adv should be False when entropy_method is max')
            if not stop_entropy_gradient:
                raise ValueError('stop_gradient should be True when entropy_method is max')
        if entropy_method == 'no_entropy' and policy_ent_coeff != 0.0:
            raise ValueError('policy_ent_coeff should be zero when there is no entropy method')

    # ------------------------------------------------------------------------------------------
    # process_samples (:612-659)
    # ------------------------------------------------------------------------------------------
    def _dev(self):
        return next(self.policy.parameters()).device

    def process_samples(self, itr, paths):
        """-> (obs [P,T,N*d], avail (None = ones), actions [P,T,N], rewards [P,T] f32, valids [P] int32,
        baselines [P,T], returns [P,T], dist_adjs [P,T,N,N]|None, channels [P,T,L,N,N]|None), all on device.
        Zero padding for obs/actions/rewards/returns, ONE padding for the masks (App. A-5)."""
        dev = self._dev()
        if isinstance(paths, PathBatch):
            e = paths.engine
            lens = paths.length
            P, T = lens.numel(), int(lens.max().item())
            tau = torch.arange(T, device=dev)
            valid = tau[None, :] < lens[:, None]                                  # [P,T]
            t_idx = (paths.start[:, None] + tau[None, :]).clamp_(max=e.obs.shape[0] - 2)
            b_idx = paths.env_idx[:, None].expand(P, T)

            def gather(buf, pad):
                x = buf[t_idx, b_idx]                                              # [P,T,...]
                m = valid.reshape(P, T, *([1] * (x.dim() - 2)))
                return torch.where(m, x, torch.full((), pad, dtype=x.dtype, device=dev))
            obs = gather(e.obs, 0).reshape(P, T, -1)
            actions = gather(e.actions, 0)
            rew64 = gather(e.reward64, 0)
            dist_adjs = None if e.dist_adj is None else gather(e.dist_adj, 1)
            channels = None if e.channels is None else gather(e.channels, 1)
            valids = lens.to(torch.int32)
        else:                                                                       # reference list-of-dicts
            P = len(paths)
            T = max(len(p['rewards']) for p in paths)
            N = self.policy._n_agents
            Lh = np.asarray(paths[0]['channels']).shape[-2] // N                    # [T, L*N, N] (sampler.py:191)

            def pad(key, val, dtype, tail):
                out = torch.full((P, T) + tail, val, dtype=dtype)
                for i, p in enumerate(paths):
                    a = torch.as_tensor(np.asarray(p[key])).to(dtype).reshape((-1,) + tail)
                    out[i, :a.shape[0]] = a
                return out.to(dev)
            obs = pad('observations', 0, torch.float32, (paths[0]['observations'].shape[-1],))
            actions = pad('actions', 0, torch.int32, (N,))
            rew64 = pad('rewards', 0, torch.float64, ())
            dist_adjs = pad('dist_adjs', 1, torch.float32, (N, N))
            channels = pad('channels', 1, torch.float32, (Lh, N, N))
            valids = torch.tensor([len(p['actions']) for p in paths], dtype=torch.int32, device=dev)
        self.temp_max_path_length = T
        rewards = rew64.to(torch.float32).contiguous()                              # torch.Tensor(path['rewards']) (:645)
        returns = torch.empty(P, T, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().cm_discount_returns(P, T, L.ptr(rew64.contiguous()), L.ptr(valids), float(self.discount),
                                                L.ptr(returns), L.current_stream()), "cm_discount_returns")
        with torch.no_grad():                                                       # :653-657
            baselines = self._baseline_forward(obs, dist_adjs, channels)
        return obs, None, actions, rewards, valids, baselines, returns, dist_adjs, channels

    # baseline dispatch (:230-233, :654-657): the GNN critic takes the graph, plain baselines only obs
    def _baseline_forward(self, obs, dist_adjs, channels):
        if self.baseline.name in ['base_critic']:
            return self.baseline.forward(obs, None, dist_adjs, channels)
        return self.baseline.forward(obs)

    def _baseline_loss(self, obs, returns, dist_adjs, channels):
        if self.baseline.name in ['base_critic']:
            return self.baseline.compute_loss(obs, returns, dist_adjs, channels)
        return self.baseline.compute_loss(obs, returns)

    # ------------------------------------------------------------------------------------------
    # loss pieces
    # ------------------------------------------------------------------------------------------
    def _advantages(self, rewards, baselines, valids):
        """compute_advantages + per-path normalisation (:418-426) in one scan kernel."""
        P, T = rewards.shape
        adv = torch.empty_like(rewards)
        with torch.cuda.device(rewards.device):
            L.check(L.lib().cm_gae(P, T, L.ptr(rewards.contiguous()), L.ptr(baselines.contiguous()), L.ptr(valids),
                                   float(self.discount), float(self._gae_lambda), int(self._center_adv), self._eps,
                                   L.ptr(adv), L.current_stream()), "cm_gae")
        if self._positive_adv:
            adv = adv - adv.min()
        return adv

    @torch.no_grad()
    def _old_log_likelihood(self, obs, actions, dist_adjs, channels):
        """old_policy.log_likelihood (:561-566): one fused no-grad launch."""
        P, T = obs.shape[:2]
        N = self.policy._n_agents
        flat = lambda x: None if x is None else x.reshape(P * T, *x.shape[2:])      # noqa: E731
        _, probs, _ = self._old_policy.act_device(flat(obs), None, flat(dist_adjs), flat(channels), want_actions=False,
                                                  want_attn=False, policy_step=0)
        lp = torch.log(probs.gather(-1, actions.reshape(P * T, N, 1).long())).squeeze(-1)
        return lp.sum(-1).reshape(P, T)

    def _valid_mask(self, valids, T):
        return torch.arange(T, device=valids.device)[None, :] < valids[:, None]

    @staticmethod
    def _ll_from_probs(probs, actions):
        """log-likelihood of the taken joint action from action probabilities [P,T,N,A] -> [P,T] (:561-566)."""
        return torch.log(probs.gather(-1, actions.long().unsqueeze(-1))).squeeze(-1).sum(-1)

    @staticmethod
    def _kl_entropy(p_old, p_new):
        """KL(old || new) mean and entropy mean of `p_new` over ALL padded steps (:440-538) from two probability tensors."""
        kl = (p_old * (torch.log(p_old) - torch.log(p_new))).sum(-1).mean()
        ent = -(p_new * torch.log(p_new)).sum(-1).mean(-1).mean()
        return float(kl), float(ent)

    def _compute_loss(self, itr, obs, avail_actions, actions, rewards, valids, baselines, dist_adjs, channels,
                      advantages=None, old_ll=None, reduce=True, logits=None):
        """:390-438.  Returns -(mean over valid steps of clipped surrogate + c * entropy); with
        reduce=False returns (sum, count) for the count-weighted multi-GPU reduction."""
        T = obs.shape[1]
        if advantages is None:
            advantages = self._advantages(rewards, baselines, valids)
        if old_ll is None:
            old_ll = self._old_log_likelihood(obs, actions, dist_adjs, channels)
        if _fused_loss_ok(self.policy, obs, avail_actions, actions, valids):
            if logits is None:
                logits = self.policy._logits(obs, dist_adjs, channels)               # [P,T,N,A]
            total, count = _SurrogateFn.apply(logits, actions, old_ll, advantages, valids, self._lr_clip_range,
                                              self._policy_ent_coeff, self._entropy_regularzied)
            return (total, count) if not reduce else total / count
        probs, _ = self.policy._probs(obs, avail_actions, dist_adjs, channels)      # one trunk pass for both terms
        dist_n = Categorical(probs=probs)
        entropies = dist_n.entropy().mean(-1)                                        # policy.entropy (:121-126)
        new_ll = dist_n.log_prob(actions).sum(-1)                                    # policy.log_likelihood (:128-137)
        ratio = (new_ll - old_ll).exp()
        surrogate = ratio * advantages
        clipped = torch.clamp(ratio, min=1 - self._lr_clip_range, max=1 + self._lr_clip_range) * advantages
        objective = torch.min(surrogate, clipped)                                    # :540-589
        if self._entropy_regularzied:
            objective = objective + self._policy_ent_coeff * entropies               # :434-435
        mask = self._valid_mask(valids, T)
        total = -(objective * mask).sum()
        count = mask.sum()
        if not reduce:
            return total, count
        return total / count

    @torch.no_grad()
    def _diagnostics(self, obs, actions, valids, dist_adjs, channels):
        """KL(old || new) mean and policy entropy mean over ALL padded steps, as :440-538 compute them."""
        P, T = obs.shape[:2]
        flat = lambda x: None if x is None else x.reshape(P * T, *x.shape[2:])      # noqa: E731
        _, p_new, _ = self.policy.act_device(flat(obs), None, flat(dist_adjs), flat(channels), want_actions=False,
                                             want_attn=False, policy_step=0)
        _, p_old, _ = self._old_policy.act_device(flat(obs), None, flat(dist_adjs), flat(channels),
                                                  want_actions=False, want_attn=False, policy_step=0)
        kl = (p_old * (torch.log(p_old) - torch.log(p_new))).sum(-1).mean()
        ent = -(p_new * torch.log(p_new)).sum(-1).mean(-1).mean()
        return float(kl), float(ent)

    def _log_performance(self, itr, paths, returns, valids):
        """The progress.csv columns of centralized_ma_ppo.py:286-366 (per-path sums -> means over paths).
        With a PathBatch everything is reduced on the device and one small vector comes to the host."""
        N = self.policy._n_agents
        if isinstance(paths, PathBatch):
            e, dev = paths.engine, self._dev()
            lens = paths.length
            P, T = lens.numel(), int(lens.max().item())
            tau = torch.arange(T, device=dev)
            valid = tau[None, :] < lens[:, None]
            t_idx = (paths.start[:, None] + tau[None, :]).clamp_(max=e.reward64.shape[0] - 1)
            b_idx = paths.env_idx[:, None].expand(P, T)
            rew = torch.where(valid, e.reward64[t_idx, b_idx], torch.zeros((), dtype=torch.float64, device=dev))
            undisc = rew.sum(1)                                                      # np.sum(path_rewards), f64
            det = (e.details[t_idx, b_idx].to(torch.float64) * valid[..., None]).sum(1)   # [P,6]
            succ = e.success[paths.start + lens - 1, paths.env_idx].to(torch.float64)
            if e.dist_adj is not None:
                deg = (e.dist_adj[t_idx, b_idx].sum(-1).mean(-1).to(torch.float64) * valid).sum(1) / lens
                diam = torch.zeros(P, dtype=torch.float64, device=dev)               # get_graph: diameter 0 (:234)
            else:
                deg = torch.full((P,), float(N), dtype=torch.float64, device=dev)
                diam = deg.clone()
            pp = e.env.scenario == "pp"
            nA = float(N)
            cap = det[:, 0] if pp else det[:, 0] / nA
            pen = det[:, 2] if pp else det[:, 2] / nA
            var2 = torch.zeros_like(cap) if pp else det[:, 3] / nA
            cols = torch.stack([undisc, returns[:, 0].to(torch.float64), succ, cap, lens.to(torch.float64),
                                det[:, 1] / nA, pen, det[:, 4] / nA, var2, deg, diam])
            means = cols.mean(1).tolist()
            std, mx, mn = float(undisc.std(unbiased=False)), float(undisc.max()), float(undisc.min())
            trput = float(e.env.n_empty_cells if not pp else 0)
            undisc_list = undisc.tolist()
        else:
            def col(f):
                return [f(p) for p in paths]
            dsum = lambda k: col(lambda p: float(np.sum([d[k] for d in p['rewards_details']])))   # noqa: E731
            undisc_list = col(lambda p: float(np.sum(np.asarray(p['rewards']))))
            means = [np.mean(undisc_list), float(returns[:, 0].to(torch.float64).mean()),
                     np.mean(col(lambda p: np.mean(p['success']))), np.mean(dsum('capture_cnt')),
                     np.mean(dsum('step_cnt')), np.mean(dsum('move_cnt')), np.mean(dsum('penalty_cnt')),
                     np.mean(dsum('variable')), np.mean(dsum('vars2')),
                     np.mean(col(lambda p: np.mean(p['ave_degs']))), np.mean(col(lambda p: np.mean(p['diameters'])))]
            std, mx, mn = float(np.std(undisc_list)), float(np.max(undisc_list)), float(np.min(undisc_list))
            trput = float(np.mean(col(lambda p: np.mean(p['ave_trputs']))))
            P = len(paths)
        self.episode_reward_mean.extend(undisc_list)
        keys = ("AverageReturn", "AverageDiscountedReturn", "SuccessRate", "AverageCaptureCount", "AverageStepCount",
                "AverageMovingCount", "AveragePenaltyCount", "AverageVariable", "AverageVar2", "AveDegree", "Diameter")
        out = dict(Iteration=itr, NumTrajs=P * N)                                    # :346-347
        out.update({k: float(v) for k, v in zip(keys, means)})
        out.update(StdReturn=std, MaxReturn=mx, MinReturn=mn, AveTroughput=trput)
        return out

    # ------------------------------------------------------------------------------------------
    # gradient exchange
    # ------------------------------------------------------------------------------------------
    def _allreduce_grads(self, n_valid, n_crit):
        """One RCCL all-reduce of [policy grads ‖ critic grads ‖ n_valid ‖ n_crit] through the persistent bucket
        (dist.GradBucket): nothing is allocated and nothing waits for the host per optimiser step."""
        from .dist import GradBucket
        pol, cri = list(self.policy.parameters()), list(self.baseline.parameters())
        b = getattr(self, "_bucket", None)
        if b is None or not b.matches(pol, cri) or b.flat.device != pol[0].device:
            b = self._bucket = GradBucket(pol, cri)
        b.allreduce(n_valid, n_crit)

    def _mean_grad_norm(self, grad_norm, dev):
        """Mean over the epoch's optimiser steps of the norm the reference records AFTER the clip (:256):
        |g| * min(1, max / (|g| + 1e-6)); optim.Adam hands over |g|^2 per step on the device, the rest is done once here."""
        if not grad_norm:
            return 0.0
        if not torch.is_tensor(grad_norm[0]):
            return float(np.mean(grad_norm))
        pre = torch.stack(grad_norm).to(torch.float32).sqrt()
        mx = float("inf") if self._clip_grad_norm is None else float(self._clip_grad_norm)
        return float((pre * torch.clamp(mx / (pre + 1e-6), max=1.0)).mean())

    # ------------------------------------------------------------------------------------------
    # train_once (:175-388)
    # ------------------------------------------------------------------------------------------
    def train_once(self, runner=None, itr=None, paths=None):
        if runner is not None:
            itr, paths = runner.step_itr, runner.step_path
        t_start = time.time()
        obs, avail, actions, rewards, valids, baselines, returns, dist_adjs, channels = self.process_samples(itr, paths)
        P, T = rewards.shape
        distributed = _dist_ready()
        advantages = self._advantages(rewards, baselines, valids)
        # The reference evaluates the two policies over the full batch eight times per epoch (old log-likelihood twice,
        # loss before / after, KL + entropy before / after with both nets each).  Only three distinct (weights, batch)
        # pairs are involved - last epoch's pre-update weights, this epoch's pre-update weights (which the old policy
        # becomes at :204) and the post-update weights - so three forwards are run and their outputs shared.
        shared = avail is None and all(hasattr(p, "evaluate_nograd") for p in (self.policy, self._old_policy))
        if shared:
            with torch.no_grad():
                _, p_old0 = self._old_policy.evaluate_nograd(obs, dist_adjs, channels)
                lg_new, p_new = self.policy.evaluate_nograd(obs, dist_adjs, channels)
                old_ll0 = self._ll_from_probs(p_old0, actions)
                loss_before = float(self._compute_loss(itr, obs, avail, actions, rewards, valids, baselines, dist_adjs,
                                                       channels, advantages, old_ll0, logits=lg_new))
                kl_before, _ = self._kl_entropy(p_old0, p_new)
                del p_old0, lg_new
            self._old_policy.load_state_dict(self.policy.state_dict())              # :204
            with torch.no_grad():
                old_ll = self._ll_from_probs(p_new, actions)                         # the old policy now IS the policy
        else:
            with torch.no_grad():
                old_ll0 = self._old_log_likelihood(obs, actions, dist_adjs, channels)
                loss_before = float(self._compute_loss(itr, obs, avail, actions, rewards, valids, baselines, dist_adjs,
                                                       channels, advantages, old_ll0))
                kl_before, _ = self._diagnostics(obs, actions, valids, dist_adjs, channels)
            self._old_policy.load_state_dict(self.policy.state_dict())              # :204
            with torch.no_grad():
                old_ll = self._old_log_likelihood(obs, actions, dist_adjs, channels)

        step_size = int(np.ceil(P / self._optimization_n_minibatches))
        if distributed and P < 2 * self._optimization_n_minibatches:
            raise RuntimeError("distributed update needs >= 2 paths per minibatch on every rank (equal optimiser-step counts)")
        shuffled_ids = np.random.permutation(P)                                      # :209
        grad_norm = []
        sl = lambda x, ids: None if x is None else x[ids]                            # noqa: E731
        t_opt = time.time()
        # the permutation is drawn once (:209), so the minibatches are the same path subsets in every mini-epoch:
        # gather them once instead of 10 times (each gather copies ~1/3 of the padded batch)
        minibatches = []
        for start in range(0, P, step_size):
            ids = torch.as_tensor(shuffled_ids[start:min(start + step_size, P)], device=obs.device)
            minibatches.append((obs[ids], actions[ids], rewards[ids], valids[ids], baselines[ids], sl(dist_adjs, ids),
                                sl(channels, ids), advantages[ids], old_ll[ids], returns[ids]))
        # The critic has its own trunk (a-17): its forward / backward (/ optimiser step) share nothing with the policy's
        # but the minibatch, so they run on a second HIP stream and fill the gaps of the policy's launch chain.
        two_streams = obs.is_cuda and os.environ.get("COMMARL_CRITIC_STREAM", "1") != "0"
        if two_streams and (getattr(self, "_side_stream", None) is None or self._side_stream.device != obs.device):
            self._side_stream = torch.cuda.Stream(device=obs.device)
        n_crits = [torch.tensor(float(mb[0].shape[0] * T), device=obs.device) for mb in minibatches]

        def one_step(o, a, r, v, bl, da, ch, adv_mb, oll_mb, ret_mb, n_crit):
            """One optimiser step of both nets on one minibatch (:211-268) -> the squared pre-clip gradient norm (device scalar,
            optim.Adam) or the post-clip norm (host float, any other optimiser)."""
            main = torch.cuda.current_stream(obs.device)
            side = self._side_stream if two_streams else main
            self._baseline_optimizer.zero_grad()
            self._optimizer.zero_grad()
            side.wait_stream(main)                                               # fork: both nets see the minibatch, nothing else

            def critic():                                                        # Gaussian NLL, mean over padded steps (comm_base_critic.py:88-89)
                with torch.cuda.stream(side):
                    bl_loss = self._baseline_loss(o, ret_mb, da, ch)
                    (bl_loss * n_crit if distributed else bl_loss).backward()
                    if not distributed:
                        self._baseline_optimizer.step()
            # Issue order: the policy's chain is the longer one, and a captured graph submits its nodes in capture order - with the
            # critic first the policy's first kernel of a replayed step started ~110 us late (kernel trace of the reference-batch
            # step).  The distributed step keeps the critic first: its gradients must be there when the bucket is reduced.
            if distributed:
                critic()
            loss_sum, n_valid = self._compute_loss(itr, o, None, a, r, v, bl, da, ch, adv_mb, oll_mb, reduce=False)
            if distributed:
                loss_sum.backward()
                main.wait_stream(side)
                self._allreduce_grads(n_valid, n_crit)
            else:
                (loss_sum / n_valid).backward()
            if hasattr(self._optimizer, "_norm"):                               # optim.Adam: clip + update in two launches
                mx = float("inf") if self._clip_grad_norm is None else float(self._clip_grad_norm)
                self._optimizer.step(max_norm=mx, return_norm=False)             # policy only (:253-255)
                gn = self._optimizer.norm_sq                                     # a view of the optimiser's workspace
            else:
                if self._clip_grad_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self._clip_grad_norm)
                gn = self.policy.grad_norm()
                self._optimizer.step()                                           # _optimize (:606-610)
            if distributed:
                self._baseline_optimizer.step()
            else:
                critic()
            main.wait_stream(side)
            return gn

        # The permutation is drawn once per epoch, so mini-epochs 1.. repeat mini-epoch 0's launches on the same buffers with
        # other weights: for the small (launch-bound) batches of the reference's own configuration each minibatch's step is
        # captured into a hipGraph in mini-epoch 1 and replayed from then on (_UpdateGraphs).
        graphs = _UpdateGraphs.maybe(self, minibatches, T, distributed)
        try:
            for mini_epoch in range(self._optimization_mini_epochs):
                for i, mb in enumerate(minibatches):
                    if graphs is not None and mini_epoch >= graphs.first_epoch:
                        gn = graphs.step(i, tuple(mb) + (n_crits[i],), one_step)
                    else:
                        gn = one_step(*mb, n_crits[i])
                        self._eager_stepped = True
                    grad_norm.append(gn.clone() if torch.is_tensor(gn) else gn)
        finally:
            if graphs is not None:                           # the optimisers leave the device-step mode whatever happened
                graphs.close()
        torch.cuda.synchronize(obs.device)
        epoch_time = time.time() - t_opt
        self.policy.sync_weights()

        with torch.no_grad():
            if shared:
                lg_after, p_after = self.policy.evaluate_nograd(obs, dist_adjs, channels)
                loss_after = float(self._compute_loss(itr, obs, avail, actions, rewards, valids, baselines, dist_adjs,
                                                      channels, advantages, old_ll, logits=lg_after))
                kl, entropy = self._kl_entropy(p_new, p_after)
                del lg_after, p_after, p_new
            else:
                loss_after = float(self._compute_loss(itr, obs, avail, actions, rewards, valids, baselines, dist_adjs,
                                                      channels, advantages, old_ll))
                kl, entropy = self._diagnostics(obs, actions, valids, dist_adjs, channels)
        perf = self._log_performance(itr, paths, returns, valids)
        avg_return = perf["AverageReturn"]
        self.stats = dict(perf, LossBefore=loss_before, LossAfter=loss_after,
                          dLoss=loss_before - loss_after, KLBefore=kl_before, KL=kl, Entropy=entropy,
                          GradNorm=self._mean_grad_norm(grad_norm, obs.device), EpochTime=epoch_time,
                          TrainOnceTime=time.time() - t_start, MaxPathLength=T,
                          EnvSteps=int(valids.sum().item()),
                          GPUMemoryMax=torch.cuda.max_memory_allocated(self._dev()) / 1024 ** 3)     # :372-383 (GiB)
        for k, v in self.stats.items():
            tabular.record(k, v)
        return avg_return

    def train(self, runner):
        """MABatchPolopt.train (com_marl/np/algos/ma_batch_polopt.py:72-120): sample -> train_once per epoch."""
        last_return = None
        for _epoch in runner.step_epochs():
            if getattr(runner, "flag", [0])[0]:
                break
            for _ in range(self.n_samples):
                runner.step_path = runner.obtain_samples(runner.step_itr)
                last_return = self.train_once(runner)
                runner.step_itr += 1
        return last_return
