"""Batched Predator-Prey / Coverage envs on one MI355X, behind the reference's env interface.

``GridEnvBatch`` owns B independent envs whose SoA state lives in HBM and is advanced by the
HIP step kernel (csrc/cm_env.hip) through the C ABI.  ``PredatorPreyWrapper`` /
``CoverageWrapper`` keep the constructor and attribute surface of the reference wrappers
(envs/predatorprey_wrapper.py:23-73, envs/coverage_wrapper.py:16-57) so that
``runner_pp_commDP.py`` / ``runner_co_commDP.py`` construct them unchanged; the only new
knobs are the keyword-only ``n_envs``, ``device``, ``seed`` (default: params['seed']).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def _round_int(v):
    return int(v) if float(v) == int(v) else v


class _Discrete:
    def __init__(self, n):
        self.n = n
        self.flat_dim = n


class _Box:
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, np.float32), np.asarray(high, np.float32)
        self.shape = self.low.shape
        self.flat_dim = int(np.prod(self.shape))


class EnvSpec:
    """What the nets read from garage's EnvSpec (comm_base_net.py:36-41)."""

    def __init__(self, observation_space, action_space):
        self.observation_space, self.action_space = observation_space, action_space


def make_cfg(scenario, params, n_envs, *, seed=1, env_id_offset=0, rng_mode=L.RNG_PHILOX, max_steps=None,
             max_path_length=None, channel=None):
    """params dict -> cm_env_cfg, following predator_prey.py:52-82, coverage.py:40-96 and
    env_communication.py:10-75 (signs: costs are stored as -abs(x))."""
    p = params
    c = L.EnvCfg()
    pp = scenario == "pp"
    c.scenario = L.CM_PP if pp else L.CM_CO
    c.n_envs, c.n_agents = int(n_envs), int(p["n_agents"])
    c.n_preys = int(p.get("n_preys", 0)) if pp else 0
    c.grid, c.rsen, c.load = int(p["grid_size"]), int(p["Rsen"]), int(p.get("load", 2))
    if pp:
        c.max_steps = int(max_steps if max_steps is not None else p["max_env_steps"])          # :59
    else:
        c.max_steps = int(max_steps if max_steps is not None else 400)                         # coverage.py:39,47
    c.max_path_length = int(max_path_length if max_path_length is not None else p.get("max_env_steps", c.max_steps))
    c.n_hops = int(p["n_gcn_layers"])
    pref = "tr" if p.get("mode", "train") in ("train", "restore") else "te"                    # env_communication.py:25-29
    pl = p.get(f"{pref}pl")
    if pl is None:
        raise ValueError("Loss probability is not applied (params['trpl'/'tepl'])")
    if channel is None:                                                                         # :35-43
        channel = "FC" if pl == 0 else ("FL" if pl == 1 else "IID")
        if not (0 <= pl <= 1):
            raise ValueError(f"invalid Ploss value: pl={pl}")
    c.channel = L.CHANNELS[channel]
    c.ploss, c.pgb, c.pbg = float(pl), float(p.get("Pgb", 0.0196)), float(p.get("Pbg", 0.282))
    # GE variants (env_communication.py:21,54-60,106-157): GE_INIT 1 good / 0 bad / else random; loss_apply 0 = one
    # state transition per env step shared by all hops, 1 (default) = one per GCN hop
    gi = p.get("GE_INIT", 1)
    la = p.get("loss_apply", 1)
    c.ge_flags = (0 if (la is None or la) else 1) | ((0 if gi in (1, None) else (1 if gi == 0 else 2)) << 1)
    c.rcom = int(p.get(f"{pref}Rcom", 9))
    c.obst_hard = int(p.get("obstComplex", "Easy") == "Hard")
    c.add_clock = int(p.get("add_clock") or 0)
    c.rng_mode, c.env_id_offset, c.seed = rng_mode, int(env_id_offset), int(seed)
    c.capture_reward = abs(p.get("capture_reward", 10 if pp else 2))
    c.step_cost = -abs(p.get("step_cost", 0.1 if pp else 0.0))
    c.move_cost = -abs(p.get("rm", 0))
    c.penalty = -abs(p.get("penalty", 0 if pp else 1))
    c.lazy_penalty = -abs(p.get("lazy_penalty", 1))
    c.revisit_penalty = -abs(p.get("revisit_penalty", 0.5))
    c.final_reward = 100.0                                                                      # coverage.py:92
    return c, channel, pl


class GridEnvBatch:
    """B envs stepped by one kernel launch.  All tensors are torch CUDA tensors."""

    def __init__(self, scenario, params, n_envs=1, *, device="cuda:0", seed=None, env_id_offset=0, rng_mode="philox",
                 max_steps=None, max_path_length=None, channel=None):
        assert scenario in ("pp", "co")
        self.scenario, self.params = scenario, dict(params)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.CommarlError("GridEnvBatch runs on the MI355X only (device must be cuda:k); there is no CPU path")
        seed = int(params.get("seed", 1) if seed is None else seed)
        mode = L.RNG_TAPE if rng_mode == "tape" else L.RNG_PHILOX
        self.cfg, self.channelType, self.pl = make_cfg(scenario, params, n_envs, seed=seed, env_id_offset=env_id_offset,
                                                       rng_mode=mode, max_steps=max_steps,
                                                       max_path_length=max_path_length, channel=channel)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_create(C.byref(self.cfg), C.byref(self._h)), "cm_env_create")
        c = self.cfg
        self.B, self.N, self.M, self.Lh = c.n_envs, c.n_agents, c.n_preys, c.n_hops
        self.S = c.grid if scenario == "pp" else c.grid + 2
        self.d = L.lib().cm_env_obs_dim(self._h)
        self.n_empty_cells = L.lib().cm_env_n_empty_cells(self._h)
        self.adj_const = bool(L.lib().cm_env_adj_is_const(self._h))
        self.ch_const = bool(L.lib().cm_env_channels_are_const(self._h))
        dev, B, N, M = self.device, self.B, self.N, max(self.M, 1)
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        self.obs = torch.zeros(B, N, self.d, dtype=f32, device=dev)
        self.reward = torch.zeros(B, dtype=f32, device=dev)
        self.reward64 = torch.zeros(B, dtype=torch.float64, device=dev)
        self.done = torch.zeros(B, dtype=u8, device=dev)
        self.details = torch.zeros(B, 6, dtype=i32, device=dev)
        self.dist_adj = torch.ones(B, N, N, dtype=f32, device=dev)
        self.channels = torch.ones(B, self.Lh, N, N, dtype=f32, device=dev)
        self.prey_alive = torch.ones(B, M, dtype=u8, device=dev)
        self.success_t = torch.zeros(B, dtype=i32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().cm_env_fill_constants(self._h, L.ptr(self.dist_adj), L.ptr(self.channels),
                                                  L.current_stream()), "cm_env_fill_constants")
        self._keep = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.destroy_env(h)                        # deferred while a hipGraph capture is open
            except Exception:
                pass
            self._h = None

    # ---- raw device-side API -------------------------------------------------------------
    def _out(self, over=None):
        o = dict(obs=self.obs, reward=self.reward, reward_f64=self.reward64, done=self.done, details=self.details,
                 dist_adj=None if self.adj_const else self.dist_adj,
                 channels=None if self.ch_const else self.channels,
                 prey_alive=self.prey_alive if self.M else None, success=self.success_t, path_len=None)
        if over:
            o.update(over)
        return L.StepOut(*[L.ptr(o[k]) for k in ("obs", "reward", "reward_f64", "done", "details", "dist_adj",
                                                 "channels", "prey_alive", "success", "path_len")])

    def _tape(self, tape):
        if not tape:
            return None
        dev = self.device

        def up(a, dt):
            return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev).contiguous()
        prey, spawn = up(tape.get("prey"), torch.uint8), up(tape.get("spawn"), torch.int32)
        iid, ge = up(tape.get("iid_u"), torch.float32), up(tape.get("ge_u"), torch.float32)
        gi = up(tape.get("ge_init_u"), torch.float32)
        self._keep = (prey, spawn, iid, ge, gi)
        return L.RngTape(L.ptr(prey), L.ptr(spawn), 0 if spawn is None else spawn.shape[1], 0, L.ptr(iid), L.ptr(ge),
                         L.ptr(gi))

    def reset_all(self, tape=None, out=None):
        t = self._tape(tape)
        so = self._out(out)
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_reset(self._h, C.byref(t) if t else None, C.byref(so), L.current_stream()),
                    "cm_env_reset")

    def step_device(self, actions, tape=None, out=None):
        """actions: int32 CUDA tensor [B,N].  Asynchronous; results land in self.* (or `out` overrides)."""
        assert actions.dtype == torch.int32 and actions.shape == (self.B, self.N) and actions.is_cuda
        t = self._tape(tape)
        so = self._out(out)
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_step(self._h, L.ptr(actions), C.byref(t) if t else None, C.byref(so),
                                        L.current_stream()), "cm_env_step")

    def check_status(self):
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_status(self._h), "env kernel")

    def get_state(self):
        B, N, M, S = self.B, self.N, max(self.M, 1), self.S
        st = dict(agent_pos=np.zeros((B, N, 2), np.int32), prey_pos=np.zeros((B, M, 2), np.int32),
                  prey_alive=np.zeros((B, M), np.uint8),
                  visited=np.zeros((B, S) if S <= 32 else (B, S, (S + 31) // 32), np.uint32),   # row bitmasks: 2 words from map 40 on
                  step_count=np.zeros(B, np.int32), total_capture=np.zeros(B, np.int32),
                  success=np.zeros(B, np.int32), ge_state=np.zeros((B, N, N), np.uint8),
                  rng_step=np.zeros(B, np.uint32))
        if self.M == 0:
            st["prey_pos"], st["prey_alive"] = st["prey_pos"][:, :0], st["prey_alive"][:, :0]
        cs = L.EnvState(*[st[k].ctypes.data if st[k].size else None for k in (
            "agent_pos", "prey_pos", "prey_alive", "visited", "step_count", "total_capture", "success", "ge_state",
            "rng_step")])
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_get_state(self._h, C.byref(cs)), "cm_env_get_state")
        return st

    # ---- dormant fault / delay helpers of the reference (custom_implement/env_communication.py:270-301) ----
    @property
    def agent_condition(self):
        """[B,N] uint8: PP agent_condition (predator_prey.py:258): 0 = the agent's moves are not applied."""
        out = np.zeros((self.B, self.N), np.uint8)
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_agent_condition(self._h, None, out.ctypes.data), "cm_env_agent_condition")
        return out

    @agent_condition.setter
    def agent_condition(self, cond):
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(cond, np.uint8), (self.B, self.N)))
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_agent_condition(self._h, c.ctypes.data, None), "cm_env_agent_condition")

    def apply_agent_fault(self, kind, p, r=0.0, fault_step=0, tape_u=None):
        """kind 'iid': iid_fault(n_agents, p_fault=p) (:290-292); kind 'GE': GE_fault(condition, p, r) (:294-301, one draw
        per group as written).  tape_u: the uniforms (numpy, [B,N] / [B,2]) in tape mode, else Philox site 9 at fault_step."""
        mode = {"iid": 1, "GE": 2}[kind]
        keep = None if tape_u is None else torch.as_tensor(np.ascontiguousarray(tape_u, np.float32)).to(self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_agent_fault(self._h, mode, float(p), float(r), L.ptr(keep), int(fault_step) & 0xFFFFFFFF,
                                               L.current_stream()), "cm_env_agent_fault")
            torch.cuda.current_stream().synchronize()

    def comm_delays(self, dist_adj, link_loss, delay_th=None, old_delays=None):
        """delays_init(adjacency, link_loss, delay_th) when old_delays is None, else calc_delays(adjacency, link_loss,
        old_delays) (:271-286).  CUDA tensors [B,N,N] / [B,L,N,N] / [B,N,N] int32 -> delays [B,L,N,N] int32."""
        Lh = link_loss.shape[1]
        out = torch.empty(self.B, Lh, self.N, self.N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_comm_delays(self.B, Lh, self.N, L.ptr(None if dist_adj is None else dist_adj.contiguous()),
                                           L.ptr(link_loss.contiguous()),
                                           L.ptr(None if old_delays is None else old_delays.contiguous()),
                                           int(delay_th or 0), int(old_delays is None), L.ptr(out), L.current_stream()),
                    "cm_comm_delays")
        return out

    def set_state(self, **arrays):
        if "agent_cond" in arrays:
            self.agent_condition = arrays.pop("agent_cond")
            if not arrays:
                return
        keep = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        cs = L.EnvState(*[keep[k].ctypes.data if k in keep and keep[k].size else None for k in (
            "agent_pos", "prey_pos", "prey_alive", "visited", "step_count", "total_capture", "success", "ge_state",
            "rng_step")])
        with torch.cuda.device(self.device):
            L.check(L.lib().cm_env_set_state(self._h, C.byref(cs)), "cm_env_set_state")


class _WrapperBase:
    """Attribute surface the sampler / runner touch (SURVEY.md §8b 'Env object')."""
    _scenario = None

    def __init__(self, centralized, other_agent_visible=False, *args, n_envs=1, device="cuda:0", seed=None,
                 env_id_offset=0, rng_mode="philox", max_steps=None, channel=None, **kwargs):
        params = kwargs["params"]
        self.centralized = centralized
        self._agent_visible = other_agent_visible
        self.batch = GridEnvBatch(self._scenario, params, n_envs, device=device, seed=seed, env_id_offset=env_id_offset,
                                  rng_mode=rng_mode, max_steps=max_steps, channel=channel)
        b = self.batch
        self.n_envs, self.n_agents = b.B, b.N
        self.n_preys = b.M
        self.maps = int(params["grid_size"])
        self.Rsen = int(params["Rsen"])
        self.GCNHops = self.L = b.Lh
        self.curriculum_learning = params.get("curriculum_learning")
        self.channelType, self.pl = b.channelType, b.pl
        self.pconn = 1 - self.pl
        rc = b.cfg.rcom
        self.Rcom = 0 if rc + 1 >= self.maps else rc                               # env_communication.py:71-72
        self.Rcom_th = np.float32(math.sqrt(2.0) * self.Rcom)                       # cdist([0,0],[R,R]) (:73-75)
        self.loss_apply = params.get("loss_apply")
        self.mode = params.get("mode")
        self.epoch = None
        self.total_n_epi = self.epoch_n_epi = 0
        self.positions_record = {"agent": [], "target": []}
        self.pickleable = False            # state lives in HBM; snapshots go through get_state()
        self.n_action = 5
        self.action_space = _Discrete(5)
        d = b.d
        low = np.zeros(d, np.float32)
        if self._scenario == "co":
            low[: d - 2 - int(b.cfg.add_clock)] = -1.0                              # coverage.py:117-118
        self.observation_space = _Box(np.tile(low, b.N) if centralized else low,
                                      np.ones(d * b.N if centralized else d, np.float32))
        self.spec = EnvSpec(self.observation_space, self.action_space)
        self.ave_trput = b.n_empty_cells if self._scenario == "co" else 0           # coverage.py:232
        if self._scenario == "pp":
            self.bound_return = self.n_preys * abs(params.get("capture_reward", 10))   # predator_prey.py:72
        else:
            n, ne = b.N, b.n_empty_cells                                            # coverage.py:214-219
            self.bound_return = b.cfg.capture_reward * ne / n - abs(b.cfg.step_cost) * ne / n + b.cfg.final_reward
        self.diameter = b.N if self.Rcom == 0 else 0                                # get_graph :219-223,234
        self._single = (b.B == 1)

    # -- env attributes the sampler reads each step (sampler.py:123-131) --
    @property
    def dist_adj(self):
        a = self.batch.dist_adj.cpu().numpy()
        if self.Rcom == 0:
            a = a.astype(np.float64)                                                # np.ones(...) f64 (:220)
        return a[0] if self._single else a

    @property
    def channels(self):
        c = self.batch.channels.cpu().numpy()
        return c[0] if self._single else c

    @property
    def ave_deg(self):
        if self.Rcom == 0:
            return self.n_agents
        v = self.batch.dist_adj.sum(-1).mean(-1).cpu().numpy()                      # :232
        return v[0] if self._single else v

    @property
    def success(self):
        s = self.batch.success_t.cpu().numpy()
        return int(s[0]) if self._single else s

    @property
    def agent_pos(self):
        p = self.batch.get_state()["agent_pos"]
        if self._single:
            return {i: [int(p[0, i, 0]), int(p[0, i, 1])] for i in range(self.n_agents)}
        return p

    @property
    def agent_condition(self):
        """predator_prey.py:74: ones unless a fault model was applied (GridEnvBatch.apply_agent_fault)."""
        c = self.batch.agent_condition
        return c[0].astype(np.float64) if self.batch.B == 1 else c

    @agent_condition.setter
    def agent_condition(self, cond):
        self.batch.agent_condition = cond

    def get_avail_actions(self):
        """All ones (predatorprey_wrapper.py:46-51)."""
        if not self.centralized:
            return [[1] * 5 for _ in range(self.n_agents)]
        a = np.ones(self.n_agents * 5, dtype=np.int64)
        return a if self._single else np.tile(a, (self.n_envs, 1))

    def seed(self, n):
        return [n, n + 1]

    def close(self):
        pass

    def _obs_np(self):
        o = self.batch.obs.cpu().numpy().astype(np.float64)
        o = o.reshape(self.n_envs, -1) if self.centralized else o
        return o[0] if self._single else o

    def reset(self, epoch=-1):
        """Resets ALL B envs (VecEnvExecutor.reset, vec_env_executor.py:47-54)."""
        self.epoch = epoch
        self.batch.reset_all()
        self.total_n_epi += self.n_envs
        self.epoch_n_epi += self.n_envs
        return self._obs_np()

    def _details(self, det):
        n = float(self.n_agents)
        if self._scenario == "pp":                                                  # predator_prey.py:440-448
            return dict(capture_cnt=int(det[0]), step_cnt=1, move_cnt=det[1] / n, penalty_cnt=int(det[2]),
                        variable=det[4] / n, vars2=0)
        return dict(capture_cnt=det[0] / n, step_cnt=1, move_cnt=det[1] / n, penalty_cnt=det[2] / n,   # coverage.py:308-315
                    variable=det[4] / n, vars2=det[3] / n)

    def step(self, actions):
        """actions: [N] (single env) or [B,N].  Envs that finish are auto-reset and return the
        reset observation, exactly as VecEnvExecutor.step does (vec_env_executor.py:36-43)."""
        a = torch.as_tensor(np.asarray(actions), dtype=torch.int32).reshape(self.n_envs, self.n_agents)
        self.batch.step_device(a.to(self.batch.device))
        self.batch.check_status()
        rew = self.batch.reward64.cpu().numpy()
        done = self.batch.done.cpu().numpy().astype(bool)
        det = self.batch.details.cpu().numpy()
        info = {}
        if self._scenario == "pp":
            info["prey_alive"] = self.batch.prey_alive.cpu().numpy().astype(bool)
        if self._single:
            d = self._details(det[0])
            d["reward"] = float(rew[0])
            if "prey_alive" in info:
                info["prey_alive"] = info["prey_alive"][0]
            return self._obs_np(), (float(rew[0]), d), bool(done[0]), info
        ds = [dict(self._details(det[b]), reward=float(rew[b])) for b in range(self.n_envs)]
        return self._obs_np(), (rew, ds), done, info


class PredatorPreyWrapper(_WrapperBase):
    """envs/predatorprey_wrapper.py:23 - same ctor (centralized, other_agent_visible, params=...)."""
    _scenario = "pp"


class CoverageWrapper(_WrapperBase):
    """envs/coverage_wrapper.py:16."""
    _scenario = "co"
