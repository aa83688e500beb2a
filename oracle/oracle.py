"""Python face of the CPU oracle (TEST INFRASTRUCTURE - see oracle/cm_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
``OracleEnv`` drives the sequential C restatement (libcm_oracle.so, built from
oracle/cm_oracle.c) in RNG-tape mode (pinned by tests/golden) or Philox mode (same logic,
production draw source, used to check the HIP kernels at BASELINE sizes).  The numpy
functions at the bottom restate the PPO maths (advantage normalisation, clipped
surrogate, Gaussian-NLL critic loss, Adam) with reference file:line citations.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libcm_oracle.so")

PP, CO = 0, 1
CH = {"FC": 0, "FL": 1, "IID": 2, "GE": 3}
RNG_PHILOX, RNG_TAPE = 0, 1


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("cm_oracle.c", "cm_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "scenario", "n_envs", "n_agents", "n_preys", "grid", "rsen", "load", "max_steps", "max_path_length",
        "n_hops", "rcom", "channel", "obst_hard", "add_clock", "rng_mode", "env_id_offset")] + [
        ("ploss", C.c_float), ("pgb", C.c_float), ("pbg", C.c_float), ("ge_flags", C.c_int32)] + [
        (n, C.c_double) for n in ("capture_reward", "step_cost", "move_cost", "penalty", "lazy_penalty",
                                  "revisit_penalty", "final_reward")] + [("seed", C.c_uint64)]


class State(C.Structure):
    _fields_ = [("agent_pos", C.c_void_p), ("prey_pos", C.c_void_p), ("prey_alive", C.c_void_p),
                ("visited", C.c_void_p), ("step_count", C.c_void_p), ("total_capture", C.c_void_p),
                ("success", C.c_void_p), ("ge_state", C.c_void_p), ("rng_step", C.c_void_p), ("agent_cond", C.c_void_p)]


class Tape(C.Structure):
    _fields_ = [("prey", C.c_void_p), ("spawn", C.c_void_p), ("spawn_cap", C.c_int32), ("_pad", C.c_int32),
                ("iid_u", C.c_void_p), ("ge_u", C.c_void_p), ("ge_init_u", C.c_void_p)]


class Out(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("details", C.c_void_p),
                ("dist_adj", C.c_void_p), ("channels", C.c_void_p), ("prey_alive_info", C.c_void_p)]


class PolicyW(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "n_agents", "n_hops", "enc_hidden", "emb", "h1", "h2", "h3",
                                         "n_act", "no_residual")] + [
        (n, C.c_void_p) for n in ("enc_w1", "enc_b1", "enc_w2", "enc_b2", "attn_w", "gcn_w", "gcn_b", "hd_w1",
                                  "hd_b1", "hd_w2", "hd_b2", "hd_w3", "hd_b3", "hd_w4", "hd_b4")]


class CriticW(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "n_agents", "n_hops", "enc_hidden", "emb", "dec_hidden", "no_residual",
                                         "_pad")] + [
        (n, C.c_void_p) for n in ("enc_w1", "enc_b1", "enc_w2", "enc_b2", "attn_w", "gcn_w", "gcn_b", "dec_w1",
                                  "dec_b1", "dec_w2", "dec_b2")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.cmo_obs_dim.argtypes = [C.POINTER(Cfg)]
        _lib.cmo_n_empty_cells.argtypes = [C.POINTER(Cfg)]
        _lib.cmo_reset.argtypes = [C.POINTER(Cfg), C.POINTER(State), C.POINTER(Tape), C.POINTER(Out)]
        _lib.cmo_step.argtypes = [C.POINTER(Cfg), C.POINTER(State), C.c_void_p, C.POINTER(Tape), C.POINTER(Out),
                                  C.c_int]
        _lib.cmo_ge_transition.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                           C.c_void_p]
        _lib.cmo_policy_forward.argtypes = [C.POINTER(PolicyW), C.c_int] + [C.c_void_p] * 7 + [C.c_int]
        _lib.cmo_sample_actions.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int,
                                            C.c_uint32, C.c_void_p]
        _lib.cmo_critic_forward.argtypes = [C.POINTER(CriticW), C.c_int] + [C.c_void_p] * 4 + [C.c_int]
        _lib.cmo_discount_cumsum.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p]
        _lib.cmo_gae.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib.cmo_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cmo_mlp_forward.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        _lib.cmo_group_softmax.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cmo_mlp_forward.restype = _lib.cmo_group_softmax.restype = None
        for f in ("cmo_policy_forward", "cmo_sample_actions", "cmo_critic_forward", "cmo_discount_cumsum",
                  "cmo_gae", "cmo_philox4x32_10", "cmo_ge_transition"):
            getattr(_lib, f).restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


def make_cfg(scenario, n_envs, n_agents, grid, rsen, n_preys=0, load=2, max_steps=200, max_path_length=None,
             n_hops=2, rcom=9, channel="FC", ploss=0.0, pgb=0.0196, pbg=0.282, obst="Easy", add_clock=0,
             capture_reward=None, step_cost=None, rm=0.0, penalty=None, lazy_penalty=1.0, revisit_penalty=0.5,
             final_reward=100.0, seed=1, env_id_offset=0, rng_mode=RNG_PHILOX, ge_init=1, loss_apply=1):
    """Signs follow the env constructors: costs are stored as -abs(x) (predator_prey.py:65-68,
    coverage.py:86-92).  Defaults are exp_runners/*/utils_{pp,co}.py."""
    sc = PP if scenario in ("pp", PP) else CO
    if capture_reward is None:
        capture_reward = 10.0 if sc == PP else 2.0
    if step_cost is None:
        step_cost = 0.1 if sc == PP else 0.0
    if penalty is None:
        penalty = 0.0 if sc == PP else 1.0
    c = Cfg()
    c.scenario, c.n_envs, c.n_agents, c.n_preys, c.grid, c.rsen = sc, n_envs, n_agents, n_preys, grid, rsen
    c.load, c.max_steps, c.max_path_length = load, max_steps, max_path_length or max_steps
    c.n_hops, c.rcom, c.channel = n_hops, rcom, CH[channel]
    c.obst_hard, c.add_clock, c.rng_mode, c.env_id_offset = int(obst == "Hard"), add_clock, rng_mode, env_id_offset
    c.ploss, c.pgb, c.pbg = ploss, pgb, pbg
    # GE_INIT 1 good / 0 bad / anything else random (env_communication.py:54-60); loss_apply 0 = per env step (:21)
    c.ge_flags = (0 if loss_apply else 1) | ((0 if ge_init == 1 else (1 if ge_init == 0 else 2)) << 1)
    c.capture_reward, c.step_cost, c.move_cost = abs(capture_reward), -abs(step_cost), -abs(rm)
    c.penalty, c.lazy_penalty, c.revisit_penalty = -abs(penalty), -abs(lazy_penalty), -abs(revisit_penalty)
    c.final_reward, c.seed = final_reward, seed
    return c


def cfg_from_json(js, n_envs, rng_mode=RNG_TAPE, seed=1):
    j = json.loads(js) if isinstance(js, str) else js
    return make_cfg(j["scenario"], n_envs, j["n_agents"], j["grid"], j["rsen"], n_preys=j["n_preys"], load=j["load"],
                    max_steps=j["max_steps"], max_path_length=j["max_path_length"], n_hops=j["n_hops"],
                    rcom=j["rcom"], channel=j["channel"], ploss=j["ploss"], pgb=j["pgb"], pbg=j["pbg"],
                    obst=j["obst"], add_clock=j["add_clock"], capture_reward=j["capture_reward"],
                    step_cost=j["step_cost"], rm=j["rm"], penalty=j["penalty"], lazy_penalty=j["lazy_penalty"],
                    revisit_penalty=j["revisit_penalty"], seed=seed, rng_mode=rng_mode,
                    ge_init=j.get("ge_init", 1), loss_apply=j.get("loss_apply", 1))


class OracleEnv:
    """B independent envs stepped by the sequential C restatement."""

    def __init__(self, cfg):
        self.cfg = cfg
        L = lib()
        B, N, M = cfg.n_envs, cfg.n_agents, max(cfg.n_preys, 1)
        self.B, self.N, self.M, self.L = B, N, cfg.n_preys, cfg.n_hops
        self.S = cfg.grid if cfg.scenario == PP else cfg.grid + 2
        self.d = L.cmo_obs_dim(C.byref(cfg))
        self.n_empty_cells = L.cmo_n_empty_cells(C.byref(cfg))
        self.agent_pos = np.zeros((B, N, 2), np.int32)
        self.prey_pos = np.zeros((B, M, 2), np.int32)
        self.prey_alive = np.zeros((B, M), np.uint8)
        self.VW = (self.S + 31) // 32                       # words per visited row (2 from map 40 on)
        self.visited = np.zeros((B, self.S, self.VW), np.uint32) if self.VW > 1 else np.zeros((B, self.S), np.uint32)
        self.step_count = np.zeros(B, np.int32)
        self.total_capture = np.zeros(B, np.int32)
        self.success = np.zeros(B, np.int32)
        self.ge_state = np.ones((B, N, N), np.uint8)
        self.rng_step = np.zeros(B, np.uint32)
        self.agent_cond = np.ones((B, N), np.uint8)         # PP agent_condition (predator_prey.py:74,152,258)
        self.obs = np.zeros((B, N, self.d), np.float32)
        self.reward = np.zeros(B, np.float64)
        self.done = np.zeros(B, np.uint8)
        self.details = np.zeros((B, 6), np.int32)
        self.dist_adj = np.zeros((B, N, N), np.float32)
        self.channels = np.zeros((B, self.L, N, N), np.float32)
        self.prey_alive_info = np.zeros((B, M), np.uint8)
        self._st = State(*[_p(a) for a in (self.agent_pos, self.prey_pos, self.prey_alive, self.visited,
                                           self.step_count, self.total_capture, self.success, self.ge_state,
                                           self.rng_step, self.agent_cond)])
        self._out = Out(*[_p(a) for a in (self.obs, self.reward, self.done, self.details, self.dist_adj,
                                          self.channels, self.prey_alive_info)])

    def _tape(self, prey=None, spawn=None, iid_u=None, ge_u=None, ge_init_u=None):
        self._keep = [np.ascontiguousarray(a) if a is not None else None for a in (prey, spawn, iid_u, ge_u, ge_init_u)]
        prey, spawn, iid_u, ge_u, ge_init_u = self._keep
        return Tape(_p(prey), _p(spawn), 0 if spawn is None else spawn.shape[1], 0, _p(iid_u), _p(ge_u), _p(ge_init_u))

    def reset(self, **tape):
        t = self._tape(**tape)
        rc = lib().cmo_reset(C.byref(self.cfg), C.byref(self._st), C.byref(t), C.byref(self._out))
        if rc:
            raise RuntimeError(f"cmo_reset failed: {rc}")
        return self.obs

    def step(self, actions, n_threads=1, **tape):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.B, self.N)
        t = self._tape(**tape)
        rc = lib().cmo_step(C.byref(self.cfg), C.byref(self._st), _p(a), C.byref(t), C.byref(self._out), n_threads)
        if rc:
            raise RuntimeError(f"cmo_step failed: {rc}")
        return self.obs, self.reward, self.done

    def load_state(self, **arrays):
        """Overwrite parts of the state in place (agent_pos, prey_pos, prey_alive, step_count, agent_cond ...)."""
        for k, v in arrays.items():
            getattr(self, k)[...] = v

    def visited_dense(self):
        cols = np.arange(self.S, dtype=np.uint32)
        words = self.visited.reshape(self.B, self.S, self.VW)[:, :, cols >> 5]            # [B, S, S]: the word of every column
        return ((words >> (cols & 31)[None, None, :]) & 1).astype(np.uint8)


# --------------------------------------------------------------------------------------
# dormant fault / delay helpers of custom_implement/env_communication.py:270-301: never called by the
# reference's own code; restated literally and pinned by direct-call recordings (tests/golden/faults_direct.npz)
# --------------------------------------------------------------------------------------
def iid_fault(u, p_fault):
    """env_communication.py:290-292: np.random.choice([0, 1], size=n, p=[p_fault, 1 - p_fault]) on the uniforms `u` the
    legacy generator draws for it (choice = cdf.searchsorted(u, side='right')): 0 (faulty) iff u < p_fault."""
    cdf = np.cumsum(np.array([p_fault, 1.0 - p_fault], np.float64))
    cdf /= cdf[-1]
    return np.array([0, 1])[np.searchsorted(cdf, np.asarray(u, np.float64), side="right")].astype(np.int64)


def ge_fault(cond, u_good, u_bad, p, r):
    """env_communication.py:294-301 AS WRITTEN: `size=(len(G_condition))` is the length of np.where's TUPLE (= 1), so ONE
    draw decides all currently-good agents (stay good with 1 - p) and ONE all currently-bad ones (recover with r)."""
    cond = np.asarray(cond)
    new = np.zeros_like(cond)
    g = 1 if u_good < 1.0 - p else 0              # np.random.choice([1, 0], size=1, p=[1 - p, p])
    b = 1 if u_bad < r else 0                     # np.random.choice([1, 0], size=1, p=[r, 1 - r])
    new[cond == 1] = g
    new[cond == 0] = b
    return new


def delays_init(adjacency, link_loss, delay_th):
    """env_communication.py:271-279.  adjacency [N,N], link_loss [L,N,N] -> [L,N,N]."""
    delays = [np.where(adjacency == 0, delay_th, 1)]
    for i, link in enumerate(link_loss[1:]):
        delays.append(np.where(link == 0, delays[i] + 1, 1))
    return np.array(delays)


def calc_delays(adjacency, link_loss, old_delays):
    """env_communication.py:281-286 with `old_delays` an [N,N] matrix (e.g. the last hop of the previous call): the
    function has no caller in the reference; with the [L,N,N] array delays_init returns it would broadcast to
    [L,L,N,N], so the per-matrix reading is the only shape-consistent one."""
    loss = adjacency * link_loss
    delays = [np.where(loss[0] == 0, old_delays + 1, 1)]
    for i, l in enumerate(loss[1:]):
        delays.append(np.where(l == 0, delays[i] + 1, 1))
    return np.array(delays)


# --------------------------------------------------------------------------------------
# policy / critic forward through the C restatement
# --------------------------------------------------------------------------------------
_POL_KEYS = dict(
    enc_w1="encoder._layers.0.linear.weight", enc_b1="encoder._layers.0.linear.bias",
    enc_w2="encoder._output_layers.0.linear.weight", enc_b2="encoder._output_layers.0.linear.bias",
    attn_w="attention_layer.linear_in.weight",
    hd_w1="categorical_output_layer._layers.0.linear.weight", hd_b1="categorical_output_layer._layers.0.linear.bias",
    hd_w2="categorical_output_layer._layers.1.linear.weight", hd_b2="categorical_output_layer._layers.1.linear.bias",
    hd_w3="categorical_output_layer._layers.2.linear.weight", hd_b3="categorical_output_layer._layers.2.linear.bias",
    hd_w4="categorical_output_layer._output_layers.0.linear.weight",
    hd_b4="categorical_output_layer._output_layers.0.linear.bias")
_CRIT_KEYS = dict(
    enc_w1="encoder._layers.0.linear.weight", enc_b1="encoder._layers.0.linear.bias",
    enc_w2="encoder._output_layers.0.linear.weight", enc_b2="encoder._output_layers.0.linear.bias",
    attn_w="attention_layer.linear_in.weight",
    dec_w1="baseline_aggregator._mean_module._layers.0.linear.weight",
    dec_b1="baseline_aggregator._mean_module._layers.0.linear.bias",
    dec_w2="baseline_aggregator._mean_module._output_layers.0.linear.weight",
    dec_b2="baseline_aggregator._mean_module._output_layers.0.linear.bias")


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _gcn_stack(sd, n_hops):
    if n_hops == 0:
        return np.zeros((1, 1, 1), np.float32), np.zeros((1, 1), np.float32)
    w = np.stack([_f32(sd[f"gcn_layers.{l}.weight"]) for l in range(n_hops)])
    b = np.stack([_f32(sd[f"gcn_layers.{l}.bias"]) for l in range(n_hops)])
    return np.ascontiguousarray(w), np.ascontiguousarray(b)


def policy_forward(sd, obs, avail, dist_adj, channels, n_agents, n_threads=1, want_emb=False, residual=True):
    """sd: reference-named state_dict (numpy). obs [S,N*d] or [S,N,d]; returns probs [S,N,A], attn [S,N,N]."""
    S = obs.shape[0]
    N = n_agents
    obs = _f32(obs).reshape(S, N, -1)
    d = obs.shape[2]
    n_hops = channels.shape[-3]
    arrs = {k: _f32(sd[v]) for k, v in _POL_KEYS.items() if v in sd}
    if "attn_w" not in arrs:                 # attention_type='dot' (attention_module.py:38-41): no linear_in, Q = E
        arrs["attn_w"] = np.eye(arrs["enc_w2"].shape[0], dtype=np.float32)
    arrs["gcn_w"], arrs["gcn_b"] = _gcn_stack(sd, n_hops)
    A = arrs["hd_w4"].shape[0]
    w = PolicyW()
    w.d, w.n_agents, w.n_hops, w.enc_hidden, w.emb = d, N, n_hops, arrs["enc_w1"].shape[0], arrs["enc_w2"].shape[0]
    w.h1, w.h2, w.h3, w.n_act = arrs["hd_w1"].shape[0], arrs["hd_w2"].shape[0], arrs["hd_w3"].shape[0], A
    w.no_residual = 0 if residual else 1
    for k, a in arrs.items():
        setattr(w, k, _p(a))
    avail = _f32(avail).reshape(S, N, A)
    adj = _f32(dist_adj).reshape(S, N, N)
    ch = _f32(channels).reshape(S, n_hops, N, N)
    probs = np.zeros((S, N, A), np.float32)
    attn = np.zeros((S, N, N), np.float32)
    emb = np.zeros((S, n_hops + 1, N, w.emb), np.float32) if want_emb else None
    lib().cmo_policy_forward(C.byref(w), S, _p(obs), _p(avail), _p(adj), _p(ch), _p(probs), _p(attn), _p(emb),
                             n_threads)
    return (probs, attn, emb) if want_emb else (probs, attn)


def critic_forward(sd, obs, dist_adj, channels, n_agents, n_threads=1, residual=True):
    S = obs.shape[0]
    N = n_agents
    obs = _f32(obs).reshape(S, N, -1)
    n_hops = channels.shape[-3]
    w1 = sd["baseline_aggregator._mean_module._layers.0.linear.weight"]
    if w1.shape[1] != sd["encoder._output_layers.0.linear.weight"].shape[0]:
        return critic_forward_direct(sd, obs, dist_adj, channels, n_agents, n_threads, residual)
    arrs = {k: _f32(sd[v]) for k, v in _CRIT_KEYS.items() if v in sd}
    if "attn_w" not in arrs:
        arrs["attn_w"] = np.eye(arrs["enc_w2"].shape[0], dtype=np.float32)
    arrs["gcn_w"], arrs["gcn_b"] = _gcn_stack(sd, n_hops)
    w = CriticW()
    w.d, w.n_agents, w.n_hops = obs.shape[2], N, n_hops
    w.enc_hidden, w.emb, w.dec_hidden = arrs["enc_w1"].shape[0], arrs["enc_w2"].shape[0], arrs["dec_w1"].shape[0]
    w.no_residual = 0 if residual else 1
    for k, a in arrs.items():
        setattr(w, k, _p(a))
    adj = _f32(dist_adj).reshape(S, N, N)
    ch = _f32(channels).reshape(S, n_hops, N, N)
    values = np.zeros(S, np.float32)
    lib().cmo_critic_forward(C.byref(w), S, _p(obs), _p(adj), _p(ch), _p(values), n_threads)
    return values


def critic_forward_direct(sd, obs, dist_adj, channels, n_agents, n_threads=1, residual=True):
    """aggregator_type='direct' (comm_base_critic.py:113-116): the trunk's x = E + H_L of all agents concatenated, one MLP
    [N * 64] -> hidden -> 1 on top.  Trunk through the C restatement (cmo_policy_forward's embeddings, its head unused)."""
    S, N = obs.shape[0], n_agents
    fake = {k: v for k, v in sd.items() if not k.startswith("baseline_aggregator")}
    for i, (o, k) in enumerate(((128, 64), (64, 128), (32, 64))):
        fake[f"categorical_output_layer._layers.{i}.linear.weight"] = np.zeros((o, k), np.float32)
        fake[f"categorical_output_layer._layers.{i}.linear.bias"] = np.zeros(o, np.float32)
    fake["categorical_output_layer._output_layers.0.linear.weight"] = np.zeros((5, 32), np.float32)
    fake["categorical_output_layer._output_layers.0.linear.bias"] = np.zeros(5, np.float32)
    _, _, emb = policy_forward(fake, obs, np.ones((S, N, 5), np.float32), dist_adj, channels, N, n_threads, want_emb=True,
                               residual=residual)
    x = emb[:, 0] + emb[:, -1] if residual else emb[:, -1]              # [S, N, 64]
    pre = "baseline_aggregator._mean_module."
    return mlp_forward(_mlp_layers(sd, pre, _n_hidden(sd, pre)), x.reshape(S, -1))[:, 0]


def mlp_forward(layers, x):
    """layers: list of (weight [out,in], bias [out], tanh?) in forward order; x [rows, in] -> [rows, out_last]."""
    x = _f32(x)
    rows, in_dim = x.shape
    ws = [_f32(w) for w, _, _ in layers]
    bs = [_f32(b) for _, b, _ in layers]
    out_dim = np.asarray([w.shape[0] for w in ws], np.int32)
    mask = sum(int(bool(t)) << i for i, (_, _, t) in enumerate(layers))
    wp = (C.c_void_p * len(ws))(*[_p(w) for w in ws])
    bp = (C.c_void_p * len(bs))(*[_p(b) for b in bs])
    y = np.zeros((rows, int(out_dim[-1])), np.float32)
    lib().cmo_mlp_forward(rows, in_dim, len(ws), _p(out_dim), mask, wp, bp, _p(x), _p(y))
    return y


def _mlp_layers(sd, prefix, n_hidden, out_tanh=False):
    ls = [(sd[f"{prefix}_layers.{i}.linear.weight"], sd[f"{prefix}_layers.{i}.linear.bias"], True)
          for i in range(n_hidden)]
    return ls + [(sd[f"{prefix}_output_layers.0.linear.weight"], sd[f"{prefix}_output_layers.0.linear.bias"], out_tanh)]


def _n_hidden(sd, prefix):
    n = 0
    while f"{prefix}_layers.{n}.linear.weight" in sd:
        n += 1
    return n


def group_softmax(logits, avail):
    lg = _f32(logits)
    A = lg.shape[-1]
    av = _f32(np.broadcast_to(np.asarray(avail, np.float32).reshape(-1, A), (lg.size // A, A)))
    p = np.zeros_like(lg)
    lib().cmo_group_softmax(lg.size // A, A, _p(lg), _p(av), _p(p))
    return p


def dec_policy_forward(sd, obs, avail, n_agents):
    """DecCategoricalMLPPolicy.forward (dec_categorical_mlp_policy.py:106-122): obs [S,N*d] -> probs [S,N,A]."""
    S = obs.shape[0]
    x = _f32(obs).reshape(S * n_agents, -1)
    layers = _mlp_layers(sd, "encoder.", _n_hidden(sd, "encoder."), out_tanh=True) + _mlp_layers(sd, "", _n_hidden(sd, ""))
    lg = mlp_forward(layers, x)
    return group_softmax(lg, avail).reshape(S, n_agents, -1)


def cent_policy_forward(sd, obs, avail, n_agents):
    """CentralizedCategoricalMLPPolicy.forward (centralized_categorical_mlp_policy.py:61-97)."""
    S = obs.shape[0]
    lg = mlp_forward(_mlp_layers(sd, "", _n_hidden(sd, "")), _f32(obs).reshape(S, -1))
    return group_softmax(lg.reshape(S * n_agents, -1), avail).reshape(S, n_agents, -1)


def gaussian_baseline_forward(sd, obs):
    """GaussianMLPBaseline.forward (gaussian_mlp_baseline.py:100-115): obs [R, N*d] -> values [R]."""
    pre = "module._mean_module."
    return mlp_forward(_mlp_layers(sd, pre, _n_hidden(sd, pre)), _f32(obs))[:, 0]


def sample_actions(probs, seed, env_id_offset, policy_step):
    S, N, A = probs.shape
    probs = _f32(probs)
    out = np.zeros((S, N), np.int32)
    lib().cmo_sample_actions(S, N, A, _p(probs), seed, env_id_offset, policy_step, _p(out))
    return out


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().cmo_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def ge_transition(s, u_gb, u_bg, pgb, pbg):
    s = np.ascontiguousarray(s, np.uint8)
    n = s.shape[-1]
    out = np.zeros_like(s)
    lib().cmo_ge_transition(n, _p(s), _p(_f32(u_gb)), _p(_f32(u_bg)), pgb, pbg, _p(out))
    return out


# --------------------------------------------------------------------------------------
# PPO maths (numpy restatement)
# --------------------------------------------------------------------------------------
def discount_cumsum(x, gamma):
    """garage/misc/tensor_utils.py:7-23 (f64 recurrence) -> f32 as torch.Tensor() casts it."""
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(len(x), np.float32)
    lib().cmo_discount_cumsum(len(x), _p(x), float(gamma), _p(out))
    return out


def gae(rewards, baselines, gamma, lam):
    """garage/torch/algos/_utils.py:106-113 on a padded [P,T] batch."""
    r, v = _f32(rewards), _f32(baselines)
    adv = np.zeros_like(r)
    lib().cmo_gae(r.shape[0], r.shape[1], _p(r), _p(v), gamma, lam, _p(adv))
    return adv


def normalize_advantages(adv, lens, eps=1e-8):
    """centralized_ma_ppo.py:422-426: per-path mean / biased var over the valid steps, applied to
    the whole padded row (F.batch_norm over the transposed batch)."""
    out = np.empty_like(adv)
    for p, n in enumerate(lens):
        v = adv[p, :n]
        m = v.mean(dtype=np.float32)
        var = ((v - m) ** 2).mean(dtype=np.float32)
        out[p] = (adv[p] - m) / np.sqrt(var + np.float32(eps))
    return out


def ppo_loss(adv_norm, new_ll, old_ll, entropy, lens, clip=0.1, ent_coeff=0.1):
    """centralized_ma_ppo.py:431-438,540-589: -(mean over valid steps of min(r*A, clip(r)*A) + c*H)."""
    ratio = np.exp(new_ll - old_ll)
    obj = np.minimum(ratio * adv_norm, np.clip(ratio, 1 - clip, 1 + clip) * adv_norm) + ent_coeff * entropy
    valid = np.concatenate([obj[p, :n] for p, n in enumerate(lens)])
    return -valid.mean(dtype=np.float32)


def critic_loss(values, returns, log_std=0.0, min_std=1e-6):
    """comm_base_critic.py:59-89 + gaussian_mlp_module.py:149-188: -mean log N(returns; v, sigma),
    sigma = exp(max(log_std, log min_std)); padded steps are included in the mean."""
    sigma = np.exp(max(np.float32(log_std), np.log(np.float32(min_std))))
    ll = -((returns - values) ** 2) / (2 * sigma ** 2) - np.log(sigma) - 0.5 * np.log(2 * np.pi)
    return np.float32(-ll.mean(dtype=np.float32))


def adam_step(p, g, m, v, step, lr=3e-4, b1=0.9, b2=0.999, eps=1e-5):
    """my_optimizer/_functional.py:72-98 (torch-1.9 Adam, no weight decay / amsgrad)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    return (p - (lr / bc1) * m / denom).astype(np.float32), m.astype(np.float32), v.astype(np.float32)
