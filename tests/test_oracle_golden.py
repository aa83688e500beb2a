"""Pins the CPU oracle (oracle/cm_oracle.c) against every golden vector generated from the
reference (oracle/gen_golden.py): integer state bit-exact, obs / reward exact, in RNG-tape mode."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "env_*.npz")))


def replay(path, make_env, check_every=1):
    """Shared replay loop: make_env(cfg) -> object with the OracleEnv interface."""
    z = np.load(path)
    cfgj = json.loads(str(z["cfg"]))
    T, B = z["actions"].shape[:2]
    cfg = O.cfg_from_json(cfgj, B, rng_mode=O.RNG_TAPE)
    env = make_env(cfg)
    pp = cfgj["scenario"] == "pp"

    def tape_at(prefix, t=None):
        def g(k):
            k = prefix + k
            if k not in z.files:
                return None
            return z[k] if t is None else z[k][t]
        return dict(prey=g("prey_tape"), spawn=g("spawn_tape"), iid_u=g("iid_u"), ge_u=g("ge_u"), ge_init_u=g("ge_init_u"))

    def check_state(t, where):
        np.testing.assert_array_equal(env.agent_pos, z["agent_pos"][t], err_msg=f"agent_pos {where}")
        np.testing.assert_array_equal(env.step_count, z["step_count"][t], err_msg=f"step_count {where}")
        if pp:
            alive = z["prey_alive"][t].astype(bool)
            np.testing.assert_array_equal(env.prey_alive.astype(bool), alive, err_msg=f"alive {where}")
            # a dead prey's stale position is not observable; compare live ones
            np.testing.assert_array_equal(env.prey_pos[alive], z["prey_pos"][t][alive], err_msg=f"prey_pos {where}")
        else:
            np.testing.assert_array_equal(env.visited_dense(), z["visited"][t], err_msg=f"visited {where}")
            np.testing.assert_array_equal(env.total_capture, z["total_capture"][t], err_msg=f"total_capture {where}")
        np.testing.assert_array_equal(env.obs.reshape(B, -1), z["obs"][t], err_msg=f"obs {where}")
        np.testing.assert_array_equal(env.dist_adj, z["dist_adj"][t], err_msg=f"dist_adj {where}")
        np.testing.assert_array_equal(env.channels, z["channels"][t], err_msg=f"channels {where}")

    env.reset(**tape_at("init_"))
    check_state(0, "after reset")
    has_cond = "agent_cond" in z.files             # PP agent_condition set after the reset (predator_prey.py:258 gate)
    if has_cond:
        env.load_state(agent_cond=z["agent_cond"][0])
    for t in range(T):
        env.step(z["actions"][t], **tape_at("", t))
        where = f"step {t}"
        if has_cond:                               # the env's own reset puts it back to ones (:152)
            np.testing.assert_array_equal(env.agent_cond, z["agent_cond"][t + 1], err_msg=f"agent_cond {where}")
        np.testing.assert_array_equal(env.done, z["done"][t], err_msg=f"done {where}")
        np.testing.assert_array_equal(env.reward, z["reward"][t], err_msg=f"reward {where}")
        np.testing.assert_array_equal(env.details[:, :5], z["details"][t][:, :5], err_msg=f"details {where}")
        np.testing.assert_array_equal(env.success, z["success"][t + 1], err_msg=f"success {where}")
        if pp:
            np.testing.assert_array_equal(env.prey_alive_info, z["prey_alive_info"][t], err_msg=f"info {where}")
        if t % check_every == 0 or t == T - 1:
            check_state(t + 1, where)
    return z, env


@pytest.mark.parametrize("path", ENV_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in ENV_FIXTURES])
def test_env_oracle_matches_reference(path):
    replay(path, O.OracleEnv)


def test_fixture_set_is_complete():
    names = {os.path.basename(p) for p in ENV_FIXTURES}
    for need in ("env_pp_map10_cap2.npz", "env_pp_map30_cap4.npz", "env_co_map20.npz", "env_co_map30_iid.npz",
                 "env_co_map20_ge.npz"):
        assert need in names


def test_co_constants():
    z = np.load(os.path.join(GOLDEN, "env_co_map20.npz"))
    cfg = O.cfg_from_json(str(z["cfg"]), 1)
    env = O.OracleEnv(cfg)
    assert env.n_empty_cells == int(z["n_empty_cells"]) == 320
    # bound_return (coverage.py:214-219) = cap*n_empty/N - |step|*n_empty/N + final
    n = cfg.n_agents
    assert cfg.capture_reward * env.n_empty_cells / n - abs(cfg.step_cost) * env.n_empty_cells / n + 100 == float(
        z["bound_return"])


def test_ge_transition_direct():
    z = np.load(os.path.join(GOLDEN, "ge_direct.npz"))
    s = np.ones(z["states"].shape[1:], np.uint8)
    for h in range(z["states"].shape[0]):
        s = O.ge_transition(s, z["u"][h, 0], z["u"][h, 1], float(z["pgb"]), float(z["pbg"]))
        np.testing.assert_array_equal(s, z["states"][h], err_msg=f"hop {h}")
    assert z["states"].min() == 0     # some links did go bad


def drive_adjacency(make_env, fixture, grid):
    """Place the agents on the recorded tie-heavy positions, step once with every agent staying put, and return the
    adjacency the ENV STEP emitted (update_communication_state -> get_graph, env_communication.py:218-243).  One far
    prey keeps the PP episode alive (no prey -> done -> auto-reset would emit the reset's graph instead)."""
    z = np.load(os.path.join(GOLDEN, fixture))
    out = {}
    for n in (4, 24, 26, 54, 72):
        pos = z[f"pos_{n}"]
        cfg = O.make_cfg("pp", 1, n, grid, 1, n_preys=1, rcom=9, max_steps=50, rng_mode=O.RNG_PHILOX, seed=n)
        env = make_env(cfg)
        env.reset()
        occupied = {tuple(q) for q in pos}
        prey = next((r, c) for r in range(grid - 1, -1, -1) for c in range(grid - 1, -1, -1)
                    if all((r + dr, c + dc) not in occupied for dr in (-1, 0, 1) for dc in (-1, 0, 1)))
        yield_state = dict(agent_pos=pos[None].astype(np.int32), prey_pos=np.array([[prey]], np.int32),
                           prey_alive=np.ones((1, 1), np.uint8), step_count=np.zeros(1, np.int32))
        env.load_state(**yield_state)
        env.step(np.full((1, n), 4, np.int32))
        assert not env.done[0]
        np.testing.assert_array_equal(env.agent_pos[0], pos)
        out[n] = (env.dist_adj[0].copy(), z[f"adj_{n}"], float(z[f"deg_{n}"]), pos)
    return out


def check_adjacency(results):
    for n, (got, want, deg, pos) in results.items():
        np.testing.assert_array_equal(got, want, err_msg=f"N={n}")
        assert abs(got.sum(1).mean() - deg) < 1e-5                       # ave_deg incl. self loops (:232)
        d2 = ((pos[:, None, :].astype(np.int64) - pos[None, :, :]) ** 2).sum(-1)
        assert (d2 == 162).sum() >= 2, "fixture should hold exact ties"
        assert got[d2 == 162].all() and not got[d2 > 162].any()


@pytest.mark.parametrize("fixture,grid", [("adj_ties.npz", 40), ("adj_ties_grid32.npz", 32)])
def test_adjacency_ties(fixture, grid):
    """The oracle's env step emits exactly the reference's adjacency on the tie-heavy recordings: the integer rule
    dx^2+dy^2 <= 2*Rcom^2 == f32 cdist <= Rcom_th incl. exact ties."""
    check_adjacency(drive_adjacency(O.OracleEnv, fixture, grid))


def test_fault_and_delay_helpers_direct():
    """iid_fault / GE_fault / delays_init / calc_delays (env_communication.py:270-301; never called by the reference's own
    code) against recordings of the functions called directly, on the uniforms each call consumed."""
    z = np.load(os.path.join(GOLDEN, "faults_direct.npz"))
    for i in range(z["iid_u"].shape[0]):
        n, p = int(z["iid_np"][i, 0]), float(z["iid_np"][i, 1])
        np.testing.assert_array_equal(O.iid_fault(z["iid_u"][i, :n], p), z["iid_cond"][i, :n])
    for i in range(z["ge_u"].shape[0]):
        n, p, r = int(z["ge_npr"][i, 0]), float(z["ge_npr"][i, 1]), float(z["ge_npr"][i, 2])
        got = O.ge_fault(z["ge_in"][i, :n], z["ge_u"][i, 0], z["ge_u"][i, 1], p, r)
        np.testing.assert_array_equal(got, z["ge_out"][i, :n])
    d = O.delays_init(z["delay_adj"], z["delay_links"][0], int(z["delay_th"]))
    np.testing.assert_array_equal(d, z["delays"][0])
    for k in range(1, z["delays"].shape[0]):
        d = O.calc_delays(z["delay_adj"], z["delay_links"][k], d[-1])
        np.testing.assert_array_equal(d, z["delays"][k])
    assert z["delays"].max() > int(z["delay_th"])     # a link stayed lost long enough to pass the threshold


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in O.philox(ctr, key)) == want


POLICY_FIXTURES = [("policy_pp_map10", 4), ("policy_co_map20", 24), ("policy_pp_map30", 72),
                   ("policy_co_map30_iid", 54),
                   # GCN depth 0 / 1 / 3 and the skip connection switched off (comm_categorical_mlp_policy.py:74-77)
                   ("policy_pp_map10_hops0", 4), ("policy_pp_map10_hops1_nores", 4), ("policy_co_map20_hops3", 24)]


@pytest.mark.parametrize("name,n_agents", POLICY_FIXTURES)
def test_policy_critic_forward(name, n_agents):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    pol = {k[4:]: z[k] for k in z.files if k.startswith("pol.")}
    crit = {k[5:]: z[k] for k in z.files if k.startswith("crit.")}
    S = z["obs"].shape[0]
    ones = np.ones((S, n_agents, 5), np.float32)
    res = bool(z["residual"]) if "residual" in z.files else True
    probs, attn, emb = O.policy_forward(pol, z["obs"], ones, z["adj"], z["channels"], n_agents, want_emb=True, residual=res)
    tol = dict(rtol=1e-5, atol=1e-5)      # north_star float tolerance
    np.testing.assert_allclose(probs, z["probs"], **tol)
    np.testing.assert_allclose(attn, z["attn"], **tol)
    for l in range(emb.shape[1]):
        np.testing.assert_allclose(emb[:, l], z[f"emb{l}"], **tol)
    probs_m, _ = O.policy_forward(pol, z["obs"], z["avail_masked"], z["adj"], z["channels"], n_agents, residual=res)
    np.testing.assert_allclose(probs_m, z["probs_masked"], **tol)
    # entropy = mean over agents, log-lik = sum over agents (comm_categorical_mlp_policy.py:121-137)
    ent = -(probs * np.log(probs)).sum(-1).mean(-1)
    np.testing.assert_allclose(ent, z["entropy"], **tol)
    a = z["actions"]
    ll = np.log(np.take_along_axis(probs, a[..., None], -1)[..., 0]).sum(-1)
    np.testing.assert_allclose(ll, z["loglik"], rtol=1e-5, atol=1e-5 * n_agents)
    v = O.critic_forward(crit, z["obs"], z["adj"], z["channels"], n_agents, residual=res)
    np.testing.assert_allclose(v, z["values"], rtol=1e-5, atol=1e-5 * n_agents)
    closs = O.critic_loss(v, z["returns"], log_std=float(crit["baseline_aggregator._init_std"][0]))
    np.testing.assert_allclose(closs, z["critic_loss"], rtol=1e-5)


def test_ppo_math():
    z = np.load(os.path.join(GOLDEN, "ppo_math.npz"))
    lens = z["lens"]
    g, lam = float(z["gamma"]), float(z["lam"])
    for p, n in enumerate(lens):
        r = O.discount_cumsum(z["rewards_pad"][p, :n], g)
        np.testing.assert_array_equal(r, z["returns"][p, :n])          # f64 recurrence -> exact f32
        assert (z["returns"][p, n:] == 0).all()
    adv = O.gae(z["rewards_pad"].astype(np.float32), z["baselines"], g, lam)
    np.testing.assert_allclose(adv, z["adv"], rtol=1e-5, atol=1e-5)
    advn = O.normalize_advantages(z["adv"], lens)
    np.testing.assert_allclose(advn, z["adv_norm"], rtol=1e-5, atol=1e-5)
    loss = O.ppo_loss(z["adv_norm"], z["new_ll"], z["old_ll"], z["ent"], lens)
    np.testing.assert_allclose(loss, z["loss"], rtol=1e-5)


def test_adam_matches_reference_optimizer():
    z = np.load(os.path.join(GOLDEN, "adam.npz"))
    p = z["params"][0]
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for s in range(z["grads"].shape[0]):
        p, m, v = O.adam_step(p, z["grads"][s], m, v, s + 1)
        np.testing.assert_allclose(p, z["params"][s + 1], rtol=1e-6, atol=1e-7)
