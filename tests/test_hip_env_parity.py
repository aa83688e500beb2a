"""GPU parity of the HIP env step (through the C ABI) against (a) the reference's golden
vectors in RNG-tape mode - bit-exact integer state, exact f32 obs / f64 reward - and (b) the
CPU oracle in production Philox mode at BASELINE sizes."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_golden import ENV_FIXTURES, replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need the MI355X")
    from tests import hip_adapters
    return hip_adapters


@pytest.mark.parametrize("path", ENV_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in ENV_FIXTURES])
def test_env_hip_matches_reference_golden(path, hip):
    replay(path, hip.HipEnv)


def test_adjacency_ties_through_the_hip_env_step(hip):
    """The HIP env step emits the reference's adjacency on the tie-heavy recording (positions inside a 32 x 32 grid,
    recorded from env_communication.get_graph): exact ties dx^2+dy^2 == 2*Rcom^2 are adjacent, 164 is not."""
    from tests.test_oracle_golden import check_adjacency, drive_adjacency
    check_adjacency(drive_adjacency(hip.HipEnv, "adj_ties_grid32.npz", 32))


def test_agent_faults_and_delays_direct(hip):
    """cm_env_agent_fault / cm_comm_delays against the reference's own functions called directly
    (tests/golden/faults_direct.npz: iid_fault, GE_fault incl. its one-draw-per-group quirk, delays_init, calc_delays),
    then the production Philox stream (site 9) against the same rules applied to host-side Philox draws."""
    import torch
    from com_marl_amd import envs as E
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "faults_direct.npz"))
    base = dict(load=2, max_env_steps=20, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=30, Rsen=1,
                n_gcn_layers=2, mode="train", trRcom=9, trpl=0)
    for i in range(z["iid_u"].shape[0]):
        n, p = int(z["iid_np"][i, 0]), float(z["iid_np"][i, 1])
        env = E.GridEnvBatch("pp", dict(base, n_agents=n, n_preys=1), 3, device="cuda:0", seed=1, rng_mode="tape")
        u = np.tile(z["iid_u"][i, :n].astype(np.float32), (3, 1))
        if np.any(np.abs(z["iid_u"][i, :n] - p) < 1e-6):
            continue                                       # a float32 copy of the uniform could flip a tie
        env.apply_agent_fault("iid", p, tape_u=u)
        np.testing.assert_array_equal(env.agent_condition, np.tile(z["iid_cond"][i, :n], (3, 1)))
    for i in range(z["ge_u"].shape[0]):
        n, p, r = int(z["ge_npr"][i, 0]), float(z["ge_npr"][i, 1]), float(z["ge_npr"][i, 2])
        env = E.GridEnvBatch("pp", dict(base, n_agents=n, n_preys=1), 2, device="cuda:0", seed=1, rng_mode="tape")
        env.agent_condition = np.tile(z["ge_in"][i, :n], (2, 1))
        env.apply_agent_fault("GE", p, r, tape_u=np.tile(z["ge_u"][i].astype(np.float32), (2, 1)))
        np.testing.assert_array_equal(env.agent_condition, np.tile(z["ge_out"][i, :n], (2, 1)))
    # delays: init then three calc steps, batch of 2 identical problems
    N, Lh = z["delay_adj"].shape[0], z["delay_links"].shape[1]
    env = E.GridEnvBatch("pp", dict(base, n_agents=N, n_preys=1), 2, device="cuda:0", seed=1)
    adj = torch.as_tensor(np.tile(z["delay_adj"], (2, 1, 1))).cuda()
    links = [torch.as_tensor(np.tile(z["delay_links"][k], (2, 1, 1, 1))).cuda() for k in range(z["delay_links"].shape[0])]
    d = env.comm_delays(adj, links[0], delay_th=int(z["delay_th"]))
    np.testing.assert_array_equal(d.cpu().numpy()[1], z["delays"][0])
    for k in range(1, len(links)):
        d = env.comm_delays(adj, links[k], old_delays=d[:, -1].contiguous())
        np.testing.assert_array_equal(d.cpu().numpy()[0], z["delays"][k])
    # production stream: Philox site 9, counter (global env id, fault_step, 9, idx)
    env = E.GridEnvBatch("pp", dict(base, n_agents=9, n_preys=1), 64, device="cuda:0", seed=77, env_id_offset=5)
    env.apply_agent_fault("iid", 0.4, fault_step=3)
    want = np.zeros((64, 9), np.int64)
    for b in range(64):
        us = [O.philox((5 + b, 3, 9, q), (77, 0)) for q in range(3)]
        u = np.array([(int(w) >> 8) * (1.0 / 16777216.0) for q in range(3) for w in us[q]], np.float32)[:9]
        want[b] = O.iid_fault(u, np.float32(0.4))
    np.testing.assert_array_equal(env.agent_condition, want)
    assert 0 < want.mean() < 1
    env.apply_agent_fault("GE", 0.3, 0.6, fault_step=4)
    got = env.agent_condition
    for b in range(64):
        x = O.philox((5 + b, 4, 9, 0), (77, 0))
        ug, ub = np.float32((int(x[0]) >> 8) * (1.0 / 16777216.0)), np.float32((int(x[1]) >> 8) * (1.0 / 16777216.0))
        np.testing.assert_array_equal(got[b], O.ge_fault(want[b], ug, ub, np.float32(0.3), np.float32(0.6)))


def _lockstep(cfg_kwargs, steps, hip, seed=1234, check_every=1):
    """Same Philox stream, same random actions through oracle and HIP; compare everything."""
    cfg_o = O.make_cfg(**cfg_kwargs, rng_mode=O.RNG_PHILOX, seed=seed)
    cfg_h = O.make_cfg(**cfg_kwargs, rng_mode=O.RNG_PHILOX, seed=seed)
    eo, eh = O.OracleEnv(cfg_o), hip.HipEnv(cfg_h)
    eo.reset()
    eh.reset()
    rng = np.random.RandomState(seed)
    B, N = eo.B, eo.N
    n_done = 0
    for t in range(steps):
        a = rng.randint(0, 5, size=(B, N)).astype(np.int32)
        eo.step(a, n_threads=8)
        eh.step(a)
        n_done += int(eo.done.sum())
        if t % check_every and t != steps - 1:
            continue
        w = f"step {t}"
        np.testing.assert_array_equal(eh.done, eo.done, err_msg=w)
        np.testing.assert_array_equal(eh.reward, eo.reward, err_msg=w)
        np.testing.assert_array_equal(eh.reward32, eo.reward.astype(np.float32), err_msg=w)
        np.testing.assert_array_equal(eh.details, eo.details, err_msg=w)
        np.testing.assert_array_equal(eh.agent_pos, eo.agent_pos, err_msg=w)
        np.testing.assert_array_equal(eh.step_count, eo.step_count, err_msg=w)
        np.testing.assert_array_equal(eh.success, eo.success, err_msg=w)
        np.testing.assert_array_equal(eh.rng_step, eo.rng_step, err_msg=w)
        if eo.M:
            alive = eo.prey_alive.astype(bool)
            np.testing.assert_array_equal(eh.prey_alive.astype(bool), alive, err_msg=w)
            np.testing.assert_array_equal(eh.prey_pos[alive], eo.prey_pos[alive], err_msg=w)
            np.testing.assert_array_equal(eh.prey_alive_info, eo.prey_alive_info[:, :eo.M], err_msg=w)
        else:
            np.testing.assert_array_equal(eh.visited, eo.visited, err_msg=w)
            np.testing.assert_array_equal(eh.total_capture, eo.total_capture, err_msg=w)
        np.testing.assert_array_equal(eh.obs, eo.obs, err_msg=w)
        np.testing.assert_array_equal(eh.dist_adj, eo.dist_adj, err_msg=w)
        np.testing.assert_array_equal(eh.channels, eo.channels, err_msg=w)
        if cfg_o.channel == 3:
            np.testing.assert_array_equal(eh.ge_state, eo.ge_state, err_msg=w)
    return n_done


def test_philox_pp_config2_4096_envs(hip):
    """BASELINE config 2: PP map10 sen1 den.04 cap2, 4096 envs, > 1 full episode incl. resets."""
    n = _lockstep(dict(scenario="pp", n_envs=4096, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=200),
                  steps=230, hip=hip, check_every=10)
    assert n >= 4096


def test_philox_co_config3_2048_envs(hip):
    n = _lockstep(dict(scenario="co", n_envs=2048, n_agents=24, grid=20, rsen=2, max_steps=400, max_path_length=40),
                  steps=90, hip=hip, check_every=9)
    assert n >= 2048


def test_philox_pp_config4_map30_cap4(hip):
    n = _lockstep(dict(scenario="pp", n_envs=256, n_agents=72, n_preys=72, grid=30, rsen=2, load=4, max_steps=25),
                  steps=60, hip=hip, check_every=6)
    assert n >= 256


def test_philox_co_config5_map30_iid(hip):
    n = _lockstep(dict(scenario="co", n_envs=128, n_agents=54, grid=30, rsen=2, max_steps=400, max_path_length=15,
                       channel="IID", ploss=0.3), steps=35, hip=hip, check_every=5)
    assert n >= 128


def test_philox_co_map30_ge(hip):
    n = _lockstep(dict(scenario="co", n_envs=64, n_agents=54, grid=30, rsen=2, max_steps=400, max_path_length=12,
                       channel="GE"), steps=30, hip=hip, check_every=3)
    assert n >= 64


@pytest.mark.parametrize("ge_init,loss_apply", [(1, 0), (0, 1), (0, 0), (2, 1)])
def test_philox_ge_variants(ge_init, loss_apply, hip):
    """GE_INIT good / bad / random x loss per hop / per env step on the production Philox stream."""
    n = _lockstep(dict(scenario="pp", n_envs=96, n_agents=5, n_preys=3, grid=10, rsen=1, max_steps=9, max_path_length=9,
                       channel="GE", ge_init=ge_init, loss_apply=loss_apply, pgb=0.2, pbg=0.3), steps=25, hip=hip)
    assert n >= 96


def test_ge_random_init_with_per_step_loss_is_refused(hip):
    from com_marl_amd import _lib as L
    with pytest.raises(L.CommarlError, match="shape-inconsistent"):
        hip.HipEnv(O.make_cfg("pp", 4, 4, 10, 1, n_preys=4, channel="GE", ge_init=2, loss_apply=0))


def test_philox_pp_load3_and_fl(hip):
    _lockstep(dict(scenario="pp", n_envs=512, n_agents=8, n_preys=8, grid=10, rsen=1, load=3, max_steps=30,
                   channel="FL", ploss=1.0), steps=70, hip=hip, check_every=7)


def test_odd_agent_counts_and_hard_obstacles(hip):
    _lockstep(dict(scenario="co", n_envs=100, n_agents=3, grid=10, rsen=1, max_steps=400, max_path_length=20,
                   channel="IID", ploss=0.5, obst="Hard", add_clock=1), steps=45, hip=hip, check_every=5)
    _lockstep(dict(scenario="pp", n_envs=100, n_agents=5, n_preys=7, grid=12, rsen=2, load=2, max_steps=20, rcom=3,
                   channel="GE"), steps=45, hip=hip, check_every=5)


def test_bad_action_is_reported(hip):
    cfg = O.make_cfg("pp", 4, 4, 10, 1, n_preys=4, rng_mode=O.RNG_PHILOX)
    e = hip.HipEnv(cfg)
    e.reset()
    a = np.zeros((4, 4), np.int32)
    a[2, 1] = 7
    from com_marl_amd import CommarlError
    with pytest.raises(CommarlError):
        e.step(a)


def test_philox_map40_teams(hip):
    """Maps beyond 32 cells a side (README.md:50,76; utils_pp.py:55-65): PP map 40 den .08 cap 4 -> N = M = 128, CO map 40 den .06
    -> N = 96 on the 42 x 42 grid (two visited words per row), IID loss; every env resets at least once.  The small batches take
    the 256-thread-per-env form, 1600 envs the one-wave-per-env form."""
    n = _lockstep(dict(scenario="pp", n_envs=48, n_agents=128, n_preys=128, grid=40, rsen=2, load=4, max_steps=12), steps=30,
                  hip=hip, check_every=4)
    assert n >= 48
    n = _lockstep(dict(scenario="co", n_envs=40, n_agents=96, grid=40, rsen=2, max_steps=400, max_path_length=10, channel="IID",
                       ploss=0.3), steps=24, hip=hip, check_every=4)
    assert n >= 40
    n = _lockstep(dict(scenario="co", n_envs=1600, n_agents=96, grid=40, rsen=2, max_steps=400, max_path_length=6), steps=8,
                  hip=hip, check_every=7)
    assert n >= 1600
    _lockstep(dict(scenario="pp", n_envs=8, n_agents=200, n_preys=200, grid=50, rsen=2, load=4, max_steps=6), steps=9, hip=hip,
              check_every=2)                                                   # map 50: N = M = 200


def test_tape_exhaustion_is_reported(hip):
    cfg = O.make_cfg("pp", 2, 4, 10, 1, n_preys=4, rng_mode=O.RNG_TAPE)
    e = hip.HipEnv(cfg)
    from com_marl_amd import CommarlError
    with pytest.raises(CommarlError):
        e.reset(spawn=np.full((2, 3, 2), -1, np.int32))


def test_unsupported_config_is_refused(hip):
    from com_marl_amd import CommarlError
    with pytest.raises(CommarlError):
        hip.HipEnv(O.make_cfg("pp", 2, 4, 10, 1, n_preys=4, load=5))
    with pytest.raises(CommarlError):
        hip.HipEnv(O.make_cfg("pp", 2, 4, 70, 1, n_preys=4))     # grid side > 64 (cell coordinates travel as 6-bit fields)
    hip.HipEnv(O.make_cfg("pp", 2, 4, 40, 1, n_preys=4))         # maps 40 ... 60 are built (README.md:50,76)


def test_ragged_batch_sizes_and_empty_calls(hip):
    """Batch sizes that do not fill the last wave's env groups (4 envs per wave at N=4), and S=0 calls."""
    for B in (1, 3, 7, 61):
        _lockstep(dict(scenario="pp", n_envs=B, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=11), steps=25,
                  hip=hip, check_every=1)
    _lockstep(dict(scenario="co", n_envs=5, n_agents=6, grid=10, rsen=1, max_steps=400, max_path_length=9), steps=20,
              hip=hip, check_every=1)
    import ctypes as C
    from com_marl_amd import _lib as L
    w = L.PolicyWeights()
    assert L.lib().cm_policy_forward(C.byref(w), 0, 1, None, None, None, 0, 0, 0, None, 0, None, None, None, None) == 0
    assert L.lib().cm_gae(0, 5, 1, 1, None, 0.99, 0.97, 0, 1e-8, 1, None) == 0
    assert L.lib().cm_masked_agg_forward(0, 4, 64, 1, None, None, 0, 1, None, 1, None) == 0
    assert L.lib().cm_linear_wgrad(0, 4, 4, 1, 1, 1, None, None) == 0
