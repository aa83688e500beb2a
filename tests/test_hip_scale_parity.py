"""Parity at BASELINE batch sizes: the fused policy / critic forward, the fused rollout step and the
env step against the CPU oracle at the per-GPU batches BASELINE.json quotes - 4096 x N4, 2048 x N24, 1024 x N72,
1024 x N54-IID - plus a ragged batch (4093: partial last workgroup, partial last env group), and every env-kernel
dispatch branch the batch size selects (narrow / wide, 16 / 32 / 64 lanes per env).

Tolerances: integer state, observations, masks, f64 rewards and sampled actions bit-exact; probabilities, attention
and values within 1e-5 (north_star's float bar).  Oracle = oracle/cm_oracle.c (OpenMP over envs)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

# name: (scenario, map, sen, N, M, load, loss, B)   - BASELINE.json configs[1..4], per-GPU batch
CONFIGS = {
    "cfg2_pp_map10": ("pp", 10, 1, 4, 4, 2, 0.0, 4096),
    "cfg2_ragged": ("pp", 10, 1, 4, 4, 2, 0.0, 4093),
    "cfg3_co_map20": ("co", 20, 2, 24, 0, 2, 0.0, 2048),
    "cfg4_pp_map30": ("pp", 30, 2, 72, 72, 4, 0.0, 1024),
    "cfg5_co_map30_iid": ("co", 30, 2, 54, 0, 2, 0.3, 1024),
    "cfg5_ragged": ("co", 30, 2, 54, 0, 2, 0.3, 1021),
}
# maps beyond 32 cells a side (README.md:50,76; utils_pp.py:55-65): teams too large for the one-launch forward - the rollout runs
# the layer-by-layer forward (nets._act_device_layers) and the two-word visited rows; small batches (the oracle is the slow side)
LARGE = {
    "pp_map40": ("pp", 40, 2, 128, 128, 4, 0.0, 24),
    "co_map40_iid": ("co", 40, 2, 96, 0, 2, 0.3, 16),
    "pp_map50": ("pp", 50, 2, 200, 200, 4, 0.0, 6),
}
CONFIGS.update(LARGE)
THREADS = 16


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need the MI355X")
    return torch


def _params(scen, map_, sen, N, M, load, loss, mpl):
    pp = scen == "pp"
    return dict(load=load, max_env_steps=mpl, capture_reward=10 if pp else 2, step_cost=0.1 if pp else 0, rm=0,
                penalty=0 if pp else 1, revisit_penalty=0.5, lazy_penalty=1, grid_size=map_, Rsen=sen, n_agents=N,
                n_preys=M, n_gcn_layers=2, mode="train", trRcom=9, trpl=loss, obstComplex="Easy", add_clock=0)


def _setup(torch, name, mpl=6, seed=7, id0=5):
    from com_marl_amd import envs as E, nets
    scen, map_, sen, N, M, load, loss, B = CONFIGS[name]
    env = E.GridEnvBatch(scen, _params(scen, map_, sen, N, M, load, loss, mpl), B, device="cuda:0", seed=seed,
                         max_steps=mpl if scen == "pp" else 400, max_path_length=mpl, env_id_offset=id0)
    spec = E.EnvSpec(E._Box(np.zeros(env.d * N), np.ones(env.d * N)), E._Discrete(5))
    torch.manual_seed(seed)
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=N, device="cuda:0")
    pol.set_rng(seed, env_id_offset=id0)
    oenv = O.OracleEnv(O.make_cfg(scen, B, N, map_, sen, n_preys=M, load=load, max_steps=mpl if scen == "pp" else 400,
                                  max_path_length=mpl, channel="IID" if 0 < loss < 1 else "FC", ploss=loss, seed=seed,
                                  env_id_offset=id0, rng_mode=O.RNG_PHILOX))
    return env, pol, crit, oenv


def _sd(net):
    return {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_rollout_at_baseline_batch_matches_oracle(name, torch_cuda):
    """RolloutEngine (cm_rollout_step: policy forward + sample + env step in one launch; the last step through the
    two-launch cm_policy_forward + cm_env_step) for 8 steps incl. auto-resets, every slot against the oracle driven
    by the device's own sampled actions: probs / attention 1e-5, actions == the oracle's inverse-CDF draw on the
    device probabilities, everything the env writes bit-exact."""
    torch = torch_cuda
    from com_marl_amd.rollout import RolloutEngine
    env, pol, crit, oenv = _setup(torch, name)
    B, N = env.B, env.N
    steps = 8
    eng = RolloutEngine(env, pol, steps, fused=name not in LARGE)   # force the one-launch step for every BASELINE shape (auto = teams of 4)
    eng.reset()
    for t in range(steps - 1):
        eng.step(t)
    assert (eng._fused is True) == (name not in LARGE), "no fused kernel for a BASELINE shape"
    eng._fused = False                                   # last step: the two-launch form on the same buffers
    eng.step(steps - 1)
    torch.cuda.synchronize()
    env.check_status()
    h = {k: getattr(eng, k).cpu().numpy() for k in ("obs", "actions", "probs", "attn", "reward64", "done", "details",
                                                    "path_len", "success")}
    adj = None if eng.dist_adj is None else eng.dist_adj.cpu().numpy()
    ch = None if eng.channels is None else eng.channels.cpu().numpy()
    sd = _sd(pol)
    ones = np.ones((B, N, 5), np.float32)
    oenv.reset()
    n_done = 0
    for t in range(steps):
        w = f"{name} step {t}"
        np.testing.assert_array_equal(h["obs"][t], oenv.obs, err_msg=w)
        if adj is not None:
            np.testing.assert_array_equal(adj[t], oenv.dist_adj, err_msg=w)
        if ch is not None:
            np.testing.assert_array_equal(ch[t], oenv.channels, err_msg=w)
        p_ref, a_ref = O.policy_forward(sd, oenv.obs, ones, oenv.dist_adj, oenv.channels, N, n_threads=THREADS)
        np.testing.assert_allclose(h["probs"][t], p_ref, rtol=1e-5, atol=1e-5, err_msg=w)
        np.testing.assert_allclose(h["attn"][t], a_ref, rtol=1e-5, atol=1e-5, err_msg=w)
        np.testing.assert_array_equal(h["actions"][t], O.sample_actions(h["probs"][t], 7, 5, t), err_msg=w)
        oenv.step(h["actions"][t], n_threads=THREADS)
        np.testing.assert_array_equal(h["reward64"][t], oenv.reward, err_msg=w)
        np.testing.assert_array_equal(h["done"][t], oenv.done, err_msg=w)
        np.testing.assert_array_equal(h["details"][t], oenv.details, err_msg=w)
        n_done += int(oenv.done.sum())
    np.testing.assert_array_equal(h["obs"][steps], oenv.obs)
    st = env.get_state()
    np.testing.assert_array_equal(st["agent_pos"], oenv.agent_pos)
    np.testing.assert_array_equal(st["rng_step"], oenv.rng_step)
    assert n_done >= B, "the window should contain an auto-reset of every env"


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_critic_and_greedy_forward_at_baseline_batch(name, torch_cuda):
    """cm_critic_forward and cm_policy_forward(greedy=1, probs + attention) on one BASELINE-size batch of real
    observations / masks (two env steps after a reset) against the oracle."""
    torch = torch_cuda
    env, pol, crit, oenv = _setup(torch, name, mpl=50)
    B, N = env.B, env.N
    env.reset_all()
    oenv.reset()
    rng = np.random.RandomState(3)
    for _ in range(2):
        a = rng.randint(0, 5, size=(B, N)).astype(np.int32)
        env.step_device(torch.as_tensor(a, device="cuda:0"))
        oenv.step(a, n_threads=THREADS)
    env.check_status()
    np.testing.assert_array_equal(env.obs.cpu().numpy(), oenv.obs)
    adj = None if env.adj_const else env.dist_adj
    ch = None if env.ch_const else env.channels
    act, probs, attn = pol.act_device(env.obs.view(B, -1), None, adj, ch, greedy=True, policy_step=0)
    vals = crit.values_device(env.obs.view(B, -1), adj, ch)
    ones = np.ones((B, N, 5), np.float32)
    p_ref, a_ref = O.policy_forward(_sd(pol), oenv.obs, ones, oenv.dist_adj, oenv.channels, N, n_threads=THREADS)
    v_ref = O.critic_forward(_sd(crit), oenv.obs, oenv.dist_adj, oenv.channels, N, n_threads=THREADS)
    np.testing.assert_allclose(probs.cpu().numpy(), p_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(attn.cpu().numpy(), a_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(vals.cpu().numpy().reshape(-1), np.asarray(v_ref).reshape(-1), rtol=1e-5, atol=1e-5 * N)
    # greedy = argmax of the device's own probabilities (ties cannot be decided across implementations)
    pr = probs.cpu().numpy()
    top2 = np.sort(pr, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-5
    np.testing.assert_array_equal(act.cpu().numpy()[clear], pr.argmax(-1)[clear])


# ---- env dispatch branches selected by the batch size (cm_env.hip launch()) ------------------------------------------
def _lockstep(kw, steps, check_every):
    from tests import hip_adapters
    from tests.test_hip_env_parity import _lockstep as run
    return run(kw, steps=steps, hip=hip_adapters, check_every=check_every)


def test_env_config4_1024_envs_wide_kernel(torch_cuda):
    """Config 4 at its per-GPU batch: env_kernel_wide<CM_PP,64> (<= 1536 envs of a large team)."""
    n = _lockstep(dict(scenario="pp", n_envs=1024, n_agents=72, n_preys=72, grid=30, rsen=2, load=4, max_steps=14),
                  steps=32, check_every=4)
    assert n >= 1024


def test_env_config5_1024_envs_wide_kernel(torch_cuda):
    n = _lockstep(dict(scenario="co", n_envs=1024, n_agents=54, grid=30, rsen=2, max_steps=400, max_path_length=12,
                       channel="IID", ploss=0.3), steps=28, check_every=4)
    assert n >= 1024


def test_env_narrow_pp_one_wave_per_env_above_1536_envs(torch_cuda):
    """PP teams > 8 at > 1536 envs: env_kernel<CM_PP,64> (narrow, one wave per env), incl. a ragged last group."""
    n = _lockstep(dict(scenario="pp", n_envs=1601, n_agents=20, n_preys=17, grid=16, rsen=2, load=3, max_steps=12, rcom=4,
                       channel="IID", ploss=0.2), steps=30, check_every=5)
    assert n >= 1601


def test_env_32_lanes_per_env_from_8192_envs(torch_cuda):
    """<= 32 agents at >= 8192 envs: the LPE = 32 instantiations (two envs per wave), CO and PP."""
    n = _lockstep(dict(scenario="co", n_envs=8192, n_agents=24, grid=20, rsen=2, max_steps=400, max_path_length=10),
                  steps=22, check_every=7)
    assert n >= 8192
    n = _lockstep(dict(scenario="pp", n_envs=8195, n_agents=12, n_preys=10, grid=12, rsen=1, load=2, max_steps=10, rcom=3),
                  steps=22, check_every=7)
    assert n >= 8195


# ---- branches and builds selected by environment knobs: run in fresh child processes (tests/conftest.py) --------------
@pytest.mark.parametrize("case", ["env_lpe32", "env_lpe64_small_team", "env_lpe16_mid_team", "env_wide_off", "env_small_off",
                                  "debug_library_env_goldens"])
def test_env_knob_branches_in_child_processes(case):
    """COMMARL_ENV_LPE / COMMARL_ENV_WIDE are read once per process and COMMARL_LIB selects the CM_BOUNDS debug build, so
    each case ran in its own fresh process (started by tests/conftest.py before this process touched the GPU; see
    tests/hip_child.py for what each case does): oracle lock-step / golden replay must have passed there."""
    from tests.conftest import child_result
    res = child_result(case)
    assert res["rc"] == 0, f"child case {case} failed:\n{res['tail']}"
    assert res["report"].get("ok") is True, res


def test_two_rank_train_once_equals_one_process_on_the_union():
    """The distributed branch of CentralizedMAPPO.train_once (algos.py: SUM losses, one all-reduce per optimiser step,
    division by global counts, clip after the reduce) on two gloo ranks sharing GPU 0, ragged path shards 37 / 59 envs:
    parameters == one process trained on the union after two optimiser steps as configured and after five with PPO's
    ratio clip off (its gradient discontinuity makes longer clipped chains sensitive to the last bit of a sum on any
    number of GPUs - tests/dist_train_child.py)."""
    from tests.conftest import child_result
    res = child_result("two_rank_train_once")
    assert res["rc"] == 0, f"two-rank case failed:\n{res['tail']}"
    rep = res["report"]
    assert rep.get("ok") is True and rep["max_param_diff"] <= 2e-6 and rep["n_paths"][0] != rep["n_paths"][1], rep


def test_bench_two_rank_rehearsal_line():
    """bench.py's N > 1 branch end to end (self-launch, env sharding by global id, max-over-ranks timing, train loop with the
    gradient all-reduce), weak and strong scaling, as two gloo ranks on the one card of this box: the line says `rehearsal`
    and counts ONE GPU (tests/hip_child.py::bench_two_rank_rehearsal)."""
    from tests.conftest import child_result
    res = child_result("bench_two_rank_rehearsal")
    assert res["rc"] == 0, f"two-rank bench rehearsal failed:\n{res['tail']}"
    rep = res["report"]
    assert rep.get("ok") is True and rep["weak"]["value"] > 0 and rep["strong"]["value"] > 0, rep
