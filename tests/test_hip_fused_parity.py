"""cm_rollout_step (policy forward + sample + env step in ONE launch) against the two-launch path: same device
bodies, same Philox counters -> every trajectory buffer must be bit-identical, for every BASELINE shape, with ragged
batch sizes (partial workgroups), auto-resets inside the window, greedy and sampled actions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = {
    # name: (scenario, map, sen, N, M, load, loss, B, steps, mpl)
    "pp_map10": ("pp", 10, 1, 4, 4, 2, 0.0, 203, 40, 9),
    # every workgroup full (a multiple of 8 envs): the constant-shape build with the env state prefetched in front of
    # the policy forward (rollout_step_kernel<..., FULL, PRE>, cm_fused.hip)
    "pp_map10_full": ("pp", 10, 1, 4, 4, 2, 0.0, 208, 40, 9),
    # 6 preys on the same grid: the carried rollout kernel in its run-time-shape build (the two shapes above take the map10 shape build)
    "pp_map10_m6": ("pp", 10, 1, 4, 6, 2, 0.0, 41, 30, 8),
    # one GCN hop, no skip connection, range adjacency + IID loss (masks read by the wave-owned kernel), 5 preys; ragged batch
    "pp_map10_hop1": ("pp", 10, 1, 4, 5, 2, 0.3, 77, 24, 7),
    "co_map20": ("co", 20, 2, 24, 0, 2, 0.0, 7, 14, 6),
    "pp_map30": ("pp", 30, 2, 72, 72, 4, 0.0, 3, 8, 5),
    "co_map30_iid": ("co", 30, 2, 54, 0, 2, 0.3, 3, 8, 5),
}


def _hops(shape):
    return 1 if shape.endswith("hop1") else 2


def _params(scen, map_, sen, N, M, load, loss, mpl, hops=2):
    pp = scen == "pp"
    return dict(load=load, max_env_steps=mpl, capture_reward=10 if pp else 2, step_cost=0.1 if pp else 0, rm=0,
                penalty=0 if pp else 1, revisit_penalty=0.5, lazy_penalty=1, grid_size=map_, Rsen=sen, n_agents=N,
                n_preys=M, n_gcn_layers=hops, mode="train", trRcom=9 if hops == 2 else 3, trpl=loss, obstComplex="Easy", add_clock=0)


def _run(torch, shape, fused, greedy, chunked=False, faults=False):
    from com_marl_amd import envs as E, nets
    from com_marl_amd.rollout import RolloutEngine
    scen, map_, sen, N, M, load, loss, B, steps, mpl = SHAPES[shape]
    env = E.GridEnvBatch(scen, _params(scen, map_, sen, N, M, load, loss, mpl, _hops(shape)), B, device="cuda:0", seed=3,
                         max_steps=mpl if scen == "pp" else 400, max_path_length=mpl, env_id_offset=11)
    if faults:           # some agents cannot move (predator_prey.py:257-261): their condition travels with the prefetch
        env.apply_agent_fault("iid", 0.35, fault_step=2)
        assert 0 < env.agent_condition.mean() < 1
    spec = E.EnvSpec(E._Box(np.zeros(env.d * N), np.ones(env.d * N)), E._Discrete(5))
    torch.manual_seed(3)
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, n_gcn_layers=_hops(shape), residual=_hops(shape) == 2, device="cuda:0")
    pol.set_rng(3)
    eng = RolloutEngine(env, pol, steps, fused=fused)
    eng.reset()
    if chunked:          # two persistent launches: slots 0..2, then 3..steps-1
        assert eng.steps_fused(0, 3, greedy=greedy) and eng.steps_fused(3, steps - 3, greedy=greedy)
    else:
        for t in range(steps):
            eng.step(t, greedy=greedy)
    torch.cuda.synchronize()
    env.check_status()
    bufs = {k: getattr(eng, k) for k in ("obs", "actions", "probs", "attn", "reward", "reward64", "done", "details",
                                         "prey_alive", "success", "path_len", "dist_adj", "channels")}
    out = {k: v.cpu().numpy() for k, v in bufs.items() if v is not None}
    out["state"] = env.get_state()
    return out, eng._fused


@pytest.mark.parametrize("greedy", [False, True], ids=["sample", "greedy"])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_fused_step_is_bit_identical_to_two_launches(shape, greedy):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need the MI355X")
    a, used = _run(torch, shape, True, greedy)
    assert used is True, "the library reported no fused kernel for a BASELINE shape"
    b, used_b = _run(torch, shape, False, greedy)
    assert used_b is False
    assert a["done"].any(), "the window should contain auto-resets"
    for k in sorted(b):
        if k == "state":
            for kk in b[k]:
                np.testing.assert_array_equal(a[k][kk], b[k][kk], err_msg=f"state.{kk}")
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("chunked", [False, True], ids=["steps", "chunks"])
@pytest.mark.parametrize("shape", ["pp_map10", "pp_map10_full"])
def test_fused_step_with_agent_faults(shape, chunked):
    """Faulty agents (condition 0: the move is computed, counted and not applied) through the fused step's own staging of
    the env state - ragged and full workgroups - against the two-launch path; as single-step launches and as multi-step
    launches, where the condition travels from step to step inside the wave (EnvCarry) and an auto-reset re-arms it."""
    import torch
    a, used = _run(torch, shape, True, False, faults=True, chunked=chunked)
    assert used is True
    b, _ = _run(torch, shape, False, False, faults=True)
    for k in sorted(b):
        if k == "state":
            for kk in b[k]:
                np.testing.assert_array_equal(a[k][kk], b[k][kk], err_msg=f"state.{kk}")
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_persistent_chunk_is_bit_identical_to_two_launches(shape):
    """cm_rollout_chunk: n steps inside one launch (no grid-wide sync between steps) == n x (policy, env) launches."""
    import torch
    a, used = _run(torch, shape, True, False, chunked=True)
    assert used is True
    b, _ = _run(torch, shape, False, False)
    assert a["done"].any()
    for k in sorted(b):
        if k == "state":
            for kk in b[k]:
                np.testing.assert_array_equal(a[k][kk], b[k][kk], err_msg=f"state.{kk}")
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_run_chunk_uses_the_persistent_kernel_and_matches_stepwise_launches():
    import torch
    from com_marl_amd import envs as E, nets
    from com_marl_amd.rollout import RolloutEngine
    scen, map_, sen, N, M, load, loss, B, steps, mpl = SHAPES["pp_map10"]
    outs = []
    for persistent, use_graph in ((True, False), (False, False), (False, True), (True, True), ("auto", True)):
        shards = [E.GridEnvBatch(scen, _params(scen, map_, sen, N, M, load, loss, mpl), 64, device="cuda:0", seed=3,
                                 max_steps=mpl, max_path_length=mpl, env_id_offset=64 * k) for k in range(2)]
        spec = E.EnvSpec(E._Box(np.zeros(shards[0].d * N), np.ones(shards[0].d * N)), E._Discrete(5))
        torch.manual_seed(3)
        pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, device="cuda:0")
        pol.set_rng(3)
        eng = RolloutEngine(shards, pol, 12, persistent=persistent)   # default "auto": persistent for teams of 4
        eng.reset()
        for _ in range(3):
            eng.run_chunk(use_graph=use_graph)  # graph capture does not advance the rollout: comparable slot by slot
        torch.cuda.synchronize()
        eng.env.check_status()
        assert use_graph or eng._fused is True
        outs.append([getattr(eng, k).cpu().numpy() for k in ("obs", "actions", "probs", "reward64", "done", "path_len")])
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            np.testing.assert_array_equal(x, y)


def test_fused_entry_point_reports_unavailable_and_errors():
    import ctypes as C
    import torch
    from com_marl_amd import _lib as L, envs as E, nets
    # CO sen1 (d = 29): no fused instantiation -> 1, nothing done, the engine falls back
    p = _params("co", 10, 1, 3, 0, 2, 0.0, 8)
    env = E.GridEnvBatch("co", p, 5, device="cuda:0", seed=1, max_path_length=8)
    spec = E.EnvSpec(E._Box(np.zeros(env.d * 3), np.ones(env.d * 3)), E._Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=3, device="cuda:0")
    from com_marl_amd.rollout import RolloutEngine
    eng = RolloutEngine(env, pol, 4, fused=True)
    eng.reset()
    for t in range(4):
        eng.step(t)
    torch.cuda.synchronize()
    env.check_status()
    assert eng._fused is False and int(env.get_state()["step_count"].max()) == 4
    # shape mismatch between handle and weights is an error, not a silent fallback
    spec4 = E.EnvSpec(E._Box(np.zeros(env.d * 4), np.ones(env.d * 4)), E._Discrete(5))
    pol4 = nets.CommCategoricalMLPPolicy(spec4, n_agents=4, device="cuda:0")
    w = pol4._weights_struct()
    so = env._out()
    rc = L.lib().cm_rollout_step(env._h, C.byref(w), env.obs.data_ptr(), None, None, None, 1, 0, 0, None, 0, None, None,
                                 None, None, C.byref(so), None)
    assert rc == -1 and b"does not match" in L.lib().cm_last_error()
