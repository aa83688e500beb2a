"""greedy eval protocol (eval_pp.py:9-104) and the progress.csv columns
(centralized_ma_ppo.py:286-366) on the HIP path, checked against the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    O.build()
    return torch


def _setup(torch, scenario, B, mpl, seed=7):
    from com_marl_amd import envs as E, nets
    if scenario == "pp":
        params = dict(load=2, max_env_steps=mpl, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=10,
                      Rsen=1, n_agents=4, n_preys=4, n_gcn_layers=2, mode="train", trRcom=9, trpl=0, seed=seed)
        env = E.PredatorPreyWrapper(centralized=True, params=params, n_envs=B, device="cuda:0")
    else:
        params = dict(load=2, max_env_steps=mpl, capture_reward=2, step_cost=0, rm=0, penalty=1, revisit_penalty=0.5,
                      lazy_penalty=1, grid_size=20, Rsen=2, n_agents=24, n_preys=0, n_gcn_layers=2, mode="train",
                      trRcom=9, trpl=0.3, seed=seed)
        env = E.CoverageWrapper(centralized=True, params=params, n_envs=B, device="cuda:0")
    torch.manual_seed(seed)
    pol = nets.CommCategoricalMLPPolicy(env.spec, n_agents=env.n_agents, device="cuda:0")
    return env, pol, params


def _oracle_greedy_episodes(env, pol, scenario, B, T, seed, mpl, rcom=9, channel=None):
    """The reference eval loop (eval_pp.py:25-91) on the oracle, one env at a time semantics:
    argmax of the policy probabilities, stop at the first done."""
    N = env.n_agents
    ch = channel or ("IID" if scenario == "co" else "FC")
    cfg = O.make_cfg(scenario, B, N, env.maps, env.Rsen, n_preys=env.n_preys, load=2,
                     max_steps=mpl if scenario == "pp" else 400, max_path_length=mpl, channel=ch, ploss=env.pl,
                     seed=seed, rng_mode=O.RNG_PHILOX, rcom=rcom)
    oe = O.OracleEnv(cfg)
    sd = {k: v.detach().cpu().numpy() for k, v in pol.state_dict().items()}
    oe.reset()
    alive = np.ones(B, bool)
    rewards = [[] for _ in range(B)]
    details = [[] for _ in range(B)]
    success = [[] for _ in range(B)]
    degs = [[] for _ in range(B)]
    avail = np.ones((B, N, 5), np.float32)
    min_gap = np.inf
    for t in range(T):
        pr, _ = O.policy_forward(sd, oe.obs.copy(), avail, oe.dist_adj.copy(), oe.channels.copy(), N)
        srt = np.sort(pr, -1)
        min_gap = min(min_gap, float((srt[..., -1] - srt[..., -2])[alive].min()))
        deg_before = oe.dist_adj.sum(-1).mean(-1).copy()
        oe.step(pr.argmax(-1).astype(np.int32))
        for b in np.nonzero(alive)[0]:
            rewards[b].append(float(oe.reward[b]))
            details[b].append(oe.details[b].copy())
            success[b].append(int(oe.success[b]))
            degs[b].append(float(deg_before[b] if oe.done[b] else oe.dist_adj[b].sum(-1).mean(-1)))
        alive &= ~oe.done.astype(bool)
        if not alive.any():
            break
    return rewards, details, success, degs, min_gap


@pytest.mark.parametrize("scenario", ["pp", "co"])
def test_eval_model_greedy_matches_oracle(scenario, torch_cuda):
    torch = torch_cuda
    from com_marl_amd.evaluate import eval_model, VECTORS
    B, mpl = (24, 40) if scenario == "pp" else (4, 12)
    env, pol, params = _setup(torch, scenario, B, mpl)
    n_epi = B - 1                                        # not a multiple of B: the last env's episode is dropped
    data, epi_success, epi_rewards, bound = eval_model(env, pol, 0, n_eval_episodes=n_epi, max_env_steps=mpl)
    assert len(data) == n_epi and len(epi_success) == n_epi and bound == env.bound_return
    assert set(epi_rewards) == set(VECTORS) and all(len(v) == n_epi for v in epi_rewards.values())
    rewards, details, success, degs, min_gap = _oracle_greedy_episodes(env, pol, scenario, B, mpl, 7, mpl)
    assert min_gap > 1e-5, "argmax near-tie in this seed: greedy actions are not comparable at 1e-5 parity"
    N = env.n_agents
    for b in range(n_epi):
        step_success, step_data = data[b]
        n = len(rewards[b])
        assert len(step_data["reward"]) == n, f"episode {b} length"
        np.testing.assert_array_equal(step_data["reward"], rewards[b])
        det = np.asarray(details[b], np.float64)
        cap = det[:, 0] if scenario == "pp" else det[:, 0] / N
        np.testing.assert_array_equal(step_data["capture_cnt"], cap)
        np.testing.assert_array_equal(step_data["move_cnt"], det[:, 1] / N)
        np.testing.assert_array_equal(step_data["variable"], det[:, 4] / N)
        np.testing.assert_array_equal(step_success, success[b])
        np.testing.assert_allclose(step_data["nodeDeg"], degs[b], rtol=1e-6)
        assert epi_success[b] == success[b][-1]
        assert epi_rewards["reward"][b] == pytest.approx(np.sum(rewards[b]), abs=1e-12)
        assert epi_rewards["step_cnt"][b] == n
    # a second call plays NEW episodes (the Philox step counter moved on)
    r0 = env.batch.get_state()["rng_step"].copy()
    pos0 = env.batch.get_state()["agent_pos"].copy()
    data2, *_ = eval_model(env, pol, 0, n_eval_episodes=3, max_env_steps=mpl)
    st = env.batch.get_state()
    assert (st["rng_step"] > r0).all() and len(data2) == 3
    assert (st["agent_pos"] != pos0).any()


def test_transfer_scale_eval_small_team_weights_on_a_larger_map(torch_cuda):
    """SURVEY §8f-4 (train small -> test large): the weights do not depend on the team size, so a state_dict of a
    4-agent policy evaluates on a 9-agent map-20 env built with mode='test' (te* communication parameters: range
    adjacency teRcom, IID loss tepl) - greedy episodes identical to the oracle's."""
    torch = torch_cuda
    from com_marl_amd import envs as E, nets
    from com_marl_amd.evaluate import eval_model
    mpl, B, seed = 25, 12, 4
    small = _setup(torch, "pp", 4, mpl, seed=seed)[1]                     # "trained" on map 10, N = 4
    params = dict(load=2, max_env_steps=mpl, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=20, Rsen=1,
                  n_agents=9, n_preys=7, n_gcn_layers=2, mode="test", trRcom=9, trpl=0, teRcom=5, tepl=0.25, seed=seed)
    env = E.PredatorPreyWrapper(centralized=True, params=params, n_envs=B, device="cuda:0")
    assert env.channelType == "IID" and env.pl == 0.25 and env.Rcom == 5 and not env.batch.adj_const
    big = nets.CommCategoricalMLPPolicy(env.spec, n_agents=9, device="cuda:0")
    big.load_state_dict(small.state_dict())                               # same names, same shapes: d = 21 either way
    data, epi_success, epi_rewards, _ = eval_model(env, big, 0, n_eval_episodes=B, max_env_steps=mpl)
    rewards, details, success, degs, min_gap = _oracle_greedy_episodes(env, big, "pp", B, mpl, seed, mpl, rcom=5,
                                                                       channel="IID")
    assert min_gap > 1e-5
    for b in range(B):
        np.testing.assert_array_equal(data[b][1]["reward"], rewards[b])
        np.testing.assert_array_equal(data[b][0], success[b])
        np.testing.assert_allclose(data[b][1]["nodeDeg"], degs[b], rtol=1e-6)


def test_eval_plays_several_rounds_when_more_episodes_than_envs(torch_cuda):
    from com_marl_amd.evaluate import eval_model, eval_model_co, VECTORS
    env, pol, _ = _setup(torch_cuda, "co", 3, 7)
    data, succ, rew, opt = eval_model_co(env, pol, 0, n_eval_episodes=8, max_env_steps=7)      # 3 rounds of 3 envs
    assert len(data) == 8 and len(succ) == 8 and all(len(rew[v]) == 8 for v in VECTORS)
    assert opt == [env.bound_return] * 8                                                      # eval_co.py:79,101
    assert env.eval_n_epi == 8 and all(1 <= len(d[1]["reward"]) <= 7 for d in data)
    # rounds are different episodes: the Philox reset counter advances between them
    assert any(data[i][1]["reward"] != data[i + 3][1]["reward"] or data[i][1]["variable"] != data[i + 3][1]["variable"]
               for i in range(3))
    assert np.isfinite(env.last_eval_average_reward)


def test_eval_refuses_render_and_honours_flag(torch_cuda):
    from com_marl_amd.evaluate import eval_model
    env, pol, _ = _setup(torch_cuda, "pp", 4, 10)
    with pytest.raises(NotImplementedError):
        eval_model(env, pol, 0, render=True)
    assert eval_model(env, pol, 0, flag=[1]) == (None, None, None, None)
    with pytest.raises(ValueError):
        eval_model(env, pol, 0, max_env_steps=11)


@pytest.mark.parametrize("scenario", ["pp", "co"])
def test_progress_columns_device_equals_dict_path(scenario, torch_cuda):
    """_log_performance reduced on the device == the reference's host loops over the path dicts."""
    torch = torch_cuda
    from com_marl_amd import nets
    from com_marl_amd.algos import CentralizedMAPPO
    from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler
    B, mpl = (40, 14) if scenario == "pp" else (8, 9)
    env, pol, _ = _setup(torch, scenario, B, mpl)
    crit = nets.CommBaseCritic(env.spec, n_agents=env.n_agents, device="cuda:0")
    algo = CentralizedMAPPO(env_spec=env.spec, policy=pol, baseline=crit, max_path_length=mpl, discount=0.99,
                            center_adv=True, positive_adv=False, policy_ent_coeff=0.1, entropy_method="regularized",
                            stop_entropy_gradient=False, clip_grad_norm=7, optimization_n_minibatches=3,
                            optimization_mini_epochs=2)
    smp = CentralizedMAOnPolicyVectorizedSampler(algo, env, n_envs=B)
    smp.start_worker()
    paths = smp.obtain_samples(0, batch_size=B * env.n_agents * (mpl + 2))
    _, _, _, _, valids, _, returns, _, _ = algo.process_samples(0, paths)
    dev = algo._log_performance(3, paths, returns, valids)
    host = algo._log_performance(3, [paths[i] for i in range(len(paths))], returns, valids)
    want = {"Iteration", "NumTrajs", "AverageDiscountedReturn", "AverageReturn", "SuccessRate", "AverageCaptureCount",
            "AverageStepCount", "AverageMovingCount", "AveragePenaltyCount", "AverageVariable", "AverageVar2",
            "StdReturn", "MaxReturn", "MinReturn", "AveDegree", "Diameter", "AveTroughput"}   # :346-369
    assert want <= set(dev) and set(dev) == set(host)
    assert dev["NumTrajs"] == len(paths) * env.n_agents and dev["Iteration"] == 3
    # SuccessRate: the device path takes each path's OWN env flag (B independent envs); the dict path
    # averages the [n_envs] vector the reference stores (:290, sampler.py:194) - equal only for n_envs == 1
    own = [int(paths[i]["success"][int(paths.env_idx[i])]) for i in range(len(paths))]
    assert dev["SuccessRate"] == pytest.approx(np.mean(own))
    for k in sorted(want - {"SuccessRate"}):
        tol = 1e-6 if k == "AveDegree" else 1e-9      # ave_deg is an f32 mean (env_communication.py:232)
        assert dev[k] == pytest.approx(host[k], rel=tol, abs=1e-9), k
    # the oracle's f64 returns on path 0
    p0 = paths[0]
    assert dev["AverageDiscountedReturn"] == pytest.approx(host["AverageDiscountedReturn"])
    r0 = O.discount_cumsum(p0["rewards"], 0.99)[0]
    assert float(returns[0, 0]) == pytest.approx(r0, rel=1e-6)
    algo.train_once(itr=0, paths=paths)
    assert want <= set(algo.stats) and {"LossBefore", "LossAfter", "dLoss", "KLBefore", "KL", "Entropy", "GradNorm",
                                        "EpochTime"} <= set(algo.stats)
