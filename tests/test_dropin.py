"""The reference's import lines resolve to this package after dropin.install() (CPU), and a
runner-shaped script trains end to end on the GPU with the reference's constructor calls."""
import sys
from types import SimpleNamespace

import numpy as np
import pytest


def test_install_registers_reference_import_names():
    saved = {k: v for k, v in sys.modules.items() if k == "envs" or k.startswith(("envs.", "com_marl"))
             and not k.startswith("com_marl_amd")}
    for k in saved:
        del sys.modules[k]
    try:
        import com_marl_amd.dropin as dropin
        names = dropin.install()
        from envs import PredatorPreyWrapper, CoverageWrapper                     # noqa: F401
        from com_marl.torch.policies import CommCategoricalMLPPolicy              # noqa: F401
        from com_marl.torch.baselines import CommBaseCritic                       # noqa: F401
        from com_marl.torch.policies import DecCategoricalMLPPolicy, CentralizedCategoricalMLPPolicy   # noqa: F401
        from com_marl.torch.baselines import GaussianMLPBaseline                  # noqa: F401
        from com_marl.torch.algos import CentralizedMAPPO                         # noqa: F401
        from com_marl.sampler import CentralizedMAOnPolicyVectorizedSampler       # noqa: F401
        from eval_pp import eval_model                                            # noqa: F401
        from eval_co import eval_model as eval_model_co                           # noqa: F401
        assert "com_marl.torch.policies" in names and "eval_co" in names
        assert CommBaseCritic.__module__.startswith("com_marl_amd")
    finally:
        for k in [k for k, v in sys.modules.items() if getattr(v, "_commarl_amd", False)]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_state_dict_names_match_reference():
    """Parameter names are the checkpoint interchange format: compare against
    the names stored in the reference-generated fixture."""
    import os
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy_pp_map10.npz"))
    spec = EnvSpec(_Box(np.zeros(84), np.ones(84)), _Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=4)
    crit = nets.CommBaseCritic(spec, n_agents=4)
    assert sorted(pol.state_dict()) == sorted(k[4:] for k in z.files if k.startswith("pol."))
    assert sorted(crit.state_dict()) == sorted(k[5:] for k in z.files if k.startswith("crit."))
    for k, v in pol.state_dict().items():
        assert tuple(v.shape) == z["pol." + k].shape
    # 128 d + 39 621 policy / 128 d + 25 026 critic parameters
    assert sum(p.numel() for p in pol.parameters()) == 128 * 21 + 39621
    assert sum(p.numel() for p in crit.parameters()) == 128 * 21 + 25026


def test_variant_state_dict_names_match_reference():
    """Obs-DP / CENT nets: parameter names and shapes of the reference classes."""
    import json
    import os
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "variants_pp_map10.npz"))
    assert json.loads(str(z["regen"])) == {}                  # small config: every tensor is stored
    spec = EnvSpec(_Box(np.zeros(84), np.ones(84)), _Discrete(5))
    made = dict(dec=nets.DecCategoricalMLPPolicy(spec, 4, hidden_sizes=[128, 64, 32]),
                cent=nets.CentralizedCategoricalMLPPolicy(spec, n_agents=4, hidden_sizes=[128, 64, 32]),
                gb=nets.GaussianMLPBaseline(env_spec=spec, hidden_sizes=(64, 64, 64)))
    for tag, net in made.items():
        pre = tag + ".sd."
        assert sorted(net.state_dict()) == sorted(k[len(pre):] for k in z.files if k.startswith(pre)), tag
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == z[pre + k].shape, (tag, k)
    assert not hasattr(made["dec"], "comm") and not hasattr(made["cent"], "comm")   # sampler / algo dispatch key
    assert made["gb"].name != "base_critic"


def _args():
    return SimpleNamespace(
        grid_size=10, Rsen=1, n_agents=4, n_preys=4, load=2, max_env_steps=20, capture_reward=10, step_cost=0.1, rm=0,
        penalty=0, n_gcn_layers=2, mode="train", trRcom=9, teRcom=9, trpl=0, tepl=0, channelType="FC", loss_apply=1,
        curriculum_learning=0, n_groups=1, n_nodes=1, calc_diameter=False, encoder_hidden_sizes=[128], embedding_dim=64,
        attention_type="general", residual=1, gcn_bias=1, categorical_mlp_hidden_sizes=[128, 64, 32],
        aggregator_type="sum", discount=0.99, center_adv=1, positive_adv=0, gae_lambda=0.97, ent=0.1,
        entropy_method="regularized", clip_grad_norm=7, opt_n_minibatches=3, opt_mini_epochs=10, device="cuda:0",
        agent_visible=1, n_envs=64, seed=1, bs=64 * 4 * 20, n_epochs=2)


@pytest.mark.gpu
def test_runner_shaped_training_script():
    import torch
    assert torch.cuda.is_available()
    import com_marl_amd.dropin as dropin
    dropin.install(force=True)
    from envs import PredatorPreyWrapper
    from com_marl.torch.policies import CommCategoricalMLPPolicy
    from com_marl.torch.baselines import CommBaseCritic
    from com_marl.torch.algos import CentralizedMAPPO
    from com_marl.sampler import CentralizedMAOnPolicyVectorizedSampler
    args = _args()
    # --- the body of train_predatorprey (runner_pp_commDP.py:89-153), constructor calls unchanged ---
    env = PredatorPreyWrapper(centralized=True, grid_shape=(args.grid_size, args.grid_size), n_agents=args.n_agents,
                              n_preys=args.n_preys, max_steps=args.max_env_steps, step_cost=args.step_cost,
                              prey_capture_reward=args.capture_reward, penalty=args.penalty,
                              other_agent_visible=bool(args.agent_visible), params=vars(args),
                              n_envs=args.n_envs, device=args.device)              # <- the two new kwargs
    policy = CommCategoricalMLPPolicy(env.spec, n_agents=args.n_agents, encoder_hidden_sizes=args.encoder_hidden_sizes,
                                      embedding_dim=args.embedding_dim, attention_type=args.attention_type,
                                      n_gcn_layers=args.n_gcn_layers, residual=bool(args.residual),
                                      gcn_bias=bool(args.gcn_bias),
                                      categorical_mlp_hidden_sizes=args.categorical_mlp_hidden_sizes,
                                      name='comm_categorical_mlp_policy', device=args.device)
    baseline = CommBaseCritic(env.spec, n_agents=args.n_agents, encoder_hidden_sizes=args.encoder_hidden_sizes,
                              embedding_dim=args.embedding_dim, attention_type=args.attention_type,
                              n_gcn_layers=args.n_gcn_layers, residual=bool(args.residual), gcn_bias=bool(args.gcn_bias),
                              aggregator_type=args.aggregator_type, device=args.device)
    algo = CentralizedMAPPO(env_spec=env.spec, policy=policy, baseline=baseline, max_path_length=args.max_env_steps,
                            discount=args.discount, center_adv=bool(args.center_adv),
                            positive_adv=bool(args.positive_adv), gae_lambda=args.gae_lambda,
                            policy_ent_coeff=args.ent, entropy_method=args.entropy_method,
                            stop_entropy_gradient=True if args.entropy_method == 'max' else False,
                            clip_grad_norm=args.clip_grad_norm, optimization_n_minibatches=args.opt_n_minibatches,
                            optimization_mini_epochs=args.opt_mini_epochs, device=args.device)
    runner = dropin.SimpleRunner()
    runner.setup(algo, env, sampler_cls=CentralizedMAOnPolicyVectorizedSampler, sampler_args={'n_envs': args.n_envs})
    ret = runner.train(n_epochs=args.n_epochs, batch_size=args.bs)
    assert np.isfinite(ret) and len(runner.history) == 2 and runner.total_env_steps >= 2 * 64 * 20
    assert runner.history[-1]["LossAfter"] < runner.history[-1]["LossBefore"]


@pytest.mark.gpu
def test_single_env_wrapper_keeps_reference_shapes():
    """n_envs=1: reset()/step() return exactly the reference's shapes and types."""
    import torch
    assert torch.cuda.is_available()
    from com_marl_amd.envs import CoverageWrapper, PredatorPreyWrapper
    a = vars(_args())
    env = PredatorPreyWrapper(centralized=True, params=a, n_envs=1, device="cuda:0")
    obs = env.reset()
    assert obs.shape == (4 * 21,) and obs.dtype == np.float64
    assert env.dist_adj.shape == (4, 4) and env.dist_adj.dtype == np.float64 and env.channels.shape == (2, 4, 4)
    assert env.get_avail_actions().shape == (20,) and env.ave_deg == 4 and isinstance(env.agent_pos, dict)
    o, (r, det), done, info = env.step(np.array([0, 1, 2, 4]))
    assert o.shape == (84,) and isinstance(r, float) and isinstance(done, bool) and info["prey_alive"].shape == (4,)
    assert set(det) == {"reward", "capture_cnt", "step_cnt", "move_cnt", "penalty_cnt", "variable", "vars2"}
    assert det["move_cnt"] == 0.75 and det["reward"] == r
    assert env.bound_return == 40 and env.observation_space.flat_dim == 84 and env.action_space.n == 5
    co = dict(a, grid_size=20, Rsen=2, n_agents=24, capture_reward=2, step_cost=0, penalty=1, revisit_penalty=0.5,
              lazy_penalty=1, obstComplex="Easy", add_clock=0, trpl=0.3)
    cenv = CoverageWrapper(centralized=True, params=co, n_envs=3, device="cuda:0")
    o = cenv.reset()
    assert o.shape == (3, 24 * 77) and cenv.channelType == "IID" and cenv.dist_adj.shape == (3, 24, 24)
    assert cenv.ave_trput == 320 and abs(cenv.bound_return - (2 * 320 / 24 + 100)) < 1e-9
    o, (r, det), done, info = cenv.step(np.zeros((3, 24), np.int64))
    assert r.shape == (3,) and done.shape == (3,) and len(det) == 3 and info == {}
    with pytest.raises(Exception):
        env.step(np.array([0, 1, 2, 9]))                                   # 'Action Not found!' (:255)
