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
        from eval_pp import eval_simple, eval_model                               # noqa: F401  (runner_pp_commDP.py:35)
        from eval_co import eval_simple as es_co, eval_model as eval_model_co     # noqa: F401
        assert "com_marl.torch.policies" in names and "eval_co" in names
        assert CommBaseCritic.__module__.startswith("com_marl_amd")
    finally:
        for k in [k for k, v in sys.modules.items() if getattr(v, "_commarl_amd", False)]:
            del sys.modules[k]
        sys.modules.update(saved)


def _purge_aliases():
    saved = {k: v for k, v in sys.modules.items()
             if (k in ("envs", "garage", "dowel", "eval_pp", "eval_co") or k.startswith(("envs.", "com_marl", "garage.")))
             and not k.startswith("com_marl_amd")}
    for k in saved:
        del sys.modules[k]
    return saved


def _restore_aliases(saved):
    for k in [k for k, v in sys.modules.items() if getattr(v, "_commarl_amd", False) or k in ("garage", "dowel")
              or k.startswith("garage.")]:
        del sys.modules[k]
    sys.modules.update(saved)
    import com_marl_amd.dropin as dropin
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, dropin._BaseSamplerHook)]


def _stub_garage_base():
    """garage.sampler.base as the reference ships it (garage/sampler/base.py:4-49): Sampler(abc.ABC) with three
    abstract methods, BaseSampler(Sampler) storing algo / env."""
    import abc
    import types

    class Sampler(abc.ABC):
        @abc.abstractmethod
        def start_worker(self):
            pass

        @abc.abstractmethod
        def obtain_samples(self, itr, batch_size, whole_paths):
            pass

        @abc.abstractmethod
        def shutdown_worker(self):
            pass

    class BaseSampler(Sampler):
        def __init__(self, algo, env):
            self.algo, self.env = algo, env
    m = types.ModuleType("garage.sampler.base")
    m.Sampler, m.BaseSampler = Sampler, BaseSampler
    return m


def _runner_gates(sampler_cls, sampler, BaseSampler):
    """The two gates of the reference runner, restated: LocalRunner.make_sampler (garage/experiment/
    local_runner.py:181-189) and LocalRunnerWrapper.obtain_samples (com_marl/experiment/local_runner_wrapper.py:41-47)."""
    return issubclass(sampler_cls, BaseSampler), isinstance(sampler, BaseSampler)


@pytest.mark.parametrize("order", ["garage_first", "install_first"])
def test_sampler_is_a_base_sampler_for_the_reference_runner(order, tmp_path):
    """The runner takes its BaseSampler branch for our sampler whichever of `import garage...` / `dropin.install()`
    comes first: registration happens at install() when garage.sampler.base is loaded already, else by an import
    hook the moment it is."""
    import types
    saved = _purge_aliases()
    try:
        import com_marl_amd.dropin as dropin
        from com_marl_amd.sampler import CentralizedMAOnPolicyVectorizedSampler as S
        if order == "garage_first":
            sys.modules["garage.sampler.base"] = _stub_garage_base()
            dropin.install()
            base = sys.modules["garage.sampler.base"]
        else:
            # a real importable garage/sampler/base.py on sys.path, imported AFTER install()
            pkg = tmp_path / "garage" / "sampler"
            pkg.mkdir(parents=True)
            (tmp_path / "garage" / "__init__.py").write_text("")
            (pkg / "__init__.py").write_text("")
            (pkg / "base.py").write_text(
                "import abc\n"
                "class Sampler(abc.ABC):\n"
                "    @abc.abstractmethod\n"
                "    def start_worker(self): pass\n"
                "    @abc.abstractmethod\n"
                "    def obtain_samples(self, itr, batch_size, whole_paths): pass\n"
                "    @abc.abstractmethod\n"
                "    def shutdown_worker(self): pass\n"
                "class BaseSampler(Sampler):\n"
                "    def __init__(self, algo, env):\n"
                "        self.algo, self.env = algo, env\n")
            dropin.install()
            sys.path.insert(0, str(tmp_path))
            try:
                import importlib
                base = importlib.import_module("garage.sampler.base")
            finally:
                sys.path.remove(str(tmp_path))
        from com_marl.sampler import CentralizedMAOnPolicyVectorizedSampler as S2
        assert S2 is S
        shell = types.SimpleNamespace(batch=types.SimpleNamespace(B=3, N=4), spec=None)
        smp = S(algo=types.SimpleNamespace(policy=None, max_path_length=9), env=shell, n_envs=3)   # the runner's ctor call
        assert _runner_gates(S, smp, base.BaseSampler) == (True, True)
        assert not issubclass(dict, base.BaseSampler)
    finally:
        _restore_aliases(saved)


def test_reference_runner_accepts_the_sampler_class():
    """With the reference tree present (build container only): the reference's own LocalRunnerWrapper.setup() builds
    our sampler through the BaseSampler branch of make_sampler, and its obtain_samples() gate holds."""
    import os
    import types
    if not os.path.isdir(os.environ.get("COMMARL_REFERENCE", "/root/reference")):
        pytest.skip("reference tree not present (GPU box)")
    saved = _purge_aliases()
    try:
        import importlib
        import tempfile
        from oracle import ref_loader as R
        R.load_reference_ppo()
        lr = importlib.import_module("garage.experiment.local_runner")
        sys.modules["garage.experiment"].LocalRunner = lr.LocalRunner
        import com_marl_amd.dropin as dropin
        dropin.install(force=True)
        wrap = importlib.import_module("com_marl.experiment.local_runner_wrapper")     # the reference's, via __path__
        assert wrap.__file__.startswith(os.environ.get("COMMARL_REFERENCE", "/root/reference"))
        from com_marl.sampler import CentralizedMAOnPolicyVectorizedSampler as S
        assert S.__module__.startswith("com_marl_amd")
        cfg = types.SimpleNamespace(snapshot_dir=tempfile.mkdtemp(), snapshot_mode="none", snapshot_gap=1)
        runner = wrap.LocalRunnerWrapper(cfg, eval=False, save_env=False)
        algo = types.SimpleNamespace(policy=types.SimpleNamespace(centralized=True), max_path_length=9, sampler_cls=S)
        env = types.SimpleNamespace(batch=types.SimpleNamespace(B=3, N=4), spec=None, n_agents=4)
        runner.setup(algo, env, sampler_cls=S, sampler_args={"n_envs": 3})            # runner_pp_commDP.py:145-151
        assert type(runner._sampler) is S and runner._sampler._n_envs == 3
        assert isinstance(runner._sampler, sys.modules["garage.sampler.base"].BaseSampler)
    finally:
        _restore_aliases(saved)
        for k in [k for k in sys.modules if k in ("gym", "akro", "pyprind", "send2trash", "pynvml", "custom_implement")
                  or k.startswith(("gym.", "custom_implement."))]:
            del sys.modules[k]


def test_runner_contract_recording():
    """What the reference's LocalRunnerWrapper.train() does to a sampler and to `paths` (recorded from the reference
    itself by oracle/gen_runner_contract.py): a class that is only BaseSampler.register()-ed takes the BaseSampler
    branches, and `paths` is only iterated with p['rewards'] read per path."""
    import json
    import os
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "runner_contract.json")))
    evs = z["events"]
    names = [e[0] for e in evs]
    assert "sampler.from_worker_factory" not in names
    assert evs[0] == ["sampler.__init__", [["n_envs", 3]], "Algo", "Env"]
    assert ["runner.setup done", "RecSampler", True] in evs
    calls = [e for e in evs if e[0] == "sampler.obtain_samples"]
    assert [c[2:] for c in calls] == [[48, True, []], [48, True, []]]               # (itr, batch_size), no agent_update
    assert {tuple(e) for e in evs if e[0].startswith("path")} == {("paths.__iter__",), ("path.__getitem__", "rewards")}
    assert z["total_env_steps_increments"] == [sum(x) for x in reversed(z["path_lengths"])]   # itr 1, then itr 2


def test_tabular_forwards_to_dowel():
    """Timers and progress columns reach dowel.tabular when dowel is importable (sampler.py:236-239 /
    centralized_ma_ppo.py:345-385 record into it)."""
    import types
    from com_marl_amd.sampler import _Tabular
    had = sys.modules.get("dowel")
    rows = {}
    sys.modules["dowel"] = types.SimpleNamespace(tabular=types.SimpleNamespace(record=rows.__setitem__))
    try:
        t = _Tabular()
        t.record("PolicyExecTime", 0.25)
        assert rows == {"PolicyExecTime": 0.25} and t.rows == rows
    finally:
        if had is None:
            del sys.modules["dowel"]
        else:
            sys.modules["dowel"] = had
    t = _Tabular()
    t._dowel = False                        # dowel absent: rows are still kept
    t.record("EnvExecTime", 1.0)
    assert t.rows == {"EnvExecTime": 1.0}


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


@pytest.mark.gpu
def test_paths_replay_of_the_reference_runner_contract():
    """The operations the reference runner applies to `paths` (tests/golden/runner_contract.json, recorded from
    LocalRunnerWrapper.train): iterate, p['rewards'], len - on a PathBatch they give the same total as the device-side
    index and copy ONE buffer (the f64 rewards) to the host, not the trajectory; the three timers are real."""
    import json
    import os
    import torch
    from com_marl_amd import sampler as S
    from tests.test_hip_ppo_parity import _small_setup
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "runner_contract.json")))
    ops = {tuple(e) for e in z["events"] if e[0].startswith("path")}
    assert ops == {("paths.__iter__",), ("path.__getitem__", "rewards")}
    env, pol, crit, algo, smp = _small_setup(torch, B=64, mpl=12, scenario="pp")
    smp.start_worker()
    paths = smp.obtain_samples(1, 64 * 4 * 12)                          # (itr, batch_size) as recorded
    total = sum([len(p['rewards']) for p in paths])                      # local_runner_wrapper.py:50-52
    assert total == int(paths.length.sum().item()) and total >= 64 * 12
    assert paths.host_buffers == ["reward64"]
    rows = S.tabular.rows
    assert rows["PolicyExecTime"] > 0 and rows["EnvExecTime"] > 0 and rows["ProcessExecTime"] >= 0
    gpu = rows["PolicyExecTime"] + rows["EnvExecTime"]
    assert 1e-5 < gpu < 5.0
    # a path is a real dict with the reference's keys; values appear on first access
    p0 = paths[0]
    assert isinstance(p0, dict) and "observations" in p0 and len(p0) == 15
    assert p0["observations"].shape == (len(p0["rewards"]), 4 * 21)
    assert "obs" in paths.host_buffers and "attn" not in paths.host_buffers
    import pickle
    q = pickle.loads(pickle.dumps(p0))
    assert type(q) is dict and set(q) == set(p0.keys())
    np.testing.assert_array_equal(q["actions"], p0["actions"])


def test_truncate_paths_walks_like_the_reference():
    """whole_paths=False (sampler :245 -> garage/sampler/utils.py:91-140): trailing paths beyond the batch are dropped, the
    last kept path is cut - and a path dict with Com-MARL's extra keys raises the reference's ValueError."""
    import numpy as np
    import pytest
    from com_marl_amd.sampler import truncate_paths
    mk = lambda n: dict(observations=np.zeros((n, 3)), actions=np.zeros((n, 2)), rewards=np.arange(n, dtype=np.float64),   # noqa: E731
                        env_infos=dict(a=np.zeros(n)), agent_infos=dict(p=np.zeros((n, 2, 5))))
    out = truncate_paths([mk(5), mk(4), mk(6), mk(2)], 11)
    assert [len(p['rewards']) for p in out] == [5, 4, 2] and out[-1]['agent_infos']['p'].shape == (2, 2, 5)
    out = truncate_paths([mk(5), mk(4)], 100)
    assert [len(p['rewards']) for p in out] == [5, 4]
    assert truncate_paths([], 5) == []
    bad = mk(4)
    bad = dict(observations=bad['observations'], actions=bad['actions'], avail_actions=np.ones((4, 10)), rewards=bad['rewards'])
    with pytest.raises(ValueError, match="Unexpected key avail_actions found in path"):
        truncate_paths([mk(3), bad], 5)
