"""Drop-in aliases: makes the reference's import lines resolve to this package.

The reference's runner scripts import the hot-path classes by these names
(exp_runners/predatorprey/runner_pp_commDP.py:22-28, runner_co_commDP.py same block):

    from envs import PredatorPreyWrapper                      # / CoverageWrapper
    from com_marl.torch.policies import CommCategoricalMLPPolicy
    from com_marl.torch.baselines import CommBaseCritic
    from com_marl.torch.algos import CentralizedMAPPO
    from com_marl.sampler import CentralizedMAOnPolicyVectorizedSampler
    from eval_pp import eval_model                            # / eval_co  (greedy evaluation, §8f-1)

``install()`` registers module objects under exactly those names (and nothing else of the
reference: experiment runner, logging, snapshotting stay the reference's own).  A maintainer who
wants the MI355X path puts ``import com_marl_amd.dropin; com_marl_amd.dropin.install()`` before those
imports and passes ``n_envs`` / ``device`` to the env wrapper.
"""
import sys
import types


def install(force=False):
    from . import algos, envs, evaluate, nets, sampler

    def mod(name, **attrs):
        if name in sys.modules and not force and not getattr(sys.modules[name], "_commarl_amd", False):
            raise RuntimeError(f"{name} is already imported from elsewhere; call install() before the reference imports "
                               "or pass force=True")
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m._commarl_amd = True
        m.__path__ = []
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, m)
        return m

    mod("envs", PredatorPreyWrapper=envs.PredatorPreyWrapper, CoverageWrapper=envs.CoverageWrapper)
    mod("envs.predatorprey_wrapper", PredatorPreyWrapper=envs.PredatorPreyWrapper)
    mod("envs.coverage_wrapper", CoverageWrapper=envs.CoverageWrapper)
    mod("com_marl")
    mod("com_marl.torch")
    mod("com_marl.torch.policies", CommCategoricalMLPPolicy=nets.CommCategoricalMLPPolicy,
        DecCategoricalMLPPolicy=nets.DecCategoricalMLPPolicy,                           # runner_pp_obsDP.py:28
        CentralizedCategoricalMLPPolicy=nets.CentralizedCategoricalMLPPolicy)           # runner_pp_cent.py:28
    mod("com_marl.torch.baselines", CommBaseCritic=nets.CommBaseCritic, GaussianMLPBaseline=nets.GaussianMLPBaseline)
    mod("com_marl.torch.modules", CommBaseNet=nets.CommBaseNet, AttentionModule=nets.AttentionModule,
        GraphConvolutionModule=nets.GraphConvolutionModule, GaussianMLPModule=nets.GaussianMLPModule)
    mod("com_marl.torch.algos", CentralizedMAPPO=algos.CentralizedMAPPO)
    mod("com_marl.sampler", CentralizedMAOnPolicyVectorizedSampler=sampler.CentralizedMAOnPolicyVectorizedSampler)
    mod("eval_pp", eval_model=evaluate.eval_model, VECTORS=evaluate.VECTORS)          # exp_runners/predatorprey/eval_pp.py:9
    mod("eval_co", eval_model=evaluate.eval_model_co, VECTORS=evaluate.VECTORS)       # exp_runners/coverage/eval_co.py:9
    return sorted(k for k, v in sys.modules.items() if getattr(v, "_commarl_amd", False))


class SimpleRunner:
    """The slice of LocalRunner / LocalRunnerWrapper the algo touches (garage/experiment/local_runner.py:
    181-232,373-457; com_marl/experiment/local_runner_wrapper.py:28-59): setup(), train(), step_epochs(),
    obtain_samples(), step_itr / step_path / total_env_steps.  Logging and snapshots are out of scope."""

    def __init__(self):
        self.step_itr, self.step_path, self.total_env_steps = 0, None, 0
        self.flag = [0]
        self.hybrid_mode = False
        self.history = []

    def setup(self, algo, env, sampler_cls=None, sampler_args=None, hybrid_mode=False, devices=None, flag=None):
        self._algo, self._env = algo, env
        self.flag = flag if flag is not None else [0]
        sampler_cls = sampler_cls or algo.sampler_cls
        self._sampler = sampler_cls(algo, env, **(sampler_args or {}))

    def obtain_samples(self, itr, batch_size=None):
        paths = self._sampler.obtain_samples(itr, batch_size or self._batch_size)
        self.total_env_steps += sum(len(p["rewards"]) for p in paths) if not hasattr(paths, "length") \
            else int(paths.length.sum().item())
        return paths

    def step_epochs(self):
        self._sampler.start_worker()
        try:
            for epoch in range(self._n_epochs):
                yield epoch
                self.history.append(dict(self._algo.stats, TotalEnvSteps=self.total_env_steps))
        finally:
            self._sampler.shutdown_worker()

    def train(self, n_epochs, batch_size):
        self._n_epochs, self._batch_size = n_epochs, batch_size
        return self._algo.train(self)
