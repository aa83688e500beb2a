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
import importlib.abc
import importlib.machinery
import sys
import types


def _natural_path(name):
    """Directories the import system would search for sub-modules of package `name` if this package did not stand
    in for it (the reference's own `com_marl/`, `com_marl/torch/`, `envs/` ... when its tree is on sys.path), found
    WITHOUT executing anything.  Our stand-in modules carry these as `__path__`, so everything of the reference that
    is not on the hot path keeps importing from the reference (`com_marl.experiment.local_runner_wrapper`,
    `com_marl.np.algos`, `envs.ma_gym` ...)."""
    parent, _, _leaf = name.rpartition(".")
    try:
        search = None
        if parent:
            pm = sys.modules.get(parent)
            search = list(getattr(pm, "__path__", [])) if pm is not None else []
            if not search:
                return []
        spec = importlib.machinery.PathFinder.find_spec(name, search)
        return list(spec.submodule_search_locations or []) if spec is not None else []
    except Exception:
        return []


def register_base_sampler(base_module=None):
    """Make the sampler a (virtual) subclass of garage's ``BaseSampler`` so that the reference's runner takes its
    ``BaseSampler`` branch: ``issubclass(sampler_cls, BaseSampler)`` in LocalRunner.make_sampler
    (garage/experiment/local_runner.py:181-189) and ``isinstance(self._sampler, BaseSampler)`` in
    LocalRunnerWrapper.obtain_samples (com_marl/experiment/local_runner_wrapper.py:41-47).  ``BaseSampler`` is an
    ``abc.ABC`` (garage/sampler/base.py:4,35), so ``register`` is all it takes.  Returns True when registered."""
    from .sampler import CentralizedMAOnPolicyVectorizedSampler
    m = base_module or sys.modules.get("garage.sampler.base")
    base = getattr(m, "BaseSampler", None)
    if base is None or not hasattr(base, "register"):
        return False
    base.register(CentralizedMAOnPolicyVectorizedSampler)
    return True


class _BaseSamplerHook(importlib.abc.MetaPathFinder):
    """Registers the sampler the moment ``garage.sampler.base`` is imported (install() may run before the runner's
    garage imports): delegates the lookup to the remaining finders and wraps the loader's exec_module."""
    TARGET = "garage.sampler.base"

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.TARGET:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is None or spec.loader is None:
                continue
            inner = spec.loader

            class _Loader(importlib.abc.Loader):
                def create_module(self, spec):
                    return inner.create_module(spec) if hasattr(inner, "create_module") else None

                def exec_module(self, module):
                    inner.exec_module(module)
                    register_base_sampler(module)
            spec.loader = _Loader()
            return spec
        return None


def install(force=False):
    from . import algos, envs, evaluate, nets, sampler

    def mod(name, **attrs):
        if name in sys.modules and not force and not getattr(sys.modules[name], "_commarl_amd", False):
            raise RuntimeError(f"{name} is already imported from elsewhere; call install() before the reference imports "
                               "or pass force=True")
        path = _natural_path(name)
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m._commarl_amd = True
        m.__path__ = path
        m.__package__ = name
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
    mod("eval_pp", eval_model=evaluate.eval_model, eval_simple=evaluate.eval_simple,        # exp_runners/predatorprey/
        VECTORS=evaluate.VECTORS)                                                             # eval_pp.py:9,107
    mod("eval_co", eval_model=evaluate.eval_model_co, eval_simple=evaluate.eval_simple_co,  # exp_runners/coverage/
        VECTORS=evaluate.VECTORS)                                                             # eval_co.py:9,104
    # the runner's BaseSampler gates (local_runner.py:181, local_runner_wrapper.py:41): now if garage is already
    # imported, otherwise the moment it is
    if not register_base_sampler() and not any(isinstance(f, _BaseSamplerHook) for f in sys.meta_path):
        sys.meta_path.insert(0, _BaseSamplerHook())
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
