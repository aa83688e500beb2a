"""Host-only loader for the Python reference (TEST INFRASTRUCTURE, never shipped, never on the GPU box).

Imports the reference's hot-path leaf modules from ``/root/reference`` so that
``oracle/gen_golden.py`` can emit golden vectors.
The reference depends on packages that are not installed here (gym, akro, dowel,
pyprind, ...) and two of its ``__init__`` files do not import as shipped, so this
loader (1) injects minimal stand-in modules for those third-party names, (2)
registers the reference's packages as bare namespace objects so their
``__init__.py`` is never executed, then (3) imports only the leaf files on the path.

Nothing here is copied from the reference; it only arranges ``sys.modules``.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("COMMARL_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _bare_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def _install_third_party_stubs():
    import numpy as np

    # ---- gym ---------------------------------------------------------------
    class Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = shape
            self.dtype = dtype

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low)
            self.high = np.asarray(high)
            super().__init__(self.low.shape, dtype)

        @property
        def flat_dim(self):
            return int(np.prod(self.low.shape))

    class Discrete(Space):
        def __init__(self, n):
            self.n = n
            super().__init__((), np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

    class Env:
        metadata = {}

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

    space_mod = _mod("gym.spaces.space", Space=Space)
    spaces = _mod("gym.spaces", Space=Space, Box=Box, Discrete=Discrete, space=space_mod)

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    def hash_seed(seed=None, max_bytes=8):
        return int(seed or 0)

    seeding = _mod("gym.utils.seeding", np_random=np_random, hash_seed=hash_seed)
    gutils = _mod("gym.utils", seeding=seeding)

    class _Registry:
        def all(self):
            return []

    registration = _mod("gym.envs.registration", register=lambda *a, **k: None)
    genvs = _mod("gym.envs", registry=_Registry(), registration=registration)
    _mod("gym", Env=Env, Wrapper=Wrapper, spaces=spaces, utils=gutils, envs=genvs)

    # ---- akro --------------------------------------------------------------
    def from_gym(space):
        return space

    _mod("akro", Box=Box, Discrete=Discrete, from_gym=from_gym)

    # ---- dowel -------------------------------------------------------------
    class _Logger:
        def log(self, *a, **k):
            pass

        def add_output(self, *a, **k):
            pass

        def has_output_type(self, *a, **k):
            return True

        def dump_all(self, *a, **k):
            pass

        def prefix(self, *_a):
            import contextlib
            return contextlib.nullcontext()

    class _Tabular:
        def __init__(self):
            self.rows = {}

        def record(self, k, v):
            self.rows[k] = v

        def prefix(self, *_a):
            import contextlib
            return contextlib.nullcontext()

        def clear(self):
            self.rows = {}

    _mod("dowel", logger=_Logger(), tabular=_Tabular(), StdOutput=object)

    # ---- misc --------------------------------------------------------------
    class ProgBar:
        active = False

        def __init__(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

        def stop(self):
            pass

    _mod("pyprind", ProgBar=ProgBar)
    _mod("send2trash", send2trash=lambda *a, **k: None)
    _mod("pynvml")


def load_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF} (it only exists in the build container)")
    if "com_marl" in sys.modules and hasattr(sys.modules["com_marl"], "_commarl_ref_ns"):
        return sys.modules["com_marl"]._commarl_ref_ns
    _install_third_party_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)

    j = os.path.join
    for name, rel in [
        ("garage", "garage"), ("garage.torch", "garage/torch"),
        ("garage.torch.modules", "garage/torch/modules"), ("garage.torch.algos", "garage/torch/algos"),
        ("garage.misc", "garage/misc"), ("garage.np", "garage/np"),
        ("garage.np.baselines", "garage/np/baselines"), ("garage.np.algos", "garage/np/algos"),
        ("garage.sampler", "garage/sampler"), ("garage.experiment", "garage/experiment"),
        ("garage.tf", "garage/tf"),
        ("com_marl", "com_marl"), ("com_marl.torch", "com_marl/torch"),
        ("com_marl.torch.modules", "com_marl/torch/modules"),
        ("com_marl.torch.policies", "com_marl/torch/policies"),
        ("com_marl.torch.baselines", "com_marl/torch/baselines"),
        ("com_marl.torch.algos", "com_marl/torch/algos"),
        ("com_marl.np", "com_marl/np"), ("com_marl.np.algos", "com_marl/np/algos"),
        ("com_marl.sampler", "com_marl/sampler"),
        ("custom_implement", "custom_implement"),
        ("envs", "envs"), ("envs.ma_gym", "envs/ma_gym"), ("envs.ma_gym.envs", "envs/ma_gym/envs"),
    ]:
        _bare_pkg(name, j(REF, rel))
    _mod("garage.tf.samplers")
    _mod("envs.utils", standard_eval=None)
    sys.modules["envs"].utils = sys.modules["envs.utils"]

    imp = importlib.import_module
    ns = types.SimpleNamespace()

    # garage leaves on the path
    ns.mlp_module = imp("garage.torch.modules.mlp_module")
    sys.modules["garage.torch.modules"].MLPModule = ns.mlp_module.MLPModule
    gutils = imp("garage.torch.algos._utils")
    for n in ("_Default", "compute_advantages", "filter_valids", "make_optimizer", "pad_to_last"):
        setattr(sys.modules["garage.torch.algos"], n, getattr(gutils, n))
    ns.compute_advantages = gutils.compute_advantages
    ns.filter_valids = gutils.filter_valids
    ns.pad_to_last = gutils.pad_to_last
    ns.tensor_utils = imp("garage.misc.tensor_utils")
    sys.modules["garage.misc"].tensor_utils = ns.tensor_utils

    # env + comm model
    ns.env_communication = imp("custom_implement.env_communication")
    ns.ge = imp("custom_implement.gilbert_elliot_loss_model")
    ns.pp_wrapper = imp("envs.predatorprey_wrapper")
    ns.co_wrapper = imp("envs.coverage_wrapper")
    ns.PredatorPreyWrapper = ns.pp_wrapper.PredatorPreyWrapper
    ns.CoverageWrapper = ns.co_wrapper.CoverageWrapper
    ns.pp_module = sys.modules["envs.ma_gym.envs.predator_prey.predator_prey"]
    ns.co_module = sys.modules["envs.ma_gym.envs.coverage.coverage"]

    # GNN modules, policy, critic
    cat = imp("com_marl.torch.modules.categorical_mlp_module")
    base = imp("com_marl.torch.modules.comm_base_net")
    sys.modules["com_marl.torch.modules"].CategoricalMLPModule = cat.CategoricalMLPModule
    sys.modules["com_marl.torch.modules"].CommBaseNet = base.CommBaseNet
    ns.CommBaseNet = base.CommBaseNet
    ns.CommCategoricalMLPPolicy = imp(
        "com_marl.torch.policies.comm_categorical_mlp_policy").CommCategoricalMLPPolicy
    ns.CommBaseCritic = imp("com_marl.torch.baselines.comm_base_critic").CommBaseCritic
    ns.pad_one_to_last = imp("com_marl.torch.algos.utils").pad_one_to_last
    ns.Adam = imp("com_marl.torch.algos.my_optimizer.adam").Adam
    sys.modules["com_marl"]._commarl_ref_ns = ns
    return ns


def load_reference_ppo(ns=None):
    """Additionally import the PPO algo + sampler (needs a few more stand-ins)."""
    ns = ns or load_reference()
    if hasattr(ns, "CentralizedMAPPO"):
        return ns
    imp = importlib.import_module
    g = sys.modules["garage"]
    g.log_performance = None
    g.TrajectoryBatch = None

    class LinearFeatureBaseline:  # dummy: only used in isinstance() checks
        pass

    sys.modules["garage.np.baselines"].LinearFeatureBaseline = LinearFeatureBaseline
    tu = imp("garage.torch.utils")
    sys.modules["garage.torch"].utils = tu
    for n in dir(tu):
        if not n.startswith("_"):
            setattr(sys.modules["garage.torch"], n, getattr(tu, n))
    sys.modules["garage.tf.samplers"].BatchSampler = object
    try:
        imp("garage.experiment.deterministic")
        sys.modules["garage.experiment"].deterministic = sys.modules["garage.experiment.deterministic"]
        smp = imp("com_marl.sampler.centralized_ma_on_policy_vectorized_sampler")
        sys.modules["com_marl.sampler"].CentralizedMAOnPolicyVectorizedSampler = \
            smp.CentralizedMAOnPolicyVectorizedSampler
        ns.ReferenceSampler = smp.CentralizedMAOnPolicyVectorizedSampler
        ns.VecEnvExecutor = sys.modules["garage.sampler.vec_env_executor"].VecEnvExecutor
        pol = imp("com_marl.np.algos.ma_batch_polopt")
        sys.modules["com_marl.np.algos"].MABatchPolopt = pol.MABatchPolopt
        ns.CentralizedMAPPO = imp("com_marl.torch.algos.centralized_ma_ppo").CentralizedMAPPO
    except Exception as e:  # pragma: no cover - diagnostic
        import traceback
        traceback.print_exc()
        ns.ppo_import_error = e
    return ns


def load_reference_variants(ns=None):
    """Additionally import the Obs-DP / CENT policies and the plain Gaussian baseline."""
    ns = ns or load_reference()
    if hasattr(ns, "DecCategoricalMLPPolicy"):
        return ns
    imp = importlib.import_module
    _bare_pkg("garage.torch.policies", os.path.join(REF, "garage/torch/policies"))
    sys.modules["garage.torch.policies"].Policy = imp("garage.torch.policies.base").Policy
    gm = imp("garage.torch.modules.gaussian_mlp_module")
    sys.modules["garage.torch.modules"].GaussianMLPModule = gm.GaussianMLPModule
    ns.DecCategoricalMLPPolicy = imp("com_marl.torch.policies.dec_categorical_mlp_policy").DecCategoricalMLPPolicy
    ns.CentralizedCategoricalMLPPolicy = imp(
        "com_marl.torch.policies.centralized_categorical_mlp_policy").CentralizedCategoricalMLPPolicy
    ns.GaussianMLPBaseline = imp("com_marl.torch.baselines.gaussian_mlp_baseline").GaussianMLPBaseline
    return ns


def make_env_spec(obs_dim_total, n_actions=5):
    """EnvSpec stand-in: the nets read only observation_space.flat_dim / action_space.n."""
    import numpy as np
    akro = sys.modules["akro"]
    return types.SimpleNamespace(
        observation_space=akro.Box(np.zeros(obs_dim_total), np.ones(obs_dim_total)),
        action_space=akro.Discrete(n_actions))
