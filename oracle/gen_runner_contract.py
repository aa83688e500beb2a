#!/usr/bin/env python3
"""Records what the reference's OWN runner does with a sampler and its paths (TEST INFRASTRUCTURE; runs in the build
container only, where /root/reference exists; writes tests/golden/runner_contract.json).

The reference's LocalRunnerWrapper (com_marl/experiment/local_runner_wrapper.py) on top of garage's LocalRunner
(garage/experiment/local_runner.py) is executed here, unmodified, for a two-epoch `train()` against recording
doubles: a sampler class that is NOT a real subclass of garage's BaseSampler but only `BaseSampler.register`-ed (the
mechanism com_marl_amd.dropin uses), an algo whose train() is the reference's epoch loop
(com_marl/np/algos/ma_batch_polopt.py:72-120), and a `paths` object that logs every operation applied to it.
The recording pins:
  * which branch `make_sampler` (local_runner.py:181-189) and `obtain_samples` (local_runner_wrapper.py:41-47) take
    for a registered class, and the constructor / obtain_samples call signatures they use;
  * every operation the runner applies to `paths` and to each path (iteration, p['rewards'], len) and the resulting
    `total_env_steps` arithmetic (local_runner_wrapper.py:49-57);
  * the sampler life-cycle calls (start_worker / shutdown_worker) around the epoch loop (local_runner.py:410-457).
tests/test_dropin.py replays it against com_marl_amd's sampler and PathBatch.
"""
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EVENTS = []


def ev(*a):
    EVENTS.append(list(a))


class RecPath:
    def __init__(self, i, n):
        self.i, self.n = i, n

    def __getitem__(self, k):
        ev("path.__getitem__", k)
        return [0.0] * self.n

    def __getattr__(self, k):
        ev("path.getattr", k)
        raise AttributeError(k)


class RecPaths:
    def __init__(self, lens):
        self.paths = [RecPath(i, n) for i, n in enumerate(lens)]

    def __iter__(self):
        ev("paths.__iter__")
        return iter(self.paths)

    def __len__(self):
        ev("paths.__len__")
        return len(self.paths)

    def __getitem__(self, i):
        ev("paths.__getitem__", repr(i))
        return self.paths[i]

    def __getattr__(self, k):
        ev("paths.getattr", k)
        raise AttributeError(k)


def main():
    from oracle import ref_loader as R
    import importlib
    R.load_reference_ppo()
    lr = importlib.import_module("garage.experiment.local_runner")
    sys.modules["garage.experiment"].LocalRunner = lr.LocalRunner
    wrap = importlib.import_module("com_marl.experiment.local_runner_wrapper")
    base = sys.modules["garage.sampler.base"]

    class RecSampler:                                   # plain class: a virtual subclass only
        def __init__(self, algo, env, **kw):
            ev("sampler.__init__", sorted(kw.items()), type(algo).__name__, type(env).__name__)
            self.lens = [[7, 5, 9], [4, 6]]

        def start_worker(self):
            ev("sampler.start_worker")

        def shutdown_worker(self):
            ev("sampler.shutdown_worker")

        def obtain_samples(self, itr, batch_size=None, whole_paths=True, **kw):
            ev("sampler.obtain_samples", itr, batch_size, whole_paths, sorted(kw))
            return RecPaths(self.lens[itr % 2])

        @classmethod
        def from_worker_factory(cls, *a, **k):          # the branch a non-BaseSampler class falls into
            ev("sampler.from_worker_factory")
            raise RuntimeError("not a BaseSampler: the runner took the worker-factory branch")

    assert not issubclass(RecSampler, base.BaseSampler)
    base.BaseSampler.register(RecSampler)

    class Pol:
        centralized = True
        _n_agents = 4
        latest_feature = None
        gcn_latest_feature = None

    class Algo:
        max_path_length = 9
        sampler_cls = RecSampler

        def __init__(self):
            self.policy, self._old_policy = Pol(), Pol()
            self.n_samples = 1

        def train(self, runner):                         # ma_batch_polopt.py:72-120 without the eval hook
            last = None
            for _ in runner.step_epochs():
                for _ in range(self.n_samples):
                    runner.step_path = runner.obtain_samples(runner.step_itr)
                    last = 0.0
                    runner.step_itr += 1
            return last

    class Env:
        n_agents = 4

    tmp = tempfile.mkdtemp()
    cfg = types.SimpleNamespace(snapshot_dir=tmp, snapshot_mode="none", snapshot_gap=1)
    runner = wrap.LocalRunnerWrapper(cfg, eval=False, save_env=False)
    algo = Algo()
    runner.setup(algo, Env(), sampler_cls=RecSampler, sampler_args={"n_envs": 3})
    ev("runner.setup done", type(runner._sampler).__name__, bool(isinstance(runner._sampler, base.BaseSampler)))
    steps = []
    orig = runner.obtain_samples

    def spy(itr, batch_size=None):
        before = runner._stats.total_env_steps
        out = orig(itr, batch_size)
        steps.append(runner._stats.total_env_steps - before)
        ev("runner.total_env_steps +=", steps[-1])
        return out
    runner.obtain_samples = spy
    try:
        runner.train(n_epochs=2, batch_size=48)
        ev("runner.train done")
    except Exception as e:                               # logging / snapshot plumbing past the hot path may not run here
        ev("runner.train stopped", type(e).__name__, str(e)[:200])
    out = dict(
        source="reference LocalRunnerWrapper.train (2 epochs) against recording doubles; see oracle/gen_runner_contract.py",
        registered_via="BaseSampler.register(cls)", path_lengths=[[7, 5, 9], [4, 6]], batch_size=48,
        total_env_steps_increments=steps, events=EVENTS)
    dst = os.path.join(ROOT, "tests", "golden", "runner_contract.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)
    for e in EVENTS:
        print("  ", e)


if __name__ == "__main__":
    main()
