"""Child-process cases of the -m gpu suite: things that need a FRESH process because the library reads the knob once
(COMMARL_ENV_LPE, COMMARL_ENV_WIDE), selects another build (COMMARL_LIB = the CM_BOUNDS debug library) or needs several
ranks (gloo, all on GPU 0).  tests/conftest.py starts them one after the other BEFORE the pytest process itself touches
the GPU and keeps their reports; the tests only assert on the reports.

    python -m tests.hip_child <case>        -> one JSON line on stdout: {"ok": true, ...}
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# case -> (environment overrides, description)
CASES = {
    "env_lpe32": (dict(COMMARL_ENV_LPE="32"), "32 lanes per env forced: env_kernel<*,32> at small batches"),
    "env_lpe64_small_team": (dict(COMMARL_ENV_LPE="64"), "one wave per env forced for teams of 4 / 6: wide and narrow"),
    "env_lpe16_mid_team": (dict(COMMARL_ENV_LPE="16"), "16 lanes per env forced for teams of 12 / 16"),
    "env_wide_off": (dict(COMMARL_ENV_WIDE="0"), "narrow one-wave-per-env kernels for large teams at small batches"),
    "env_small_off": (dict(COMMARL_ENV_SMALL="0"), "occupancy-tile walk kept for small PP teams (the A/B of pp_small_step)"),
    "debug_library_env_goldens": (dict(COMMARL_LIB=os.path.join(ROOT, "com-marl_amd", "libcommarl_hip_dbg.so")),
                                  "range-checked (-DCM_BOUNDS) build: env goldens + Philox lock-step"),
    "bench_two_rank_rehearsal": (dict(COMMARL_DIST_BACKEND="gloo"),
                                 "bench.py --gpus 2 started without a launcher (self-launch), two gloo ranks on GPU 0: weak and --scaling strong"),
    "two_rank_train_once": ({}, "CentralizedMAPPO.train_once on 2 gloo ranks (both on GPU 0) == 1 process on the union (2 clipped steps; 5 steps without the ratio clip)"),
}


def _lock(kw, steps, check_every=3):
    from tests import hip_adapters
    from tests.test_hip_env_parity import _lockstep
    return _lockstep(kw, steps=steps, hip=hip_adapters, check_every=check_every)


def env_lpe32():
    n = _lock(dict(scenario="co", n_envs=301, n_agents=24, grid=20, rsen=2, max_steps=400, max_path_length=9), 22)
    n += _lock(dict(scenario="pp", n_envs=203, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=9), 22)
    n += _lock(dict(scenario="pp", n_envs=130, n_agents=20, n_preys=17, grid=16, rsen=2, load=3, max_steps=9, rcom=4,
                    channel="IID", ploss=0.2), 22)
    # register-resident loops, 8-entity build (pp_small_step<32, 8>), capture rule of load 3 (neighbouring preys count)
    n += _lock(dict(scenario="pp", n_envs=150, n_agents=7, n_preys=8, grid=11, rsen=1, load=3, max_steps=9, rcom=3), 22)
    return dict(dones=n)


def env_small_off():
    n = _lock(dict(scenario="pp", n_envs=203, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=9), 22)
    n += _lock(dict(scenario="pp", n_envs=100, n_agents=5, n_preys=7, grid=12, rsen=2, load=3, max_steps=9, rcom=3), 22)
    return dict(dones=n)


def env_lpe64_small_team():
    n = _lock(dict(scenario="pp", n_envs=203, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=9), 22)   # wide
    n += _lock(dict(scenario="pp", n_envs=1603, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=9), 22)  # narrow
    n += _lock(dict(scenario="co", n_envs=50, n_agents=6, grid=10, rsen=1, max_steps=400, max_path_length=9,
                    channel="GE"), 22)
    return dict(dones=n)


def env_lpe16_mid_team():
    n = _lock(dict(scenario="pp", n_envs=300, n_agents=12, n_preys=10, grid=12, rsen=1, load=2, max_steps=9, rcom=3), 22)
    n += _lock(dict(scenario="co", n_envs=77, n_agents=16, grid=20, rsen=2, max_steps=400, max_path_length=9,
                    channel="IID", ploss=0.3), 22)
    return dict(dones=n)


def env_wide_off():
    n = _lock(dict(scenario="pp", n_envs=128, n_agents=72, n_preys=72, grid=30, rsen=2, load=4, max_steps=9), 22)
    n += _lock(dict(scenario="co", n_envs=96, n_agents=54, grid=30, rsen=2, max_steps=400, max_path_length=9,
                    channel="IID", ploss=0.3), 22)
    return dict(dones=n)


def debug_library_env_goldens():
    from com_marl_amd import _lib as L
    assert L.LIB_PATH.endswith("libcommarl_hip_dbg.so"), L.LIB_PATH
    from tests import hip_adapters
    from tests.test_oracle_golden import ENV_FIXTURES, replay
    for path in ENV_FIXTURES:
        replay(path, hip_adapters.HipEnv)
    n = _lock(dict(scenario="pp", n_envs=515, n_agents=4, n_preys=4, grid=10, rsen=1, load=2, max_steps=11), 25)
    n += _lock(dict(scenario="co", n_envs=64, n_agents=54, grid=30, rsen=2, max_steps=400, max_path_length=9,
                    channel="GE"), 20)
    n += _lock(dict(scenario="pp", n_envs=40, n_agents=72, n_preys=72, grid=30, rsen=2, load=4, max_steps=9), 20)
    return dict(fixtures=len(ENV_FIXTURES), dones=n, lib=os.path.basename(L.LIB_PATH))


def bench_two_rank_rehearsal():
    """`python bench.py --gpus 2` with no launcher: bench.py starts its two ranks itself (torch.distributed.run, before anything touched
    the GPU), they rendezvous over gloo on ONE card, run the sharded rollout + the train loop with the gradient all-reduce, and
    rank 0 prints the contract line - marked as a rehearsal, not an N-GPU measurement."""
    import subprocess
    out = {}
    for scaling in ("weak", "strong"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "4", "--envs", "512",
               "--no-cpu-baseline", "--scaling", scaling]
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, COMMARL_DIST_BACKEND="gloo"))
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        per_gpu = 512 if scaling == "weak" else 256
        assert d["rehearsal"] is True and d["ranks"] == 2 and d["n_gpus"] == 1 and d["backend"] == "gloo", d
        assert d["scaling"] == scaling and d["config"]["envs_per_gpu"] == per_gpu and d["config"]["total_envs"] == 2 * per_gpu
        assert d["value"] > 0 and d["steps"] == 24 and d["config"]["graphs"]["capture_in_timed_region"] is False
        tl = d["train_loop"]
        assert tl and "error" not in tl and tl["value"] > 0 and "all-reduce" in tl["schedule"], tl
        assert "cpu_baseline" not in d and "configs" not in d                   # single-GPU extras stay out of multi-rank lines
        out[scaling] = dict(value=d["value"], train_loop=tl["value"])
    return out


def two_rank_train_once():
    from tests.dist_train_child import run_two_rank_case
    return run_two_rank_case()


def main():
    case = sys.argv[1]
    import torch
    # device_count() does not initialise the GPU: the two-rank case starts its ranks before this process does
    assert torch.cuda.device_count() > 0, "child cases need the MI355X"
    out = dict(globals()[case]())
    out["ok"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
