import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- child-process cases of the -m gpu suite (tests/hip_child.py) ---------------------------------------------------
# Env knobs read once per process, the CM_BOUNDS debug library and the two-rank update need FRESH processes.  They are
# started here, one after the other, at session start - i.e. before this pytest process has touched the GPU (a process
# that has initialised the GPU must not exec another program on this pool) - and the tests assert on the kept reports.
_CHILD = {}


def _run_child(case):
    import json
    import subprocess
    from tests.hip_child import CASES
    env = dict(os.environ)
    env.update(CASES[case][0])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run([sys.executable, "-m", "tests.hip_child", case], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=900)
        rc, out, err = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -9, e.stdout or "", (e.stderr or "") + "\n[timeout]"
    report = {}
    for line in reversed((out or "").strip().splitlines()):
        try:
            report = json.loads(line)
            break
        except ValueError:
            continue
    return dict(rc=rc, report=report, tail=((out or "")[-1500:] + "\n" + (err or "")[-3000:]))


def _gpu_selected(config):
    expr = (config.getoption("-m") or "").strip()
    return "gpu" in expr and "not gpu" not in expr


def pytest_sessionstart(session):
    if not _gpu_selected(session.config) or session.config.getoption("--collect-only", False):
        return
    if os.environ.get("COMMARL_SKIP_CHILD_CASES"):
        return
    import torch
    if torch.cuda.device_count() == 0:          # (does not initialise the GPU)
        return
    from tests.hip_child import CASES
    for case in CASES:
        _CHILD[case] = _run_child(case)


def child_result(case):
    if case not in _CHILD:
        import torch
        if torch.cuda.is_initialized():
            pytest.fail(f"child case '{case}' did not run at session start (run the suite as `pytest tests -m gpu`); "
                        "it cannot be started now: this process has initialised the GPU")
        _CHILD[case] = _run_child(case)
    return _CHILD[case]
