"""com_marl_amd - MI355X-native batched rollout + GNN-PPO hot path of Com-MARL.

Host-side mirror of the reference's operator interface (same class names, ctor kwargs and
argument meaning) over the C ABI of libcommarl_hip.so (include/commarl.h).  The HIP library
is mandatory: nothing here falls back to a CPU implementation.
"""
from . import _lib
from ._lib import CommarlError, lib  # noqa: F401

__all__ = ["lib", "CommarlError", "envs", "nets", "sampler", "algos"]


def __getattr__(name):
    import importlib
    if name in ("envs", "nets", "sampler", "algos", "dist", "dropin"):
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
