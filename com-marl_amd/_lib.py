"""ctypes binding of libcommarl_hip.so (the C ABI declared in include/commarl.h).

There is no fallback: if the HIP library is missing or does not load, importing the
binding raises.  Nothing in this package routes through oracle/ or a CPU path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# COMMARL_LIB selects another build of the same ABI (e.g. the -DCM_BOUNDS checked build)
LIB_PATH = os.environ.get("COMMARL_LIB") or os.path.join(HERE, "libcommarl_hip.so")

CM_PP, CM_CO = 0, 1
PACK_F32, PACK_F16, PACK_WAVE, PACK_CHECK, PACK_ALL = 1, 2, 4, 8, 15      # cm_*_pack_sections (include/commarl.h)
CHANNELS = {"FC": 0, "FL": 1, "IID": 2, "GE": 3}
RNG_PHILOX, RNG_TAPE = 0, 1


class EnvCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "scenario", "n_envs", "n_agents", "n_preys", "grid", "rsen", "load", "max_steps", "max_path_length",
        "n_hops", "rcom", "channel", "obst_hard", "add_clock", "rng_mode", "env_id_offset")] + [
        ("ploss", C.c_float), ("pgb", C.c_float), ("pbg", C.c_float), ("ge_flags", C.c_int32)] + [
        (n, C.c_double) for n in ("capture_reward", "step_cost", "move_cost", "penalty", "lazy_penalty",
                                  "revisit_penalty", "final_reward")] + [("seed", C.c_uint64)]


class RngTape(C.Structure):
    _fields_ = [("prey", C.c_void_p), ("spawn", C.c_void_p), ("spawn_cap", C.c_int32), ("_pad", C.c_int32),
                ("iid_u", C.c_void_p), ("ge_u", C.c_void_p), ("ge_init_u", C.c_void_p)]


class StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "reward", "reward_f64", "done", "details", "dist_adj", "channels",
                                          "prey_alive", "success", "path_len")]


class EnvState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("agent_pos", "prey_pos", "prey_alive", "visited", "step_count",
                                          "total_capture", "success", "ge_state", "rng_step")]


class PolicyWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "n_agents", "n_hops", "enc_hidden", "emb", "h1", "h2", "h3", "n_act",
                                         "no_residual")] + [
        (n, C.c_void_p) for n in ("enc_w1t", "enc_b1", "enc_w2t", "enc_b2", "attn_wt", "gcn_w", "gcn_b", "hd_w1t",
                                  "hd_b1", "hd_w2t", "hd_b2", "hd_w3t", "hd_b3", "hd_w4t", "hd_b4", "mfma_pack")]


class CriticWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "n_agents", "n_hops", "enc_hidden", "emb", "dec_hidden",
                                         "no_residual", "_pad")] + [
        (n, C.c_void_p) for n in ("enc_w1t", "enc_b1", "enc_w2t", "enc_b2", "attn_wt", "gcn_w", "gcn_b", "dec_w1t",
                                  "dec_b1", "dec_w2t", "dec_b2", "mfma_pack")]


class ChunkStrides(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("obs", "actions", "probs", "attn", "reward", "reward_f64", "done", "details",
                                         "dist_adj", "channels", "prey_alive", "success", "path_len")]


MLP_MAX_LAYERS = 6


class FwdSaves(C.Structure):
    """cm_fwd_saves: device pointers of the activations the training forward stores (include/commarl.h)."""
    _fields_ = [("a1", C.c_void_p), ("e", C.c_void_p), ("q", C.c_void_p), ("hw", C.c_void_p * 4), ("h", C.c_void_p * 4),
                ("x1", C.c_void_p), ("x2", C.c_void_p), ("x3", C.c_void_p), ("out", C.c_void_p), ("probs", C.c_void_p)]


class MlpWeights(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("n_layers", C.c_int32), ("out_dim", C.c_int32 * MLP_MAX_LAYERS),
                ("tanh_mask", C.c_int32), ("_pad", C.c_int32), ("wt", C.c_void_p * MLP_MAX_LAYERS),
                ("b", C.c_void_p * MLP_MAX_LAYERS), ("mfma_pack", C.c_void_p)]


# every symbol include/commarl.h declares, with its signature
_SIGNATURES = {
    "cm_abi_version": (C.c_int, []),
    "cm_last_error": (C.c_char_p, []),
    "cm_env_create": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(C.c_void_p)]),
    "cm_env_destroy": (C.c_int, [C.c_void_p]),
    "cm_env_obs_dim": (C.c_int, [C.c_void_p]),
    "cm_env_n_empty_cells": (C.c_int, [C.c_void_p]),
    "cm_env_adj_is_const": (C.c_int, [C.c_void_p]),
    "cm_env_channels_are_const": (C.c_int, [C.c_void_p]),
    "cm_env_fill_constants": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_env_reset": (C.c_int, [C.c_void_p, C.POINTER(RngTape), C.POINTER(StepOut), C.c_void_p]),
    "cm_env_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RngTape), C.POINTER(StepOut), C.c_void_p]),
    "cm_env_status": (C.c_int, [C.c_void_p]),
    "cm_env_get_state": (C.c_int, [C.c_void_p, C.POINTER(EnvState)]),
    "cm_env_set_state": (C.c_int, [C.c_void_p, C.POINTER(EnvState)]),
    "cm_policy_forward": (C.c_int, [C.POINTER(PolicyWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_critic_forward": (C.c_int, [C.POINTER(CriticWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "cm_rollout_step": (C.c_int, [C.c_void_p, C.POINTER(PolicyWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(RngTape), C.POINTER(StepOut), C.c_void_p]),
    "cm_rollout_chunk": (C.c_int, [C.c_void_p, C.POINTER(PolicyWeights), C.c_int32, C.POINTER(ChunkStrides), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(StepOut), C.c_void_p]),
    "cm_rollout_chunk_tail": (C.c_int, [C.c_void_p, C.POINTER(PolicyWeights), C.c_int32, C.POINTER(ChunkStrides), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(StepOut), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "cm_policy_pack_bytes": (C.c_size_t, [C.POINTER(PolicyWeights)]),
    "cm_policy_pack": (C.c_int, [C.POINTER(PolicyWeights), C.c_void_p, C.c_void_p]),
    "cm_policy_pack_sections": (C.c_int, [C.POINTER(PolicyWeights), C.c_void_p, C.c_int32, C.c_void_p]),
    "cm_critic_pack_sections": (C.c_int, [C.POINTER(CriticWeights), C.c_void_p, C.c_int32, C.c_void_p]),
    "cm_critic_pack_bytes": (C.c_size_t, [C.POINTER(CriticWeights)]),
    "cm_critic_pack": (C.c_int, [C.POINTER(CriticWeights), C.c_void_p, C.c_void_p]),
    "cm_mlp_policy_forward": (C.c_int, [C.POINTER(MlpWeights), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_mlp_pack_bytes": (C.c_size_t, [C.POINTER(MlpWeights)]),
    "cm_mlp_pack": (C.c_int, [C.POINTER(MlpWeights), C.c_void_p, C.c_void_p]),
    "cm_mlp_value_forward": (C.c_int, [C.POINTER(MlpWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_masked_agg_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_masked_agg_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "cm_masked_agg_backward_r": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p]),
    "cm_attention_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_attention_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_linear_wgrad": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "cm_policy_forward_saved_wave": (C.c_int, [C.POINTER(PolicyWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(FwdSaves), C.c_void_p]),
    "cm_critic_forward_saved_wave": (C.c_int, [C.POINTER(CriticWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.POINTER(FwdSaves), C.c_void_p]),
    "cm_policy_forward_saved": (C.c_int, [C.POINTER(PolicyWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(FwdSaves), C.c_void_p]),
    "cm_critic_forward_saved": (C.c_int, [C.POINTER(CriticWeights), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(FwdSaves), C.c_void_p]),
    "cm_env_agent_condition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_env_agent_fault": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p]),
    "cm_comm_delays": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p]),
    "cm_chunk_tail": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cm_linear_act_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_void_p]),
    "cm_linear_act_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_encoder_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_multi_copy_t": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_multi_adam_step": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "cm_gauss_nll_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "cm_gauss_nll_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_adam_bias_corrections": (None, [C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "cm_multi_adam_step_dev": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p]),
    "cm_ppo_surrogate": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cm_discount_returns": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                      C.c_void_p]),
    "cm_gae": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32,
                         C.c_float, C.c_void_p, C.c_void_p]),
}
EXPORTED = tuple(_SIGNATURES)

_lib = None


class CommarlError(RuntimeError):
    pass


def lib():
    """The loaded library.  Fails loudly when the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CommarlError(
                f"{LIB_PATH} not found: build it with `make -C com-marl_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.cm_abi_version() != 3:
            raise CommarlError("libcommarl_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cm_last_error()
        raise CommarlError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Raw device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes contiguous buffers"
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


# ---- hipGraph captures must not contain a hipFree: env handles released while a capture is open (a garbage-collected
# GridEnvBatch -> cm_env_destroy -> hipFree invalidated a capture in round 2) are parked here and destroyed afterwards ----
_capture_depth = 0
_deferred_env_handles = []


def destroy_env(handle):
    """cm_env_destroy now, or after the innermost open capture_guard() has closed."""
    if _capture_depth > 0:
        _deferred_env_handles.append(handle)
        return
    lib().cm_env_destroy(handle)


class capture_guard:
    """``with capture_guard(): <stream capture>``: keeps the cyclic collector off while the capture is open and defers every
    env-handle destruction requested meanwhile (a refcount reaching zero) to the exit.  (No gc.collect() here: a full
    collection costs ~50 ms in a process that has torch loaded, and a sampler captures dozens of span graphs.)"""

    def __enter__(self):
        global _capture_depth
        import gc
        self._gc_was_on = gc.isenabled()
        gc.disable()
        _capture_depth += 1
        return self

    def __exit__(self, *exc):
        global _capture_depth
        import gc
        _capture_depth -= 1
        if self._gc_was_on:
            gc.enable()
        if _capture_depth == 0:
            while _deferred_env_handles:
                lib().cm_env_destroy(_deferred_env_handles.pop())
        return False
