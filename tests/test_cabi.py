"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/commarl.h declares (no compute calls without a GPU), and the product package never
imports the oracle."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "commarl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from com_marl_amd import _lib
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libcommarl_hip.so does not export {n}"
    assert set(names) == set(_lib.EXPORTED), "ctypes binding and header disagree"
    assert lib.cm_abi_version() == 3
    assert isinstance(lib.cm_last_error(), (bytes, type(None)))


def test_struct_layouts_match_header_sizes():
    import ctypes as C
    from com_marl_amd import _lib
    assert C.sizeof(_lib.EnvCfg) == 16 * 4 + 4 * 4 + 7 * 8 + 8
    assert C.sizeof(_lib.RngTape) == 6 * 8
    assert C.sizeof(_lib.StepOut) == 10 * 8
    assert C.sizeof(_lib.EnvState) == 9 * 8
    assert C.sizeof(_lib.PolicyWeights) == 10 * 4 + 16 * 8
    assert C.sizeof(_lib.CriticWeights) == 8 * 4 + 12 * 8
    assert C.sizeof(_lib.MlpWeights) == 2 * 4 + 6 * 4 + 2 * 4 + 13 * 8


def test_argument_errors_without_gpu():
    """Pure argument validation paths return error codes + text, no exception crosses the ABI."""
    import ctypes as C
    from com_marl_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.cm_env_create(None, C.byref(h)) == -1
    assert b"null" in lib.cm_last_error()
    cfg = _lib.EnvCfg()
    cfg.scenario = 7
    assert lib.cm_env_create(C.byref(cfg), C.byref(h)) == -1
    assert lib.cm_policy_forward(None, 4, None, None, None, None, 0, 0, 0, None, 0, None, None, None, None) == -1
    assert lib.cm_gae(0, 0, None, None, None, 0.99, 0.97, 0, 1e-8, None, None) == -1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no product file may import, load or link it."""
    pkg = os.path.join(ROOT, "com-marl_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libcm_oracle|cm_oracle\.h|cmo_[a-z_]+\(", re.M)
    n = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                n += 1
                assert not bad.search(open(os.path.join(dirpath, f)).read()), f"{f} touches the oracle"
    assert n >= 8


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from com_marl_amd import envs, CommarlError
    with pytest.raises(CommarlError):
        envs.GridEnvBatch("pp", dict(n_agents=4, n_preys=4, grid_size=10, Rsen=1, n_gcn_layers=2, trpl=0,
                                     max_env_steps=200), 2, device="cpu")


def test_graft_entry_build_runs():
    """The driver's build check: compiles (no-op when up to date), imports the package, resolves every symbol."""
    import __graft_entry__ as g
    g.build()


def test_bench_contract_helpers():
    """bench.py imports on a machine without a GPU and its byte / FLOP accounting matches SURVEY.md §8(d)."""
    import bench
    assert set(bench.CONFIGS) == {"pp_map10", "co_map20", "pp_map30", "co_map30"}
    c = bench.CONFIGS["pp_map10"]
    b_env, b_pol = bench.algorithmic_bytes(c, 21, adj_const=True, ch_const=True)
    assert (b_env, b_pol, b_env + b_pol) == (501, 496, 997)                 # config 2: 997 B per env-step
    c4 = bench.CONFIGS["pp_map30"]
    b_env, b_pol = bench.algorithmic_bytes(c4, 53, adj_const=False, ch_const=True)
    assert abs((b_env + b_pol) - 97.2e3) < 0.5e3                            # config 4: ~97.2 KB
    assert bench.policy_flops(c, 21) == 340224                              # 2N(128d + 39 621 + 192N) at d=21, N=4
    assert bench.host_cores() >= 1


def test_env_kernels_use_no_flat_or_scratch_addressing(tmp_path):
    """Static guard against the one GPU abort on record (round 1, gpurun_out/smoke.log: HSA_STATUS_ERROR_MEMORY_
    APERTURE_VIOLATION in env_kernel<0>, private_seg_size=88, group_seg_size=288): that fault class
    can only be raised by FLAT / SCRATCH instructions whose address lands in the LDS or scratch aperture beyond the
    wave's allocation - `ds_*` accesses out of range are dropped silently.  The env step therefore addresses LDS
    through integer offsets only (ds ops) and keeps no dynamically indexed local arrays (no scratch): pinned here on
    the gfx950 ISA of every env kernel instantiation."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "com-marl_amd", "csrc", "cm_env.hip")
    out = tmp_path / "cm_env.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-w", "-o", str(out), src])
    asm = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*env_kernel\S*)", asm)
    assert len(kernels) >= 8, kernels                         # PP / CO x 16 / 32 / 64 lanes + the two wide forms
    assert not re.search(r"^\s+(flat_(load|store|atomic)|scratch_)", asm, re.M)
    assert len(re.findall(r"^\s+ds_", asm, re.M)) > 1000
    for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)", asm):
        assert m.group(1) == "0"


def test_fused_step_kernels_use_no_flat_or_scratch_addressing(tmp_path):
    """The same guard for the fused rollout step (cm_fused.hip: the env body runs inside the policy's workgroup): every
    rollout_step_kernel instantiation - the kernels of the default rollout path - must request no private segment and
    contain no flat / scratch instruction.  (The opt-in persistent chunk kernels are not covered: the all-f32 large-team
    form spills two registers.)"""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "com-marl_amd", "csrc", "cm_fused.hip")
    out = tmp_path / "cm_fused.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-w", "-o", str(out), src])
    asm = out.read_text()
    seen = 0
    for blk in re.split(r"\n\s+- \.agpr_count:", asm)[1:]:                 # one metadata block per kernel
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "rollout_step_kernel" not in name:
            continue
        seen += 1
        assert re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1) == "0", name
        assert re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1) == "0", name
    assert seen >= 10, seen                                               # five shapes x two policy bodies (+ the full-workgroup builds)
    bodies = list(re.finditer(r"^(_ZN2cm19rollout_step_kernel\S+):[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.M | re.S))
    assert len(bodies) == seen                                            # (the label line carries a trailing "; @name" comment)
    for m in bodies:
        assert not re.search(r"^\s+(flat_(load|store|atomic)|scratch_)", m.group(2), re.M), m.group(1)


def test_wave_owned_rollout_kernels_use_no_flat_or_scratch_addressing(tmp_path):
    """The same guard for the default rollout path of teams of 4 (cm_rollout_w.hip: policy forward + sample + env step of a
    wave's four envs, for a whole chunk of steps per launch): every rollout_w_kernel instantiation must request no private
    segment, spill nothing and contain no flat / scratch instruction - with 256 VGPRs + ~120 AGPRs in use (one wave per SIMD,
    the 128 -> 64 head layer resident in registers) this is the kernel closest to the register limit."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "com-marl_amd", "csrc", "cm_rollout_w.hip")
    out = tmp_path / "cm_rollout_w.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-w", "-o", str(out), src])
    asm = out.read_text()
    seen = 0
    for blk in re.split(r"\n\s+- \.agpr_count:", asm)[1:]:                 # one metadata block per kernel
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "rollout_w_kernel" not in name:
            continue
        seen += 1
        assert re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1) == "0", name
        assert re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1) == "0", name
    assert seen == 18, seen     # 1 / 2 hops x (env prefetch on / off x full / ragged workgroups + the tape variant + carried full / ragged, generic and map10 shape)
    bodies = list(re.finditer(r"^(_ZN2cm16rollout_w_kernel\S+):[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.M | re.S))
    assert len(bodies) == seen                                            # (the label line carries a trailing "; @name" comment)
    for m in bodies:
        assert not re.search(r"^\s+(flat_(load|store|atomic)|scratch_)", m.group(2), re.M), m.group(1)
        assert len(re.findall(r"^\s+s_barrier", m.group(2), re.M)) == 1, "one workgroup barrier per launch (behind the weight staging)"
