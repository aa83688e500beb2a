"""cm_policy_forward_saved_wave against cm_policy_forward_saved, save by save:  python tools/wave_saves_check.py [envs] [hops]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

L = importlib.import_module("com_marl_amd._lib")
nets = importlib.import_module("com_marl_amd.nets")
E = importlib.import_module("com_marl_amd.envs")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 91
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, d, dev = 4, 21, torch.device("cuda:0")
spec = E.EnvSpec(E._Box(np.zeros(N * d), np.ones(N * d)), E._Discrete(5))
torch.manual_seed(2)
pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, n_gcn_layers=hops, device=dev)
obs = torch.rand(S * N, d, device=dev)
R = S * N
out = {}
for tag in ("wave", "tiled"):
    z = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=dev)   # noqa: E731
    t = dict(a1=z(R, 128), e=z(R, 64), q=z(R, 64), hw=[z(R, 64) for _ in range(hops)], h=[z(R, 64) for _ in range(hops)],
             x1=z(R, 128), x2=z(R, 64), x3=z(R, 32), out=z(R, 5))
    attn = z(S, N, N)
    sv = L.FwdSaves()
    sv.a1, sv.e, sv.q, sv.x1, sv.out, sv.x2, sv.x3 = (t[k].data_ptr() for k in ("a1", "e", "q", "x1", "out", "x2", "x3"))
    for l in range(hops):
        sv.hw[l], sv.h[l] = t["hw"][l].data_ptr(), t["h"][l].data_ptr()
    pol._train_fwd, pol._train_fwd_wave = True, tag == "wave"
    w = pol._weights_struct()
    fn = L.lib().cm_policy_forward_saved_wave if tag == "wave" else L.lib().cm_policy_forward_saved
    L.check(fn(C.byref(w), S, L.ptr(obs), None, None, L.ptr(attn), C.byref(sv), L.current_stream()), tag)
    torch.cuda.synchronize()
    pol._train_fwd = pol._train_fwd_wave = False
    flat = dict(attn=attn, **{k: v for k, v in t.items() if not isinstance(v, list)})
    for l in range(hops):
        flat[f"hw{l}"], flat[f"h{l}"] = t["hw"][l], t["h"][l]
    out[tag] = {k: v.cpu().numpy() for k, v in flat.items()}
for k in out["tiled"]:
    a, b = out["wave"][k], out["tiled"][k]
    bad = np.isnan(a).sum()
    print(f"{k:6s} max|diff| {np.nanmax(np.abs(a - b)):.3e}   nan in wave {bad}   max|tiled| {np.abs(b).max():.3f}", flush=True)
