"""cm_masked_agg_backward (teams of 4) at S envs, with / without the bias gradient, for COMMARL_AGG4_BLOCKS (read once per process).
   python tools/agg_bwd_sweep.py S [N]      (N != 4: the MFMA N x N kernel, with an adjacency mask)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

L = importlib.import_module("com_marl_amd._lib")
S = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
attn = torch.softmax(torch.randn(S, N, N, device=dev), -1)
adj = (torch.rand(S, N, N, device=dev) < 0.5).float() if N != 4 else None
hw = torch.randn(S, N, 64, device=dev)
out = torch.tanh(torch.randn(S, N, 64, device=dev))
d_out = torch.randn(S, N, 64, device=dev)
d_attn = torch.empty(S, N, N, device=dev)
d_hw = torch.empty_like(hw)
d_bias = torch.zeros(64, device=dev)
for bias in (True, False):
    def run():
        L.check(L.lib().cm_masked_agg_backward(S, N, 64, L.ptr(attn), L.ptr(adj), None, 0, L.ptr(hw), L.ptr(out), None, L.ptr(d_out), L.ptr(d_attn),
                                               L.ptr(d_hw), L.ptr(d_bias) if bias else None, L.current_stream()), "agg_bwd")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"S={S} N={N} bias={bias} blocks={os.environ.get('COMMARL_AGG4_BLOCKS', 'default')}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
