"""Time of the large-team aggregation / attention backward kernels at a training batch (default N = 72, 68 k envs), with
COMMARL_NXN_STOP=k returning after phase k of the aggregation kernel (1 staging, 2 normalisation, 3 d_hw)."""
import os, sys
sys.path.insert(0, '.')
import torch
from com_marl_amd import _lib as L
N = int(os.environ.get("N", 72)); S = int(os.environ.get("S", 68000))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
attn = torch.softmax(torch.randn(S, N, N, device=dev), -1)
adj = (torch.rand(S, N, N, device=dev) < 0.7).float()
ch = (torch.rand(S, 2, N, N, device=dev) < 0.7).float()
hw = torch.randn(S, N, 64, device=dev); out = torch.tanh(torch.randn(S, N, 64, device=dev)); e = torch.randn(S, N, 64, device=dev) * 0.1
dout = torch.randn(S, N, 64, device=dev)
da = torch.empty_like(attn); dhw = torch.empty_like(hw); db = torch.zeros(64, device=dev)
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
agg = t(lambda: lib.cm_masked_agg_backward(S, N, 64, attn.data_ptr(), adj.data_ptr(), ch.data_ptr(), 2 * N * N, hw.data_ptr(), out.data_ptr(),
                                           e.data_ptr(), dout.data_ptr(), da.data_ptr(), dhw.data_ptr(), db.data_ptr(), st()))
q = torch.randn(S, N, 64, device=dev); dq = torch.empty_like(q); de = torch.empty_like(q); dm = torch.randn(S, N, N, device=dev)
att = t(lambda: lib.cm_attention_backward(S, N, 64, q.data_ptr(), e.data_ptr(), attn.data_ptr(), dm.data_ptr(), hw.data_ptr(), dout.data_ptr(),
                                          dq.data_ptr(), de.data_ptr(), st()))
print(f"N={N} S={S} stop={os.environ.get('COMMARL_NXN_STOP', '0')}: agg_bwd {agg:.2f} ms  attn_bwd {att:.2f} ms")
