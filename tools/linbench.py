"""Times of the fused dense-layer kernels of the PPO update (cm_linear_act_forward / backward) against the torch ops they
replace, at the update's row count: R = 1.1 M agent rows."""
import os, sys
sys.path.insert(0, '.')
import torch
from com_marl_amd import _lib as L
R = int(os.environ.get("ROWS", 548000))
dev = "cuda:0"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for K, O in ((128, 64), (64, 128), (64, 64), (21, 128), (64, 32), (32, 128)):
    x = torch.randn(R, K, device=dev); w = torch.randn(O, K, device=dev) * 0.1; b = torch.zeros(O, device=dev)
    y = torch.empty(R, O, device=dev); dy = torch.randn(R, O, device=dev); dx = torch.empty(R, K, device=dev)
    dw = torch.zeros(O, K, device=dev); db = torch.zeros(O, device=dev)
    gb = lambda *ts: sum(a.numel() for a in ts) * 4 / 1e9
    f = t(lambda: lib.cm_linear_act_forward(R, K, O, x.data_ptr(), w.data_ptr(), 0, b.data_ptr(), 1, y.data_ptr(), st()))
    bfull = t(lambda: lib.cm_linear_act_backward(R, K, O, x.data_ptr(), w.data_ptr(), 0, dy.data_ptr(), None, y.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), st()))
    bnodx = t(lambda: lib.cm_linear_act_backward(R, K, O, x.data_ptr(), w.data_ptr(), 0, dy.data_ptr(), None, y.data_ptr(), None, dw.data_ptr(), db.data_ptr(), st()))
    bnoact = t(lambda: lib.cm_linear_act_backward(R, K, O, x.data_ptr(), w.data_ptr(), 0, dy.data_ptr(), None, None, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), st()))
    wg = t(lambda: lib.cm_linear_wgrad(R, O, K, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), st()))
    tf = t(lambda: torch.tanh(torch.nn.functional.linear(x, w, b)))
    def tb():
        dz = dy * (1 - y * y); return dz @ w
    tbt = t(tb)
    print(f"K={K:3d} O={O:3d}: fwd {f:6.0f} us ({gb(x, y) / f * 1e3:5.2f} TB/s)  torch fwd {tf:6.0f} | bwd {bfull:6.0f} us ({gb(dy, y, x, dx) / bfull * 1e3:5.2f} TB/s) "
          f"no-dx {bnodx:6.0f}  no-act {bnoact:6.0f}  old wgrad alone {wg:6.0f}  torch tanh'+dgrad {tbt:6.0f}")
