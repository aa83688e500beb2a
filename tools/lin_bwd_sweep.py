"""cm_linear_act_backward at a row count, for a sweep of COMMARL_LIN2_BLOCKS (one process per setting: the override is read once).
   python tools/lin_bwd_sweep.py R K O"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib  # noqa: E402

import torch  # noqa: E402

L = importlib.import_module("com_marl_amd._lib")
R, K, O = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.tanh(torch.randn(R, K, device=dev))
y = torch.tanh(torch.randn(R, O, device=dev)) if os.environ.get("NO_ACT") is None else None
dy = torch.randn(R, O, device=dev)
w = torch.randn(O, K, device=dev) * 0.1
dx = torch.empty_like(x)
dw = torch.zeros_like(w)
db = torch.zeros(O, device=dev)


def run():
    L.check(L.lib().cm_linear_act_backward(R, K, O, L.ptr(x), L.ptr(w), 0, L.ptr(dy), None, L.ptr(y), L.ptr(dx), L.ptr(dw), L.ptr(db),
                                           L.current_stream()), "bwd")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print(f"R={R} K={K} O={O} blocks={os.environ.get('COMMARL_LIN2_BLOCKS', os.environ.get('COMMARL_LIN_BLOCKS', 'default'))}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
