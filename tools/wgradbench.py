"""cm_linear_wgrad vs the library GEMM for the update's weight-gradient shapes."""
import sys; sys.path.insert(0, '.')
import torch
from com_marl_amd.nets import _wgrad
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
R = 1371 * 200 * 4
for P, Q in ((128, 21), (64, 128), (64, 64), (128, 64), (64, 128), (32, 64), (5, 32), (1, 64)):
    a = torch.randn(R, P, device="cuda"); b = torch.randn(R, Q, device="cuda")
    t1 = timeit(lambda: _wgrad(a, b, True))
    t2 = timeit(lambda: (a.t() @ b, a.sum(0)))
    fl = 2 * R * P * Q
    print(f"R={R} P={P} Q={Q}: hip {t1:8.1f} us ({fl/t1/1e6:6.1f} TF)   torch {t2:8.1f} us ({fl/t2/1e6:6.1f} TF)   bytes/HBM-floor {R*(P+Q)*4/4e12*1e6:6.1f} us", flush=True)
