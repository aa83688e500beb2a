import sys; sys.path.insert(0,'.')
import torch, numpy as np
from com_marl_amd.nets import _LinearActFn
torch.manual_seed(0)
for (K,O,act,layout) in ((21,128,1,0),(128,64,1,0),(64,64,0,1),(64,32,1,0),(32,5,0,0),(64,1,0,0)):
    R=3456; S=1332
    x=torch.randn(R,K,device='cuda'); dy=torch.randn(R,O,device='cuda')*1e-3
    w=(torch.randn(O,K,device='cuda')*0.2) if layout==0 else (torch.randn(K,O,device='cuda')*0.2)
    b=torch.randn(O,device='cuda')*0.1 if layout==0 else None
    def run(xs,dys):
        xr=xs.clone().requires_grad_(); wr=w.clone().requires_grad_(); br=None if b is None else b.clone().requires_grad_()
        y=_LinearActFn.apply(xr,wr,br,act,layout); y.backward(dys)
        return y.detach(), xr.grad, wr.grad, None if br is None else br.grad
    yu,dxu,dwu,dbu=run(x,dy)
    ya,dxa,dwa,dba=run(x[:S],dy[:S]); yb,dxb,dwb,dbb=run(x[S:],dy[S:])
    print(K,O,act,layout, "fwd", float((torch.cat([ya,yb])-yu).abs().max()), "dx", float((torch.cat([dxa,dxb])-dxu).abs().max()),
          "dw", float((dwa+dwb-dwu).abs().max()/dwu.abs().max()), "db", None if dbu is None else float((dba+dbb-dbu).abs().max()/dbu.abs().max()))
    yu2,dxu2,dwu2,_=run(x,dy)
    print("   repeat: dw", float((dwu2-dwu).abs().max()/dwu.abs().max()), "dx", float((dxu2-dxu).abs().max()))
