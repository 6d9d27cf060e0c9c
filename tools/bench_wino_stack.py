import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, time, os, ctypes
import decnet_amd
from decnet_amd import _lib
L=_lib.lib()
dev=torch.device("cuda:0")
B=int(os.environ.get("SB","8")); D,H,W,C=8,20,36,216
g=torch.Generator().manual_seed(0)
layers=[]
for i in range(7):
    w=(torch.randn(C,C,3,3,3,generator=g)*(2.0/(27*C))**0.5).to(dev)
    u=torch.empty(L.decnet_conv3d_wino_weight_floats(C,2),device=dev)
    L.decnet_conv3d_wino_pack_weight(w.data_ptr(),u.data_ptr(),C,C,2,None)
    layers.append(dict(u=u,scale=torch.ones(C,device=dev),shift=torch.zeros(C,device=dev)))
x=torch.randn(B,D,H,W,C,device=dev)
y=torch.empty_like(x)
n=L.decnet_conv3d_wino_stack_workspace_floats(B,D,H,W,C,2)
ws=torch.empty(n,device=dev)
arr=ctypes.c_void_p*7
u,sc,sh=(arr(*[p[k].data_ptr() for p in layers]) for k in ("u","scale","shift"))
def stack():
    rc=L.decnet_conv3d_wino_stack_bn_act(x.data_ptr(),u,sc,sh,7,1,4,y.data_ptr(),ws.data_ptr(),B,D,H,W,C,2,None); assert rc==0,rc
a=torch.empty_like(x); b=torch.empty_like(x); c=torch.empty_like(x)
def conv(i,s,d,r=None):
    p=layers[i]
    rc=L.decnet_conv3d_wino_bn_act(s.data_ptr(),p["u"].data_ptr(),p["scale"].data_ptr(),p["shift"].data_ptr(),r.data_ptr() if r is not None else None,d.data_ptr(),ws.data_ptr(),B,D,H,W,C,C,1,2,None); assert rc==0
def seq():
    conv(0,x,a);conv(1,a,c);conv(2,c,a);conv(3,a,b);conv(4,b,a,c);conv(5,a,b);conv(6,b,c)
for name,f in (("seq",seq),("stack",stack)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True);e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record();torch.cuda.synchronize()
    print(name, "%.3f ms per 7 layers"%(e0.elapsed_time(e1)/20))
print("maxdiff", float((y-c).abs().max()), float(c.abs().max()))
