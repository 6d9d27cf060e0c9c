import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from decnet_amd.ext import SpaMat as ESM
from decnet_amd.modules.SparseMatching.build.lib import SpaMat as sm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B_, C, H, W, p = 2, 8, 5, 300, 0.6
L = torch.relu(torch.randn(B_, C, H, W, generator=g)).to(dev)
R = torch.relu(torch.randn(B_, C, H, W, generator=g)).to(dev)
rm = (torch.rand(B_, H, W, generator=g) < p).float().to(dev)
tm = (torch.rand(B_, H, W, generator=g) < p).float().to(dev)
torch.cuda.synchronize()
def show(tag, o, s, m):
    torch.cuda.synchronize()
    off = rm == 0
    print(tag, "o at masked-off: n7=%d n0=%d other=%d | s n7=%d | at active: n7=%d" % (
        int((o[off] == 7).sum()), int((o[off] == 0).sum()), int(((o[off] != 7) & (o[off] != 0)).sum()),
        int((s[off] == 7).sum()), int((o[~off] == 7).sum())))
for D in (216, np.int64(216)):
    for name, fn in (("ctypes", ESM.sparse_matching_cuda_forward), ("compiled", sm.sparse_matching_cuda_forward)):
        o, s, m = (torch.full_like(rm, 7.0) for _ in range(3))
        torch.cuda.synchronize()
        fn(L, R, rm, tm, o, s, m, D)
        show("default-stream %-8s D=%r" % (name, type(D).__name__), o, s, m)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            o, s, m = (torch.full_like(rm, 7.0) for _ in range(3))
            fn(L, R, rm, tm, o, s, m, D)
        side.synchronize()
        show("side-stream    %-8s D=%r" % (name, type(D).__name__), o, s, m)
        side = torch.cuda.Stream()
        o, s, m = (torch.full_like(rm, 7.0) for _ in range(3))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            fn(L, R, rm, tm, o, s, m, D)
        side.synchronize()
        show("side, prefilled %-8s D=%r" % (name, type(D).__name__), o, s, m)
