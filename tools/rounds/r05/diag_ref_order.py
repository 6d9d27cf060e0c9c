import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import ref
import decnet_amd
dev = torch.device('cuda:0')
B, C, H, W, D = 2, 3, 1, 636, 270
def run(tag):
    g = torch.Generator(device="cpu").manual_seed(1670)
    L, R = (torch.relu(torch.randn(B, C, H, W, generator=g) * 0.5) for _ in range(2))
    print(tag, "input checksum", float(L.double().sum()), float(R.double().sum()), torch.backends.cpu.get_cpu_capability())
    rm = tm = torch.ones(B, H, W)
    ro, _, rmx = ref.spamat_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
    o, _, _, m = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
    Ld, Rd = L.double().numpy(), R.double().numpy()
    truth = np.zeros((B, H, W))
    for b in range(B):
        for x in range(W):
            ds = np.arange(0, min(D, x + 1))
            c = (Ld[b, :, 0, x][:, None] * Rd[b, :, 0][:, x - ds]).sum(0)
            e = np.exp(c - max(1e-6, c.max()))
            truth[b, 0, x] = (1e-6 + (e * ds).sum()) / (1e-6 + e.sum())
    print(tag, "e_ref %.4g e_hip %.4g" % (np.abs(ro.cpu().numpy() - truth).mean(), np.abs(o.cpu().numpy() - truth).mean()))
run("before")
if "--mine" in sys.argv:
    from decnet_amd.modules.SparseMatching.build.lib import SpaMat
    run("after-my-module")
