"""decnet_warp_disparity at [8,8,540,972]: how much of its time is the incoherence of the disparity map?  (The benchmark's
untrained network produces noise; a trained one produces piecewise-smooth maps.)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from decnet_amd import _lib
from decnet_amd.ops import _stream
dev = torch.device("cuda:0")
B, C, H, W = 8, 8, 540, 972
r = torch.randn(B, C, H, W, device=dev)
out = torch.empty_like(r)
L = _lib.lib()
xs = torch.arange(W, device=dev).float().view(1, 1, W)
ys = torch.arange(H, device=dev).float().view(1, H, 1)
maps = {
    "uniform noise 0..216": torch.rand(B, H, W, device=dev) * 216,
    "gaussian noise around 60, sigma 30": (60 + 30 * torch.randn(B, H, W, device=dev)).clamp(0, 216),
    "smooth ramp + sine (sub-pixel gradients)": (40 + 0.05 * xs + 10 * torch.sin(ys / 40) + torch.zeros(B, 1, 1, device=dev)).expand(B, H, W).contiguous(),
    "piecewise constant blocks of 64 x 64": (torch.randint(0, 200, (B, (H + 63) // 64, (W + 63) // 64), device=dev).float()
                                             .repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :H, :W]).contiguous(),
    "constant 37.5": torch.full((B, H, W), 37.5, device=dev),
}
for name, d in maps.items():
    d = d.contiguous()
    with torch.cuda.device(dev):
        for _ in range(3):
            _lib.check(L.decnet_warp_disparity(r.data_ptr(), d.data_ptr(), out.data_ptr(), B, C, H, W, _stream(r)), "warp")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.decnet_warp_disparity(r.data_ptr(), d.data_ptr(), out.data_ptr(), B, C, H, W, _stream(r))
        e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-45s %.4f ms  %.2f TB/s of its 2 x %.0f MB + disparity" % (name, ms, (2 * r.numel() * 4 + d.numel() * 4) / ms / 1e9, r.numel() * 4 / 1e6))
