import os, sys, torch
sys.path.insert(0, '.')
from decnet_amd.model import Unit
dev = torch.device('cuda:0')
for (cin, cout, B, H, W) in ((72, 24, 16, 60, 108), (216, 72, 16, 20, 36)):
    u = Unit(cin, cout, 3, stride=3, transposed=True).to(dev).eval()
    x = torch.randn(B, cin, H, W, device=dev)
    for flag in ("1", "0"):
        os.environ["DECNET_CONV2D_MFMA"] = flag
        with torch.no_grad():
            for _ in range(3): u(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): u(x)
            e1.record(); e1.synchronize()
        print((cin, cout, B, H, W), "mfma" if flag == "1" else "library", "%.4f ms" % (e0.elapsed_time(e1) / 20))
