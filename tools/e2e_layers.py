"""Per-module GPU time of the whole-graph forward (decnet_amd.model), batch B at 972x540:
forward hooks with events around every leaf conv / Unit.  python tools/e2e_layers.py [B]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from decnet_amd.model import get_model, Unit  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    torch.manual_seed(17)
    model = get_model(name="sparsedensenetrefinementmask", max_disp=bench.MAX_DISP, base_channels=8, cost_func="cor",
                      grad_method="detach", num_stage=4, down_scale=3, step=[-1., 1., 1., 1.],
                      samp_num=[-1., 12., 10., 6.], sample_spa_size_list=[-1, 3, 5, 7],
                      down_func_name="bicubic", weights=[1., 1., 1., 1.], if_overmask=False, skip_stage_id=4,
                      use_detail=True, thold=0.5).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(17)
    left = torch.randn(B, 3, bench.PAD_H, bench.PAD_W, device=dev, generator=g)
    right = torch.randn(B, 3, bench.PAD_H, bench.PAD_W, device=dev, generator=g)
    rec = collections.OrderedDict()
    state = {}

    def pre(name):
        def f(m, inp):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            x = inp[0]
            shp = tuple(x.shape) if torch.is_tensor(x) else ("cat",) + tuple(tuple(t.shape) for t in x)
            state[name] = (e, shp)
        return f

    def post(name):
        def f(m, inp, out):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            e0, shp = state[name]
            rec.setdefault(name, []).append((e0, e1, shp, m))
        return f
    for name, m in model.named_modules():
        if isinstance(m, Unit):
            m.register_forward_pre_hook(pre(name))
            m.register_forward_hook(post(name))
    with torch.no_grad():
        for _ in range(2):
            model(left, right)
        rec.clear()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        model(left, right)
        t1.record()
        torch.cuda.synchronize()
    print("forward %.2f ms" % t0.elapsed_time(t1))
    rows = []
    for name, calls in rec.items():
        for e0, e1, shp, m in calls:
            c = m.conv
            rows.append((e0.elapsed_time(e1), name, shp, type(c).__name__, tuple(c.weight.shape), c.stride, c.dilation))
    tot = sum(r[0] for r in rows)
    print("Unit modules: %d calls, %.2f ms" % (len(rows), tot))
    for r in sorted(rows, key=lambda r: -r[0])[:120]:
        print("%7.3f ms  %-42s in=%s %s w=%s s=%s d=%s" % r)


if __name__ == "__main__":
    main()
