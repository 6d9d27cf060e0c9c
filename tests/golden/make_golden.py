#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python classes.

Run in the build container only (needs /root/reference; the GPU box has neither the
reference nor any need for this script -- it consumes the committed .npz data):

    python tests/golden/make_golden.py

What is real reference execution and what is not:
  * stage0_small.npz / stage0_c216.npz -- GetCostVolume, CostRegNetNoDown and
    disparity_regression are the reference's classes (modules/submodule.py) run on
    torch CPU.  These pin oracle/stage0.py and, through it, the HIP stage-0 kernels.
  * net_54x243.npz -- the full reference graph (SparseDenseNetRefinementMask, demo.sh
    hyper-parameters, torch.manual_seed(17) init as demo.py:70) run with forward hooks.
    The compiled CUDA ops (`..build.lib.SpaMat/SpaVar`) cannot be built here (no nvcc),
    so they are stubbed with oracle/spamat_oracle.c: the arrays recorded at the
    sparse_matching / sparse_var call sites pin the CALL-SITE contract (argument order,
    shapes, mask values, per-stage max_disp, numpy.int64 type -- SURVEY.md S13), not
    the kernel arithmetic.
Missing third-party imports of the reference (torchvision, cv2, visdom) are replaced
by empty stub modules; none of their functions is on the recorded path.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
import oracle                      # noqa: E402
from oracle import stage0 as o0    # noqa: E402


class _ExtStub:
    """Stands in for the pybind modules of SM_cuda.cpp:29-33 / SV_cuda.cpp:34-38."""

    @staticmethod
    def sparse_matching_cuda_forward(ref, tar, rm, tm, out, ssum, mx, max_disp):
        ref, tar, rm, tm = (t.float() for t in (ref, tar, rm, tm))       # (a .double() graph: see fp32_noise)
        o, s, m = oracle.spamat_forward(ref, tar, rm, tm, max_disp)
        out.copy_(torch.from_numpy(o)); ssum.copy_(torch.from_numpy(s)); mx.copy_(torch.from_numpy(m))
        return 1

    @staticmethod
    def sparse_var_cuda_forward(ref, tar, rm, tm, disp, out, ssum, mx, max_disp):
        ref, tar, rm, tm, disp = (t.float() for t in (ref, tar, rm, tm, disp))
        o, s, m = oracle.spavar_forward(ref, tar, rm, tm, disp, max_disp)
        out.copy_(torch.from_numpy(o)); ssum.copy_(torch.from_numpy(s)); mx.copy_(torch.from_numpy(m))
        return 1


def install_stubs():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms", "cv2", "visdom"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for pkg, attr in (("modules.SparseMatching.build", "SpaMat"), ("modules.SparseVar.build", "SpaVar")):
        b = types.ModuleType(pkg)
        lib = types.ModuleType(pkg + ".lib")
        setattr(lib, attr, _ExtStub)
        b.lib = lib
        sys.modules[pkg] = b
        sys.modules[pkg + ".lib"] = lib
    sys.path.insert(0, REF)


def bn_randomise(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            n = m.num_features
            m.weight.data = torch.rand(n, generator=g) + 0.5
            m.bias.data = torch.randn(n, generator=g) * 0.1
            m.running_mean.data = torch.randn(n, generator=g) * 0.1
            m.running_var.data = torch.rand(n, generator=g) + 0.5


def load_params(module, params):
    i = 0
    for seq in (module.conv0, module.conv1, module.conv2):
        for unit in seq:
            unit.conv.weight.data = params[i]["w"].clone()
            g, b, m, v = params[i]["bn"]
            unit.bn.weight.data, unit.bn.bias.data = g.clone(), b.clone()
            unit.bn.running_mean.data, unit.bn.running_var.data = m.clone(), v.clone()
            i += 1


def flat_params(params):
    d = {}
    for i, p in enumerate(params):
        d["w%d" % i] = p["w"].numpy()
        for k, t in zip(("gamma", "beta", "mean", "var"), p["bn"]):
            d["bn%d_%s" % (i, k)] = t.numpy()
    return d


def stage0_case(sub, C, B, H, W, D, seed, store_params, cost_func="cor"):
    torch.manual_seed(seed)
    left = torch.relu(torch.randn(B, C, H, W))
    right = torch.relu(torch.randn(B, C, H, W))
    gcv = sub.GetCostVolume(warp_ope="homgrp", cost_func=cost_func)
    reg = sub.CostRegNetNoDown(in_channels=C, base_channels=2 * C, cost_func=cost_func, down_scale=3)
    params = o0.random_params(C, seed + 1000)
    load_params(reg, params)
    w_pre = None
    if cost_func == "cat":                             # submodule.py:618-619
        w_pre = o0.random_w_pre(C, seed + 1000)
        assert tuple(reg.conv_pre.weight.shape) == tuple(w_pre.shape)
        reg.conv_pre.weight.data = w_pre.clone()
    reg.eval()
    with torch.no_grad():
        ds = sub.get_disp_samples(D, left, stage_id=0)
        cv = gcv(left, right, disp_samples=ds, max_disp=D)
        r = reg(cv.clone())
        pred = sub.disparity_regression(r, ds)
    out = dict(left=left.numpy(), right=right.numpy(), max_disp=np.int64(D), cost_vol=cv.numpy(),
               reg=r.numpy(), pred=pred.numpy(), param_seed=np.int64(seed + 1000),
               w0_checksum=np.float64(params[0]["w"].double().sum().item()),
               w7_checksum=np.float64(params[7]["w"].double().abs().sum().item()))
    if store_params:
        out.update(flat_params(params))
    if w_pre is not None:
        out["w_pre_checksum"] = np.float64(w_pre.double().abs().sum().item())
        if store_params:
            out["w_pre"] = w_pre.numpy()
    return out


def net_case():
    from modules import get_model
    torch.manual_seed(17)                                                 # demo.py:70
    model = get_model(name="sparsedensenetrefinementmask", max_disp=216, base_channels=8,
                      cost_func="cor", grad_method="detach", num_stage=4, down_scale=3,
                      step=[-1., 1., 1., 1.], samp_num=[-1., 12., 10., 6.],
                      sample_spa_size_list=[-1, 3, 5, 7], down_func_name="bicubic",
                      weights=[1., 1., 1., 1.], if_overmask=False, skip_stage_id=4,
                      use_detail=True, thold=0.5)
    model.eval()
    rec = {}

    def hook(tag):
        def fn(mod, inp, out):
            names = ("ref", "tar", "rmask", "tmask", "disparity", "max_disp") if len(inp) == 6 else \
                    ("ref", "tar", "rmask", "tmask", "max_disp")
            for n, v in zip(names, inp):
                if n == "max_disp":
                    rec["%s_max_disp" % tag] = np.int64(v)
                    rec["%s_max_disp_type" % tag] = np.array(type(v).__name__)
                else:
                    rec["%s_%s" % (tag, n)] = v.detach().numpy().copy()
            rec["%s_out" % tag] = out.detach().numpy().copy()
        return fn

    for i in range(3):
        model.sparse_matching[i].register_forward_hook(hook("sm%d" % (i + 1)))
        model.sparse_var[i].register_forward_hook(hook("sv%d" % (i + 1)))
    g = torch.Generator().manual_seed(1717)
    H, W = 54, 243
    left = torch.randn(1, 3, H, W, generator=g)
    right = torch.randn(1, 3, H, W, generator=g)
    disp = torch.zeros(1, H, W)
    lm = [torch.ones(1, H // 9, W // 9), torch.ones(1, H // 3, W // 3), torch.ones(1, H, W)]
    with torch.no_grad():
        pred = model(left, right, disp, lm, lm, is_check=False, is_eval=False)[-1]
    rec["pred"] = pred.numpy()
    # SpaVar is called with the very tensors SpaMat got: record that as a
    # fact and drop the duplicate arrays, and note disparity == SpaMat's output.
    for i in (1, 2, 3):
        same = all(np.array_equal(rec["sm%d_%s" % (i, n)], rec["sv%d_%s" % (i, n)])
                   for n in ("ref", "tar", "rmask", "tmask"))
        rec["sv%d_same_inputs_as_sm" % i] = np.bool_(same)
        rec["sv%d_disparity_is_sm_out" % i] = np.bool_(
            np.array_equal(rec["sv%d_disparity" % i], rec["sm%d_out" % i]))
        if same:
            for n in ("ref", "tar", "rmask", "tmask"):
                del rec["sv%d_%s" % (i, n)]
    return rec


E2E_KW = dict(name="sparsedensenetrefinementmask", max_disp=216, base_channels=2, cost_func="cor",
              grad_method="detach", num_stage=4, down_scale=3, step=[-1., 1., 1., 1.],
              samp_num=[-1., 12., 10., 6.], sample_spa_size_list=[-1, 3, 5, 7], down_func_name="bicubic",
              weights=[1., 1., 1., 1.], if_overmask=False, skip_stage_id=4, use_detail=True, thold=0.5)


def e2e_inputs(seed=99, H=54, W=243):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 3, H, W, generator=g), torch.randn(1, 3, H, W, generator=g)


def e2e_case(cost_func="cor"):
    """Whole reference graph (base_channels=2 so that the run is small) with seeded synthetic
    parameters (tests/golden/netparams.py -- regenerated by the test, not stored): the network's
    final disparity and the per-stage sparse results.  SpaMat/SpaVar are the oracle stub."""
    from modules import get_model
    from netparams import fill_state_dict
    model = get_model(**dict(E2E_KW, cost_func=cost_func))
    model.load_state_dict(fill_state_dict(model.state_dict()))
    model.eval()
    rec = {}
    for i in range(3):
        model.sparse_matching[i].register_forward_hook(
            lambda m, inp, out, i=i: rec.__setitem__("sparse%d" % (i + 1), out.numpy().copy()) or
            rec.__setitem__("lmask%d" % (i + 1), inp[2].numpy().copy()))
    left, right = e2e_inputs()
    H, W = left.shape[-2:]
    lm = [torch.ones(1, H // 9, W // 9), torch.ones(1, H // 3, W // 3), torch.ones(1, H, W)]
    with torch.no_grad():
        pred = model(left, right, torch.zeros(1, H, W), lm, lm, is_check=False, is_eval=False)[-1]
    rec["pred"] = pred.numpy()
    rec["input_checksum"] = np.float64(left.double().sum().item() + right.double().abs().sum().item())
    rec["w_checksum"] = np.float64(sum(v.double().abs().sum().item() for v in model.state_dict().values()))
    return rec


def full_inputs(B, C, H, W, seed):
    """Stage-0 feature maps of a full-size case from numpy's frozen RandomState stream (the same on every host; the .npz
    stores only their CRC32): post-ReLU N(0, 1) like the real extractor's outputs."""
    rs = np.random.RandomState(seed)
    left = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    right = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    return left, right


def stage0_full_case(sub, C, B, H, W, D, seed, cost_func="cor"):
    """BASELINE config 2's full stage-0 shape (216 channels, 20 x 36, D = 8) through the REFERENCE's classes; every op is
    per sample; B = 8 is the whole batch of the bench step.  Outputs only (the cost volume alone would be 40 MB)."""
    import zlib
    ln, rn = full_inputs(B, C, H, W, seed)
    left, right = torch.from_numpy(ln), torch.from_numpy(rn)
    gcv = sub.GetCostVolume(warp_ope="homgrp", cost_func=cost_func)
    reg = sub.CostRegNetNoDown(in_channels=C, base_channels=2 * C, cost_func=cost_func, down_scale=3)
    params = o0.random_params(C, seed + 1000)
    load_params(reg, params)
    if cost_func == "cat":
        reg.conv_pre.weight.data = o0.random_w_pre(C, seed + 1000)
    reg.eval()
    with torch.no_grad():
        ds = sub.get_disp_samples(D, left, stage_id=0)
        cv = gcv(left, right, disp_samples=ds, max_disp=D)
        r = reg(cv.clone())
        pred = sub.disparity_regression(r, ds)
    return dict(shape=np.array([B, C, H, W], np.int64), max_disp=np.int64(D), input_seed=np.int64(seed),
                input_crc=np.int64(zlib.crc32(ln.tobytes() + rn.tobytes())), param_seed=np.int64(seed + 1000),
                w0_checksum=np.float64(params[0]["w"].double().sum().item()),
                w7_checksum=np.float64(params[7]["w"].double().abs().sum().item()),
                cost_vol_abs_sum=np.float64(cv.double().abs().sum().item()), reg=r.numpy(), pred=pred.numpy())


def main():
    install_stubs()
    import modules.submodule as sub
    if "--only-costfunc" in sys.argv:
        # cost_func "ssd" (demo.py:31's default) and "cat" (+ CostRegNetNoDown.conv_pre): the reference's own classes again
        for cf in ("ssd", "cat"):
            np.savez_compressed(os.path.join(HERE, "stage0_%s_small.npz" % cf),
                                **stage0_case(sub, C=12, B=2, H=5, W=9, D=4, seed=5, store_params=True, cost_func=cf))
            np.savez_compressed(os.path.join(HERE, "stage0_%s_c216.npz" % cf),
                                **stage0_case(sub, C=216, B=1, H=4, W=7, D=8, seed=7, store_params=False, cost_func=cf))
            np.savez_compressed(os.path.join(HERE, "stage0_cfg2_%s_full.npz" % cf),
                                **stage0_full_case(sub, C=216, B=8, H=20, W=36, D=8, seed=11, cost_func=cf))
            # config 3 (KITTI) and config 4 (Middlebury half-res: D = 10, the volume does not fit the fused head kernel's
            # LDS, so the stand-alone volume kernel + the stack run instead)
            np.savez_compressed(os.path.join(HERE, "stage0_cfg3_%s_full.npz" % cf),
                                **stage0_full_case(sub, C=216, B=4, H=14, W=46, D=8, seed=12, cost_func=cf))
            np.savez_compressed(os.path.join(HERE, "stage0_cfg4_%s_full.npz" % cf),
                                **stage0_full_case(sub, C=216, B=1, H=38, W=56, D=10, seed=13, cost_func=cf))
            sys.path.insert(0, HERE)                   # the whole graph with the other stage-0 volume
            np.savez_compressed(os.path.join(HERE, "e2e_bc2_54x243_%s.npz" % cf), **e2e_case(cf))
        return
    if "--only-stage0-full" in sys.argv:               # (minutes of CPU: 203 GFLOP through torch's Conv3d)
        np.savez_compressed(os.path.join(HERE, "stage0_cfg2_full.npz"),
                            **stage0_full_case(sub, C=216, B=8, H=20, W=36, D=8, seed=11))
        # config 3 (KITTI, 4 pairs per GPU) and config 4 (Middlebury half-res, max_disp 270 -> D = 10)
        np.savez_compressed(os.path.join(HERE, "stage0_cfg3_full.npz"),
                            **stage0_full_case(sub, C=216, B=4, H=14, W=46, D=8, seed=12))
        np.savez_compressed(os.path.join(HERE, "stage0_cfg4_full.npz"),
                            **stage0_full_case(sub, C=216, B=1, H=38, W=56, D=10, seed=13))
        return
    np.savez_compressed(os.path.join(HERE, "stage0_small.npz"),
                        **stage0_case(sub, C=12, B=2, H=5, W=9, D=4, seed=5, store_params=True))
    np.savez_compressed(os.path.join(HERE, "stage0_c216.npz"),
                        **stage0_case(sub, C=216, B=1, H=4, W=7, D=8, seed=7, store_params=False))
    rec = net_case()
    for k in sorted(rec):
        v = rec[k]
        if getattr(v, "ndim", 0) >= 2 and k.endswith("mask"):
            print(k, v.shape, "density %.3f" % float((v != 0).mean()))
    np.savez_compressed(os.path.join(HERE, "net_54x243.npz"), **rec)
    sys.path.insert(0, HERE)
    e2e = e2e_case()
    for i in (1, 2, 3):
        print("e2e lmask%d density %.3f" % (i, float((e2e["lmask%d" % i] != 0).mean())))
    np.savez_compressed(os.path.join(HERE, "e2e_bc2_54x243.npz"), **e2e)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
