"""oracle/stage0.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

torch-CPU restatement of the reference's stage-0 dense path (coarsest level):

  get_disp_samples (stage-0 branch)       modules/submodule.py:389-390
  GetCostVolume.get_warped_feats_by_homgrp modules/submodule.py:479-510
  GetCostVolume.cost_computation_cor       modules/submodule.py:518-522
  GetCostVolume.cost_computation_cat / ssd modules/submodule.py:512-516, 524-530
  CostRegNetNoDown.conv_pre (cost_func cat) modules/submodule.py:618-619, 651-652
  Conv3dUnit / CostRegNetNoDown.forward    modules/submodule.py:90-123, 608-662
  disparity_regression                     modules/submodule.py:766-777

The arithmetic of conv3d / batch_norm / grid_sample / softmax lives in PyTorch
(README pins 1.6.0; no lockfile), a third-party dependency outside /root/reference,
so those calls are made here through the same torch.nn.functional entry points the
reference uses.  ``warp_right_closed_form`` restates grid_sample's bilinear /
zero-padding / align_corners=False arithmetic independently so the
HIP kernel has a formula-level target as well.

PINNED by tests/golden/stage0_*.npz, produced by tests/golden/make_golden.py from
the imported reference classes themselves (tests/test_oracle_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5           # nn.BatchNorm3d default, submodule.py:111


def disp_samples(max_disp, B, H, W, dtype=torch.float32):
    """submodule.py:389-390 -> [B,D,H,W] with value d on plane d."""
    return torch.arange(int(max_disp), dtype=dtype).view(1, -1, 1, 1).expand(B, -1, H, W)


def warp_right(right, samples):
    """submodule.py:491-503: grid_sample(align_corners=False default) on coordinates
    normalised the align_corners=True way.  -> [B,C,S,H,W]"""
    B, C, H, W = right.shape
    S = samples.shape[1]
    pos_y, pos_x = torch.meshgrid(torch.arange(H, dtype=right.dtype),
                                  torch.arange(W, dtype=right.dtype), indexing="ij")
    pos_x = pos_x.reshape(1, 1, H, W).repeat(B, S, 1, 1)
    pos_y = pos_y.reshape(1, 1, H, W).repeat(B, S, 1, 1)
    cx = (pos_x - samples) / ((W - 1.0) / 2.0) - 1.0
    cy = pos_y / ((H - 1.0) / 2.0) - 1.0
    grid = torch.stack([cx, cy], dim=4)
    return F.grid_sample(right, grid.view(B, S * H, W, 2), mode="bilinear", padding_mode="zeros",
                         align_corners=False).view(B, C, S, H, W)


def warp_right_closed_form(right, max_disp):
    """Same as warp_right(right, disp_samples(D)) from first principles (float32 numpy).

    cx = (x-d)/((W-1)/2) - 1 ; unnormalise (align_corners=False): ix = ((cx+1)*W - 1)/2
    cy = y/((H-1)/2) - 1     ;                                     iy = ((cy+1)*H - 1)/2
    bilinear over floor/ceil neighbours, out-of-range taps contribute 0.
    """
    r = right.detach().cpu().numpy().astype(np.float32)
    B, C, H, W = r.shape
    D = int(max_disp)
    f = np.float32
    x = np.arange(W, dtype=f)[None, :] - np.arange(D, dtype=f)[:, None]          # [D,W]
    cx = x / f((W - 1.0) / 2.0) - f(1.0)
    ix = ((cx + f(1.0)) * f(W) - f(1.0)) / f(2.0)
    y = np.arange(H, dtype=f)
    cy = y / f((H - 1.0) / 2.0) - f(1.0)
    iy = ((cy + f(1.0)) * f(H) - f(1.0)) / f(2.0)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    wx1 = (ix - x0).astype(f)
    wx0 = (f(1.0) - wx1).astype(f)
    wy1 = (iy - y0).astype(f)
    wy0 = (f(1.0) - wy1).astype(f)
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    out = np.zeros((B, C, D, H, W), f)

    def tap(yy, xx):                      # yy [H], xx [D,W] -> [B,C,D,H,W], zeros outside
        okx = (xx >= 0) & (xx < W)
        oky = (yy >= 0) & (yy < H)
        v = r[:, :, np.clip(yy, 0, H - 1)][:, :, :, np.clip(xx, 0, W - 1)]        # [B,C,H,D,W]
        v = v * okx[None, None, None] * oky[None, None, :, None, None]
        return v.transpose(0, 1, 3, 2, 4)

    wy0b, wy1b = wy0[None, None, None, :, None], wy1[None, None, None, :, None]
    wx0b, wx1b = wx0[None, None, :, None, :], wx1[None, None, :, None, :]
    # same tap order / weight products as grid_sample's CPU kernel: nw, ne, sw, se
    out += tap(y0, x0) * (wx0b * wy0b)
    out += tap(y0, x0 + 1) * (wx1b * wy0b)
    out += tap(y0 + 1, x0) * (wx0b * wy1b)
    out += tap(y0 + 1, x0 + 1) * (wx1b * wy1b)
    return torch.from_numpy(out)


def cost_volume(left, right, max_disp, cost_func="cor"):
    """GetCostVolume(warp_ope="homgrp", cost_func=...).forward -> [B,C,D,H,W] ([B,2C,D,H,W] for "cat")
    (submodule.py:532-562 with :505-509 left masking; :521 product, :514 concatenation, :527-529 ssd)."""
    B, C, H, W = left.shape
    samples = disp_samples(max_disp, B, H, W, left.dtype)
    right_vol = warp_right(right, samples)
    left_vol = left.unsqueeze(2).repeat(1, 1, samples.shape[1], 1, 1)
    pos_x = torch.arange(W, dtype=left.dtype).view(1, 1, 1, W).expand(B, samples.shape[1], H, W)
    keep = ~(pos_x < samples)                                            # x >= d
    left_vol = left_vol * keep.unsqueeze(1).to(left.dtype)
    if cost_func == "cor":
        return left_vol * right_vol
    if cost_func == "cat":
        return torch.cat((left_vol, right_vol), dim=1)
    if cost_func == "ssd":                                               # the reference's in-place sequence, out of place
        volume_sum = left_vol + right_vol
        volume_sqr = left_vol.pow(2) + right_vol.pow(2)
        return volume_sqr.div(2).sub(volume_sum.div(2).pow(2))
    raise ValueError("No such cost computation function: {}".format(cost_func))


def conv3d_unit(x, w, bn, relu):
    """Conv3dUnit.forward, eval mode (submodule.py:115-123): conv(k3,s1,p1,no bias) ->
    batch_norm(running stats) -> relu.  bn = (gamma, beta, running_mean, running_var)."""
    x = F.conv3d(x, w, None, stride=1, padding=1)
    g, b, m, v = bn
    x = F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)
    return F.relu(x) if relu else x


def cost_regularizer(x, params, w_pre=None):
    """CostRegNetNoDown.forward (submodule.py:650-662).
    params: list of 8 dicts {"w": [Co,Ci,3,3,3], "bn": (gamma,beta,mean,var)} in module
    order conv0[0..1], conv1[0..2], conv2[0..2];  w_pre: conv_pre.weight [C,2C,1,1,1] of cost_func="cat"
    (:618-619, applied first :651-652), else None.  -> [B,D,H,W]"""
    if w_pre is not None:
        x = F.conv3d(x, w_pre, None, stride=1, padding=0)
    u = lambda i, t, relu=True: conv3d_unit(t, params[i]["w"], params[i]["bn"], relu)
    o0 = u(1, u(0, x))
    o = u(4, u(3, u(2, o0))) + o0
    o = u(7, u(6, u(5, o)), relu=False)
    return o.squeeze(1)


def disparity_regression(cost, samples):
    """submodule.py:766-777"""
    return torch.sum(F.softmax(cost, dim=1) * samples, 1)


def stage0_forward(left, right, params, max_disp, cost_func="cor", w_pre=None):
    """The whole stage-0 branch of SparseDenseNetRefinementMask.forward (:127-137)."""
    B, C, H, W = left.shape
    assert (w_pre is not None) == (cost_func == "cat")
    cv = cost_volume(left, right, max_disp, cost_func)
    reg = cost_regularizer(cv, params, w_pre)
    return disparity_regression(reg, disp_samples(max_disp, B, H, W, left.dtype)), reg, cv


def params_from_module(reg):
    """Pull the 8 (weight, bn) sets out of a reference-shaped CostRegNetNoDown module
    (anything exposing conv0/conv1/conv2 Sequentials of units with .conv/.bn)."""
    out = []
    for seq in (reg.conv0, reg.conv1, reg.conv2):
        for unit in seq:
            bn = unit.bn
            out.append({"w": unit.conv.weight.detach().clone(),
                        "bn": (bn.weight.detach().clone(), bn.bias.detach().clone(),
                               bn.running_mean.detach().clone(), bn.running_var.detach().clone())})
    return out


def random_params(C, seed, bn_random=True):
    """Deterministic synthetic CostRegNetNoDown parameters (conv init as
    SparseDenseNetRefinementMask._initialize_weights :248-250: N(0, sqrt(2/(27*Co))));
    BN stats randomised (a trained net has non-trivial ones) unless bn_random=False."""
    g = torch.Generator().manual_seed(int(seed))
    out = []
    for i in range(8):
        co = 1 if i == 7 else C
        w = torch.randn(co, C, 3, 3, 3, generator=g) * float(np.sqrt(2.0 / (27 * co)))
        if bn_random:
            bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1,
                  torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5)
        else:
            bn = (torch.ones(co), torch.zeros(co), torch.zeros(co), torch.ones(co))
        out.append({"w": w, "bn": bn})
    return out


def random_w_pre(C, seed):
    """Deterministic conv_pre.weight [C,2C,1,1,1] of cost_func="cat" (init as :246-248: N(0, sqrt(2 / C)))."""
    g = torch.Generator().manual_seed(int(seed) + 4242)
    return torch.randn(C, 2 * C, 1, 1, 1, generator=g) * float(np.sqrt(2.0 / C))
