"""Independent vectorised torch restatement of SpaMat / SpaVar (test infrastructure).

Written from the math in SURVEY.md section 2b, NOT from oracle/spamat_oracle.c: it
materialises the dense [B,H,W,D] cost tensor, so it is only usable at small sizes.
Differentiable (max_cost detached, S7), so autograd of it checks the hand-written
backward kernels (SM_kernel.cu:143-195, 300-355; SV_kernel.cu:142-325).
"""
import torch


def _costs(ref, tar, rmask, tmask, max_disp):
    B, C, H, W = ref.shape
    D = int(max_disp)
    costs, valid = [], []
    for d in range(D):
        shifted = torch.zeros_like(tar)
        tm = torch.zeros_like(tmask)
        if d < W:
            shifted[..., d:] = tar[..., : W - d]
            tm[..., d:] = tmask[..., : W - d]
        costs.append((ref * shifted).sum(1))
        v = (tm != 0) & (rmask != 0)
        v[..., : min(d, W)] = False          # x - d < 0
        valid.append(v)
    return torch.stack(costs, -1), torch.stack(valid, -1)      # [B,H,W,D]


def spamat(ref, tar, rmask, tmask, max_disp):
    """-> output, sum_sim, max_cost  ([B,H,W]); zeros where ref mask is off."""
    cost, valid = _costs(ref, tar, rmask, tmask, max_disp)
    neg = torch.full_like(cost, float("-inf"))
    mx = torch.where(valid, cost, neg).max(-1).values.clamp_min(1e-6).detach()
    e = torch.where(valid, torch.exp(cost - mx[..., None]), torch.zeros_like(cost))
    d = torch.arange(cost.shape[-1], dtype=cost.dtype)
    S = 1e-6 + e.sum(-1)
    out = (1e-6 + (e * d).sum(-1)) / S
    on = rmask != 0
    z = torch.zeros_like(out)
    return torch.where(on, out, z), torch.where(on, S, z), torch.where(on, mx, z)


def spavar(ref, tar, rmask, tmask, disparity, max_disp):
    cost, valid = _costs(ref, tar, rmask, tmask, max_disp)
    neg = torch.full_like(cost, float("-inf"))
    mx = torch.where(valid, cost, neg).max(-1).values.clamp_min(1e-6).detach()
    e = torch.where(valid, torch.exp(cost - mx[..., None]), torch.zeros_like(cost))
    d = torch.arange(cost.shape[-1], dtype=cost.dtype)
    S = 1e-6 + e.sum(-1)
    dd = d - disparity[..., None]
    out = (1e-6 + (e * dd * dd).sum(-1)) / S
    on = rmask != 0
    z = torch.zeros_like(out)
    return torch.where(on, out, z), torch.where(on, S, z), torch.where(on, mx, z)
