"""Test helper: observe the model's fused SpaMat/SpaVar call whichever entry point it takes (float mask planes or the
bit-packed masks of decnet_detail_mask) -- callback(L, R, left_mask_float, right_mask_float, D, outputs)."""
import contextlib

import torch


def unpack_mask_bits(bits, W):
    """int64 [B,H,ceil(W/64)] (bit i of word w = pixel 64 w + i) -> float 0/1 [B,H,W]."""
    B, H, wpr = bits.shape
    sh = torch.arange(64, device=bits.device, dtype=torch.int64)
    return ((bits.unsqueeze(-1) >> sh) & 1).reshape(B, H, wpr * 64)[:, :, :W].float()


@contextlib.contextmanager
def spamat_spy(callback):
    import decnet_amd.model as M
    orig_f, orig_b = M.spamatvar_forward, M.spamatvar_forward_bits

    # (the graph runs this call on a side stream beside DynamicUpsampling: the callback reads the outputs, so it
    # synchronises that stream first)
    def spy_f(L, R, lm, rm, D, out=None):
        o = orig_f(L, R, lm, rm, D, out=out)
        torch.cuda.current_stream(L.device).synchronize()
        callback(L, R, lm, rm, D, o)
        return o

    def spy_b(L, R, lb, rb, D, out=None):
        o = orig_b(L, R, lb, rb, D, out=out)
        torch.cuda.current_stream(L.device).synchronize()
        W = L.shape[-1]
        callback(L, R, unpack_mask_bits(lb, W), unpack_mask_bits(rb, W), D, o)
        return o
    M.spamatvar_forward, M.spamatvar_forward_bits = spy_f, spy_b
    try:
        yield
    finally:
        M.spamatvar_forward, M.spamatvar_forward_bits = orig_f, orig_b
