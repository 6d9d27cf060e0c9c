"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU checker for the DecNet hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under
``decnet_amd/`` does (tests/test_no_oracle_in_product.py enforces it).

* ``spamat_oracle.c``  literal C restatement of SM_kernel.cu / SV_kernel.cu
  (see that file's header for the reference line map).  Pinned by reference execution:
  tests/golden/spamat_ref_*.npz are outputs of the reference's own kernels on an MI355X.
* ``ref_build.sh`` / ``ref.py``  build (hipcc -x hip, unmodified sources, into oracle/_ref/) and
  load the REFERENCE's own SpaMat / SpaVar extensions; GPU only.
* ``stage0.py``        torch-CPU restatement of the stage-0 dense path
  (submodule.py:389-390, 479-522, 608-662, 766-777), pinned by golden vectors
  generated from the imported reference (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_F = ctypes.POINTER(ctypes.c_float)


def build_ref(force=False, reference="/root/reference"):
    """oracle/_ref/SpaMat.so, SpaVar.so from the reference's own sources (oracle/ref_build.sh); only where the
    reference tree exists (the build container) -- the GPU box uses the prebuilt files.  -> True if present."""
    outs = [os.path.join(_HERE, "_ref", n) for n in ("SpaMat.so", "SpaVar.so")]
    if os.path.isdir(os.path.join(reference, "modules", "SparseMatching", "src")) and \
            (force or not all(os.path.exists(o) for o in outs)):
        subprocess.check_call([os.path.join(_HERE, "ref_build.sh")], env=dict(os.environ, DECNET_REFERENCE=reference))
    return all(os.path.exists(o) for o in outs)


def build(force=False):
    """Compile liboracle.so / liboracle_nofma.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "spamat_oracle.c")
    need = force or not all(
        os.path.exists(os.path.join(_HERE, n)) and os.path.getmtime(os.path.join(_HERE, n)) >= os.path.getmtime(src)
        for n in ("liboracle.so", "liboracle_nofma.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _lib(fma=True):
    name = "liboracle.so" if fma else "liboracle_nofma.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        i = ctypes.c_int
        lib.oracle_spamat_forward.argtypes = [_F] * 7 + [i] * 5
        lib.oracle_spamat_backward.argtypes = [_F] * 10 + [i] * 5
        lib.oracle_spavar_forward.argtypes = [_F] * 8 + [i] * 5
        lib.oracle_spavar_backward.argtypes = [_F] * 12 + [i] * 5
        _LIBS[name] = lib
    return _LIBS[name]


def _np(x):
    if hasattr(x, "detach"):  # torch tensor
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_F)


def num_threads():
    return int(_lib().oracle_num_threads())


def set_num_threads(n):
    """OpenMP threads of the C restatement (bench.py's one-thread cpu_baseline figure)."""
    for fma in (True, False):
        _lib(fma).oracle_set_num_threads(int(n))


def spamat_forward(ref, tar, rmask, tmask, max_disp, fma=True):
    """-> (output, sum_similarities, max_cost), each [B,H,W] float32 (zeros where skipped)."""
    ref, tar, rmask, tmask = map(_np, (ref, tar, rmask, tmask))
    B, C, H, W = ref.shape
    out, ssum, mx = (np.zeros((B, H, W), np.float32) for _ in range(3))
    _lib(fma).oracle_spamat_forward(_p(ref), _p(tar), _p(rmask), _p(tmask), _p(out), _p(ssum),
                                    _p(mx), B, C, H, W, int(max_disp))
    return out, ssum, mx


def spamat_backward(ref, tar, rmask, tmask, out, ssum, mx, grad_out, max_disp, fma=True):
    """-> (grad_ref, grad_tar), each [B,C,H,W]."""
    ref, tar, rmask, tmask, out, ssum, mx, grad_out = map(
        _np, (ref, tar, rmask, tmask, out, ssum, mx, grad_out))
    B, C, H, W = ref.shape
    gl, gr = np.zeros_like(ref), np.zeros_like(tar)
    _lib(fma).oracle_spamat_backward(_p(ref), _p(tar), _p(rmask), _p(tmask), _p(out), _p(ssum),
                                     _p(mx), _p(grad_out), _p(gl), _p(gr), B, C, H, W,
                                     int(max_disp))
    return gl, gr


def spavar_forward(ref, tar, rmask, tmask, disparity, max_disp, fma=True):
    ref, tar, rmask, tmask, disparity = map(_np, (ref, tar, rmask, tmask, disparity))
    B, C, H, W = ref.shape
    out, ssum, mx = (np.zeros((B, H, W), np.float32) for _ in range(3))
    _lib(fma).oracle_spavar_forward(_p(ref), _p(tar), _p(rmask), _p(tmask), _p(disparity), _p(out),
                                    _p(ssum), _p(mx), B, C, H, W, int(max_disp))
    return out, ssum, mx


def spavar_backward(ref, tar, rmask, tmask, disparity, out, ssum, mx, grad_out, max_disp, fma=True):
    """-> (grad_ref, grad_tar, grad_disparity)."""
    ref, tar, rmask, tmask, disparity, out, ssum, mx, grad_out = map(
        _np, (ref, tar, rmask, tmask, disparity, out, ssum, mx, grad_out))
    B, C, H, W = ref.shape
    gl, gr, gd = np.zeros_like(ref), np.zeros_like(tar), np.zeros_like(disparity)
    _lib(fma).oracle_spavar_backward(_p(ref), _p(tar), _p(rmask), _p(tmask), _p(disparity), _p(out),
                                     _p(ssum), _p(mx), _p(grad_out), _p(gl), _p(gr), _p(gd),
                                     B, C, H, W, int(max_disp))
    return gl, gr, gd
