"""The compiled drop-in modules `SpaMat` / `SpaVar` (decnet_amd/csrc/pybind/, built by `python -m decnet_amd.build
--pybind`): what the reference's `from ..build.lib import SpaMat` (modules/SparseMatching/functions/SpaMat.py:4) gets on
the MI355X.  Signatures: SM_cuda.cpp:7-33, SV_cuda.cpp:7-38.

CPU tests: the modules exist, load the way oracle/ref.py loads the reference's own build, export the reference's names
and refuse what they cannot run.  GPU tests: argument conventions (numpy.int64, side streams, a second layout that only
contains a copy of the module); the 22 reference-kernel fixtures run through them in tests/test_spamat_ref.py.
"""
import importlib.machinery
import importlib.util
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from decnet_amd import build as B


def load(name):
    """Load the compiled module by file, under a spec name of its own: pybind11 >= 3 caches initialised modules per
    interpreter by ``spec.name``, and the reference's own build (oracle/_ref/, loaded by oracle/ref.py in the same
    pytest process) has the same plain names.  Checked: what comes back is this repository's module."""
    path = B.pybind_path(name)
    assert os.path.exists(path), "%s missing: python -m decnet_amd.build --pybind" % path
    spec_name = "decnet_compiled_dropin." + name
    loader = importlib.machinery.ExtensionFileLoader(spec_name, path)
    spec = importlib.util.spec_from_loader(spec_name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    assert hasattr(mod, "decnet_version") and os.path.realpath(mod.__file__) == os.path.realpath(path), mod
    return mod


def test_modules_sit_where_the_reference_build_leaves_them():
    # compile.sh:24-28: `python setup.py build; cp -r build/lib* build/lib` inside modules/SparseMatching
    assert B.pybind_path("SpaMat").endswith(os.path.join("modules", "SparseMatching", "build", "lib", "SpaMat.so"))
    assert B.pybind_path("SpaVar").endswith(os.path.join("modules", "SparseVar", "build", "lib", "SpaVar.so"))


def test_exports_and_version():
    sm, sv = load("SpaMat"), load("SpaVar")
    for f in ("sparse_matching_cuda_forward", "sparse_matching_cuda_backward"):
        assert callable(getattr(sm, f))
    for f in ("sparse_var_cuda_forward", "sparse_var_cuda_backward"):
        assert callable(getattr(sv, f))
    import decnet_amd
    assert sm.decnet_version() == decnet_amd.version()          # the same library the ctypes path binds


def test_reference_import_statement():
    from decnet_amd.modules.SparseMatching.build.lib import SpaMat
    from decnet_amd.modules.SparseVar.build.lib import SpaVar
    assert SpaMat.__name__.endswith("SpaMat") and SpaVar.__name__.endswith("SpaVar")
    assert SpaMat.__file__ == B.pybind_path("SpaMat") and hasattr(SpaMat, "decnet_version")


def test_same_plain_name_as_the_reference_build_cannot_be_confused():
    """The hazard the loaders above are built around, demonstrated: under ONE spec name pybind11's module cache hands
    back the first module whatever file the second load names; under distinct names both files load."""
    if not os.path.exists(B.pybind_path("SpaVar")):
        pytest.skip("modules not built")
    a = load("SpaVar")
    path = B.pybind_path("SpaMat")
    loader = importlib.machinery.ExtensionFileLoader("decnet_compiled_dropin.SpaVar", B.pybind_path("SpaVar"))
    again = importlib.util.module_from_spec(importlib.util.spec_from_loader("decnet_compiled_dropin.SpaVar", loader))
    assert again is a or again.__file__ == a.__file__           # same name -> the cached module
    b = load("SpaMat")
    assert b is not a and b.__file__ == path


def test_cpu_tensors_are_refused_not_run():
    sm, sv = load("SpaMat"), load("SpaVar")
    f, m = torch.zeros(1, 8, 4, 20), torch.zeros(1, 4, 20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sm.sparse_matching_cuda_forward(f, f, m, m, m, m, m, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sm.sparse_matching_cuda_backward(f, f, m, m, m, m, m, m, f, f, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sv.sparse_var_cuda_forward(f, f, m, m, m, m, m, m, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sv.sparse_var_cuda_backward(f, f, m, m, m, m, m, m, m, f, f, m, 8)
    with pytest.raises(TypeError):                                # pybind: wrong arity, as the reference's module
        sm.sparse_matching_cuda_forward(f, f, m, m, m, m, 8)


# ----------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _case(dev, B_=2, C=8, H=5, W=300, seed=3, p=0.6):
    g = torch.Generator().manual_seed(seed)
    L = torch.relu(torch.randn(B_, C, H, W, generator=g)).to(dev)
    R = torch.relu(torch.randn(B_, C, H, W, generator=g)).to(dev)
    rm = (torch.rand(B_, H, W, generator=g) < p).float().to(dev)
    tm = (torch.rand(B_, H, W, generator=g) < p).float().to(dev)
    go = torch.randn(B_, H, W, generator=g).to(dev)
    return L, R, rm, tm, go


@pytest.mark.gpu
def test_compiled_module_equals_ctypes_path_bit_for_bit(dev):
    """Same library, same kernels: the two bindings must agree exactly (forward, backward, SpaVar), also with
    max_disp as numpy.int64 (the model passes numpy integers) and on a non-default stream."""
    from decnet_amd.ext import SpaMat as ESM, SpaVar as ESV
    sm, sv = load("SpaMat"), load("SpaVar")
    L, R, rm, tm, go = _case(dev)
    D = np.int64(216)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o, s, m = (torch.full_like(rm, 7.0) for _ in range(3))
        assert sm.sparse_matching_cuda_forward(L, R, rm, tm, o, s, m, D) == 1
        gl, gr = torch.full_like(L, 7.0), torch.full_like(R, 7.0)
        assert sm.sparse_matching_cuda_backward(L, R, rm, tm, o, s, m, go, gl, gr, D) == 1
        v, vs, vm = (torch.full_like(rm, 7.0) for _ in range(3))
        assert sv.sparse_var_cuda_forward(L, R, rm, tm, o, v, vs, vm, D) == 1
        vgl, vgr, vgd = torch.full_like(L, 7.0), torch.full_like(R, 7.0), torch.full_like(rm, 7.0)
        assert sv.sparse_var_cuda_backward(L, R, rm, tm, o, v, vs, vm, go, vgl, vgr, vgd, D) == 1
    side.synchronize()
    o2, s2, m2 = (torch.empty_like(rm) for _ in range(3))
    assert ESM.sparse_matching_cuda_forward(L, R, rm, tm, o2, s2, m2, D) == 1
    gl2, gr2 = torch.empty_like(L), torch.empty_like(R)
    assert ESM.sparse_matching_cuda_backward(L, R, rm, tm, o2, s2, m2, go, gl2, gr2, D) == 1
    v2, vs2, vm2 = (torch.empty_like(rm) for _ in range(3))
    assert ESV.sparse_var_cuda_forward(L, R, rm, tm, o2, v2, vs2, vm2, D) == 1
    vgl2, vgr2, vgd2 = torch.empty_like(L), torch.empty_like(R), torch.empty_like(rm)
    assert ESV.sparse_var_cuda_backward(L, R, rm, tm, o2, v2, vs2, vm2, go, vgl2, vgr2, vgd2, D) == 1
    torch.cuda.synchronize()
    for a, b in ((o, o2), (s, s2), (m, m2), (gl, gl2), (gr, gr2), (v, v2), (vs, vs2), (vm, vm2), (vgl, vgl2),
                 (vgr, vgr2), (vgd, vgd2)):
        assert torch.equal(a, b)
    assert float(o.abs().max()) > 1.0 and not (o == 7.0).any()


@pytest.mark.gpu
def test_checks_replace_the_reference_s_silent_out_of_bounds(dev):
    sm = load("SpaMat")
    L, R, rm, tm, go = _case(dev, H=3, W=40)
    o, s, m = (torch.empty_like(rm) for _ in range(3))
    with pytest.raises(RuntimeError, match="tar_feas has shape"):
        sm.sparse_matching_cuda_forward(L, R[:, :4].contiguous(), rm, tm, o, s, m, 16)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        sm.sparse_matching_cuda_forward(L.transpose(2, 3), R, rm, tm, o, s, m, 16)
    with pytest.raises(RuntimeError, match="must be float32"):
        sm.sparse_matching_cuda_forward(L, R, rm.double(), tm, o, s, m, 16)
    with pytest.raises(RuntimeError, match="ref_mask has shape"):
        sm.sparse_matching_cuda_forward(L, R, rm[:, :2].contiguous(), tm, o, s, m, 16)
    with pytest.raises(RuntimeError, match="max_disp"):
        sm.sparse_matching_cuda_forward(L, R, rm, tm, o, s, m, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sm.sparse_matching_cuda_forward(L, R, rm.cpu(), tm, o, s, m, 16)


@pytest.mark.gpu
def test_a_reference_shaped_checkout_imports_a_copy(dev, tmp_path):
    """A package tree laid out like the reference's (modules/SparseMatching/{functions,build/lib}) that holds nothing
    of this repository but a COPY of SpaMat.so: the Function file's relative import (written here, two lines) finds
    it, the module finds libdecnet_hip.so through its absolute rpath, and an autograd step runs -- in a fresh
    interpreter, so nothing of decnet_amd is imported."""
    root = tmp_path / "checkout"
    pk = root / "modules" / "SparseMatching"
    (pk / "functions").mkdir(parents=True)
    (pk / "build" / "lib").mkdir(parents=True)
    for d in (root / "modules", pk, pk / "functions"):
        (d / "__init__.py").write_text("")
    shutil.copy(B.pybind_path("SpaMat"), pk / "build" / "lib" / "SpaMat.so")
    (pk / "functions" / "probe.py").write_text("from ..build.lib import SpaMat\n")
    script = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r)
        from modules.SparseMatching.functions.probe import SpaMat
        assert "decnet_amd" not in sys.modules
        g = torch.Generator().manual_seed(5)
        L = torch.relu(torch.randn(1, 8, 4, 250, generator=g)).cuda()
        R = torch.relu(torch.randn(1, 8, 4, 250, generator=g)).cuda()
        m = torch.ones(1, 4, 250).cuda()
        o, s, mx = (torch.zeros_like(m) for _ in range(3))
        assert SpaMat.sparse_matching_cuda_forward(L, R, m, m, o, s, mx, 216) == 1
        gl, gr = torch.zeros_like(L), torch.zeros_like(R)
        assert SpaMat.sparse_matching_cuda_backward(L, R, m, m, o, s, mx, torch.ones_like(m), gl, gr, 216) == 1
        torch.cuda.synchronize()
        assert float(o.max()) > 1.0 and float(gl.abs().max()) > 0
        print("ok", SpaMat.__file__)
    """ % str(root))
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, cwd=str(tmp_path),
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ok" in r.stdout and str(root) in r.stdout
