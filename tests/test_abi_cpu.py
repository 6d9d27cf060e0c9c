"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/decnet_hip.h declares; argument validation happens before anything is launched;
the product package never touches oracle/."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "decnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(decnet_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from decnet_amd import build
    path = build.build()                      # hipcc cross-compiles for gfx950 without a GPU
    return ctypes.CDLL(path)


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("decnet_spamat_forward", "decnet_spamat_backward", "decnet_spavar_forward",
              "decnet_spavar_backward", "decnet_spamatvar_forward", "decnet_costvol_forward",
              "decnet_conv3d_bn_act", "decnet_conv3d_cout1_softargmax", "decnet_version"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), "libdecnet_hip.so does not export %s" % s


def test_python_binding_covers_every_declared_symbol():
    from decnet_amd import _lib
    assert sorted(list(_lib.SIGNATURES) + ["decnet_version"]) == declared_symbols()


def test_version_and_argument_validation_without_gpu(lib):
    lib.decnet_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.decnet_version()
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.decnet_spamat_forward.argtypes = [P] * 7 + [I] * 5 + [P]
    one = ctypes.c_void_p(64)
    # null pointer / bad shape are rejected before any HIP call
    assert lib.decnet_spamat_forward(None, one, one, one, one, one, one, 1, 1, 1, 1, 1, None) == -1
    assert lib.decnet_spamat_forward(one, one, one, one, one, one, one, 1, 1, 0, 1, 1, None) == -2
    assert lib.decnet_spamat_forward(one, one, one, one, one, one, one, 1, 1, 1, 1, 0, None) == -2
    assert lib.decnet_conv3d_packed_cout(216) == 224
    assert lib.decnet_conv3d_packed_cout(300) == -1


def test_product_never_imports_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "decnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt \
                        or "spamat_oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, "product code references the oracle: %s" % bad


def test_cpu_tensors_are_refused():
    import torch
    import decnet_amd
    x = torch.zeros(1, 2, 3, 4)
    m = torch.ones(1, 3, 4)
    with pytest.raises(decnet_amd.DecnetHipError):
        decnet_amd.SpaMat()(x, x, m, m, 2)


def test_conv_algo_choice_and_sizes(lib, monkeypatch):
    """Host logic of the Conv3d algorithm choice (decnet_amd.stage0.conv_algo) and the size helpers of the
    three Winograd variants (no GPU call)."""
    from decnet_amd.stage0 import conv_algo, WINO_VARIANT
    monkeypatch.delenv("DECNET_CONV_ALGO", raising=False)
    assert conv_algo(8) == "winograd444" and conv_algo(10) == "winograd444"      # stage 0 of configs 1-5
    assert conv_algo(2) == "winograd4" and conv_algo(5) == "winograd4" and conv_algo(1) == "winograd4"
    monkeypatch.setenv("DECNET_CONV_ALGO", "direct")
    assert conv_algo(8) == "direct"
    monkeypatch.setenv("DECNET_CONV_ALGO", "fft")
    with pytest.raises(ValueError):
        conv_algo(8)
    lib.decnet_conv3d_wino_weight_floats.restype = ctypes.c_size_t
    lib.decnet_conv3d_wino_workspace_floats.restype = ctypes.c_size_t
    for name, pts, (od, oh) in (("winograd", 64, (2, 2)), ("winograd4", 144, (2, 4)), ("winograd444", 216, (4, 4))):
        v = WINO_VARIANT[name]
        assert lib.decnet_conv3d_wino_weight_floats(216, v) >= pts * 224 * 224
        tiles = 8 * -(-8 // od) * -(-20 // oh) * -(-36 // oh)
        assert lib.decnet_conv3d_wino_workspace_floats(8, 8, 20, 36, 216, 216, v) == pts * tiles * (224 + 224)
    assert lib.decnet_conv3d_wino_weight_floats(216, 3) == 0


def test_hip_lib_override_fails_loudly_when_the_file_is_missing(tmp_path):
    """DECNET_HIP_LIB points the loader at another build of the library (tools/dev_obj.sh experiment builds); a path that
    does not exist is an error at first use, never a silent fallback to the default build or to the CPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from decnet_amd import _lib\n"
            "try:\n"
            "    _lib.lib()\n"
            "except Exception as e:\n"
            "    print('RAISED', type(e).__name__)\n"
            "else:\n"
            "    print('LOADED')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DECNET_HIP_LIB=str(tmp_path / "nope.so")),
                       capture_output=True, text=True, timeout=120)
    assert "RAISED" in r.stdout and "LOADED" not in r.stdout, r.stdout + r.stderr
