"""Build libdecnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m decnet_amd.build [--force] [--pybind]

The library lands in decnet_amd/lib/ (git-ignored, but it travels with gpurun snapshots).
--pybind also builds the two compiled torch modules the reference's autograd Functions import
(`from ..build.lib import SpaMat`, modules/SparseMatching/functions/SpaMat.py:4; SpaVar likewise) into
decnet_amd/modules/Sparse{Matching,Var}/build/lib/ -- the place compile.sh:24-28 leaves the reference's own.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdecnet_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


# per-source extra flags
#   spamat_mfma.hip: -fno-honor-nans drops the canonicalising v_max that fmaxf otherwise needs on
#   every MFMA result (the kernel's VALU passes are the bottleneck); see the file header.
#   -fno-slp-vectorize (round 5): the SLP vectoriser packs the softmax passes into v_pk_{add,mul,fma}_f32 pairs and pays
#   for them with ~140 v_mov per 16-tile chunk (operands have to sit in aligned register pairs); packed fp32 issues no
#   faster than two scalar ops on this chip.  Stage 3, mask density 0.4: 0.169 -> 0.151 ms (profiles/r05d_*).
EXTRA_FLAGS = {"spamat_mfma.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ (one object each) and link one shared library."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    common = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall",
              "-Wno-unused-function"]
    procs, objs = [], []
    live = {os.path.basename(src) + ".o" for src in sources()}
    for old in glob.glob(os.path.join(obj_dir, "*.o")):      # objects whose source is gone (pruned experiments)
        if os.path.basename(old) not in live:
            os.remove(old)
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = common + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def build_ubench(force=False):
    """tools/ubench/softmax_rate.bin and bwd_tile_rate.bin: the issue-floor microbenchmarks bench.py runs beside the
    dense cost-volume kernel and the dense-row backward (stand-alone HIP programs; child processes of bench.py, never
    linked into the library).  Returns the forward one's path."""
    root = os.path.dirname(HERE)
    outs = []
    for name, flags in (("softmax_rate", ["-fno-honor-nans"]), ("bwd_tile_rate", [])):
        src = os.path.join(root, "tools", "ubench", name + ".hip")
        out = os.path.join(root, "tools", "ubench", name + ".bin")
        if not os.path.exists(src):
            outs.append(None)
            continue
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-O3", "-w"] + flags + [src, "-o", out])
        outs.append(out)
    return outs[0]


# ---- the compiled drop-in modules (SURVEY.md 8b "Build boundary") --------------------------------------------
PYBIND_SRC = os.path.join(CSRC, "pybind")
PYBIND_MODULES = {          # module name -> (source, directory of the reference layout it lands in)
    "SpaMat": ("SpaMat_ext.cpp", os.path.join(HERE, "modules", "SparseMatching", "build", "lib")),
    "SpaVar": ("SpaVar_ext.cpp", os.path.join(HERE, "modules", "SparseVar", "build", "lib")),
}


def pybind_path(name):
    return os.path.join(PYBIND_MODULES[name][1], name + ".so")


def build_pybind(force=False, verbose=False):
    """SpaMat.so / SpaVar.so: pybind11 modules with the at::Tensor signatures of SM_cuda.cpp:7-33 / SV_cuda.cpp:7-38
    over the C ABI.  Host-only C++ (g++), linked against libdecnet_hip.so; the library is found through an rpath
    relative to the module ($ORIGIN) and, for a copy placed into a reference checkout, the absolute build path."""
    import sysconfig
    import torch
    build(force=False)
    troot = os.path.dirname(torch.__file__)
    inc = ["-I" + os.path.join(troot, "include"), "-I" + os.path.join(troot, "include", "torch", "csrc", "api", "include"),
           "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include",
           "-I" + os.path.join(os.path.dirname(HERE), "include")]
    defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-DTORCH_API_INCLUDE_EXTENSION_H",
            # the libstdc++ string ABI torch itself was built with (the other one links, then fails at import with
            # undefined c10 symbols)
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    stamp = os.path.join(PYBIND_SRC, ".built_for")             # torch version + ABI the modules were compiled against
    want = "%s abi%d" % (torch.__version__, int(torch._C._GLIBCXX_USE_CXX11_ABI))
    try:
        with open(stamp) as f:
            force = force or f.read().strip() != want
    except OSError:
        force = True
    cxx = os.environ.get("CXX", "g++")
    deps = [os.path.join(PYBIND_SRC, "torch_boundary.h"), os.path.join(os.path.dirname(HERE), "include", "decnet_hip.h")]
    procs, outs = [], []
    for name, (src, out_dir) in PYBIND_MODULES.items():
        src = os.path.join(PYBIND_SRC, src)
        out = pybind_path(name)
        outs.append(out)
        if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps + [src]):
            continue
        os.makedirs(out_dir, exist_ok=True)
        cmd = [cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-attributes"] + defs + \
            ["-DTORCH_EXTENSION_NAME=" + name] + inc + [src, "-o", out + ".tmp",
             "-L" + LIB_DIR, "-ldecnet_hip", "-L" + os.path.join(troot, "lib"), "-lc10", "-ltorch", "-ltorch_cpu",
             "-ltorch_python", "-lc10_hip", "-ltorch_hip",
             "-Wl,-rpath,$ORIGIN/../../../../lib", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath," + os.path.join(troot, "lib")]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, out, subprocess.Popen(cmd)))
    for cmd, out, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.replace(out + ".tmp", out)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--pybind" in sys.argv:
        for p in build_pybind(force="--force" in sys.argv, verbose=True):
            print(p)
