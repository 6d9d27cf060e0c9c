"""Build libdecnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m decnet_amd.build [--force]

The library lands in decnet_amd/lib/ (git-ignored, but it travels with gpurun snapshots).
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
EXTRA_FLAGS = {"spamat_mfma.hip": ["-fno-honor-nans"]}


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
    """tools/ubench/softmax_rate.bin: the VALU-floor microbenchmark bench.py runs beside the dense cost-volume
    kernel (a stand-alone HIP program; a child process of bench.py, never linked into the library)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "ubench", "softmax_rate.hip")
    out = os.path.join(root, "tools", "ubench", "softmax_rate.bin")
    if not os.path.exists(src):
        return None
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-O3", "-w", "-fno-honor-nans", src, "-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
