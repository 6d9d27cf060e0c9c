"""Generate tests/golden/spamat_ref_*.npz by RUNNING THE REFERENCE'S OWN KERNELS on an MI355X.

    oracle/ref_build.sh                               # build container: reference .cu/.cpp -> oracle/_ref/*.so (unmodified)
    gpurun -- python tests/golden/make_spamat_ref_golden.py gpurun_out/spamat_ref
    cp gpurun_out/spamat_ref/*.npz tests/golden/

Each fixture holds, per case of spamat_ref_cases.py, the OUTPUTS of
  sparse_matching_cuda_forward / _backward   (SM_cuda.cpp:7-27  -> SM_kernel.cu:22-125, 143-195, 300-355)
  sparse_var_cuda_forward / _backward        (SV_cuda.cpp:7-32  -> SV_kernel.cu:22-124, 142-325)
called with the protocol of functions/SpaMat.py / SpaVar.py (zero-filled outputs), plus a CRC32
of the seeded inputs.  SpaVar runs twice: around mu = SpaMat's output (what the net does,
SparseDenseNetRefinementMask.py:188-192) and around mu = output + N(0,1).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

import spamat_ref_cases as K  # noqa: E402


def run_case(name, dev):
    import torch
    from oracle import ref
    x = K.make_inputs(name)
    t = {k: torch.from_numpy(v).to(dev) for k, v in x.items() if k != "max_disp"}
    D = x["max_disp"]
    out, ssum, mx = ref.spamat_forward(t["L"], t["R"], t["rm"], t["tm"], D)
    gl, gr = ref.spamat_backward(t["L"], t["R"], t["rm"], t["tm"], out, ssum, mx, t["g"], D)
    res = dict(out=out, ssum=ssum, mx=mx, gl=gl, gr=gr)
    for tag, mu in (("v0", out.clone()), ("v1", (out + t["mu_noise"]).contiguous())):
        v, vs, vm = ref.spavar_forward(t["L"], t["R"], t["rm"], t["tm"], mu, D)
        vgl, vgr, vgd = ref.spavar_backward(t["L"], t["R"], t["rm"], t["tm"], mu, v, vs, vm, t["g"], D)
        res.update({tag + "_var": v, tag + "_ssum": vs, tag + "_mx": vm,
                    tag + "_gl": vgl, tag + "_gr": vgr, tag + "_gd": vgd})
    res = {k: v.cpu().numpy() for k, v in res.items()}
    res["crc"] = K.crc(x["L"], x["R"], x["rm"], x["tm"], x["g"], x["mu_noise"])
    res["max_disp"] = np.int64(D)
    return res


def provenance():
    import subprocess
    import torch
    p = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0),
         "arch": torch.cuda.get_device_properties(0).gcnArchName}
    try:
        p["hipcc"] = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True,
                                    text=True).stdout.splitlines()[0]
    except Exception as e:  # pragma: no cover
        p["hipcc"] = repr(e)
    for n in ("SpaMat.so", "SpaVar.so"):
        with open(os.path.join(ROOT, "oracle", "_ref", n), "rb") as f:
            p["sha256_" + n] = hashlib.sha256(f.read()).hexdigest()
    return p


def main(out_dir):
    import torch
    from oracle import ref
    assert torch.cuda.is_available() and ref.available(), "needs the MI355X and oracle/_ref/*.so"
    dev = torch.device("cuda:0")
    os.makedirs(out_dir, exist_ok=True)
    prov = provenance()
    print(prov)
    for group, names in K.GROUPS.items():
        blob = {"provenance": np.array([f"{k}={v}" for k, v in prov.items()])}
        for name in names:
            for k, v in run_case(name, dev).items():
                blob[f"{name}/{k}"] = v
            print("ran", name, flush=True)
        path = os.path.join(out_dir, f"spamat_ref_{group}.npz")
        np.savez_compressed(path, **blob)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "spamat_ref"))
