// decnet_amd/csrc/pybind/torch_boundary.h -- the torch side of the drop-in boundary.
//
// The reference's autograd Functions import compiled pybind modules:
//     from ..build.lib import SpaMat      (modules/SparseMatching/functions/SpaMat.py:4)
//     from ..build.lib import SpaVar      (modules/SparseVar/functions/SpaVar.py:4)
// built by setup.py:7-19 / compile.sh:24-28 from SM_cuda.cpp:7-33 and SV_cuda.cpp:7-38.  SpaMat_ext.cpp and
// SpaVar_ext.cpp in this directory are those two modules for the MI355X: the same module names, function names,
// positional at::Tensor / int arguments and return value (1), forwarding to the C ABI of libdecnet_hip.so
// (include/decnet_hip.h).  Host-only C++ (no device code): compiled with g++ and linked against the library.
//
// What the reference's modules do not do and these do (SURVEY.md 8b "Preconditions / errors", "Threading / streams"):
//   * TORCH_CHECK of device, dtype, contiguity and shape of every tensor (the reference passes data_ptr<float>()
//     of whatever it is given, SM_kernel.cu:369-376);
//   * a device guard on ref_feas.device() and the CURRENT stream of that device (the reference launches on the
//     legacy default stream and relies on the Python side's torch.cuda.device_of, functions/SpaMat.py:24);
//   * a non-zero return code of the C ABI becomes a c10::Error (Python RuntimeError); the reference always returns 1.
#pragma once
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "decnet_hip.h"

namespace decnet_boundary {

struct Dims { int B, C, H, W; };

inline void check_f32_cuda(const at::Tensor &t, const char *name, const at::Device &dev) {
    TORCH_CHECK(t.defined(), name, " is undefined");
    TORCH_CHECK(t.is_cuda(), name, " is on ", t.device(), ": the MI355X HIP path has no CPU fallback");
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

// feature maps [B,C,H,W]
inline Dims check_feats(const at::Tensor &ref, const at::Tensor &tar) {
    TORCH_CHECK(ref.defined() && ref.is_cuda(), "ref_feas must be a tensor on the GPU (no CPU fallback)");
    TORCH_CHECK(ref.dim() == 4, "ref_feas must be [B,C,H,W], got ", ref.sizes());
    const at::Device dev = ref.device();
    check_f32_cuda(ref, "ref_feas", dev);
    check_f32_cuda(tar, "tar_feas", dev);
    TORCH_CHECK(tar.sizes() == ref.sizes(), "tar_feas has shape ", tar.sizes(), ", expected ", ref.sizes());
    TORCH_CHECK(ref.numel() < (int64_t(1) << 31), "index space beyond 2^31 elements");
    return Dims{(int)ref.size(0), (int)ref.size(1), (int)ref.size(2), (int)ref.size(3)};
}

// per-pixel planes [B,H,W]
inline void check_plane(const at::Tensor &t, const char *name, const at::Tensor &ref, const Dims &d) {
    check_f32_cuda(t, name, ref.device());
    TORCH_CHECK(t.dim() == 3 && t.size(0) == d.B && t.size(1) == d.H && t.size(2) == d.W, name, " has shape ",
                t.sizes(), ", expected [", d.B, ", ", d.H, ", ", d.W, "]");
}

inline void check_like_feats(const at::Tensor &t, const char *name, const at::Tensor &ref) {
    check_f32_cuda(t, name, ref.device());
    TORCH_CHECK(t.sizes() == ref.sizes(), name, " has shape ", t.sizes(), ", expected ", ref.sizes());
}

inline void check_rc(int rc, const char *what) {
    TORCH_CHECK(rc == 0, what, rc < 0 ? ": rejected arguments, DECNET_ERR code " : ": HIP launch failed, hipError_t ", rc);
}

// the stream torch work on `t`'s device is currently enqueued on, as the void* the C ABI takes
inline void *current_stream(const at::Tensor &t) {
    return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;

}  // namespace decnet_boundary
