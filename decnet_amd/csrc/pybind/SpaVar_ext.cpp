// decnet_amd/csrc/pybind/SpaVar_ext.cpp -- the compiled module `SpaVar` the reference imports
// (modules/SparseVar/functions/SpaVar.py:4), for the MI355X.  Replaces SV_cuda.cpp:7-38; see torch_boundary.h.
#include "torch_boundary.h"

namespace db = decnet_boundary;

// SV_cuda.cpp:7-17 (get_max_cost + sparse_var_forward, SV_kernel.cu:329-359)
static int sparse_var_cuda_forward(at::Tensor ref_feas, at::Tensor tar_feas, at::Tensor ref_mask, at::Tensor tar_mask,
                                   at::Tensor disparity, at::Tensor output, at::Tensor sum_similarities,
                                   at::Tensor max_cost, int max_disp) {
    const db::Dims d = db::check_feats(ref_feas, tar_feas);
    db::check_plane(ref_mask, "ref_mask", ref_feas, d);
    db::check_plane(tar_mask, "tar_mask", ref_feas, d);
    db::check_plane(disparity, "disparity", ref_feas, d);
    db::check_plane(output, "output", ref_feas, d);
    db::check_plane(sum_similarities, "sum_similarities", ref_feas, d);
    db::check_plane(max_cost, "max_cost", ref_feas, d);
    TORCH_CHECK(max_disp >= 1, "max_disp must be >= 1, got ", max_disp);
    db::DeviceGuard guard(ref_feas.device());
    const int rc = decnet_spavar_forward(ref_feas.data_ptr<float>(), tar_feas.data_ptr<float>(),
                                         ref_mask.data_ptr<float>(), tar_mask.data_ptr<float>(),
                                         disparity.data_ptr<float>(), output.data_ptr<float>(),
                                         sum_similarities.data_ptr<float>(), max_cost.data_ptr<float>(), d.B, d.C,
                                         d.H, d.W, max_disp, db::current_stream(ref_feas));
    db::check_rc(rc, "decnet_spavar_forward");
    return 1;
}

// SV_cuda.cpp:19-32 (sparse_var_{ref,tar,dis}_backward, SV_kernel.cu:361-410)
static int sparse_var_cuda_backward(at::Tensor ref_feas, at::Tensor tar_feas, at::Tensor ref_mask, at::Tensor tar_mask,
                                    at::Tensor disparity, at::Tensor output, at::Tensor sum_similarities,
                                    at::Tensor max_cost, at::Tensor grad_output, at::Tensor grad_ref_feas,
                                    at::Tensor grad_tar_feas, at::Tensor grad_disparity, int max_disp) {
    const db::Dims d = db::check_feats(ref_feas, tar_feas);
    db::check_plane(ref_mask, "ref_mask", ref_feas, d);
    db::check_plane(tar_mask, "tar_mask", ref_feas, d);
    db::check_plane(disparity, "disparity", ref_feas, d);
    db::check_plane(output, "output", ref_feas, d);
    db::check_plane(sum_similarities, "sum_similarities", ref_feas, d);
    db::check_plane(max_cost, "max_cost", ref_feas, d);
    db::check_plane(grad_output, "grad_output", ref_feas, d);
    db::check_like_feats(grad_ref_feas, "grad_ref_feas", ref_feas);
    db::check_like_feats(grad_tar_feas, "grad_tar_feas", ref_feas);
    db::check_plane(grad_disparity, "grad_disparity", ref_feas, d);
    TORCH_CHECK(max_disp >= 1, "max_disp must be >= 1, got ", max_disp);
    db::DeviceGuard guard(ref_feas.device());
    const int rc = decnet_spavar_backward(ref_feas.data_ptr<float>(), tar_feas.data_ptr<float>(),
                                          ref_mask.data_ptr<float>(), tar_mask.data_ptr<float>(),
                                          disparity.data_ptr<float>(), output.data_ptr<float>(),
                                          sum_similarities.data_ptr<float>(), max_cost.data_ptr<float>(),
                                          grad_output.data_ptr<float>(), grad_ref_feas.data_ptr<float>(),
                                          grad_tar_feas.data_ptr<float>(), grad_disparity.data_ptr<float>(), d.B,
                                          d.C, d.H, d.W, max_disp, db::current_stream(ref_feas));
    db::check_rc(rc, "decnet_spavar_backward");
    return 1;
}

PYBIND11_MODULE(SpaVar, m) {
    m.doc() = "DecNet SparseVar on MI355X (gfx950): drop-in for the module built from SV_cuda.cpp";
    m.def("sparse_var_cuda_forward", &sparse_var_cuda_forward, "sparse var forward (HIP, gfx950)");
    m.def("sparse_var_cuda_backward", &sparse_var_cuda_backward, "sparse var backward (HIP, gfx950)");
    m.def("decnet_version", [] { return std::string(decnet_version()); });
}
