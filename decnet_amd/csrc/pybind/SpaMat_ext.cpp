// decnet_amd/csrc/pybind/SpaMat_ext.cpp -- the compiled module `SpaMat` the reference imports
// (modules/SparseMatching/functions/SpaMat.py:4), for the MI355X.  Replaces SM_cuda.cpp:7-33; see torch_boundary.h.
#include "torch_boundary.h"

namespace db = decnet_boundary;

// SM_cuda.cpp:7-15 (get_max_cost + sparse_matching_forward, SM_kernel.cu:359-387)
static int sparse_matching_cuda_forward(at::Tensor ref_feas, at::Tensor tar_feas, at::Tensor ref_mask,
                                        at::Tensor tar_mask, at::Tensor output, at::Tensor sum_similarities,
                                        at::Tensor max_cost, int max_disp) {
    const db::Dims d = db::check_feats(ref_feas, tar_feas);
    db::check_plane(ref_mask, "ref_mask", ref_feas, d);
    db::check_plane(tar_mask, "tar_mask", ref_feas, d);
    db::check_plane(output, "output", ref_feas, d);
    db::check_plane(sum_similarities, "sum_similarities", ref_feas, d);
    db::check_plane(max_cost, "max_cost", ref_feas, d);
    TORCH_CHECK(max_disp >= 1, "max_disp must be >= 1, got ", max_disp);
    db::DeviceGuard guard(ref_feas.device());
    const int rc = decnet_spamat_forward(ref_feas.data_ptr<float>(), tar_feas.data_ptr<float>(),
                                         ref_mask.data_ptr<float>(), tar_mask.data_ptr<float>(),
                                         output.data_ptr<float>(), sum_similarities.data_ptr<float>(),
                                         max_cost.data_ptr<float>(), d.B, d.C, d.H, d.W, max_disp,
                                         db::current_stream(ref_feas));
    db::check_rc(rc, "decnet_spamat_forward");
    return 1;                                            // what the reference's function returns
}

// SM_cuda.cpp:17-27 (sparse_matching_ref_backward + sparse_matching_tar_backward, SM_kernel.cu:389-429)
static int sparse_matching_cuda_backward(at::Tensor ref_feas, at::Tensor tar_feas, at::Tensor ref_mask,
                                         at::Tensor tar_mask, at::Tensor output, at::Tensor sum_similarities,
                                         at::Tensor max_cost, at::Tensor grad_output, at::Tensor grad_ref_feas,
                                         at::Tensor grad_tar_feas, int max_disp) {
    const db::Dims d = db::check_feats(ref_feas, tar_feas);
    db::check_plane(ref_mask, "ref_mask", ref_feas, d);
    db::check_plane(tar_mask, "tar_mask", ref_feas, d);
    db::check_plane(output, "output", ref_feas, d);
    db::check_plane(sum_similarities, "sum_similarities", ref_feas, d);
    db::check_plane(max_cost, "max_cost", ref_feas, d);
    db::check_plane(grad_output, "grad_output", ref_feas, d);
    db::check_like_feats(grad_ref_feas, "grad_ref_feas", ref_feas);
    db::check_like_feats(grad_tar_feas, "grad_tar_feas", ref_feas);
    TORCH_CHECK(max_disp >= 1, "max_disp must be >= 1, got ", max_disp);
    db::DeviceGuard guard(ref_feas.device());
    const int rc = decnet_spamat_backward(ref_feas.data_ptr<float>(), tar_feas.data_ptr<float>(),
                                          ref_mask.data_ptr<float>(), tar_mask.data_ptr<float>(),
                                          output.data_ptr<float>(), sum_similarities.data_ptr<float>(),
                                          max_cost.data_ptr<float>(), grad_output.data_ptr<float>(),
                                          grad_ref_feas.data_ptr<float>(), grad_tar_feas.data_ptr<float>(), d.B,
                                          d.C, d.H, d.W, max_disp, db::current_stream(ref_feas));
    db::check_rc(rc, "decnet_spamat_backward");
    return 1;
}

PYBIND11_MODULE(SpaMat, m) {
    m.doc() = "DecNet SparseMatching on MI355X (gfx950): drop-in for the module built from SM_cuda.cpp";
    m.def("sparse_matching_cuda_forward", &sparse_matching_cuda_forward, "sparse matching forward (HIP, gfx950)");
    m.def("sparse_matching_cuda_backward", &sparse_matching_cuda_backward, "sparse matching backward (HIP, gfx950)");
    m.def("decnet_version", [] { return std::string(decnet_version()); });
}
