// Test extension: the header-only libtorch binding (include/vsloss_torch.h) behind a pybind11 module, so a
// GPU test can drive the C++ path exactly as a libtorch trainer would and compare it with the ctypes binding.
#include <torch/extension.h>

#include "vsloss_torch.h"

static std::vector<torch::Tensor> loss(torch::Tensor disp, torch::Tensor target, std::vector<torch::Tensor> sources,
                                       torch::Tensor K, torch::Tensor inv_K, std::vector<torch::Tensor> poses,
                                       torch::Tensor noise, bool automask, bool avg, bool no_ssim, double warp_scale,
                                       std::vector<double> scale_weights) {
  vsl::LossOptions opt;
  opt.automask = automask;
  opt.avg_reprojection = avg;
  opt.no_ssim = no_ssim;
  opt.warp_scale = (float)warp_scale;
  opt.scale_weights.assign(scale_weights.begin(), scale_weights.end());
  auto r = vsl::view_synthesis_loss(disp, target, sources, K, inv_K, poses, opt, noise);
  return {r.first, r.second};
}

PYBIND11_MODULE(vsl_torch_ext, m) { m.def("loss", &loss); }
