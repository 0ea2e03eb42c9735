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

// several heads in one launch: returns {losses, identity_selection, depth_0, sample_0, color_0, depth_1, ...}
// (empty tensors where an output was not asked for)
static std::vector<torch::Tensor> loss_heads(std::vector<torch::Tensor> disps, torch::Tensor target,
                                             std::vector<torch::Tensor> sources, torch::Tensor K, torch::Tensor inv_K,
                                             std::vector<torch::Tensor> poses, torch::Tensor noise, bool automask,
                                             double warp_scale, std::vector<bool> forward_only,
                                             std::vector<std::vector<double>> scale_weights, std::vector<bool> materialise) {
  vsl::LossOptions opt;
  opt.automask = automask;
  opt.warp_scale = (float)warp_scale;
  std::vector<vsl::HeadOptions> heads(disps.size());
  for (size_t h = 0; h < disps.size(); ++h) {
    heads[h].scale_weights.assign(scale_weights[h].begin(), scale_weights[h].end());
    heads[h].forward_only = forward_only[h];
    heads[h].depth = heads[h].sample = heads[h].color = materialise[h];
  }
  auto r = vsl::view_synthesis_loss_heads(disps, target, sources, K, inv_K, poses, opt, heads, noise);
  std::vector<torch::Tensor> out{r.losses, r.identity_selection};
  auto or_empty = [](const torch::Tensor& t) { return t.defined() ? t : torch::empty({0}); };
  for (size_t h = 0; h < disps.size(); ++h) {
    out.push_back(or_empty(r.depth[h]));
    out.push_back(or_empty(r.sample[h]));
    out.push_back(or_empty(r.color[h]));
  }
  return out;
}

PYBIND11_MODULE(vsl_torch_ext, m) {
  m.def("loss", &loss);
  m.def("loss_heads", &loss_heads);
}
