/*
 * vsloss_torch.h -- header-only libtorch (C++) binding of libvsloss.so.
 *
 * For the reference's C++ trainer: Main_Trainer::process_batch (cpp/main_trainer.hpp:126-143)
 * stops at line 142 after running the encoder on the batch; a loss would be computed right
 * there.  vsl::view_synthesis_loss() is that call: torch::Tensor in, torch::Tensor out,
 * differentiable with respect to the disparity and the pose matrices, stream-ordered on the
 * current HIP stream.  Tensors are only used for their device pointers and lifetime; all
 * arithmetic happens inside the C-ABI calls of vsloss.h.
 *
 * Build: add include/ to the include path and link libvsloss.so next to libtorch.
 */
#ifndef VSLOSS_TORCH_H
#define VSLOSS_TORCH_H

#include <c10/hip/HIPStream.h>
#include <torch/torch.h>

#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "vsloss.h"

namespace vsl {

struct LossOptions {
  bool automask = true;           /* not opt.disable_automasking (Python/options.py:131-134) */
  bool avg_reprojection = false;  /* Python/options.py:120-123 */
  bool no_ssim = false;           /* Python/options.py:135-138 */
  float min_depth = 0.1f, max_depth = 100.0f;
  float warp_scale = 1.0f;                  /* 0.9 in Python/Trainer_distilation.py:1234 */
  std::vector<float> scale_weights{1e-3f};  /* disparity_smoothness / 2^scale, one per opt.scales entry */
  float inv_num_scales = 1.0f;              /* 1 / num_scales (Python/Trainer.py:33,618) */
  uint64_t seed = 0;                        /* tie-break noise stream when no noise tensor is given */
};

inline void check(int rc) {
  if (rc != VSL_OK) throw std::runtime_error(std::string("libvsloss: ") + vsl_last_error_string());
}

inline torch::Tensor as_f32(const torch::Tensor& t) {
  TORCH_CHECK(t.is_cuda(), "vsloss: tensors must live on the GPU");
  return t.to(torch::kFloat32).contiguous();
}

class ViewSynthesisLossFn : public torch::autograd::Function<ViewSynthesisLossFn> {
 public:
  /* consts = {target, K, inv_K, source_0 .. source_{S-1}, [noise]}; T1..T3 are empty for unused sources */
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const torch::Tensor& disp_in,
                                                const torch::Tensor& T0, const std::optional<torch::Tensor>& T1,
                                                const std::optional<torch::Tensor>& T2,
                                                const std::optional<torch::Tensor>& T3, std::vector<torch::Tensor> consts,
                                                LossOptions opt) {
    std::vector<torch::Tensor> vars{disp_in, T0};
    for (const auto* t : {&T1, &T2, &T3})
      if (t->has_value() && t->value().defined()) vars.push_back(t->value());
    const int S = (int)vars.size() - 1;
    TORCH_CHECK((int)consts.size() == 3 + S || (int)consts.size() == 4 + S, "vsloss: consts = target, K, inv_K, sources, [noise]");
    std::vector<torch::Tensor> keep;
    for (auto& v : vars) keep.push_back(as_f32(v));
    for (auto& c : consts) keep.push_back(as_f32(c));
    const torch::Tensor& disp = keep[0];
    vsl_loss_desc d{};
    d.B = (int32_t)disp.size(0);
    d.H = (int32_t)disp.size(2);
    d.W = (int32_t)disp.size(3);
    d.S = S;
    d.flags = (opt.automask ? VSL_FLAG_AUTOMASK : 0u) | (opt.avg_reprojection ? VSL_FLAG_AVG_REPROJECTION : 0u) |
              (opt.no_ssim ? VSL_FLAG_NO_SSIM : 0u);
    d.min_depth = opt.min_depth;
    d.max_depth = opt.max_depth;
    d.eps = 1e-7f;
    d.ssim_weight = 0.85f;
    d.l1_weight = 0.15f;
    d.warp_scale = opt.warp_scale;
    d.n_scales = (int32_t)opt.scale_weights.size();
    TORCH_CHECK(d.n_scales >= 1 && d.n_scales <= VSL_MAX_SCALES, "vsloss: 1..8 scales");
    for (int i = 0; i < d.n_scales; ++i) d.scale_weight[i] = opt.scale_weights[i];
    d.inv_num_scales = opt.inv_num_scales;
    d.disp = disp.data_ptr<float>();
    const size_t c0 = vars.size();
    d.target = keep[c0 + 0].data_ptr<float>();
    d.K = keep[c0 + 1].data_ptr<float>();
    d.inv_K = keep[c0 + 2].data_ptr<float>();
    for (int s = 0; s < S; ++s) {
      d.T[s] = keep[1 + s].data_ptr<float>();
      d.source[s] = keep[c0 + 3 + s].data_ptr<float>();
    }
    d.noise = (int)consts.size() == 4 + S ? keep[c0 + 3 + S].data_ptr<float>() : nullptr;
    d.seed = opt.seed;

    const size_t bytes = vsl_loss_workspace_bytes(&d);
    auto o8 = torch::TensorOptions().dtype(torch::kUInt8).device(disp.device());
    auto of = torch::TensorOptions().dtype(torch::kFloat32).device(disp.device());
    torch::Tensor ws = torch::empty({(int64_t)bytes}, o8);
    torch::Tensor losses = torch::empty({VSL_LOSS_SLOTS}, of);
    torch::Tensor ident = opt.automask ? torch::empty({d.B, d.H, d.W}, of) : torch::empty({0}, of);
    void* stream = (void*)c10::hip::getCurrentHIPStream().stream();
    check(vsl_loss_forward(&d, ws.data_ptr(), bytes, losses.data_ptr<float>(),
                           opt.automask ? ident.data_ptr<float>() : nullptr, stream));
    keep.push_back(ws);
    ctx->save_for_backward(keep);
    ctx->saved_data["desc"] = std::string(reinterpret_cast<const char*>(&d), sizeof(d));
    ctx->saved_data["S"] = (int64_t)S;
    ctx->mark_non_differentiable({ident});
    return {losses, ident};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    auto keep = ctx->get_saved_variables();
    const std::string raw = ctx->saved_data["desc"].toStringRef();
    vsl_loss_desc d;
    std::memcpy(&d, raw.data(), sizeof(d));
    const int S = (int)ctx->saved_data["S"].toInt();
    const torch::Tensor& ws = keep.back();
    auto of = torch::TensorOptions().dtype(torch::kFloat32).device(ws.device());
    torch::Tensor g_losses = grads[0].to(torch::kFloat32).contiguous();
    torch::Tensor g_disp = torch::empty({d.B, 1, d.H, d.W}, of);
    std::vector<torch::Tensor> g_T;
    float* ptrs[VSL_MAX_SOURCES] = {nullptr, nullptr, nullptr, nullptr};
    for (int s = 0; s < S; ++s) {
      g_T.push_back(torch::empty({d.B, 4, 4}, of));
      ptrs[s] = g_T[s].data_ptr<float>();
    }
    void* stream = (void*)c10::hip::getCurrentHIPStream().stream();
    check(vsl_loss_backward(&d, ws.data_ptr(), (size_t)ws.numel(), g_losses.data_ptr<float>(), g_disp.data_ptr<float>(),
                            ptrs, stream));
    torch::autograd::variable_list out{g_disp};
    for (int s = 0; s < VSL_MAX_SOURCES; ++s) out.push_back(s < S ? g_T[s] : torch::Tensor());
    out.push_back(torch::Tensor());  // consts
    out.push_back(torch::Tensor());  // options
    return out;
  }
};

/* losses[VSL_LOSS_TOTAL] is losses["loss"] of Python/Trainer.py:619; second result = outputs["identity_selection"]. */
inline std::pair<torch::Tensor, torch::Tensor> view_synthesis_loss(const torch::Tensor& disp, const torch::Tensor& target,
                                                                  const std::vector<torch::Tensor>& sources,
                                                                  const torch::Tensor& K, const torch::Tensor& inv_K,
                                                                  const std::vector<torch::Tensor>& poses,
                                                                  const LossOptions& opt = LossOptions(),
                                                                  const torch::Tensor& noise = torch::Tensor()) {
  TORCH_CHECK(sources.size() == poses.size() && !poses.empty() && poses.size() <= VSL_MAX_SOURCES,
              "vsloss: one pose per source frame, 1..4 source frames");
  std::vector<torch::Tensor> consts{target, K, inv_K};
  for (auto& s : sources) consts.push_back(s);
  if (noise.defined() && noise.numel() > 0) consts.push_back(noise);
  auto opt_pose = [&](size_t i) { return i < poses.size() ? std::optional<torch::Tensor>(poses[i]) : std::nullopt; };
  auto out = ViewSynthesisLossFn::apply(disp, poses[0], opt_pose(1), opt_pose(2), opt_pose(3), consts, opt);
  return {out[0], out[1]};
}

}  // namespace vsl
#endif /* VSLOSS_TORCH_H */
