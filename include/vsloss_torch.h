/*
 * vsloss_torch.h -- header-only libtorch (C++) binding of libvsloss.so.
 *
 * For the reference's C++ trainer: Main_Trainer::process_batch (cpp/main_trainer.hpp:126-143)
 * stops at line 142 after running the encoder on the batch; a loss would be computed right
 * there.  vsl::view_synthesis_loss() is that call: torch::Tensor in, torch::Tensor out,
 * differentiable with respect to the disparity and the pose matrices, stream-ordered on the
 * current HIP stream; vsl::view_synthesis_loss_heads() is the same for several disparity heads in
 * one launch (per-scale disparities, a student head with a forward-only teacher head), with the
 * optional depth / sampling-grid / warped-frame outputs of Trainer.generate_images_pred.  Tensors are only used for their device pointers and lifetime; all
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


/* ---------------------------------------------------------------------------------------------------------------
 * Several disparity heads over the same images and poses in one launch (vsl_loss_forward_heads /
 * vsl_loss_backward_heads): the per-scale disparities of a trainer that feeds a different one to every scale
 * (the loops of Python/Trainer.py:424 and :534), or the student head with a forward-only teacher head beside it
 * (Python/Trainer_distilation.py:845-870,1221-1224).  Each head can also hand back what
 * Trainer.generate_images_pred keeps in `outputs` (Python/Trainer.py:436,464,467), written by the same launch.
 * ------------------------------------------------------------------------------------------------------------- */
struct HeadOptions {
  std::vector<float> scale_weights{1e-3f};  /* disparity_smoothness / 2^scale for this head's loss slots */
  bool forward_only = false;                /* a teacher head: evaluated, never differentiated             */
  bool depth = false;                       /* also return outputs[("depth",0)]   [B,1,H,W]                */
  bool sample = false;                      /* also return outputs[("sample",f)]  [S,B,H,W,2]              */
  bool color = false;                       /* also return outputs[("color",f)]   [S,B,3,H,W]              */
};

struct HeadsResult {
  torch::Tensor losses;              /* [n_heads, VSL_LOSS_SLOTS]; row h, slot VSL_LOSS_TOTAL = that head's losses["loss"] */
  torch::Tensor identity_selection;  /* [n_heads, B, H, W] (empty without auto-masking)                                    */
  std::vector<torch::Tensor> depth, sample, color;  /* one entry per head, undefined where not asked for */
};

class ViewSynthesisLossHeadsFn : public torch::autograd::Function<ViewSynthesisLossHeadsFn> {
 public:
  /* vars = {disp_0 .. disp_{n-1}, T_0 .. T_{S-1}}; consts = {target, K, inv_K, source_0 .. source_{S-1}, [noise]} */
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, at::TensorList vars,
                                                std::vector<torch::Tensor> consts, LossOptions opt,
                                                std::vector<HeadOptions> heads) {
    const int n = (int)heads.size();
    const int S = (int)vars.size() - n;
    TORCH_CHECK(n >= 1 && n <= VSL_MAX_HEADS && S >= 1 && S <= VSL_MAX_SOURCES, "vsloss: 1..8 heads, 1..4 source frames");
    TORCH_CHECK((int)consts.size() == 3 + S || (int)consts.size() == 4 + S, "vsloss: consts = target, K, inv_K, sources, [noise]");
    std::vector<torch::Tensor> keep;
    for (const auto& v : vars) keep.push_back(as_f32(v));
    for (auto& c : consts) keep.push_back(as_f32(c));
    const torch::Tensor& d0 = keep[0];
    vsl_loss_desc d{};
    d.B = (int32_t)d0.size(0);
    d.H = (int32_t)d0.size(2);
    d.W = (int32_t)d0.size(3);
    d.S = S;
    d.flags = (opt.automask ? VSL_FLAG_AUTOMASK : 0u) | (opt.avg_reprojection ? VSL_FLAG_AVG_REPROJECTION : 0u) |
              (opt.no_ssim ? VSL_FLAG_NO_SSIM : 0u);
    d.min_depth = opt.min_depth;
    d.max_depth = opt.max_depth;
    d.eps = 1e-7f;
    d.ssim_weight = 0.85f;
    d.l1_weight = 0.15f;
    d.warp_scale = opt.warp_scale;
    d.n_scales = 1;
    d.inv_num_scales = opt.inv_num_scales;
    d.disp = d0.data_ptr<float>();  // only read by the workspace query
    const size_t c0 = vars.size();
    d.target = keep[c0 + 0].data_ptr<float>();
    d.K = keep[c0 + 1].data_ptr<float>();
    d.inv_K = keep[c0 + 2].data_ptr<float>();
    for (int s = 0; s < S; ++s) {
      d.T[s] = keep[n + s].data_ptr<float>();
      d.source[s] = keep[c0 + 3 + s].data_ptr<float>();
    }
    d.noise = (int)consts.size() == 4 + S ? keep[c0 + 3 + S].data_ptr<float>() : nullptr;
    d.seed = opt.seed;

    const size_t bytes = vsl_loss_workspace_bytes(&d), stride = (bytes + 255) / 256 * 256;
    auto o8 = torch::TensorOptions().dtype(torch::kUInt8).device(d0.device());
    auto of = torch::TensorOptions().dtype(torch::kFloat32).device(d0.device());
    torch::Tensor ws = torch::empty({(int64_t)(stride * n)}, o8);
    torch::Tensor losses = torch::empty({n, VSL_LOSS_SLOTS}, of);
    torch::Tensor ident = torch::empty({opt.automask ? n : 0, d.B, d.H, d.W}, of);
    std::vector<vsl_head> table(n);
    torch::autograd::variable_list out{losses, ident}, extras;
    for (int h = 0; h < n; ++h) {
      TORCH_CHECK(keep[h].sizes() == d0.sizes(), "vsloss: every disparity must be [B,1,H,W] of one shape");
      vsl_head& hd = table[h];
      std::memset(&hd, 0, sizeof(hd));
      hd.disp = keep[h].data_ptr<float>();
      hd.workspace = (char*)ws.data_ptr() + stride * h;
      hd.workspace_bytes = bytes;
      hd.losses = losses.data_ptr<float>() + (size_t)VSL_LOSS_SLOTS * h;
      hd.identity_selection = opt.automask ? ident.data_ptr<float>() + (size_t)d.B * d.H * d.W * h : nullptr;
      hd.n_scales = (int32_t)heads[h].scale_weights.size();
      TORCH_CHECK(hd.n_scales >= 1 && hd.n_scales <= VSL_MAX_SCALES, "vsloss: 1..8 scale weights per head");
      for (int i = 0; i < hd.n_scales; ++i) hd.scale_weight[i] = heads[h].scale_weights[i];
      hd.flags = heads[h].forward_only ? VSL_HEAD_NO_GRAD : 0u;
      torch::Tensor dep, smp, col;
      if (heads[h].depth) {
        dep = torch::empty({d.B, 1, d.H, d.W}, of);
        hd.depth = dep.data_ptr<float>();
      }
      if (heads[h].sample) {
        smp = torch::empty({S, d.B, d.H, d.W, 2}, of);
        for (int s = 0; s < S; ++s) hd.sample[s] = smp.data_ptr<float>() + (size_t)s * d.B * d.H * d.W * 2;
      }
      if (heads[h].color) {
        col = torch::empty({S, d.B, 3, d.H, d.W}, of);
        for (int s = 0; s < S; ++s) hd.color[s] = col.data_ptr<float>() + (size_t)s * d.B * 3 * d.H * d.W;
      }
      for (auto* t : {&dep, &smp, &col}) extras.push_back(t->defined() ? *t : torch::empty({0}, of));
    }
    void* stream = (void*)c10::hip::getCurrentHIPStream().stream();
    check(VSL_LOSS_FORWARD_HEADS(&d, table.data(), n, stream));
    keep.push_back(ws);
    ctx->save_for_backward(keep);
    ctx->saved_data["desc"] = std::string(reinterpret_cast<const char*>(&d), sizeof(d));
    ctx->saved_data["heads"] = std::string(reinterpret_cast<const char*>(table.data()), sizeof(vsl_head) * n);
    ctx->saved_data["S"] = (int64_t)S;
    ctx->saved_data["n"] = (int64_t)n;
    torch::autograd::variable_list nondiff{ident};
    for (auto& e : extras) {
      out.push_back(e);
      nondiff.push_back(e);
    }
    ctx->mark_non_differentiable(nondiff);
    return out;  // losses, identity_selection, then (depth, sample, color) of every head
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    auto keep = ctx->get_saved_variables();
    vsl_loss_desc d;
    std::memcpy(&d, ctx->saved_data["desc"].toStringRef().data(), sizeof(d));
    const int S = (int)ctx->saved_data["S"].toInt(), n = (int)ctx->saved_data["n"].toInt();
    std::vector<vsl_head> table(n);
    std::memcpy(table.data(), ctx->saved_data["heads"].toStringRef().data(), sizeof(vsl_head) * n);
    const torch::Tensor& ws = keep.back();
    auto of = torch::TensorOptions().dtype(torch::kFloat32).device(ws.device());
    torch::Tensor g_losses = grads[0].to(torch::kFloat32).contiguous();
    torch::Tensor g_disp = torch::empty({n, d.B, 1, d.H, d.W}, of);
    torch::Tensor g_T = torch::zeros({n, S, d.B, 4, 4}, of);  // rows of forward-only heads stay zero in the sum below
    for (int h = 0; h < n; ++h) {
      vsl_head& hd = table[h];
      hd.grad_losses = g_losses.data_ptr<float>() + (size_t)VSL_LOSS_SLOTS * h;
      if (hd.flags & VSL_HEAD_NO_GRAD) continue;
      hd.grad_disp = g_disp.data_ptr<float>() + (size_t)d.B * d.H * d.W * h;
      for (int s = 0; s < S; ++s) hd.grad_T[s] = g_T.data_ptr<float>() + (size_t)16 * d.B * (h * S + s);
    }
    void* stream = (void*)c10::hip::getCurrentHIPStream().stream();
    check(VSL_LOSS_BACKWARD_HEADS(&d, table.data(), n, stream));
    // every head with gradients differentiates the same pose matrices: their gradients add up
    torch::Tensor g_pose = g_T.sum(0);
    torch::autograd::variable_list out;
    for (int h = 0; h < n; ++h) out.push_back((table[h].flags & VSL_HEAD_NO_GRAD) ? torch::Tensor() : g_disp[h]);
    for (int s = 0; s < S; ++s) out.push_back(g_pose[s]);
    out.push_back(torch::Tensor());  // consts
    out.push_back(torch::Tensor());  // options
    out.push_back(torch::Tensor());  // head options
    return out;
  }
};

inline HeadsResult view_synthesis_loss_heads(const std::vector<torch::Tensor>& disps, const torch::Tensor& target,
                                             const std::vector<torch::Tensor>& sources, const torch::Tensor& K,
                                             const torch::Tensor& inv_K, const std::vector<torch::Tensor>& poses,
                                             const LossOptions& opt, const std::vector<HeadOptions>& heads,
                                             const torch::Tensor& noise = torch::Tensor()) {
  TORCH_CHECK(!disps.empty() && disps.size() == heads.size(), "vsloss: one HeadOptions per disparity");
  TORCH_CHECK(sources.size() == poses.size() && !poses.empty() && poses.size() <= VSL_MAX_SOURCES,
              "vsloss: one pose per source frame, 1..4 source frames");
  std::vector<torch::Tensor> vars;
  for (size_t h = 0; h < disps.size(); ++h) vars.push_back(heads[h].forward_only ? disps[h].detach() : disps[h]);
  for (auto& p : poses) vars.push_back(p);
  std::vector<torch::Tensor> consts{target, K, inv_K};
  for (auto& s : sources) consts.push_back(s);
  if (noise.defined() && noise.numel() > 0) consts.push_back(noise);
  auto out = ViewSynthesisLossHeadsFn::apply(at::TensorList(vars), consts, opt, heads);
  HeadsResult r;
  r.losses = out[0];
  r.identity_selection = out[1];
  for (size_t h = 0; h < disps.size(); ++h) {
    auto pick = [&](int k, bool on) { return on ? out[2 + 3 * h + k] : torch::Tensor(); };
    r.depth.push_back(pick(0, heads[h].depth));
    r.sample.push_back(pick(1, heads[h].sample));
    r.color.push_back(pick(2, heads[h].color));
  }
  return r;
}

}  // namespace vsl
#endif /* VSLOSS_TORCH_H */
