/*
 * vsloss.h -- C ABI of libvsloss.so: the Monodepth2-style view-synthesis loss
 * (forward + backward) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (sharjeel1999/Monocular-depth-estimation) has no FFI for this
 * path: it is eager PyTorch.  Each entry point below therefore names the
 * reference Python code it replaces; a maintainer binds these symbols with
 * ctypes (Python trainers) or links them from libtorch C++ (cpp/main_trainer.hpp,
 * where process_batch ends at line 142 before any loss exists).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - images / disparities are float32, NCHW-contiguous; matrices are row-major
 *     [B,4,4] float32;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued on it, nothing synchronises the host;
 *   - functions allocate nothing: the caller owns a workspace of
 *     vsl_loss_workspace_bytes() bytes (256-byte aligned) that must stay alive
 *     and untouched between vsl_loss_forward and vsl_loss_backward;
 *   - return value 0 = success, negative = error (see VSL_ERR_*); the message of
 *     the calling thread's last error is returned by vsl_last_error_string().
 */
#ifndef VSLOSS_H
#define VSLOSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSL_MAX_SOURCES 4
#define VSL_MAX_SCALES 8
#define VSL_MAX_HEADS 8

/* flag bits of vsl_loss_desc.flags (Python/options.py:105-135) */
#define VSL_FLAG_AUTOMASK 1u         /* == not opt.disable_automasking  (Trainer.py:553,587) */
#define VSL_FLAG_AVG_REPROJECTION 2u /* opt.avg_reprojection            (Trainer.py:562,582) */
#define VSL_FLAG_NO_SSIM 4u          /* opt.no_ssim                     (Trainer.py:489)     */
#define VSL_FLAG_NO_GRAD 8u          /* forward only: gradients are not prepared             */

#define VSL_OK 0
#define VSL_ERR_INVALID_ARGUMENT (-1)
#define VSL_ERR_WORKSPACE_TOO_SMALL (-2)
#define VSL_ERR_LAUNCH_FAILED (-3)

/* slots of the `losses` output of vsl_loss_forward */
#define VSL_LOSS_TOTAL 0  /* losses["loss"]            (Trainer.py:618-619) */
#define VSL_LOSS_PHOTO 1  /* to_optimise.mean()        (Trainer.py:605)     */
#define VSL_LOSS_SMOOTH 2 /* get_smooth_loss(norm_disp, color) (Trainer.py:612) */
#define VSL_LOSS_SCALE0 3 /* losses["loss/{scale}"] for the i-th entry of scale_weight (Trainer.py:616) */
#define VSL_LOSS_SLOTS (3 + VSL_MAX_SCALES)

/*
 * One fused evaluation of
 *   Trainer.generate_images_pred  (Python/Trainer.py:419-479)  +
 *   Trainer.compute_losses        (Python/Trainer.py:518-620)
 * for ONE disparity tensor.  The reference reads outputs[("disp",0)] in every
 * iteration of its `for scale in self.opt.scales` loop, so all its scales share
 * the photometric term; that is expressed here by n_scales / scale_weight.
 */
typedef struct vsl_loss_desc {
  int32_t B, H, W; /* batch, height, width (H >= 2, W >= 2)                     */
  int32_t S;       /* number of source frames, 1..VSL_MAX_SOURCES (frame_ids[1:]) */
  uint32_t flags;  /* VSL_FLAG_*                                                 */

  float min_depth, max_depth; /* disp_to_depth (layers.py:16-25); defaults 0.1, 100  */
  float eps;                  /* Project3D eps (layers.py:180); default 1e-7          */
  float ssim_weight;          /* 0.85 (Trainer.py:493)                                */
  float l1_weight;            /* 0.15 (Trainer.py:493)                                */
  float warp_scale;           /* 1.0, or 0.9 as in Trainer_distilation.py:1234        */

  int32_t n_scales;                     /* len(opt.scales), 1..VSL_MAX_SCALES         */
  float scale_weight[VSL_MAX_SCALES];   /* disparity_smoothness / 2**scale (Trainer.py:614) */
  float inv_num_scales;                 /* 1 / self.num_scales (Trainer.py:33,618) = 1 */

  const float* disp;                     /* [B,1,H,W] sigmoid disparity, outputs[("disp",0)] */
  const float* target;                   /* [B,3,H,W] inputs[("image",0)]                    */
  const float* source[VSL_MAX_SOURCES];  /* [B,3,H,W] inputs[("image",f)], f in frame_ids[1:] */
  const float* K;                        /* [B,4,4] inputs["K"]                              */
  const float* inv_K;                    /* [B,4,4] inputs["inv_K"]                          */
  const float* T[VSL_MAX_SOURCES];       /* [B,4,4] outputs[("cam_T_cam",f)] or inputs["stereo_T"] */
  const float* noise;                    /* [B,S,H,W] ([B,1,H,W] with AVG_REPROJECTION) N(0,1) tie-break
                                            noise of Trainer.py:589, or NULL: the kernel then draws it from
                                            a counter-based generator keyed by `seed`               */
  uint64_t seed;
  const float* identity_min;             /* optional [B,H,W] written by vsl_identity_min for the same images,
                                            noise/seed and flags (lets several disparity heads share it);
                                            NULL: vsl_loss_forward computes it itself                  */
} vsl_loss_desc;

/* Bytes of workspace vsl_loss_forward / vsl_loss_backward need for this shape. */
size_t vsl_loss_workspace_bytes(const vsl_loss_desc* desc);

/*
 * Forward: fills losses[VSL_LOSS_SLOTS] (device floats) and, when not NULL,
 * identity_selection [B,H,W] (outputs["identity_selection"], Trainer.py:601-603;
 * only written with VSL_FLAG_AUTOMASK).  Unless VSL_FLAG_NO_GRAD is set it also
 * leaves in the workspace everything vsl_loss_backward needs.
 */
int vsl_loss_forward(const vsl_loss_desc* desc, void* workspace, size_t workspace_bytes, float* losses,
                     float* identity_selection, void* stream);

/*
 * The part of the auto-mask that depends on neither disparity nor pose: for every
 * pixel the minimum over the sources of the identity-reprojection error plus
 * 1e-5 * noise (Trainer.py:553-566,587-590; with VSL_FLAG_AVG_REPROJECTION the mean
 * over sources).  Reads B,H,W,S, flags, ssim/l1 weights, target, source[], noise/seed.
 * identity_min is [B,H,W] float32.
 */
int vsl_identity_min(const vsl_loss_desc* desc, float* identity_min, void* stream);

/*
 * Backward (the autograd pass that `.backward()` runs over Trainer.py:419-620):
 * grad_disp [B,1,H,W]; grad_T[s] [B,4,4] for each source (entries, or the array
 * itself, may be NULL to skip).  grad_losses is a device array of VSL_LOSS_SLOTS
 * floats holding d(objective)/d(losses[i]) for every slot of the forward output
 * (normally 1 in VSL_LOSS_TOTAL and 0 elsewhere), or NULL for exactly that
 * default.  `grad_T` is a HOST array of S device pointers.
 */
int vsl_loss_backward(const vsl_loss_desc* desc, const void* workspace, size_t workspace_bytes,
                      const float* grad_losses, float* grad_disp, float* const* grad_T, void* stream);

/* ------------------------------------------------------------------------
 * Per-module entry points: one for each layers.py primitive, forward and
 * backward, for callers that use the modules one at a time.  All tensors are
 * float32, contiguous, on the device.
 * ---------------------------------------------------------------------- */

/* disp_to_depth (Python/layers.py:16-25): scaled = 1/max + (1/min - 1/max) * disp, depth = 1/scaled.
 * scaled / depth (and g_scaled / g_depth) may be NULL when not wanted. */
int vsl_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, long long n, float min_depth, float max_depth,
                          void* stream);
int vsl_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* g_disp, long long n,
                          float min_depth, float max_depth, void* stream);

/* BackprojectDepth.forward (Python/layers.py:161-174): depth [B,1,H,W], inv_K [B,4,4] -> points [B,4,H*W]. */
int vsl_backproject_fwd(const float* depth, const float* inv_K, float* points, int B, int H, int W, void* stream);
int vsl_backproject_bwd(const float* g_points, const float* inv_K, float* g_depth, int B, int H, int W, void* stream);

/* Project3D.forward (Python/layers.py:188-202): points [B,4,H*W], K, T [B,4,4] -> grid [B,H,W,2].
 * Backward: g_points [B,4,H*W], g_T, g_K [B,4,4] (each may be NULL). */
int vsl_project_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W, float eps,
                    void* stream);
size_t vsl_project_bwd_workspace_bytes(int B, int H, int W);
int vsl_project_bwd(const float* points, const float* K, const float* T, const float* g_grid, float* g_points, float* g_T,
                    float* g_K, void* workspace, size_t workspace_bytes, int B, int H, int W, float eps, void* stream);

/* F.grid_sample(image, grid, mode="bilinear", padding_mode="border", align_corners=False)
 * (Python/Trainer.py:467): image [B,C,Hi,Wi], grid [B,Ho,Wo,2] -> out [B,C,Ho,Wo].  Backward: g_grid
 * and/or g_image (NULL to skip); g_image is accumulated with float atomics (feature warping,
 * Python/Train_distillation_new.py:1159-1182), so its last bits depend on the arrival order. */
int vsl_grid_sample_fwd(const float* image, const float* grid, float* out, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        void* stream);
int vsl_grid_sample_bwd(const float* image, const float* grid, const float* g_out, float* g_grid, float* g_image, int B,
                        int C, int Hi, int Wi, int Ho, int Wo, void* stream);

/* SSIM.forward (Python/layers.py:245-262) over `planes` = B*C images of H x W: out = clamp((1-SSIM)/2, 0, 1).
 * Backward gives the gradient with respect to x; SSIM is symmetric, so swap x and y for the other one. */
int vsl_ssim_fwd(const float* x, const float* y, float* out, int planes, int H, int W, void* stream);
int vsl_ssim_bwd(const float* x, const float* y, const float* g_out, float* g_x, int planes, int H, int W, void* stream);

/* get_smooth_loss (Python/layers.py:211-226): disp [B,1,H,W], img [B,C,H,W] -> loss (device scalar).
 * Backward: g_disp = g_loss * d loss / d disp (img carries no gradient in the reference's use). */
size_t vsl_smooth_workspace_bytes(int B, int H, int W);
int vsl_smooth_fwd(const float* disp, const float* img, float* loss, void* workspace, size_t workspace_bytes, int B, int C,
                   int H, int W, void* stream);
int vsl_smooth_bwd(const float* disp, const float* img, const float* g_loss, float* g_disp, int B, int C, int H, int W,
                   void* stream);

/* Message of the last error raised on the calling thread ("" if none). */
const char* vsl_last_error_string(void);

/*
 * Several disparity heads over the SAME images, poses, intrinsics and flags in one call: what a
 * trainer that feeds a different disparity to every entry of opt.scales runs per step
 * (the `for scale in self.opt.scales` loops of Python/Trainer.py:424 and :534; the student and
 * teacher heads of Python/Trainer_distilation.py:1221-1224).  One grid covers the tiles of all
 * heads, one finalize and one gradient launch serve all of them.  From `desc` the calls read
 * everything except disp, n_scales and scale_weight, which each head carries itself; with
 * VSL_FLAG_AUTOMASK and desc->identity_min == NULL the identity minimum is computed once (into
 * the first head's workspace) and shared.  Results are bit-identical to one
 * vsl_loss_forward / vsl_loss_backward call per head.
 */
/* vsl_head.flags: this head is evaluated forward only (a teacher head next to a student head,
 * Python/Trainer_distilation.py:845-870); vsl_loss_backward_heads skips it */
#define VSL_HEAD_NO_GRAD 1u

typedef struct vsl_head {
  const float* disp;          /* [B,1,H,W] this head's sigmoid disparity                          */
  void* workspace;            /* vsl_loss_workspace_bytes(desc) bytes, 256-byte aligned, per head   */
  size_t workspace_bytes;
  float* losses;              /* out [VSL_LOSS_SLOTS] (forward)                                    */
  float* identity_selection;  /* out [B,H,W] or NULL (forward)                                     */
  int32_t n_scales;                    /* loss slots of this head, 1..VSL_MAX_SCALES               */
  float scale_weight[VSL_MAX_SCALES];  /* disparity_smoothness / 2**scale (Trainer.py:614)         */
  uint32_t flags;                      /* VSL_HEAD_*                                                */
  const float* grad_losses;   /* in [VSL_LOSS_SLOTS] or NULL = d/d losses[VSL_LOSS_TOTAL] (backward) */
  float* grad_disp;           /* out [B,1,H,W] or NULL (backward)                                  */
  float* grad_T[VSL_MAX_SOURCES]; /* out [B,4,4] each or NULL (backward)                           */
  /* What Trainer.generate_images_pred leaves in `outputs` for validation dumps and the depth metrics
   * (Python/Trainer.py:436,464,467; read at :129-132,627), written by the same forward launch when the
   * pointer is not NULL: */
  float* depth;                   /* out [B,1,H,W]   outputs[("depth",0)]                           */
  float* sample[VSL_MAX_SOURCES]; /* out [B,H,W,2]   outputs[("sample",f)], the sampling grid        */
  float* color[VSL_MAX_SOURCES];  /* out [B,3,H,W]   outputs[("color",f)], the warped source frame   */
} vsl_head;

int vsl_loss_forward_heads(const vsl_loss_desc* desc, const vsl_head* heads, int n_heads, void* stream);
/*
 * vsl_loss_forward for callers that cannot allocate device memory (code under HIP graph capture,
 * tight validation / serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Forward only (as if VSL_FLAG_NO_GRAD were set; there is no matching backward): the
 * work area is a VSL_STATIC_SCRATCH_BYTES array that the library itself holds in device memory, on
 * every GPU it runs on, until the process ends.  The call allocates and frees nothing.  Shapes
 * whose work area, vsl_loss_workspace_bytes(desc), exceeds VSL_STATIC_SCRATCH_BYTES are rejected
 * with VSL_ERR_WORKSPACE_TOO_SMALL (640 x 192 with batch 2 fits).  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams or threads) race; calls on one
 * stream, or ordered by events, are safe.  Results equal vsl_loss_forward's bit for bit.
 */
#define VSL_STATIC_SCRATCH_BYTES (4u << 20)
int vsl_loss_forward_static(const vsl_loss_desc* desc, float* losses, float* identity_selection, void* stream);

int vsl_loss_backward_heads(const vsl_loss_desc* desc, const vsl_head* heads, int n_heads, void* stream);

/*
 * The same two calls with the caller's structure sizes spelled out: desc_bytes = sizeof(vsl_loss_desc) and
 * head_bytes = sizeof(vsl_head) as the CALLER was compiled.  vsl_head has grown between library versions (128 ->
 * 200 bytes with the optional outputs); a binary built against another layout gets VSL_ERR_INVALID_ARGUMENT here
 * instead of having the bytes of its table read as other fields (and stored through).  Callers that may meet a
 * library other than the one they were built with should use these, through the macros below; such a caller must
 * be recompiled against the new header, relinking is not enough.
 */
int vsl_loss_forward_heads_checked(const vsl_loss_desc* desc, size_t desc_bytes, const vsl_head* heads, size_t head_bytes,
                                   int n_heads, void* stream);
int vsl_loss_backward_heads_checked(const vsl_loss_desc* desc, size_t desc_bytes, const vsl_head* heads, size_t head_bytes,
                                    int n_heads, void* stream);
#define VSL_LOSS_FORWARD_HEADS(desc, heads, n_heads, stream) \
  vsl_loss_forward_heads_checked((desc), sizeof(vsl_loss_desc), (heads), sizeof(vsl_head), (n_heads), (stream))
#define VSL_LOSS_BACKWARD_HEADS(desc, heads, n_heads, stream) \
  vsl_loss_backward_heads_checked((desc), sizeof(vsl_loss_desc), (heads), sizeof(vsl_head), (n_heads), (stream))

/*
 * Pose preparation in front of the loss: transformation_from_parameters (Python/layers.py:28-45, with
 * rot_from_axisangle :64-103 and get_translation_matrix :48-61) for every source frame of a batch in one
 * launch, as Trainer.predict_pose calls it (Python/Trainer.py:393 inverted for negative frame ids in
 * "pairs" mode, :414 otherwise).  axisangle[s] / translation[s] point at the first of B rows of 3 floats,
 * `aa_stride` / `tr_stride` floats apart (the slices axisangle[:, i] of a [B,F,1,3] network output have
 * stride 3*F).  t_scale, if not NULL, is a [B] factor on the translation: the mean inverse depth of the
 * "posecnn" pose model (Python/Trainer.py:446-455).
 */
typedef struct vsl_pose_desc {
  int32_t B, n_src;                            /* images, source frames (1..VSL_MAX_SOURCES)          */
  int32_t aa_stride, tr_stride;                /* floats between consecutive images, >= 3             */
  uint32_t invert_mask;                        /* bit s set: source s uses invert=True                 */
  uint32_t reserved;
  const float* axisangle[VSL_MAX_SOURCES];     /* [B,3] strided                                       */
  const float* translation[VSL_MAX_SOURCES];   /* [B,3] strided                                       */
  const float* t_scale;                        /* [B] or NULL                                         */
} vsl_pose_desc;

/* T[s]: out [B,4,4] contiguous. */
int vsl_pose_fwd(const vsl_pose_desc* desc, float* const* T, void* stream);
/* g_T[s]: in [B,4,4]; g_axisangle[s], g_translation[s]: out [B,3] contiguous (tables or entries may be
 * NULL); g_t_scale: out [B] summed over the sources, or NULL. */
int vsl_pose_bwd(const vsl_pose_desc* desc, const float* const* g_T, float* const* g_axisangle,
                 float* const* g_translation, float* g_t_scale, void* stream);

/*
 * The "posecnn" pose model's preparation (Python/Trainer.py:446-455) in two launches instead of three: the per-image
 * mean of 1/depth of `disp` [B,1,H,W] (per-chunk partial sums), then ONE launch that finishes the means, scales the
 * translations with them and builds every pose matrix.  desc->t_scale is ignored; t_scale_out [B] receives the means
 * (pass it as desc->t_scale to vsl_pose_bwd, whose g_t_scale then goes to vsl_mean_inv_depth_bwd).  Results equal
 * vsl_mean_inv_depth_fwd followed by vsl_pose_fwd bit for bit.  workspace: vsl_mean_inv_depth_workspace_bytes(B)
 * bytes, 8-byte aligned.
 */
int vsl_pose_fwd_posecnn(const vsl_pose_desc* desc, const float* disp, int H, int W, float min_depth, float max_depth,
                         float* t_scale_out, float* const* T, void* workspace, size_t workspace_bytes, void* stream);

/* Per-image mean of 1/depth with depth = disp_to_depth(disp)[1] (Python/Trainer.py:451-452):
 * disp [B,1,H,W] -> out [B].  Backward: g_disp [B,1,H,W] = (or +=, if accumulate) g_out[b] * d out[b] / d disp. */
size_t vsl_mean_inv_depth_workspace_bytes(int B);
int vsl_mean_inv_depth_fwd(const float* disp, float* out, void* workspace, size_t workspace_bytes, int B, int H, int W,
                           float min_depth, float max_depth, void* stream);
int vsl_mean_inv_depth_bwd(const float* disp, const float* g_out, float* g_disp, int accumulate, int B, int H, int W,
                           float min_depth, float max_depth, void* stream);

/*
 * Feature warping at a coarse scale (Python/Train_distillation_new.py:1159-1182, the same block in
 * Python/train_controlled_diffusion.py:1106-1129): disp_to_depth -> BackprojectDepth -> Project3D ->
 * F.grid_sample(feats, grid, padding_mode="border") of a C-channel float32 feature map that requires a
 * gradient, fused into one pass (the points and the grid are never written out).  `disp` is the disparity
 * already resized to H x W (the F.interpolate of :1159 stays with the caller); K, inv_K are used as given
 * (the reference passes the full-resolution intrinsics at every scale).
 */
typedef struct vsl_warp_desc {
  int32_t B, C, H, W;
  float min_depth, max_depth, eps;   /* eps = 1e-7 (Python/layers.py:180) */
  uint32_t reserved;
  const float* feats;                /* [B,C,H,W] */
  const float* disp;                 /* [B,1,H,W] */
  const float* K;                    /* [B,4,4]   */
  const float* inv_K;                /* [B,4,4]   */
  const float* T;                    /* [B,4,4]   */
} vsl_warp_desc;

size_t vsl_warp_features_workspace_bytes(const vsl_warp_desc* desc);
/* out [B,C,H,W]; sample: the sampling grid [B,H,W,2] or NULL. */
int vsl_warp_features_fwd(const vsl_warp_desc* desc, float* out, float* sample, void* stream);
/* g_out [B,C,H,W] -> g_feats [B,C,H,W] (float atomics: last bits depend on arrival order), g_disp [B,1,H,W],
 * g_T [B,4,4]; each may be NULL.  workspace: vsl_warp_features_workspace_bytes(desc) bytes, 8-byte aligned. */
int vsl_warp_features_bwd(const vsl_warp_desc* desc, const float* g_out, float* g_feats, float* g_disp, float* g_T,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same with up to VSL_MAX_WARP_TENSORS feature maps that share the sampling coordinates: what
 * Trainer.generate_images_pred does for a source frame when the affinity term is on -- F.grid_sample of the
 * colour frame (3 channels) and of the 25-channel affinity map with one grid (Python/Trainer.py:467-473) -- as one
 * launch.  `out`, `g_out`, `g_feats` are HOST arrays of n device pointers; a NULL entry of g_out means that
 * tensor has no upstream gradient, a NULL entry (or table) of g_feats that its own gradient is not wanted.
 */
#define VSL_MAX_WARP_TENSORS 4
typedef struct vsl_warp_multi_desc {
  int32_t B, H, W, n;                          /* n = number of tensors, 1..VSL_MAX_WARP_TENSORS */
  float min_depth, max_depth, eps;
  uint32_t reserved;
  int32_t C[VSL_MAX_WARP_TENSORS];
  const float* feats[VSL_MAX_WARP_TENSORS];    /* [B,C[i],H,W] */
  const float* disp;                           /* [B,1,H,W] */
  const float* K;                              /* [B,4,4]   */
  const float* inv_K;                          /* [B,4,4]   */
  const float* T;                              /* [B,4,4]   */
} vsl_warp_multi_desc;

size_t vsl_warp_multi_workspace_bytes(const vsl_warp_multi_desc* desc);
int vsl_warp_multi_fwd(const vsl_warp_multi_desc* desc, float* const* out, float* sample, void* stream);
int vsl_warp_multi_bwd(const vsl_warp_multi_desc* desc, const float* const* g_out, float* const* g_feats, float* g_disp,
                       float* g_T, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Trainer.compute_affinity_reprojection_loss (Python/Trainer.py:497-516): mean over all elements of
 * (pred - target * drop)^2 with drop = pred where rand < keep_below and 1 elsewhere (rand is the U[0,1) tensor of
 * :502; the segmentation sums of :508-510 do not enter the value).  loss: one device float.  Backward: g_pred =
 * g_loss[0] * d loss / d pred (the reference differentiates through `drop` as well).  workspace:
 * vsl_affinity_loss_workspace_bytes(n) bytes, 8-byte aligned.  Deterministic (fixed-order float64 sums).
 */
size_t vsl_affinity_loss_workspace_bytes(long long n);
int vsl_affinity_loss_fwd(const float* pred, const float* target, const float* rand, float keep_below, long long n,
                          float* loss, void* workspace, size_t workspace_bytes, void* stream);
int vsl_affinity_loss_bwd(const float* pred, const float* target, const float* rand, float keep_below, long long n,
                          const float* g_loss, float* g_pred, void* stream);

/*
 * Depth metrics of a step, Trainer.compute_depth_losses (Python/Trainer.py:622-662) with compute_depth_errors
 * (Python/layers.py:265-283): the predicted depth is resized to the ground truth's size (bilinear,
 * align_corners=False) and clamped, the mask is depth_gt > 0 inside crop rows [y0,y1) x columns [x0,x1)
 * (Python slice semantics; the reference hard-codes 375 x 1242 and the Garg/Eigen crop 153:371, 44:1197), the
 * prediction is scaled by median(gt) / median(pred) over the whole batch (lower middle element, as torch.median)
 * and clamped again, then the seven means.  No host synchronisation, deterministic.
 */
#define VSL_METRIC_ABS_REL 0
#define VSL_METRIC_SQ_REL 1
#define VSL_METRIC_RMSE 2
#define VSL_METRIC_RMSE_LOG 3
#define VSL_METRIC_A1 4
#define VSL_METRIC_A2 5
#define VSL_METRIC_A3 6
#define VSL_METRIC_SUM 7        /* depth_errors of Trainer.py:659: the sum of the seven                    */
#define VSL_METRIC_COUNT 8      /* number of masked pixels (as float)                                      */
#define VSL_METRIC_MEDIAN_GT 9
#define VSL_METRIC_MEDIAN_PRED 10
#define VSL_METRIC_SLOTS 11

typedef struct vsl_depth_metrics_desc {
  int32_t B, H, W;                 /* predicted depth [B,1,H,W], outputs[("depth",0)]              */
  int32_t Hg, Wg;                  /* ground truth [B,1,Hg,Wg], inputs["depth_gt"]                 */
  int32_t crop_y0, crop_y1, crop_x0, crop_x1;
  float clamp_min, clamp_max;      /* 1e-3, 80 (Trainer.py:628-629,648)                            */
  uint32_t reserved;
  const float* depth;
  const float* depth_gt;
} vsl_depth_metrics_desc;

size_t vsl_depth_metrics_workspace_bytes(const vsl_depth_metrics_desc* desc);
/* metrics: out [VSL_METRIC_SLOTS] device floats (NaN means for an empty mask, where the reference raises);
 * workspace: vsl_depth_metrics_workspace_bytes(desc) bytes, 256-byte aligned. */
int vsl_depth_metrics(const vsl_depth_metrics_desc* desc, float* metrics, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Library version, e.g. 200 for 0.2.0 (the first with the 200-byte vsl_head), 210 for 0.2.1 (the *_checked calls). */
int vsl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VSLOSS_H */
