/*
 * gpnerf_hip.h -- C ABI of the MI355X (gfx950) per-ray render path of GP-NeRF.
 *
 * This is the drop-in boundary for the reference's per-ray hot path
 * (libs/renders + libs/nerfheads).  The reference is pure Python: there is no
 * existing FFI; each entry point below names the reference function(s) it
 * replaces (paths relative to the reference root).  The library is loaded with
 * ctypes.CDLL by gp-nerf_amd/_lib.py; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer marked "device" is HBM memory owned by the caller and only
 *     borrowed for the call; "host" pointers are ordinary CPU memory;
 *   - entry points never allocate, never synchronise and never throw; kernels are
 *     enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *   - return value: 0 on success, a negative GPNERF_E_* code otherwise;
 *     gpnerf_strerror() gives the text;
 *   - all arithmetic is fp32 (the reference's dtype); V = 3 source views and
 *     C = 32 feature channels are compiled in (rgb_fc's 96 = 3*32 inputs hard-wire
 *     them in the reference too: libs/nerfheads/trainhead.py:96,143).
 */
#ifndef GPNERF_HIP_H
#define GPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPNERF_VIEWS 3
#define GPNERF_CH 32
#define GPNERF_LEVELS 4

#define GPNERF_OK 0
#define GPNERF_E_ARG (-1)      /* null pointer / bad size */
#define GPNERF_E_LAUNCH (-2)   /* hipLaunchKernel failed; see hipGetLastError */
#define GPNERF_E_DEVICE (-3)   /* not a gfx950 device / no device */

#define GPNERF_FOLD_FIRST_LEVEL 2   /* vol_folded / gpnerf_fold_volumes: the two coarse levels of the four */

/* flags of gpnerf_render_fused */
#define GPNERF_FLAG_NEG_RAY 1u     /* Projector(neg_ray=True): a sample is in front of a source view iff h_z < 0
                                      (BaseRender.py:317-320, demo_render.py:550-553).  The dense renderer pairs it with
                                      GPNERF_FLAG_FLIP_SAMPLES; the progressive renderer's integral never flips */
#define GPNERF_FLAG_FLIP_SAMPLES 16u /* raw2outputs(neg=True): rgb and sigma reversed along the ray before compositing, z not
                                      (BaseRender.py:86-88,101); rgb_in_map pairs the weights with the un-flipped rgb_in (:147) */
#define GPNERF_FLAG_EARLY_TERM 2u  /* a ray stops at the first sample at whose start its transmittance T < term_eps (not in the
                                      reference; everything dropped is bounded by term_eps, depth by term_eps * far).  With the
                                      workspace and at least one round of wavefronts the samples are walked in 16-sample segments,
                                      one launch each, the rays still alive re-packed 32 to a wavefront (same bits per ray in any
                                      ray order); otherwise a 32-ray tile stops once all its rays have */
#define GPNERF_FLAG_SPLIT_F16 8u   /* dense layers on f16 MFMA with every fp32 operand split into f16 hi + lo (three MFMAs per
                                      k-step, f32 accumulation): ~fp32 accuracy (1e-6 on rgb), 3/16 of the fp32 MFMA cost.
                                      Needs frame->head_blob_split; operands must stay below the f16 range (65504):
                                      GPNERF_FLAG_SPLIT_GUARD checks it */
#define GPNERF_FLAG_SPLIT_GUARD 32u /* with GPNERF_FLAG_SPLIT_F16: track the largest magnitude that becomes an MFMA operand in every
                                      32-ray tile (raw features, cross-view mean / variance, every activation); tiles in which it
                                      reaches the f16 range are rendered again by the fp32 form in a second launch of the same
                                      call, so the result never depends on the range of the data.  Needs the workspace
                                      (gpnerf_render_workspace_bytes) and frame->head_blob_ref */
#define GPNERF_FLAG_OCC_CULL 4u    /* the progressive renderer's per-sample rules (libs/renders/demo_render.py): grid coordinates
                                      with its literal voxel size 0.005 instead of frame->voxel (:87-95), a sample is evaluated
                                      only where the occupancy volume (frame->occ) interpolates to > 0 (:270-283), culled
                                      samples carry alpha = 0, colour is kept only where alpha > 1e-14 (:317,:329-341);
                                      ray_mask counts kept samples only.  With the workspace and no per-sample output (weights,
                                      raw) the keep decisions are made in a pass before the launch and the tiles are handed out
                                      longest first; same bits either way */

#define GPNERF_FLAG_REF_ORDER 64u   /* the fp32 form in the REFERENCE's arithmetic order even when the frame carries folded volumes:
                                      every dense layer as sgemm's chain (k ascending from zero, bias last) in the unscaled
                                      domain, x / 3 with IEEE rounding, multiply-then-add trilinear taps, no folded levels.
                                      On trained parameters it sits at the op-for-op CPU oracle's distance from the reference
                                      where the folded form is 5-10 x further.  Needs frame->head_blob_ref */

#define GPNERF_FLAG_NO_EXITS 128u   /* diagnostic: every layer of every sample is evaluated -- without the fp32 forms' bit-exact exits
                                      (the sigma feature layer in empty space; the colour branch run only for samples whose weight
                                      alpha * T is not zero; the sample loop ended where every ray's transmittance is exactly 0:
                                      GpnerfOutputs.step_stats) -- the A/B that shows they change no bit */
#define GPNERF_FLAG_SHARED_DEVICE 256u /* other processes' kernels share this device: no launch of this call waits for its own
                                      workgroups.  (By default a launch on the tile queue that lists its colour work lets its own
                                      wavefronts evaluate the list once they have no tile left, every wavefront reporting before any
                                      leaves: fine on a device the process has to itself -- one process per GPU -- or shares with
                                      kernels that end on their own, a deadlock hazard only against another tenant's kernel that
                                      waits the same way while holding compute units.  With this flag the list goes to a second
                                      kernel: the same bits, 1-10 % slower.) */
#define GPNERF_FLAG_RESERVE_CUS(n) (((uint32_t)(n) & 0xffu) << 24)
                                   /* bits 24..31: plan the launch for n fewer compute units (rounded down to a multiple of 8: one
                                      per XCD round).  The persistent workgroups then leave n CUs idle for kernels of other
                                      streams -- a pipelined evaluation loop runs the NEXT frame's encoder and volume builder
                                      there (an experiment of the pipelined evaluation loop: profiles/r05/d_pipeline.txt).  The maps are those of
                                      a chip with n fewer CUs: the same bits unless that makes the launch split tiles (see `workspace`) */

/* Per-frame constants (everything render_rays reads that does not depend on the ray).
 * Layouts are channels-last so that one bilinear / trilinear tap is one contiguous
 * 128-byte line; gpnerf_relayout_* produce them from the reference's NCHW tensors. */
typedef struct GpnerfFrame {
    const float* vol[GPNERF_LEVELS];        /* device; level k: [D_k][H_k][W_k][32]; SparseConvNet.py:111 `.dense()` */
    int32_t vol_dhw[GPNERF_LEVELS][3];
    const float* featmaps;                  /* device; [V][fh][fw][32]; encoder output, BaseRender.py:222 */
    int32_t feat_h, feat_w;
    const float* imgs;                      /* device; [V][H][W][4] = r,g,b,0 in [0,1]; BaseRender.py:231 */
    int32_t img_h, img_w;
    float proj[GPNERF_VIEWS][12];           /* rows 0..2 of K4 @ P4, row-major 3x4; BaseRender.py:233-247,314 */
    float Rh[9];                            /* batch['Rh'][0], row-major; BaseRender.py:52-60 */
    float Th[3];
    float bounds_min[3];                    /* batch['bounds'][0,0], SMPL-frame xyz; BaseRender.py:65 */
    float voxel[3];                         /* cfg.dataset.voxel_size (applied in d,h,w order); BaseRender.py:67 */
    int32_t out_sh[3];                      /* batch['out_sh'] d,h,w; BaseRender.py:69-70 */
    const float* head_blob;                 /* device or NULL (needed only by the folded form and gpnerf_fold_volumes);
                                               gpnerf_pack_head() image, gpnerf_head_blob_floats() floats */
    const float* head_blob_split;           /* device or NULL; gpnerf_pack_head_split() image (GPNERF_FLAG_SPLIT_F16) */
    const float* occ;                       /* device or NULL; [D_1][H_1][W_1] occupancy `masks3d` at level-1 size
                                               (SparseConvNet.py:135-139), read only with GPNERF_FLAG_OCC_CULL */
    const float* vol_folded[GPNERF_LEVELS]; /* device or NULL; levels GPNERF_FOLD_FIRST_LEVEL.. (all of them or none; the finer
                                               levels' entries are ignored): [D_k][H_k][W_k][64], written by gpnerf_fold_volumes
                                               from vol[] and head_blob.  With them the fp32 form of gpnerf_render_fused
                                               interpolates these levels' share of the sigma feature layer's pre-activation
                                               instead of running it per sample (same result up to fp32 rounding) */
    const float* head_blob_ref;             /* device or NULL; gpnerf_pack_head_ref() image: the fp32 form in the reference's
                                               summation order (GPNERF_FLAG_REF_ORDER, frames without vol_folded, and the
                                               fix-up launch of GPNERF_FLAG_SPLIT_GUARD) */
} GpnerfFrame;

/* The per-ray MLP parameters in PyTorch layout (weight [out][in] row-major, bias [out]),
 * host memory.  Names follow the reference state_dict. */
typedef struct GpnerfHeadParams {
    const float *geo_w, *geo_b;   /* sigmahead.out_geometry_fc.0  64x128 */
    const float *b1_w, *b1_b;     /* rgbhead.base_fc.0            64x105 */
    const float *b2_w, *b2_b;     /* rgbhead.base_fc.2            32x64  */
    const float *v1_w, *v1_b;     /* rgbhead.vis_fc.0             32x32  */
    const float *v2_w, *v2_b;     /* rgbhead.vis_fc.2             32x32  */
    const float *r1_w, *r1_b;     /* rgbhead.rgb_fc.0             32x96  */
    const float *r2_w, *r2_b;     /* rgbhead.rgb_fc.2             16x32  */
    const float *r3_w, *r3_b;     /* rgbhead.rgb_fc.4              3x16  */
    const float *d1_w, *d1_b;     /* rgbhead.out_geometry_fc.0    64x134 */
    const float *d2_w, *d2_b;     /* rgbhead.out_geometry_fc.2    32x64  */
    const float *d3_w, *d3_b;     /* rgbhead.out_geometry_fc.4    16x32  */
    const float *d4_w, *d4_b;     /* rgbhead.out_geometry_fc.6     1x16  */
} GpnerfHeadParams;

/* Outputs of Renderer.render_rays (BaseRender.py:148-156), all device, [N,...] row-major.
 * rgb/depth/acc/disp are required, the rest may be NULL. */
typedef struct GpnerfOutputs {
    float* rgb;        /* [N,3]  rgb_map   */
    float* depth;      /* [N]    depth_map */
    float* acc;        /* [N]    acc_map   */
    float* disp;       /* [N]    disp_map  */
    float* weights;    /* [N,S]  ret['alpha'] (= weights) */
    float* z_vals;     /* [N,S]  */
    float* rgb_in;     /* [N,9]  rgb_in_map (view-major, then rgb) */
    uint8_t* ray_mask; /* [N]    raw2outputs' mask: #samples with >1 valid view > 8 */
    float* raw;        /* [N,S,4] NeRFHead.forward output (rgb, sigma), un-flipped sample order */
    int32_t* samples_done; /* [N]  diagnostic: samples the ray's wavefront evaluated (S unless early termination / culling
                              skipped some); with it the launch never splits a tile's samples over several wavefronts */
    uint32_t* step_stats;  /* [8] or NULL, zeroed by the caller; the launch ADDS, per wavefront step of 32 samples:
                              [0] steps the launch answers for (sample-loop steps + [3]);
                              [1] steps without the sigma feature layer: all four volume levels exactly zero in all 32 samples
                                  (reference-order form: ELU(bias) without the layer's MFMAs), or counted in [3];
                              [2] [0] MINUS [5];
                              [3] steps settled BEHIND the sample loop, without a gather or an MFMA, because every ray's
                                  transmittance was exactly 0 (zero weights and the ray_mask count are all they still owe);
                              [4] volume LEVELS left out of the sigma feature layer (each a quarter of the layer; 0..4 per
                                  sample-loop step: a level whose 16 features are zero in all 32 samples adds fma(w, 0, s) = s);
                              [5] evaluations of the colour branch on 32 samples: one per step where it runs in the step, one
                                  per colour pass where it is deferred -- a sample whose weight alpha * T is exactly 0 adds
                                  fma(0, rgb, c) = c to the colour map, so only the samples that need it are evaluated, 32 at a
                                  time: listed for the launch as a whole where the workspace has room for the list (fp32 forms
                                  on the tile queue) and evaluated by the launch's own wavefronts once they have no tile left
                                  (launches on the tile queue with no remainder launch behind them) or by a second kernel (exactly ceil(listed / 32) evaluations),
                                  out of a queue per wavefront otherwise (never with `raw`, never under GPNERF_FLAG_NO_EXITS);
                              [6], [7] reserved (0).
                              All of it is bit-exact; bench.py prices its roofline on the work done:
                              sample-loop steps x (everything but the colour branch) - [4] x layer / 4 + [5] x colour branch */
} GpnerfOutputs;

/* Number of floats of the packed head image. */
int64_t gpnerf_head_blob_floats(void);

/* Re-arrange the PyTorch-layout parameters into the LDS image the kernels stage
 * (MFMA A-operand order).  Host-side, model-load time.
 * Replaces nothing in the reference; it is what load_state_dict is to nn.Linear. */
int gpnerf_pack_head(const GpnerfHeadParams* params_host, float* blob_host);

/* The same parameters for the reference-order fp32 form (GPNERF_FLAG_REF_ORDER): gpnerf_head_blob_floats() floats, rows and
 * columns in the order libs/nerfheads/trainhead.py's nn.Linear layers accumulate them on the CPU (sgemm), nothing pre-scaled. */
int gpnerf_pack_head_ref(const GpnerfHeadParams* params_host, float* blob_host);

/* The same parameters as f16 hi/lo pairs in v_mfma_f32_32x32x16_f16 A-operand order (GPNERF_FLAG_SPLIT_F16). */
int64_t gpnerf_head_blob_split_floats(void);
int gpnerf_pack_head_split(const GpnerfHeadParams* params_host, float* blob_host);

/* sigmahead.out_geometry_fc (trainhead.py:39-40,58) is linear in the 4 x 32 volume features, and F.grid_sample
 * (SparseConvNet.py:113-116) is linear in the voxels: Linear(sum_t w_t v_t) = sum_t w_t Linear(v_t).  For the coarse levels
 * k >= GPNERF_FOLD_FIRST_LEVEL this applies the layer's 32 columns of level k (without the bias) to every voxel of vol[k] --
 * once per frame instead of once per sample -- and writes 64 values per voxel in the order the sample loop accumulates them.
 * out: host array of GPNERF_LEVELS device pointers (entries below GPNERF_FOLD_FIRST_LEVEL unused), D_k * H_k * W_k * 64 floats. */
int gpnerf_fold_volumes(const GpnerfFrame* frame, float* const* out, void* stream);

/* Fused sample -> gather -> MLP -> composite over N rays.
 * Replaces Renderer.batchify_rays + render_rays with is_train=False
 * (libs/renders/BaseRender.py:110-184): get_sampling_points :35-50, pts_to_can_pts :52-60,
 * get_grid_coords :62-73, Projector.compute :326-363 (sample part), SparseConvNet.forward's
 * trilinear sampling (libs/nerfheads/networks/SparseConvNet.py:113-122),
 * NeRFSigmaHead.out_geometry_fc + NeRFRGBHead.forward (libs/nerfheads/trainhead.py:39-40,58,118-145)
 * and raw2outputs :75-107, rgb_in_map :147.
 *   rays: device [N][8] = origin(3), direction(3, un-normalised), near, far (BaseRender.py:250)
 *   term_eps: transmittance threshold, read only with GPNERF_FLAG_EARLY_TERM
 *   ray_order: optional device [N] list of distinct row indices (NULL = identity): launch slot i renders the ray in row
 *     ray_order[i] of `rays`; inputs are read and outputs written at that row, so results do not depend on the order.  In the
 *     plain case it is a permutation of 0..N-1; it may also pick N rows out of larger `rays` / output arrays (the progressive
 *     renderer passes every pixel's ray and the list of selected pixels: rows not listed are neither read nor written).
 *     It only decides which 32 rays share a wavefront and which 256 share a workgroup: pass image patches
 *     (e.g. 32x8 pixels per workgroup) so neighbouring rays hit the same cache lines.
 *   workspace: optional device scratch of gpnerf_render_workspace_bytes() bytes (NULL = none).  With it the launch balances
 *     its load: frames of more than one round of workgroups run as persistent workgroups that pull 32-ray tiles from a queue
 *     in the workspace (a tile's cost varies under early termination / culling; results are unchanged, bit for bit), and
 *     frames too small to fill the chip let 2, 4 or 8 wavefronts share the samples of one tile and merge their partial
 *     composites (second small launch); the transmittance product is then associated per segment, a ~1e-7 relative
 *     difference (never with GPNERF_FLAG_EARLY_TERM).  On the tile queue the fp32 forms also keep the LIST of the samples whose
 *     colour branch has to run there (32 bytes per sample of the launch: an entry and a result; launches of up to 2^26 samples)
 *     -- the sample loop then only lists them, the list is evaluated 32 entries per wavefront step, balanced whatever the rays
 *     (by the same launch's wavefronts as they run out of tiles, or by a second kernel: behind the segment launches of early
 *     termination, and always under GPNERF_FLAG_SHARED_DEVICE), and a last kernel adds every ray's terms in sample order: the colour map's bits are those of the loop
 *     that evaluates them in place, which is what a workspace too small for the list gets (gpnerf_render_workspace_bytes
 *     includes it).
 *     The workspace is private to the call until the stream reaches its end. */
int gpnerf_render_fused(const GpnerfFrame* frame, const float* rays, int64_t n_rays, int32_t n_samples,
                        uint32_t flags, float term_eps, const int32_t* ray_order, const GpnerfOutputs* out,
                        void* workspace, size_t workspace_bytes, void* stream);
/* Bytes of workspace gpnerf_render_fused can use for this launch (0: it would not split); any smaller amount is valid -- the
 * launch keeps to the forms that fit. */
size_t gpnerf_render_workspace_bytes(int64_t n_rays, int32_t n_samples);
/* GPNERF_FLAG_SPLIT_GUARD keeps its state in the last gpnerf_render_guard_bytes(n_rays) bytes of the workspace (start rounded down
 * to 256): word 0 = number of flagged tiles once the stream has passed the call, words 64.. = one flag per 32-ray tile. */
size_t gpnerf_render_guard_bytes(int64_t n_rays);

/* Stage entry points (the same device code as the fused kernel, one reference function per launch).
 *
 * get_sampling_points + pts_to_can_pts + get_grid_coords (libs/renders/BaseRender.py:35-73), jitter off:
 *   pts [N][S][3] world points, z_vals [N][S], grid [N][S][3] normalised volume coords (xyz); any may be NULL. */
int gpnerf_sample_points(const GpnerfFrame* frame, const float* rays, int64_t n_rays, int32_t n_samples, float* pts,
                         float* z_vals, float* grid, void* stream);
/* SparseConvNet.forward's F.grid_sample over the 4 dense levels
 * (libs/nerfheads/networks/SparseConvNet.py:113-122): grid [P][3] -> vol_feat [P][128] (level-major). */
int gpnerf_sample_volume(const GpnerfFrame* frame, const float* grid, int64_t n_points, float* vol_feat, void* stream);
/* Projector.compute for sample points (libs/renders/BaseRender.py:326-363, without the SMPL-vertex branch):
 *   pts [P][3] world -> rgb_feat [P][V][35] = (rgb, 32 features), mask [P][V] (0/1). */
int gpnerf_project_gather(const GpnerfFrame* frame, const float* pts, int64_t n_points, int32_t neg_ray, float* rgb_feat,
                          float* mask, void* stream);

/* NeRFHead.forward on already-gathered features (libs/nerfheads/trainhead.py:159-163, with the
 * sparse volume replaced by its sampled features): P points.
 *   vol_feat [P][128] (level-major), rgb_feat [P][V][35], mask [P][V] (0/1 floats), all device
 *   raw [P][4] = rgb, sigma.   head_blob_ref: the gpnerf_pack_head_ref() image (reference summation order). */
int gpnerf_head_forward(const float* head_blob_ref, const float* vol_feat, const float* rgb_feat, const float* mask,
                        int64_t n_points, float* raw, void* stream);

/* The two halves of the head as the reference's progressive renderer calls them (libs/renders/demo_render.py:295-326):
 * gpnerf_sigma_features = NeRFSigmaHead.test_forward (libs/nerfheads/trainhead.py:61-76) after its volume sampling:
 *   vol_feat [P][128], rgb_feat [P][V][35] -> sigma_feat [P][64] = ELU(Linear(vol_feat)), globalfeat [P][134] =
 *   [sigma_feat, mean over views (35), population variance over views (35)];
 * gpnerf_rgb_head_forward = NeRFRGBHead.forward (:118-145): sigma_feat [P][64], rgb_feat [P][V][35], mask [P][V] ->
 *   raw [P][4] = (rgb_out, sigma_out).  All device; head_blob_ref: the gpnerf_pack_head_ref() image. */
int gpnerf_sigma_features(const float* head_blob_ref, const float* vol_feat, const float* rgb_feat, int64_t n_points,
                          float* sigma_feat, float* globalfeat, void* stream);
int gpnerf_rgb_head_forward(const float* head_blob_ref, const float* sigma_feat, const float* rgb_feat, const float* mask,
                            int64_t n_points, float* raw, void* stream);

/* Renderer.raw2outputs (BaseRender.py:75-107) alone.  raw [N][S][4], z [N][S],
 * nvalid [N][S] = per-sample number of valid views (may be NULL), all device. */
int gpnerf_composite(const float* raw, const float* z_vals, const float* nvalid, int64_t n_rays, int32_t n_samples,
                     int32_t neg, const GpnerfOutputs* out, void* stream);

/* get_rays + get_near_far (libs/datasets/data_utils.py:47-63,96-130) for one target camera, in the precision the dataset
 * runs them in (sample_ray's test branch, :294-300): float64 camera products rounded once to float32 rays, float64 plane hits
 * and on-box tests (`bounds + [-0.01, 0.01]` promotes them, :98), float32 norm_ray, distances rounded to float32.
 * mask_at_box, rays, near and far are bit-exact against the reference's numpy run (tests/golden/rays_*.npz).
 *   Kinv, Rinv: host 3x3 row-major float64 inverses (np.linalg.inv of K, R); cam_o: host [3] float64 camera centre -Rinv @ T;
 *   bounds: host [2][3] float32 world AABB (un-padded).
 *   rays: device [H*W][8]; hit: device [H*W] uint8 (mask_at_box).  Rays are written at their
 *   pixel index; the caller keeps the hit ones in raster order. */
int gpnerf_make_rays(int32_t H, int32_t W, const double* Kinv, const double* Rinv, const double* cam_o,
                     const float* bounds, float* rays, uint8_t* hit, void* stream);

/* SparseConvNet.encode's occupancy volume (libs/nerfheads/networks/SparseConvNet.py:135-139):
 * occ[d][h][w] = sum over the 4 levels of (channel sum of level k, nearest-upsampled to level-1 size).
 * Reads frame->vol / vol_dhw (channels-last); occ: device [D_1][H_1][W_1]. */
int gpnerf_build_occupancy(const GpnerfFrame* frame, float* occ, void* stream);

/* Progressive ray selection of the inference renderer (libs/renders/demo_render.py:166-200): every level-1 voxel with
 * occ > threshold (SparseConvNet.py:140: 0.1) is mapped to a world point (voxel index * 2 * voxel + bounds_min, then
 * @ Rh^T + Th), projected with the target camera (pose 3x4 row-major [R|T], K 3x3), and its 4 neighbouring pixels
 * (truncation toward zero, clamped) are marked in pixel_sel (device [img_h*img_w], cleared here).  world_minmax: device
 * int32[6] = order-preserving integer images of min xyz / max xyz of the world points (decode: i >= 0 ? bits : bits ^ 0x7FFFFFFF).
 * voxel_xyz, bounds_min, Rh, Th, pose, K: host. */
int gpnerf_select_pixels(const float* occ, int32_t D, int32_t H, int32_t W, float threshold, const float* voxel_xyz,
                         const float* bounds_min, const float* Rh, const float* Th, const float* pose, const float* K,
                         int32_t img_h, int32_t img_w, uint8_t* pixel_sel, int32_t* world_minmax, void* stream);
/* The inference renderer's on-device get_rays / near-far (libs/renders/demo_render.py:201-239): pixel_camera = xy1 @ Kinv^T,
 * pixel_world = (pixel_camera - T) @ R, rays_o = (-R^T) @ T, with every length-3 product accumulated as torch's CPU `@` does
 * (fused multiply-adds over k = 0,1,2), the box used as given (no +-0.01), directions not clamped, distances by torch.norm's
 * formula, and under neg_ray the second distance negated.  Kinv: host 3x3 (batch['target_K_inv']); pose: host 3x4 row-major
 * [R|T] (batch['target_pose']); the box: bounds, host [2][3], or -- when world_minmax_dev is not NULL -- what
 * gpnerf_select_pixels left on the device, decoded and z-padded by 0.05 (demo_render.py:168-175) inside the kernel, so the two
 * launches need no host round trip between them.  pixel_sel: optional device [H*W] mask of the pixels to consider
 * (others get hit = 0).  Bit-exact against the reference's CPU run (tests/golden/demo_*.npz). */
int gpnerf_make_rays_demo(int32_t H, int32_t W, const float* Kinv, const float* pose, const float* bounds,
                          const int32_t* world_minmax_dev, int32_t neg_ray, const uint8_t* pixel_sel, float* rays, uint8_t* hit,
                          void* stream);

/* ---- per-frame sparse convolution pyramid (gpnerf_volume.hip), replacing the external spconv v1.2.1 calls of
 * libs/nerfheads/networks/SparseConvNet.py:22-111 (SubMConv3d / SparseConv3d + BatchNorm1d + ReLU, .dense()).
 * A sparse tensor is: features [M][C] fp32, coords [M][3] int32 (d,h,w), and a dense int32 index grid [D][H][W]
 * (row of the site, -1 = inactive).  Row counts may live on the device (m_dev, NULL = use m_cap) so a chain of levels
 * needs no host synchronisation; m_cap bounds every launch.  All pointers device unless noted; dims: host int32[3].
 * Parity unpinned: spconv is not in the reference tree. */
/* grid <- -1, then grid[coords[i]] = i (highest row wins for duplicate voxels). */
int gpnerf_sparse_index(const int32_t* coords, const int32_t* m_dev, int32_t m_cap, const int32_t* dims, int32_t* grid,
                        void* stream);
/* 3x3x3 conv + folded BatchNorm + ReLU over the sites listed in out_coords.
 * strided = 0: submanifold (SubMConv3d; in/out share sites and grid), strided = 1: SparseConv3d(k=3, s=2, p=1) reading
 * the finer level (in_grid/in_dims) at the coarser sites of out_coords.  weight [27][cin][cout] (spconv's [3,3,3,Ci,Co]). */
int gpnerf_sparse_conv3(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                        const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const float* weight, int32_t cout,
                        const float* bn_scale, const float* bn_shift, float* out, void* stream);
/* The same convolution on the matrix cores (32 output sites per wavefront, v_mfma_f32_32x32x2_f32; cin a multiple of 8,
 * cin, cout <= 32).  packed_weight: device copy of gpnerf_sparse_pack_weight()'s image of the [27][cin][cout] weight
 * (host side, model-load time; gpnerf_sparse_packed_weight_floats(cin) floats).  Same result as gpnerf_sparse_conv3 up to
 * fp32 summation order. */
int64_t gpnerf_sparse_packed_weight_floats(int32_t cin);
int gpnerf_sparse_pack_weight(const float* weight_host, int32_t cin, int32_t cout, float* packed_host);
int gpnerf_sparse_conv3_mfma(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                             const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const float* packed_weight,
                             int32_t cout, const float* bn_scale, const float* bn_shift, float* out, void* stream);
/* The same convolution in the encoder's split-precision arithmetic (cin = 16 or 32, cout <= 32): every fp32 operand as f16 hi + lo,
 * three v_mfma_f32_32x32x16_f16 per 16 input channels, f32 accumulation -- the fp32 matrix instruction above runs at 1/16 of the
 * f16 rate.  packed_weight16: device copy of gpnerf_sparse_pack_weight16()'s image (host side, model-load time;
 * gpnerf_sparse_packed_weight16_bytes(cin) bytes; refuses |w| >= 15.99): per (tap, 16-channel chunk) the scaled weights as f16
 * hi | lo and once more in fp32.  A wavefront that gathers a value beyond the f16 range (|x| >= 4 094) runs that tap on the fp32
 * instructions with the fp32 copy.  Within 2^-21 (relative to the sum of |terms|) of gpnerf_sparse_conv3_mfma. */
int64_t gpnerf_sparse_packed_weight16_bytes(int32_t cin);
int gpnerf_sparse_pack_weight16(const float* weight_host, int32_t cin, int32_t cout, void* packed_host);
int gpnerf_sparse_conv3_mfma16(int32_t strided, const float* in, int32_t cin, const int32_t* in_grid, const int32_t* in_dims,
                               const int32_t* out_coords, const int32_t* m_dev, int32_t m_cap, const void* packed_weight16,
                               int32_t cout, const float* bn_scale, const float* bn_shift, float* out, void* stream);
/* The whole pyramid of SparseConvNet.py:90-111 behind two calls (the entries above, in the reference's order, enqueued from
 * native code: ~60 launches without a trip through the host language between them).  The caller owns every buffer.
 *   gpnerf_sparse_pyramid_plan: everything that depends on the vertices' voxel coordinates only -- the full-resolution index grid,
 *     every coarse level's site list + grid (gpnerf_sparse_down_sites) and the zeroed dense volumes; may run on another stream
 *     beside the image encoder.
 *   gpnerf_sparse_pyramid_run: double_conv at full resolution, the duplicate merge, then per level strided conv + double_conv +
 *     scatter into vol[i] (channels-last [D_i][H_i][W_i][ch_i]).  convs: 2 + 3 * n_levels entries in network order.
 * feat_a / feat_b: two float buffers of max(m0, cap[i]) * 32 each (ping-pong).  code: [m0][code_ch] per-vertex features.
 * Rows that share a voxel (two vertices rounded into one 5 mm cell) follow spconv v1.2.1's rulebook as oracle/producers_ref.py's
 * header recalls it: a submanifold convolution gives the voxel's OWNER row (the highest) the sum over ALL rows of its neighbour
 * voxels and every other row of the voxel only its own centre term; a strided convolution takes every row. */
#define GPNERF_PYRAMID_MAX_LEVELS 4
typedef struct GpnerfSparseConv {
    int32_t strided, cin, cout;
    int32_t form;                 /* 2: gpnerf_sparse_conv3_mfma16 (weight = its packed image), 1: _mfma, 0: gpnerf_sparse_conv3 (raw [27][cin][cout]) */
    const void* weight;
    const float* bn_scale;
    const float* bn_shift;
    const float* weight_raw;      /* device [27][cin][cout] fp32, spconv's own layout: needed for the two vertex-level convolutions
                                     (rows that share a voxel are recomputed from it), may be NULL for the others */
} GpnerfSparseConv;
typedef struct GpnerfPyramid {
    int32_t n_levels, m0;
    int32_t dims0[3];
    int32_t dims[GPNERF_PYRAMID_MAX_LEVELS][3];
    int32_t cap[GPNERF_PYRAMID_MAX_LEVELS];
    int32_t ch[GPNERF_PYRAMID_MAX_LEVELS];
    const int32_t* coords0;       /* [m0][3] (d, h, w) */
    int32_t* grid0;               /* dims0 cells */
    int32_t* dup_scratch;         /* 9 * m0 */
    int32_t* grid[GPNERF_PYRAMID_MAX_LEVELS];
    int32_t* coords[GPNERF_PYRAMID_MAX_LEVELS];     /* [cap][3] */
    int32_t* m[GPNERF_PYRAMID_MAX_LEVELS];          /* device row counts */
    float* vol[GPNERF_PYRAMID_MAX_LEVELS];
    float* feat_a;
    float* feat_b;
    float* feat_c;                /* m0 * 32 floats: the vertex level's input with every shared voxel's rows summed into its owner */
} GpnerfPyramid;
int gpnerf_sparse_pyramid_plan(const GpnerfPyramid* p, void* stream);
int gpnerf_sparse_pyramid_run(const GpnerfPyramid* p, const float* code, int32_t code_ch, const GpnerfSparseConv* convs, int32_t n_convs,
                              void* stream);
/* Before the first strided conv: feat[owner] += feat[i] for every row i whose voxel is indexed by another row (two
 * vertices rounded into one voxel).  spconv's strided rulebook takes every input row; its submanifold lookups one.
 * The rows of a voxel are added in ascending row order (deterministic); scratch: device int32[9 * m], overwritten (a count
 * and eight row slots per row). */
int gpnerf_sparse_merge_duplicates(float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, int32_t m,
                                   const int32_t* dims, int32_t* scratch, void* stream);
/* Active sites of the next (half-resolution) level: out_grid / out_coords / m_out_dev from the finer level's coords. */
int gpnerf_sparse_down_sites(const int32_t* in_coords, const int32_t* m_in_dev, int32_t m_in_cap, const int32_t* out_dims,
                             int32_t* out_grid, int32_t* out_coords, int32_t* m_out_dev, int32_t m_out_cap, void* stream);
/* .dense(): zero-filled [D][H][W][C] volume with the active sites' features (channels-last, as GpnerfFrame.vol wants). */
int gpnerf_sparse_to_dense(const float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, const int32_t* m_dev,
                           int32_t m_cap, const int32_t* dims, float* vol_ndhwc, void* stream);
/* The same in two steps, for callers that lay out a frame's pyramid BEFORE its features exist (the structure of the pyramid --
 * index grids, coarse site lists, zeroed dense volumes -- depends on the vertices' voxel coordinates only, so it can be enqueued
 * on a side stream while the image encoder runs): gpnerf_zero_volume now, gpnerf_sparse_scatter_dense(prezeroed = 1) later. */
int gpnerf_zero_volume(float* vol_ndhwc, int32_t channels, const int32_t* dims, void* stream);
int gpnerf_sparse_scatter_dense(const float* feat, int32_t channels, const int32_t* coords, const int32_t* grid, const int32_t* m_dev,
                                int32_t m_cap, const int32_t* dims, float* vol_ndhwc, int32_t prezeroed, void* stream);

/* Vertex-code attention of the volume builder (libs/nerfheads/trainhead.py:48-52, networks/MultiHeadAttention.py:61-98 with
 * sum=False): q [n][d_model] vertex codes, kv [n][views][kv_dim] the vertices' per-view features, weights in PyTorch
 * layout (w_qs [d_model][d_model], w_ks / w_vs [d_model][kv_dim], fc [d_model][d_model]); out [n][d_model].  All device.
 * d_model, kv_dim <= 64, views <= 4, d_model / n_head a power of two. */
int gpnerf_vertex_attention(const float* q, const float* kv, const float* w_qs, const float* w_ks, const float* w_vs,
                            const float* fc, int32_t n, int32_t d_model, int32_t kv_dim, int32_t n_head, int32_t views,
                            float* out, void* stream);

/* The image encoder on channels-last activations (gpnerf_conv.hip), all tensors device fp32 [N][H][W][C].
 * conv2d_nhwc = nn.Conv2d(cin, cout, ks, stride, padding=ks/2, padding_mode='reflect') (UNet.py:6-14,108-115,154-155) as an
 *   implicit GEMM on the matrix cores, in one of two arithmetic forms chosen by `exact` (the SAME kernels, tiles, staging and
 *   fused InstanceNorm tables either way; `packed` must be the image gpnerf_conv_pack_weight wrote for that form):
 *     exact = 1  fp32 operands on v_mfma_f32_32x32x2_f32: every dot product an fp32 FMA chain over (channel block, tap, channel),
 *                the reference's own arithmetic up to the order of the sum; no operand range, range_flag is not looked at.  The
 *                encoder's default (ResUNet.precision = "fp32"): the end-to-end chain then stays inside 1e-4 of the reference.
 *     exact = 0  fp32 operands split into f16 hi + lo on v_mfma_f32_32x32x16_f16 (three MFMAs per k-step, f32 accumulation:
 *                ~23 bits per operand, 3/16 of the matrix time); operands must stay below the f16 range, which InstanceNorm'd /
 *                ReLU'd activations of images do -- see range_flag.  The fast mode (ResUNet.precision = "split").
 *   ks in {1, 3, 7}, stride in {1, 2}; cin a multiple of 16, or < 8 for the 7x7/2 stem; cout a multiple of 4.
 *   packed: gpnerf_conv_pack_weight()'s device image of the PyTorch weight [cout][cin][ks][ks] (gpnerf_conv_packed_bytes()
 *   bytes; re-pack when the parameter changes); bias: [cout] or NULL.  y: [N][Ho][Wo][cout], Ho = (H + 2 (ks/2) - ks) / stride + 1.
 *   tile_stats: NULL, or [N][gpnerf_conv_out_tiles()][cout][3] floats that receive every workgroup tile's per-channel sum, sum of
 *   squares, and M2 (sum of squares about the tile's own mean) of the outputs -- the statistics the InstanceNorm behind the
 *   convolution needs, without re-reading y.  out_table below takes the variance as E[y^2] - mean^2 from the sums where that is
 *   well conditioned and from the M2's (merged as Chan et al.) on a channel whose values are nearly constant over the image.
 * instance_norm_act_nhwc = act(InstanceNorm2d(x; gamma, beta, eps, biased variance, no running statistics) [+ residual]),
 *   act 0 none / 1 ReLU / 2 ELU (UNet.py:38-53,117-120,180-183); statistics from a double-precision pass over x, added up in a
 *   fixed order (deterministic): the stand-alone operator, what the fused tables of conv2d_norm_nhwc are tested against;
 *   scratch: gpnerf_instance_norm_nhwc_scratch_bytes() bytes.
 * conv2d_norm_nhwc = conv2d_nhwc with the InstanceNorms on either side of it fused in (a residual unit is conv - norm - ReLU -
 *   conv - norm, UNet.py:38-53):
 *     in_table  NULL, or [N][3][cin] floats (mean, gamma * rstd, beta per channel, what out_table below produces): the
 *               convolution then reads act((x - mean) * scale + beta) instead of x while it stages its input, act = ReLU for
 *               in_act 1, identity for 0 -- the normalised tensor is never written.  3x3 and 1x1 convolutions with cin % 16 == 0.
 *     out_table NULL, or [N][3][cout] floats that receive mean / gamma * rstd / beta of InstanceNorm2d(y; gamma, beta, eps): the
 *               last workgroup to finish an (image, 32..64-channel group) merges that group's tile_stats rows in double, in a fixed
 *               order (deterministic), so no separate reduction launch follows the convolution.  Needs tile_stats, gamma, beta and
 *               `counters`: at least N * ceil(cout / 32) uint32 words that are zero before the call; they are zero again after it.
 * conv2d_norm_cat_nhwc = the 3x3 stride-1 convolution of conv2d_norm_nhwc on the channel concatenation [x (cin_a channels), x_b (cin_b)]
 *   of two NHWC tensors of one size (UNet.py:199-211's torch.cat([up, skip], 1) in front of iconv3 / iconv2), read in place: the
 *   concatenated tensor is never written.  cin_a, cin_b multiples of 16; `packed` is the image of the [cout][cin_a + cin_b][3][3] weight.
 * norm_apply_nhwc = act((x - mean) * scale + beta [+ residual]) from such a table; with res_table the residual is itself
 *   normalised on the fly ((residual - rmean) * rscale + rbeta: the projected shortcut's InstanceNorm, UNet.py:48-51).
 * upsample2x_nhwc = F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) (UNet.py:129).
 * range_flag (the three convolutions, exact = 0 only): NULL, or ONE uint32 word (device memory, or pinned host memory the device can
 *   write) that the call sets to 1 -- it never clears it -- when an operand was beyond what the f16 hi/lo split holds (an
 *   activation with |x| >= 4095, a weight with |w| >= 16, or a non-finite input): such an operand splits into f16 infinities and
 *   the outputs it meets are NaN, which the call finds in the sums of the InstanceNorm table (out_table given) or in its
 *   accumulators (no norm behind it).  A result produced with the flag raised must be discarded and the convolution chain run
 *   again with exact = 1 (fp32 operands, no range), so that parameters of any size are served (a trained InstanceNorm scale times sqrt(h w) can
 *   exceed the range on a one-hot image; ordinary images stay orders of magnitude below it).
 * conv2d_nhwc_exact = the same nn.Conv2d on fp32 operands (v_mfma_f32_32x32x2_f32, an fp32 FMA chain per output, any odd ks,
 *   any stride, any channel counts), from the PyTorch weight [cout][cin][ks][ks] as it is, one scalar load per operand: the
 *   independent restatement the tests hold the fused exact = 1 form against (and a way to run shapes the fused kernels do not
 *   cover); ~20x slower, not on the encoder's path. */
int64_t gpnerf_conv_packed_bytes(int32_t cout, int32_t cin, int32_t ks);
int gpnerf_conv_pack_weight(const float* weight, int32_t cout, int32_t cin, int32_t ks, int32_t exact, void* packed, void* stream);
int32_t gpnerf_conv_out_tiles(int32_t h, int32_t w, int32_t cin, int32_t ks, int32_t stride);
int gpnerf_conv2d_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const void* packed, const float* bias,
                       int32_t cout, int32_t ks, int32_t stride, float* y, float* tile_stats, uint32_t* range_flag, int32_t exact,
                       void* stream);
int gpnerf_conv2d_norm_cat_nhwc(const float* x, int32_t cin_a, const float* x_b, int32_t cin_b, int32_t n, int32_t h, int32_t w,
                                const void* packed, const float* bias, int32_t cout, float* y, float* tile_stats, const float* gamma,
                                const float* beta, float eps, float* out_table, uint32_t* counters, uint32_t* range_flag, int32_t exact,
                                void* stream);
int gpnerf_conv2d_norm_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const float* in_table, int32_t in_act,
                            const void* packed, const float* bias, int32_t cout, int32_t ks, int32_t stride, float* y, float* tile_stats,
                            const float* gamma, const float* beta, float eps, float* out_table, uint32_t* counters, uint32_t* range_flag,
                            int32_t exact, void* stream);
int gpnerf_conv2d_nhwc_exact(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, const float* weight, const float* bias,
                             int32_t cout, int32_t ks, int32_t stride, float* y, void* stream);
int gpnerf_norm_apply_nhwc(const float* x, const float* table, const float* residual, const float* res_table, int32_t n, int64_t hw,
                           int32_t c, int32_t act, float* out, void* stream);
int64_t gpnerf_instance_norm_nhwc_scratch_bytes(int32_t n, int64_t hw, int32_t c);
int gpnerf_instance_norm_act_nhwc(const float* x, const float* gamma, const float* beta,
                                  const float* residual, int32_t n, int64_t hw, int32_t c, float eps, int32_t act, float* out,
                                  void* scratch, void* stream);
int gpnerf_upsample2x_nhwc(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* out, void* stream);

/* Channels-last re-layouts of the per-frame tensors (device -> device). */
int gpnerf_relayout_volume(const float* ncdhw, float* ndhwc, int32_t D, int32_t H, int32_t W, void* stream);
int gpnerf_relayout_featmaps(const float* nchw, float* nhwc, int32_t V, int32_t H, int32_t W, void* stream);
/* src_imgs [V][3][H][W] in [-1,1] -> [V][H][W][4] = x*0.5+0.5 (BaseRender.py:231), 4th lane 0 */
int gpnerf_relayout_images(const float* nchw, float* nhwc4, int32_t V, int32_t H, int32_t W, void* stream);
/* Launch order of a frame's rays: order[q] = row (in the caller's ray list, which follows the raster order of the kept pixels of
 * `mask`, as ZjumocapDataset.py:505's mask_at_box does) of the q-th ray when the H x W image is walked in patches of
 * patch_w x patch_h pixels (patch_w <= 64, patch_h <= 32), patches and the pixels inside a patch in raster order -- the 32 rays of
 * a wavefront then cover a compact block (gpnerf_render_fused's ray_order).  mask: H*W bytes, non-zero = kept; n: the caller's ray
 * count; scratch: gpnerf_patch_order_scratch_bytes() bytes, overwritten.  Three launches, no host round trip; a mask that does not
 * keep exactly n pixels yields the identity 0 .. n-1.  Results of a render do not depend on the order; it is a locality choice. */
int64_t gpnerf_patch_order_scratch_bytes(int32_t H, int32_t W, int32_t patch_w, int32_t patch_h);
int gpnerf_patch_order(const uint8_t* mask, int32_t H, int32_t W, int32_t patch_w, int32_t patch_h, int32_t n, int32_t* scratch,
                       int32_t* order, void* stream);

/* Layout of the head image, for tools and tests: table[4*l + {0,1,2,3}] = k-steps, 32-row output tiles,
 * weight offset, bias offset (in floats) of MFMA layer l = GEO,D1,D2,D3,BS,BV,B2,V1,V2,R1,R2 (11 layers),
 * then table[44..47] = offsets of the 16->1 / 16->3 VALU tails (D4 weights, D4 bias, R3 weights, R3 bias).
 * table: host, 48 int32. */
int gpnerf_head_layout(int32_t* table);

const char* gpnerf_strerror(int code);
/* compile-time facts for callers / tests */
int32_t gpnerf_rays_per_tile(void);   /* rays one wavefront renders together (32) */
const char* gpnerf_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* GPNERF_HIP_H */
