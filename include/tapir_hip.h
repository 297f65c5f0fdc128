/* libtapir_hip -- C ABI of the MI355X (gfx950) TAPIR inference hot path.
 *
 * The reference (google-deepmind/tapnet) has no FFI layer: its boundary for
 * this path is the Python API of tapnet/models/tapir_model.py.  Every entry
 * point below replaces one reference function (cited per declaration); the
 * Python mirror of that API lives in tapnet_amd/tapir_model.py and only
 * marshals pointers through ctypes.
 *
 * Conventions
 *  - return 0 (TAPIR_OK) or a negative error code; never throws; the message
 *    for the last error of a context is available from tapir_last_error().
 *  - every tensor argument is a DEVICE pointer owned by the caller (torch),
 *    dense row-major, float32 unless stated; the library owns only the
 *    context (weights + workspaces).  tapir_set_weight takes HOST memory.
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream).  No host synchronisation inside.
 *    Workspaces grow on demand (hipMalloc): call tapir_reserve() first if the
 *    call is going to be captured into a hipGraph.
 *  - one context per device; calls on one context are not thread-safe.
 *  - layouts follow the reference: feature grids [B,T,h,w,C] channels-last,
 *    query points (t,y,x), tracks (x,y), tokens ordered (b, q, t).
 */
#ifndef TAPIR_HIP_H_
#define TAPIR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAPIR_OK 0
#define TAPIR_ERR_INVALID (-1)      /* bad argument / shape                           */
#define TAPIR_ERR_HIP (-2)          /* a HIP runtime call failed                      */
#define TAPIR_ERR_UNSUPPORTED (-3)  /* shape outside what the kernels are built for   */
#define TAPIR_ERR_WEIGHTS (-4)      /* weights missing / wrong shape / not finalized  */

#define TAPIR_F32 0   /* exact-f32 MFMA (v_mfma_f32_16x16x4_f32): parity build       */
#define TAPIR_BF16 1  /* bf16 operands, f32 accumulate (v_mfma_f32_16x16x32_bf16)    */

#define TAPIR_MAX_LEVELS 8   /* 1 + number of refinement resolutions */

typedef struct tapir_ctx tapir_ctx;

/* Mirrors the constructor kwargs of TAPIR.__init__ that matter after the
 * backbone (tapnet/models/tapir_model.py:299-317). */
typedef struct tapir_cfg {
  int pyramid_level;          /* 0 (TAPIR ckpt) or 1 (BootsTAPIR / causal)    */
  int num_pips_iter;          /* default 4                                     */
  int num_mixer_blocks;       /* default 12                                    */
  int use_causal_conv;        /* online model                                  */
  float softmax_temperature;  /* 20.0 default, 10.0 BootsTAPIR                 */
  int initial_h, initial_w;   /* initial_resolution, default 256 x 256         */
  int dtype;                  /* TAPIR_F32 or TAPIR_BF16                        */
} tapir_cfg;

int tapir_create(tapir_ctx** out, const tapir_cfg* cfg, int device);
void tapir_destroy(tapir_ctx* ctx);
const char* tapir_last_error(const tapir_ctx* ctx);
const char* tapir_version(void);

/* Weights.  `name` is the reference's torch state_dict key
 * (tapnet/torch/tapir_model.py:115-137, e.g.
 * "torch_pips_mixer.blocks.3.conv_channels_mixer.mlp2_up.weight"); `data` is
 * float32 HOST memory in the state_dict's own layout.  Backbone keys
 * (resnet_torch.*, extra_convs.*) are accepted and ignored: the backbone runs
 * in PyTorch-ROCm.  tapir_finalize_weights uploads / re-lays-out everything
 * and fails with TAPIR_ERR_WEIGHTS if a tensor the hot path needs is missing. */
int tapir_set_weight(tapir_ctx* ctx, const char* name, const float* data,
                     const int64_t* shape, int ndim);
int tapir_finalize_weights(tapir_ctx* ctx);

/* Pre-sizes every workspace for B clips x Q queries x T frames (largest grid
 * h x w of the low-resolution pyramid), so later calls allocate nothing. */
int tapir_reserve(tapir_ctx* ctx, int B, int Q, int T, int max_lowres_h, int max_lowres_w);

/* Counted pin: on != 0 adds a pin, on == 0 removes one.  While pinned, a call that would have to
 * GROW a workspace fails with TAPIR_ERR_INVALID instead of reallocating it.  Set after tapir_reserve() and before capturing calls into a
 * hipGraph: the graph holds the workspace pointers, so a later, larger call on the same context
 * must not free them (it gets an error; reserve the largest shape first or use a second context).
 * Calls on one context also share these workspaces: one stream at a time per context. */
int tapir_pin_workspaces(tapir_ctx* ctx, int on);

/* einsum('bnc,bthwc->tbnhw') of TAPIR.tracks_from_cost_volume
 * (tapir_model.py:433) -- the north_star's "build_cost_volume".
 * qfeat [B,Q,C], grid [B,T,h,w,C] -> volume [B,Q,T,h,w] (the reference's
 * 'tbnhw' is a permuted view of this). */
int tapir_build_cost_volume(tapir_ctx* ctx, const float* qfeat, const float* grid,
                            int B, int Q, int T, int h, int w, int C,
                            float* volume, void* stream);

/* TAPIR.tracks_from_cost_volume (tapir_model.py:399-471) incl.
 * model_utils.heatmaps_to_points / soft_argmax_heatmap (model_utils.py:209-314).
 * query_points: NULL or [B,Q,3] (t,y,x) in initial_resolution coordinates.
 * points [B,Q,T,2] (x,y) in initial_resolution pixels; occlusion / expected_dist
 * [B,Q,T] logits. */
int tapir_tracks_from_cost_volume(tapir_ctx* ctx, const float* qfeat, const float* grid,
                                  const float* query_points, int B, int Q, int T,
                                  int h, int w,
                                  float* points, float* occlusion, float* expected_dist,
                                  void* stream);

/* TAPNet.tracks_from_cost_volume (tapnet/models/tapnet_model.py:111-171), num_heads 1, 2 or 4 (= the input
 * channels of hid1.weight [16,num_heads,3,3]; channel c of the features belongs to head c % num_heads): the TAPIR
 * head kernel with the TAP-Net differences (no ReLU after the stride-2 convolution, ONE occlusion
 * logit, softmax temperature 10).  Needs the head's weights under the names
 * "tapnet_cost_volume_track_mods.{hid1,hid2,hid3,hid4,occ_out}.{weight,bias}" (torch layout of the
 * TAPIR head; occ_out [1,16]) before tapir_finalize_weights.  qfeat [B,Q,256], grid [B,T,h,w,256]
 * (h, w <= 32: one fused kernel), query_points NULL or [B,Q,3] (t,y,x) in initial_resolution
 * coordinates -> points [B,Q,T,2] (x,y), occlusion [B,Q,T] logits. */
int tapir_tapnet_tracks_from_cost_volume(tapir_ctx* ctx, const float* qfeat, const float* grid,
                                         const float* query_points, int B, int Q, int T, int h, int w,
                                         float* points, float* occlusion, void* stream);

/* TAP-Net's forward-backward cycle-consistency tracker -- tapnet/training/supervised_point_prediction.py:443-546, the
 * evaluation path of prediction_algo != 'cost_volume_regressor' (SURVEY.md 8 f4): no learned head.  Forward: soft arg max
 * of softmax(temperature * einsum('bnc,bthwc->bnthw')) per frame, the query's own frame overridden by the query point
 * (:453-469); the grid is sampled bilinearly at the tracked point of every frame (:473-496); backward: each sampled
 * vector against the grid of the frame its query came from (:501-531); a point whose backward track lands more than
 * dist_threshold (48) pixels from the query is occluded: logit +10, else -10 (:533-539; the intended indexing of :537).
 * Both passes are the row-streamed cost-volume kernel without heads -- no [B,N,T,h,w] tensor exists.
 * query_feats [B,Q,256], feature_grid [B,T,h,w,256] (rows of up to 32 cells on up to 32 rows -- 64 rows for w <= 16 --
 * or rows of 33..64 cells on up to 64 rows; TAPIR_ERR_UNSUPPORTED otherwise), query_points [B,Q,3] (t,y,x) in
 * img_h x img_w pixels; tracks [B,Q,T,2] (x,y), occlusion [B,Q,T], inverse_tracks [B,Q,T,2] or NULL.  No weights needed.
 * Queries are processed in chunks (at most 256 MiB of sampled vectors at a time, the reference's eval_chunk_size loop).
 * Its workspaces are not part of tapir_reserve(): call it once at the largest shape before tapir_pin_workspaces(). */
int tapir_cycle_consistency_tracks(tapir_ctx* ctx, const float* query_feats, const float* feature_grid,
                                   const float* query_points, int B, int Q, int T, int h, int w, int img_h, int img_w,
                                   float softmax_temperature, float dist_threshold, float* tracks, float* occlusion,
                                   float* inverse_tracks, void* stream);
/* One grid of TAPIR.get_query_features (tapir_model.py:781-849;
 * model_utils.interp mode='nearest' :177-206): trilinear sample of
 * grid [B,T,h,w,C] at query_points [B,Q,3] (t,y,x) given in video pixels
 * (video_h x video_w) -> out [B,Q,C]. */
int tapir_get_query_features(tapir_ctx* ctx, const float* grid, const float* query_points,
                             int B, int Q, int T, int h, int w, int C,
                             int video_h, int video_w, float* out, void* stream);

/* PIPSMLPMixer (tapir_model.py:127-156): x [N,T,Cin] -> out [N,T,388], with
 * Cin = 388 + 49*(2+pyramid_level).  Causal state (use_causal_conv only), all
 * optional: ctx1_* [num_blocks,N,2,512], ctx2_* [num_blocks,N,2,2048]
 * (= the reference's block_i_causal_1 / _2 entries, :48-73); *_in NULL = zeros,
 * *_out NULL = not requested.  in and out must not alias. */
int tapir_pips_mixer(tapir_ctx* ctx, const float* x, int N, int T, float* out,
                     const float* ctx1_in, const float* ctx2_in,
                     float* ctx1_out, float* ctx2_out, void* stream);

typedef struct tapir_pyramid {
  int n_levels;                    /* 2 + pyramid_level                              */
  const float* query[3];           /* [B,Q,C_l]: hires(128), lowres(256), lowres(256) */
  const float* grid[3];            /* [B,T,h_l,w_l,C_l]                               */
  int h[3], w[3], C[3];
} tapir_pyramid;

/* TAPIR.refine_pips (tapir_model.py:473-624), one refinement iteration.
 * pos [B,Q,T,2] (x,y) in orig (= initial_resolution) pixels, occ / expd [B,Q,T];
 * last_iter NULL (first iteration of a level) or [B,Q,T,384].
 * Outputs: pos_out, occ_out, expd_out, feats_out [B,Q,T,384] (may alias the
 * inputs).  Causal state as in tapir_pips_mixer with N = B*Q. */
int tapir_refine_pips(tapir_ctx* ctx, const tapir_pyramid* pyr, int B, int Q, int T,
                      const float* pos, const float* occ, const float* expd,
                      const float* last_iter, int orig_h, int orig_w,
                      int resized_h, int resized_w,
                      float* pos_out, float* occ_out, float* expd_out, float* feats_out,
                      const float* ctx1_in, const float* ctx2_in,
                      float* ctx1_out, float* ctx2_out, void* stream);

typedef struct tapir_traj_args {
  int B, Q, T;
  int n_levels;                               /* len(feature_grids.lowres)                 */
  const float* lowres[TAPIR_MAX_LEVELS];      /* [B,T,h_i,w_i,256]  (L2-normalised)        */
  const float* hires[TAPIR_MAX_LEVELS];       /* [B,T,2h_i,2w_i,128]                       */
  int lowres_h[TAPIR_MAX_LEVELS], lowres_w[TAPIR_MAX_LEVELS];
  int hires_h[TAPIR_MAX_LEVELS], hires_w[TAPIR_MAX_LEVELS];
  int res_h[TAPIR_MAX_LEVELS], res_w[TAPIR_MAX_LEVELS];   /* feature_grids.resolutions      */
  const float* q_lowres[TAPIR_MAX_LEVELS];    /* [B,Q,256]                                 */
  const float* q_hires[TAPIR_MAX_LEVELS];     /* [B,Q,128]                                 */
  const float* query_points;                  /* NULL or [B,Q,3] (t,y,x) in video pixels   */
  int video_h, video_w;
  /* causal state, all optional: [num_iters, num_blocks, B*Q, 2, 512 | 2048] */
  const float* ctx1_in; const float* ctx2_in;
  float* ctx1_out; float* ctx2_out;
  /* outputs: one slice per iteration (0 = cost-volume initialisation), video pixels */
  float* tracks;          /* [num_iters+1, B, Q, T, 2] */
  float* occlusion;       /* [num_iters+1, B, Q, T]    */
  float* expected_dist;   /* [num_iters+1, B, Q, T]    */
} tapir_traj_args;

/* TAPIR.estimate_trajectories (tapir_model.py:858-1066): cost-volume
 * initialisation on level 0, then num_pips_iter refinement iterations on each
 * further level, all enqueued on `stream`.  The query permutation / chunking of
 * the reference (:938-952) does not change per-query results and is not
 * reproduced. */
int tapir_estimate_trajectories(tapir_ctx* ctx, const tapir_traj_args* args, void* stream);

/* Measurement support (bench.py).  When enabled, every launch of the kernel classes below is
 * timed with two hipEvents on the caller's stream -- filled in by the dispatch itself with the
 * kernel's start / stop timestamps (hipExtLaunchKernelGGL; the value rocprofv3 reports as the
 * kernel's duration), or recorded as markers around the launches when a class takes more than
 * one launch (split-K GEMMs of the online model); tapir_profile_read() synchronises on them and
 * returns the number of launches and their summed duration since the previous read of that class. */
#define TAPIR_PROF_GEMM_UP 0    /* mixer mlp2_up GEMM   [R,512]x[512,2048] + bias + GELU  */
#define TAPIR_PROF_GEMM_DOWN 1  /* mixer mlp2_down GEMM [R,2048]x[2048,512] + bias + skip */
#define TAPIR_PROF_MIX 2        /* LN + temporal depthwise convs + LN (mix_kernel)         */
#define TAPIR_PROF_PATCH 3      /* pyramid patch correlation (patch_corr_kernel)           */
#define TAPIR_PROF_CV_HEADS 4   /* cost-volume heads (cv_heads_kernel)                     */
#define TAPIR_PROF_CV_GEMM 5    /* cost-volume einsum GEMM                                 */
#define TAPIR_PROF_MIXER 6      /* track-resident fused PIPs mixer (mixer_fused_kernel): one launch
                                   per refinement iteration = input Linear + all blocks + output Linear */
/* backbone kernel classes (conv_fused.hpp, backbone.hpp).  Events cannot be recorded inside a captured hipGraph: these
 * are read from EAGER launches (tapnet_amd.backbone with TAPIR_BACKBONE_GRAPH=0; bench.py's untimed profiling pass). */
#define TAPIR_PROF_STEM 7        /* 7x7 / stride-2 stem (stem_conv_kernel)                                  */
#define TAPIR_PROF_CONV3_C64 8   /* 3x3 stride-1 block convolutions, 64 -> 64 channels (conv_fused_kernel)  */
#define TAPIR_PROF_CONV3_C128 9  /* ... 128 -> 128                                                         */
#define TAPIR_PROF_CONV3_C256 10 /* ... 256 -> 256                                                         */
#define TAPIR_PROF_CONV_OTHER 11 /* the 3x3 stride-2 convolutions and the 1x1 projections                   */
#define TAPIR_PROF_L2NORM 12     /* L2 normalisation + operand-type copies (l2norm_kernel)                  */
#define TAPIR_PROF_KINDS 13
/* on: bit mask of kernel classes (1 << TAPIR_PROF_*) to bracket with events; -1 = all, 0 = off. */
int tapir_profile_enable(tapir_ctx* ctx, int on);
int tapir_profile_read(tapir_ctx* ctx, int kind, double* total_ms, int64_t* launches);
/* Times only every stride-th launch of an enabled class (default 1 = every launch; resets the launch counters).  A timed launch is
 * dispatched with start / stop signals and costs ~12 us of idle device on either side of it: bench.py samples the dominant class
 * with a stride co-prime to the four refinement launches of a clip inside its timed region (averages per launch, not totals). */
int tapir_profile_stride(tapir_ctx* ctx, int stride);

/* Feature backbone, the memory-bound half (TAPIR.get_feature_grids, tapir_model.py:626-729; ResNet
 * blocks, tapnet/models/resnet.py:152-257).  Every convolution of the ResNet (stem, 3x3, strided, 1x1)
 * has a HIP kernel in both element types (tapir_conv_fused / tapir_stem_conv below); these three entry
 * points are what is left between them -- the final L2 normalisation, and the statistics / normalise
 * passes around a convolution whose shape does not fit the HIP kernels and goes to MIOpen.  Tensors are NHWC in the
 * context's element type (f32, or bf16 bits for TAPIR_BF16); channel counts: C / (8 bf16 | 4 f32)
 * must be a power of two <= 256 (<= 64 for tapir_l2_normalize).
 *
 * tapir_inorm_stats: per-(image, channel) statistics of hk.InstanceNorm (resnet.py:177-181) over
 *   x = a [N,HW,C], or over x = a + b with the sum written to sum_out (the residual add that ends
 *   a block, resnet.py:256, fused with the statistics of the next block's first norm; sum_out may
 *   alias a or b).  part [N, slabs, C, 2] f32 receives one (mean, M2) summary per slab of pixels.
 * tapir_inorm_relu: y = relu((x - mean) / sqrt(var + 1e-5) * gamma + beta) (resnet.py:241-249) from
 *   those summaries (slabs of per_s pixels; per_s = 0: ceil(HW / slabs), what tapir_inorm_stats
 *   writes; the part_out of tapir_conv_fused has per_s = rows * W); ss: N * C * 2 floats of scratch
 *   owned by the caller (the merged scale / shift pairs; the context keeps no buffer of its own, so
 *   calls on different streams do not share one).  y is [N, out_h, out_w, C] with out_h >= H, out_w >= W: rows / columns past
 *   H / W are not written (pass a zero-initialised buffer with out_h = H+1, out_w = W+1 to get the
 *   XLA "SAME" padding of a stride-2 3x3 convolution, which pads on the high side only).
 *   y_sub, if not NULL, [N, H/2, W/2, C] receives the pixels with even h and w (input of the
 *   stride-2 1x1 projection, resnet.py:243).
 * tapir_l2_normalize: out f32 [pixels, C] = x / sqrt(max(sum_c x^2, 1e-12)) (tapir_model.py:709-720). */
int tapir_inorm_stats(tapir_ctx* ctx, const void* a, const void* b, void* sum_out, float* part,
                      int N, int HW, int C, int slabs, void* stream);
int tapir_inorm_relu(tapir_ctx* ctx, const void* x, const float* part, const float* gamma,
                     const float* beta, float* ss, void* y, void* y_sub, int N, int H, int W, int C,
                     int slabs, int per_s, int out_h, int out_w, void* stream);
int tapir_l2_normalize(tapir_ctx* ctx, const void* x, float* out, long pixels, int C, void* stream);
/* The same, and -- bf16 build -- the normalised values once more in the hot path's operand type: row-major
 * [pixels, C] bf16 (out_op) and, for the 256-channel low-res map, in the cost-volume kernel's tile order (out_tiled:
 * [frame][tile of 16 cells][32 chunks of 8 channels][16 cells][8], frames of cells_per_frame cells, zero-initialised
 * by the caller, ((cells + 15) / 16) * 16 * 256 elements per frame).  Either may be NULL.  These are the copies
 * tapir_estimate_trajectories / tapir_tracks_from_cost_volume otherwise make themselves by re-reading the f32 grids
 * (151 MB per 48-frame 256 x 256 clip); tapir_set_staged_grid tells the context that they exist. */
int tapir_l2_normalize_staged(tapir_ctx* ctx, const void* x, float* out, void* out_op, void* out_tiled, long pixels,
                              int C, int cells_per_frame, void* stream);
/* Registers operand-type copies of an f32 feature grid (written by tapir_l2_normalize_staged, same stream order) for
 * the calls that follow: where a call is handed grid_f32 it reads grid_op / grid_tiled instead of casting.  The caller
 * guarantees that the copies match the f32 grid; tapir_clear_staged_grids forgets all of them (the Python layer
 * registers the borrowed grids of ONE TAPIR.__call__ and clears them before it returns).  bf16 build only. */
int tapir_set_staged_grid(tapir_ctx* ctx, const float* grid_f32, const void* grid_op, const void* grid_tiled);
int tapir_clear_staged_grids(tapir_ctx* ctx);

/* The convolutions of the ResNet blocks (tapnet/models/resnet.py:185-257: self.conv_0 / self.conv_1 /
 * self.proj_conv of BlockV2 with the InstanceNorm + relu in front of them, :241-243 / :248-249, and the
 * residual add :256): one HIP implicit-GEMM kernel per convolution (bf16 MFMA in bf16 contexts, exact-f32
 * MFMA in f32 contexts; tensors in the context's element type) with the
 * normalisation of its INPUT folded into the operand load and the residual add + the statistics of its
 * OUTPUT (for the next norm) into the epilogue.  Supported: 3x3 stride 1 with C -> C channels,
 * C in {64,128,256}; 1x1 stride 1 with 64 -> 64 / 256 -> 256; 3x3 and 1x1 stride 2 with 64 -> 128 /
 * 128 -> 256; XLA "SAME" padding.  (The 7x7 stem has its own entry point below.)
 * tapir_conv_plan : output rows per workgroup tile and tiles per image for an [H, W, cin] INPUT map
 *   (TAPIR_ERR_UNSUPPORTED when the shape does not fit: keep that convolution on MIOpen).
 * tapir_conv_pack : w = the reference's [cout, cin, ks, ks] f32 kernel (torch OIHW, host memory)
 *   -> device-resident packed fragment streams.  A pack (and a tapir_stem_pack) belongs to the caller's
 *   backbone object, NOT to the hot-path weights: it survives tapir_set_weight / tapir_finalize_weights
 *   (a second load of hot-path weights does not invalidate a backbone or the hipGraphs that captured
 *   its pointers) and lives until tapir_conv_free(ctx, wstream) or tapir_destroy.
 * tapir_conv_fused: y [N, ceil(H/stride), ceil(W/stride), cout] =
 *   conv(relu(instance_norm(x; part_in, gamma, beta))) (+ shortcut, 3x3 stride 1 only), rounded to the
 *   context's element type.
 *   part_in [N, slabs_in, cin, 2] are (mean, M2) summaries of x per slab of per_s_in pixels
 *   (0: ceil(HW / slabs_in)) -- from tapir_inorm_stats or from a previous call's part_out;
 *   ss: N * cin * 2 floats of scratch owned by the caller (the merged scale / shift; part_in == NULL: ss already
 *   holds the pairs of this input and norm from a previous call -- conv_0 and proj_conv of a block -- and is
 *   used as it is); part_out, if not
 *   NULL, [N, tiles, cout, 2] receives the summaries of y per tile (rows * W_out pixels each).
 *
 * The _nn forms ("next norm") also merge the summaries of y into the (a, b) pairs of the InstanceNorm that READS y,
 * inside the same launch: every workgroup publishes its tile summary write-through and takes a ticket from the
 * image's arrival counter; the one that draws the last ticket merges the image's summaries exactly as the separate
 * merge kernel does (bit-identical pairs) and resets the counter -- no waiting anywhere.  next->ss [N, cout, 2]
 * (pass it as `ss` with part_in = NULL to the call that consumes y), next->arrive [N] int32, ZERO before the first
 * launch (every launch leaves it zero).  next == NULL: the plain forms. */
typedef struct tapir_next_norm {
  const float* gamma;   /* [cout] scale of the norm that reads y */
  const float* beta;    /* [cout] offset */
  float* ss;            /* [N, cout, 2] out */
  int* arrive;          /* [N] arrival counters */
} tapir_next_norm;
int tapir_conv_plan(tapir_ctx* ctx, int H, int W, int cin, int cout, int ks, int stride, int* rows, int* tiles);
/* Few-frame clips (the online model's single frame: tapnet/live_demo.py:51-77, tapir_model.py:1156-1203; bf16 contexts):
 * on = 1 makes tapir_conv_fused* take the K-split form of csrc/conv_small.hpp -- a workgroup per (tile of whole rows <= 128
 * pixels, group of 64 output channels), its four waves splitting the taps -- and tapir_conv_plan answer with THAT form's tile
 * geometry, for every shape the form covers (output rows of <= 128 pixels); 0 (default) = the many-frame kernels.  The forms
 * differ in summation order (not bit-identical): set it from the frame count of the WHOLE clip (tapnet_amd.backbone: clips of
 * fewer than 4 frames), never per shard or chunk, and plan / allocate the summaries after setting it.  The dual launch
 * (tapir_conv_fused_dual_nn) has no few-frame form and returns TAPIR_ERR_UNSUPPORTED while the mode is on.
 * While it is on, also: (1) tapir_xconv / tapir_xconv_nt run both convolutions of an ExtraConvs block (256 -> 1024 + bias +
 * gelu, 1024 -> 256 + bias + skip) in the same form where the pack was built for 256-channel chunks; (2) a tapir_conv_fused*
 * call that is handed summaries (part_in != NULL, slabs_in <= 32 * 256 / cin) merges them into the input norm's (a, b) pairs
 * INSIDE the consuming launch, in every workgroup's prologue -- no inorm_finalize launch, `ss` is neither written nor read --
 * which is cheaper on a one-frame launch than the producer-side merge of the _nn forms (still accepted): pass next = NULL to
 * the producer and its part_out to the consumer (tapnet_amd.backbone does). */
int tapir_conv_set_small(tapir_ctx* ctx, int on);
int tapir_conv_pack(tapir_ctx* ctx, const float* w, int cout, int cin, int ks, void** wstream);
int tapir_conv_free(tapir_ctx* ctx, void* wstream);
int tapir_conv_fused(tapir_ctx* ctx, const void* x, const float* part_in, int slabs_in, int per_s_in,
                     const float* gamma, const float* beta, float* ss, const void* wstream,
                     const void* shortcut, void* y, float* part_out, int N, int H, int W, int cin,
                     int cout, int ks, int stride, void* stream);
int tapir_conv_fused_nn(tapir_ctx* ctx, const void* x, const float* part_in, int slabs_in, int per_s_in,
                        const float* gamma, const float* beta, float* ss, const void* wstream,
                        const void* shortcut, void* y, float* part_out, int N, int H, int W, int cin,
                        int cout, int ks, int stride, const tapir_next_norm* next, void* stream);
/* conv_0 (3x3) AND proj_conv (1x1, same stride) of the first block of a group -- resnet.py:232-247: both read
 * relu(norm(x)) -- in ONE launch: the projection is the centre tap of the tile the 3x3 convolution stages anyway, so
 * the second staging of the input (and a launch of the dependency chain) disappears.  w3 [cout,cin,3,3], w1 [cout,cin,1,1]
 * (torch layouts) -> one stream; supported: 64->64 and 256->256 at stride 1, 64->128 and 128->256 at stride 2
 * (TAPIR_ERR_UNSUPPORTED otherwise).  y = the 3x3 result with its statistics (part_out / next as above), y_proj = the
 * projection (the block's shortcut; no statistics).  Both outputs are bit-identical to the two separate launches. */
int tapir_conv_pack_dual(tapir_ctx* ctx, const float* w3, const float* w1, int cout, int cin, int stride, void** wstream);
int tapir_conv_fused_dual_nn(tapir_ctx* ctx, const void* x, const float* part_in, int slabs_in, int per_s_in,
                             const float* gamma, const float* beta, float* ss, const void* wstream_dual, void* y,
                             void* y_proj, float* part_out, int N, int H, int W, int cin, int cout, int stride,
                             const tapir_next_norm* next, void* stream);

/* The stem of the ResNet (resnet.py:356-364: initial_conv, 7x7 / stride 2 / SAME, 3 -> 64 channels):
 * x = the f32 frames [N,H,W,3] as the model receives them (bf16 contexts: rounded to bf16 on load, as the
 * library path's cast does), y [N, ceil(H/2), ceil(W/2), 64] in the context's element type, part_out [N, tiles, 64, 2] the
 * (mean, M2) summaries of y per tile of rows * W_out pixels (input of the first InstanceNorm).
 * tapir_stem_pack takes the reference's [64, 3, 7, 7] f32 kernel (host memory). */
int tapir_stem_plan(tapir_ctx* ctx, int H, int W, int* rows, int* tiles);
int tapir_stem_pack(tapir_ctx* ctx, const float* w, void** wstream);
int tapir_stem_conv(tapir_ctx* ctx, const float* x, const void* wstream, void* y, float* part_out, int N,
                    int H, int W, void* stream);
int tapir_stem_conv_nn(tapir_ctx* ctx, const float* x, const void* wstream, void* y, float* part_out, int N,
                       int H, int W, const tapir_next_norm* next, void* stream);

/* BootsTAPIR's ExtraConvs (tapnet/models/tapir_model.py:159-186, class ExtraConvs; torch twin
 * tapnet/torch/nets.py:25-89): five blocks  y = LayerNorm(x) * scale + offset;  r = gelu(conv3x3(y) + b);
 * x = y + conv3x3(r) + b'  on the low-resolution map.  NHWC tensors in the context's element type.
 * tapir_layernorm_affine: hk.LayerNorm(axis=-1, create_scale=True, create_offset=True) per pixel (:176),
 *   eps 1e-5; C / (8 bf16 | 4 f32) a power of two <= 64.
 * tapir_xconv_plan : output rows per workgroup tile, tiles per image and input channels per LDS chunk for
 *   an [H, W, cin] map (W <= 94: a one-row tile of the wide form has to fit LDS; cin, cout multiples of 256),
 *   TAPIR_ERR_UNSUPPORTED otherwise.  Two forms of the kernel: 64 or 128 pixels per workgroup (the latter wherever
 *   it gives a tile more rows; TAPIR_XCONV_NT=4|8 in the environment of tapir_create forces one, for A/B runs).
 * tapir_xconv_pack : w = the reference's [cout, cin, 3, 3] f32 kernel (torch OIHW, host memory) -> packed
 *   fragment streams for chunks of `cch` input channels (the value tapir_xconv_plan returned for the map it will
 *   be used on: tapir_xconv returns TAPIR_ERR_INVALID for a pack built with another chunk width); owned like
 *   a tapir_conv_pack (tapir_conv_free).
 * tapir_xconv      : y [N, H, W, cout] = conv3x3_SAME(x [N, H, W, cin]) + bias, then gelu (tanh form,
 *   jax.nn.gelu :184) if `gelu`, or + skip [N, H, W, cout] if skip != NULL (:185), rounded to the element
 *   type.  (hk.Conv2D(1024, 3) / hk.Conv2D(256, 3): stride 1, SAME padding, with bias.) */
int tapir_layernorm_affine(tapir_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y,
                           long pixels, int C, void* stream);
int tapir_xconv_plan(tapir_ctx* ctx, int H, int W, int cin, int cout, int* rows, int* tiles, int* cch);
/* The same with the frame count of the WHOLE clip (never of one launch or shard: the forms add the input channels in
 * different orders): short clips take the 64-pixel form even where the 128-pixel one fits (all their workgroups are
 * resident at once and the smaller ones finish first; frames x tiles x (cout / 256) < 512).  *form = 4 | 8, to be handed to
 * tapir_xconv_nt together with a pack built for *cch. */
int tapir_xconv_plan_frames(tapir_ctx* ctx, int frames, int H, int W, int cin, int cout, int* rows, int* tiles, int* cch,
                            int* form);
int tapir_xconv_pack(tapir_ctx* ctx, const float* w, int cout, int cin, int cch, void** wstream);
int tapir_xconv(tapir_ctx* ctx, const void* x, const void* wstream, const float* bias, const void* skip,
                void* y, int N, int H, int W, int cin, int cout, int gelu, void* stream);
/* tapir_xconv with the kernel form named (form = 4 | 8; 0 = as tapir_xconv chooses: the many-frames choice of
 * tapir_xconv_plan).  CONTRACT: `form` must be the value tapir_xconv_plan_frames returned for the WHOLE clip's frame count
 * (batch x frames of the call, or the global frame count of a sharded clip), the same for every launch, chunk and shard
 * of that clip, with a pack built for the *cch of that same plan call.  The call only checks that the pack's chunk width
 * fits the form; the two forms add the input channels in different orders, so mixing them still computes the
 * convolution but the results of chunks / shards are then no longer bit-identical to the whole clip's.  Consequently
 * BootsTAPIR features are bit-stable for a given (batch x frames) only: one 16-frame clip and two batched 16-frame
 * clips may take different forms (tapnet_amd.backbone passes max(B x T, global frames)); a C caller that follows
 * plan -> pack -> tapir_xconv (form 0) gets the many-frames form whatever its clip length, i.e. other last bits than the
 * Python backbone on short clips -- use tapir_xconv_plan_frames + tapir_xconv_nt to match it. */
int tapir_xconv_nt(tapir_ctx* ctx, const void* x, const void* wstream, const float* bias, const void* skip,
                   void* y, int N, int H, int W, int cin, int cout, int gelu, int form, void* stream);

/* Kernel-level hooks for the micro-benchmarks (tools/kbench.py) and the tile-shape tests; no
 * reference counterpart.  One launch of the engine's MFMA GEMM  C = epi(A . W^T + bias):
 * A [M,lda], W [N,ldw] in the context's operand type (f32 or bf16 bits), bias [N] f32 or NULL;
 * epi 0: C f32 = acc + bias;  1: C operand type = gelu_tanh(acc + bias);
 * 2: C f32 = acc + bias + resid [M,ldr] f32.  tile: 0 = automatic, 1 = 192x128 (4 LDS stages), 2 = 128x128,
 * 3 = 192x64, 4 = 192x128 (3 stages), 5 = 192x128 wave-specialised (4 DMA + 8 MFMA waves),
 * 6 = 192x256, 7 = 256x128 (3 stages), 8 = 128x128 with 8 waves, 9 = 192x64 with 8 waves,
 * 10 = 128x64 with 8 waves (three workgroups per CU), 11 = 256x128 with 16 waves, 3 stages and
 * fragment prefetch, 12 = 128x128 8 waves 3 stages + prefetch, 13 = 256x128 16 waves 2 stages,
 * 14 = 256x128 16 waves 3 stages, 15 = 128x128 8 waves 3 stages, 16 = 256x256 8 waves 2 stages; bits 8..19 of `tile`, when non-zero,
 * cap the persistent grid (tests); bit 20 (bf16 build, epi 1 | 2, trace buffer set, tiles 1 / 3 / 6 / 8):
 * the TRACE build of the kernel, whose waves write their per-k-step cycle totals
 * ([workgroup*16 + wave][8] int64: copy issue, fragment reads + MFMAs, wait for own copies, barrier,
 * epilogue, lifetime) into the trace buffer.  K must be a multiple of 64 (bf16) / 32 (f32); N, ldc
 * multiples of 4. */
int tapir_debug_gemm(tapir_ctx* ctx, const void* A, long lda, const void* W, long ldw,
                     const float* bias, const float* resid, long ldr, void* C, long ldc,
                     int M, int N, int K, int epi, int tile, void* stream);
/* One launch of the token-mixing kernel of mixer block `block` (LN, temporal depthwise
 * convs, GELU, group sum, skip, LN): x_in [N,T,512] f32 -> x_out [N,T,512] f32 and
 * xn [N*T,512] in the operand type.  Non-causal contexts only.  tc: 0 = production form (one
 * (track, 12-frame chunk) unit per workgroup), > 0 = the persistent double-buffered form on at most
 * tc workgroups (tests: several units per workgroup). */
/* Phase tracing for tools/kbench.py: while a device buffer is set, tapir_debug_mix writes
 * wall-clock stamps (100 MHz) per work unit (int64 [units][6]), tapir_debug_gemm per workgroup
 * (int64 [workgroups][16], or the bit-20 layout above) and the bf16 cost-volume heads kernel its
 * phase totals per workgroup (int64 [512][8]); NULL turns it off. */
int tapir_debug_set_trace(tapir_ctx* ctx, void* device_buffer);
/* Which implementation tapir_pips_mixer / tapir_refine_pips / tapir_estimate_trajectories use for the
 * PIPs mixer: 0 = automatic (the track-resident fused kernel for non-causal clips of <= 48 frames
 * with >= 128 tracks, separate launches otherwise), 1 = always separate
 * launches (token-mixing kernel + tiled GEMMs), 2 = always the fused kernel, 3 = always its wide form
 * (bf16: two tracks or one 49..96-frame track per workgroup) (TAPIR_ERR_UNSUPPORTED where they do
 * not apply; automatic: wide above 256 tracks and for clips of 49..96 frames).  Both are HIP paths; tests and tools/kbench.py A/B them.
 * -DTAPIR_EXPERIMENTS builds add 4 (timing-only pair simulation) and 5 (the half-CU kernel, mixer_fused_half.hpp). */
int tapir_debug_set_mixer_mode(tapir_ctx* ctx, int mode);
/* Cost-volume stage (tapir_tracks_from_cost_volume and the first stage of
 * tapir_estimate_trajectories): 0 = automatic (ONE kernel -- einsum on the matrix cores into LDS +
 * heads, no volume in HBM -- for grids of up to 64 x 64 cells; other shapes take the path below, up to 1600 cells),
 * 1 = einsum GEMM into a workspace followed by the heads kernel (round-1 path; tools A/B it),
 * 2 = the fused kernel in its pixel-tiled form (costvol_fused.hpp: one map at a time over the whole
 * workgroup) also where automatic picks the row-streamed form (costvol_rows.hpp: every wave owns whole
 * maps; rows of up to 32 cells). */
int tapir_debug_set_cv_mode(tapir_ctx* ctx, int mode);
/* tools/kbench.py --what contraction: the contraction phase of the row-streamed cost-volume kernel alone (cost maps into
 * LDS, no heads); scratch = B*T*ceil(Q/8) floats. */
int tapir_debug_contraction(tapir_ctx* ctx, const float* qfeat, const float* grid, int B, int Q, int T, int h, int w,
                            float* scratch, void* stream);
/* Few-row GEMMs of the mixer (the online model, M = points x 1 frame <= 512 rows): 3 (default) = 2, and the online model's
 * whole mixer (one frame, use_causal_conv, <= 256 points: every block and the final LayerNorm, tapir_model.py:33-156) as ONE
 * persistent launch of 256 workgroups in 8 clusters that meet on bounded counters (csrc/mixer_online.hpp; devices with >= 256
 * CUs, else 2); 2 = 1, and the channel MLP of a block (up-projection + gelu + down-projection, tapir_model.py:127-156) as ONE
 * launch whose partial outputs the next consumer of the residual stream adds (csrc/gemm.hpp mlp_small_kernel); 1 = one launch
 * of the whole-K small-tile kernel per GEMM; 0 = the split-K kernel + element-wise reduce pair of round 2 (A/B measurements).
 * Modes 2 and 3 give the same bits; 1 differs from them by the summation order of the down-projection (same tolerance to the
 * oracle).  Tests add 4 x form to mode 3 (bit 0: the workgroups that read the same weight slice on one XCD instead of a cluster per
 * XCD; bit 1: acquire + plain loads instead of sc1 loads; bit 2: one member never arrives -- the timeout path): same bits. */
int tapir_debug_set_gemm_mode(tapir_ctx* ctx, int mode);
/* The persistent launch above bounds every wait: a workgroup that waited ~2 s for its cluster (the device could not hold the 256
 * workgroups at once) writes an error word, every member leaves, and the mixer's rows are NaN -- loud in the tracks.  Reads
 * the word (0 = no error; synchronises the device) and, if it was set, clears it together with the launch's counters: until then
 * every further persistent launch of this context leaves at once with NaN rows.  Never set on an otherwise idle MI355X. */
int tapir_online_sync_error(tapir_ctx* ctx, unsigned* word);
/* The 3x3 256 -> 256 block convolutions (tapir_conv_fused*): 1 = always the flat tiling of the whole launch
 * (csrc/conv_flat.hpp: three consecutive 64-pixel slabs of the (image, row) space per workgroup) where the shape allows it,
 * 0 (default) = never, -1 = from 96 slabs per launch on.  The forms are bit-identical (tests); the flat form is faster as one
 * 48-frame launch and slower inside the 4-stream backbone (profiles/r06_ab_flat_v1.txt), hence off. */
int tapir_debug_set_conv_flat(tapir_ctx* ctx, int mode);
/* TAPIR_OK and the number of workgroups if a tapir_conv_fused* launch of N images of this shape takes the flat tiling under
 * the current mode, TAPIR_ERR_UNSUPPORTED if it takes the per-image tiling (no error message is set). */
int tapir_conv_flat_plan(tapir_ctx* ctx, int N, int H, int W, int cin, int cout, int ks, int stride, int* workgroups);
/* Defect hunting (tools/probe_two_process.py): the separate-launch mixer returns after `stages` launch groups (1 = the
 * input Linear, then per block: token mixing, up-projection, down-projection; 0 = off); the engine's workspaces
 * (which: 0 mlp_in, 1 xa, 2 xb, 3 xn, 4 hid, 5 res, 6 split-K partials) as device pointer + capacity; a launch that fills
 * every CU's LDS with a pattern. */
int tapir_debug_mixer_stop(tapir_ctx* ctx, int stages);
int tapir_debug_workspace(tapir_ctx* ctx, int which, void** p, unsigned long long* bytes);
int tapir_debug_poison_lds(tapir_ctx* ctx, unsigned pattern, void* stream);
/* refine_pips's state update (tapir_model.py:613-623): 1 (default) = applied by the output stage of the track-resident
 * mixer kernels, 0 = always the separate update kernel on the mixer's [R,388] output (A/B, tests; bit-identical). */
int tapir_debug_set_update_mode(tapir_ctx* ctx, int mode);
/* refine_pips's front half (patch correlation + mixer input rows, tapir_model.py:496-594): 0 (default) = the separate
 * patch_corr_kernel launch writing mlp_in to HBM; 1 = built by the track-resident mixer kernel in its prologue, straight
 * into LDS, where that kernel runs (bit-identical; measured equal in time: profiles/r04_ab_fuse_patch.txt). */
int tapir_debug_set_patch_mode(tapir_ctx* ctx, int mode);
int tapir_debug_mix(tapir_ctx* ctx, int block, const float* x_in, float* x_out, void* xn,
                    int N, int T, int tc, void* stream);

/* RCCL all-gather of frame-sharded feature grids over xGMI is
 * done by the host layer with torch.distributed (backend "nccl" == RCCL); the
 * library itself has no collective. */

#ifdef __cplusplus
}
#endif
#endif /* TAPIR_HIP_H_ */
