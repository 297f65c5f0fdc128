// libtapir_hip: context, weights, workspaces, stage orchestration and the C ABI
// declared in include/tapir_hip.h.  Everything is enqueued on the caller's
// stream; there is no host synchronisation and no CPU compute path.
#include "../../include/tapir_hip.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "backbone.hpp"
#include "common.hpp"
#include "conv_fused.hpp"
#include "conv_flat.hpp"
#include "conv_small.hpp"
#include "costvol.hpp"
#include "costvol_fused.hpp"
#include "costvol_rows.hpp"
#include "extra_convs.hpp"
#include "gemm.hpp"
#include "mixer.hpp"
#include "mixer_fused.hpp"
#include "mixer_fused_wide.hpp"
#include "mixer_online.hpp"
#include "pips.hpp"

using namespace tapir;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct BlockW {
  float *ln1, *w1, *b1, *w2, *b2, *ln2, *bup, *bdn;
  void *Wup, *Wdn;   // operand type: [2048,512], [512,2048]
};

}  // namespace

struct tapir_ctx {
  tapir_cfg cfg;
  int device = 0;
  std::string err;
  std::map<std::string, HostTensor> host_w;
  bool finalized = false;
  std::vector<void*> owned;         // device weight allocations (hot path: rebuilt by tapir_finalize_weights)
  std::map<const void*, int> xconv_cch;   // tapir_xconv_pack: input channels per chunk each pack was built for
  std::vector<void*> conv_owned;    // backbone weight packs (tapir_conv_pack / tapir_stem_pack): they belong to the
                                    // caller's Backbone object and outlive tapir_finalize_weights (tapir_conv_free)

  // cost-volume head weights (f32)
  CvHeadWeights cvw;
  CvHeadWeights tapnet_cvw;          // TAP-Net head (tapnet_model.py:64-107), uploaded when its weights are set
  bool tapnet_ready = false;
  int tapnet_heads = 1;              // num_heads of the TAP-Net head: input channels of its hid1
  bool tapir_ready = false;          // the TAPIR weights (heads + mixer) are uploaded
  // mixer weights
  int in_dim = 0, k0_pad = 0;       // 388 + 49*(2+pyr), padded to the GEMM k-step
  void* W0 = nullptr; float* b0 = nullptr;        // [512, k0_pad]
  std::vector<BlockW> blocks;
  float* lnF = nullptr;
  void* Wout = nullptr; float* bout = nullptr;    // [388, 512]
  // track-resident fused mixer (mixer_fused.hpp): per-wave A-fragment streams + per-block vectors
  uint4* fused_stream = nullptr; long fused_fpw = 0;
  uint4* fused_wide_stream = nullptr; long fused_wide_fpw = 0;   // bf16: the 6-tile kernel's chunking (mixer_fused_wide.hpp)
  std::vector<FusedBlockParams> fused_blocks;     // per-block vectors (passed in the kernel arguments)
  int dbg_mixer_stop = 0;                         // tapir_debug_mixer_stop: the separate-launch mixer returns after this many launch groups (0: off)
  int mixer_mode = 0;                             // 0 auto, 1 separate launches, 2 fused (tapir_debug_set_mixer_mode)
  bool cv_tiled = true;                           // row-streamed cost volume, bf16: contraction operand in tile order (TAPIR_CV_TILED=0: row-major, A/B)
  bool fuse_patch = false;                        // refine_pips's front half in the track-resident mixer's prologue (TAPIR_FUSE_PATCH=1; measured: the
                                                  // prologue costs what the separate launch costs, profiles/r04_ab_fuse_patch.txt -- opt-in)
  int cv_form = 1;                                // row-streamed cost volume: maps x waves per workgroup (costvol_rows.hpp; TAPIR_CV_FORM, A/B)
  int cv_mode = 0;                                // 0 auto (fused where it applies), 1 einsum workspace + heads kernel
  bool warm_weights = true;                       // read the track-resident mixer's weight stream once in front of a level's first iteration
                                                  // (TAPIR_WARM_WEIGHTS=0: off, A/B; warm_stream_kernel below)
  DevBuf warm_sink;                               // 4 bytes the warming kernel never writes
  int fused_min_tracks = 48;                      // fewest tracks the track-resident mixer is chosen for (TAPIR_FUSED_MIN_TRACKS).  One
                                                  // workgroup per track runs 613-640 us per launch whatever the track count; the separate
                                                  // launches take 580 / 588 / 652 / 780 / 858 us at 16 / 32 / 64 / 96 / 128 tracks x 48 frames
                                                  // (profiles/r05_kbench_mixer_small.txt): the crossover is between 32 and 64 tracks
  int cv_stream_out = 0;                          // cost-volume workspace GEMM: non-temporal stores of the f32 volume (TAPIR_CV_STREAM_OUT=1; measured
                                                  // SLOWER at the production chunk: 101 against 84 us at M = 682, profiles/r05_kbench_contraction.txt -- off)
  int fuse_update = 1;                            // track-resident mixers apply refine_pips's state update themselves (0: update_kernel; A/B, tests)
  int xconv_nt = 0;                               // ExtraConvs convolutions: 0 = xconv_plan chooses, 4 / 8 = pixel tiles per wave forced (TAPIR_XCONV_NT: tests, A/B)
  int conv_flat = 0;                              // 3x3 256 -> 256 block convolutions: 1 = the flat tiling (conv_flat.hpp) wherever it applies, 0 = never (default:
                                                  // faster as a single 48-frame launch, 74 vs 87 us, slower inside the 4-stream backbone, 4.60 vs 4.51 ms per step:
                                                  // profiles/r06_ab_flat_v1.txt), -1 = from conv_flat_min_slabs slabs per launch on (TAPIR_CONV_FLAT / tapir_debug_set_conv_flat)
  int conv_small = 0;                             // block convolutions of FEW-frame clips (the online model): 1 = conv_small_kernel (conv_small.hpp) where the shape
                                                  // allows it; set per clip by the caller's backbone (tapir_conv_set_small), follows the WHOLE clip's frame count
  int conv_flat_min_slabs = 96;                   // (TAPIR_CONV_FLAT_MIN_SLABS)
  int small_gemm = 3;                             // few-row GEMMs: 3 = 2 + the online model's mixer (one frame, causal, <= 256 rows) as ONE persistent launch over all blocks (mixer_online.hpp), 2 = 1 + the channel MLP of a block in ONE launch (mlp_small_kernel), 1 = gemm_small_kernel (one launch each), 0 = split-K + reduce
  int online_form = 0;                            // (tests, TAPIR_ONLINE_FORM) bit 0: the members of the persistent launch that read the same weight slice on ONE XCD (default: a cluster on one XCD); bit 1: acquire + plain loads instead of sc1 loads; bit 2: one member never arrives (the timeout path)
  int n_cus = 0;                                  // compute units of the device (the persistent launch needs its 256 workgroups resident at once)

  // workspaces
  DevBuf cv, mlp_in, xa, xb, xn, hid, res, pos, occ, expd, occ0, expd0, feats, qpts;
  DevBuf qf_cast, grid_cast[kMaxLevels], pooled;
  DevBuf cyc_pts, cyc_feat, cyc_map, cyc_inv;   // cycle-consistency tracker (tapir_cycle_consistency_tracks)
  struct Staged { const float* f32; const void* op; const void* tiled; };
  std::vector<Staged> staged;       // operand-type copies the caller's backbone wrote next to its f32 grids (tapir_set_staged_grid)
  DevBuf grid_tiled;                 // bf16 low-res grid in the cost-volume kernel's operand order (pips.hpp: PoolArgs::tiled)
  const float* tiled_src = nullptr;  // which grid it holds (valid together with cast_src[1])
  DevBuf splitk;    // [splits, M, N] f32 partial sums of the few-row GEMMs
  hipStream_t side = nullptr;       // the weight warm-up of a clip's first refinement iteration runs here, under the cost volume
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool warm_pending = false;        // do_estimate launched that pass: run_mixer joins it instead of launching its own
  int fuse_iter0 = 1;               // (TAPIR_FUSE_ITER0=0: iter0_kernel as its own launch behind the cost volume, as before)
  int warm_side = 1;                // (TAPIR_WARM_SIDE=0: on the caller's stream in front of the mixer, as before)
  DevBuf online_sync;               // mixer_online.hpp: cluster counters + error word (zeroed in-stream before every launch)
  OnlineBlockW* online_blocks = nullptr;   // device table of the blocks' parameters
  int pinned = 0;               // tapir_pin_workspaces count: > 0 = growth is an error (hipGraphs hold the pointers)
  void* dbg_times = nullptr;   // tools only: device buffer for kernel phase stamps (tapir_debug_set_trace)
  // which caller grid each cast slot currently holds (valid within one call)
  const float* cast_src[kMaxLevels] = {nullptr, nullptr, nullptr};

  // optional per-kernel-class timing with hipEvents on the caller's stream
  unsigned prof = 0;   // bit k: kernel class k is bracketed by events
  int prof_stride = 1;                       // every prof_stride-th launch of a class is timed (tapir_profile_stride)
  unsigned prof_count[TAPIR_PROF_KINDS] = {};
  struct ProfEv { hipEvent_t a, b; };
  std::vector<ProfEv> prof_ev[TAPIR_PROF_KINDS];   // recorded, not yet read
  std::vector<ProfEv> prof_free;                   // recycled event pairs
};

namespace {

int fail(tapir_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

// times one kernel class when profiling is on.  single = the scope holds exactly one launch that goes
// through TAPIR_LAUNCH: the dispatch then writes the kernel's own start / stop timestamps into the
// two events (LaunchTimer, common.hpp); otherwise the scope is bracketed by two event markers.
struct ProfScope {
  tapir_ctx* c; int kind; hipStream_t s; tapir_ctx::ProfEv ev; bool on, single;
  ProfScope(tapir_ctx* c_, int kind_, hipStream_t s_, bool single_ = true)
      : c(c_), kind(kind_), s(s_), on((c_->prof >> kind_) & 1u), single(single_) {
    if (on) {   // events cannot be recorded while the stream is being captured into a hipGraph: the scope is a no-op there
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) on = false;
    }
    // every prof_stride-th launch of a class carries events (tapir_profile_stride): a timed launch is dispatched with start / stop
    // signals, which costs ~12 us of idle device on either side of it (profiles/r06_ab_prof_stride.txt)
    if (on && c->prof_stride > 1 && (c->prof_count[kind]++ % (unsigned)c->prof_stride) != 0) on = false;
    if (!on) return;
    if (!c->prof_free.empty()) { ev = c->prof_free.back(); c->prof_free.pop_back(); }
    else if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
    if (single) { LaunchTimer& t = launch_timer(); t.start = ev.a; t.stop = ev.b; t.used = false; }
    else (void)hipEventRecord(ev.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    if (single) {
      LaunchTimer& t = launch_timer();
      const bool used = t.used;
      t = LaunchTimer{};
      if (!used) { (void)hipEventRecord(ev.a, s); (void)hipEventRecord(ev.b, s); }   // no launch: empty interval
    } else {
      (void)hipEventRecord(ev.b, s);
    }
    c->prof_ev[kind].push_back(ev);
  }
};

#define HIP_TRY(c, expr)                                                         \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess)                                                        \
      return fail((c), TAPIR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int ensure(tapir_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return TAPIR_OK;
  // A captured hipGraph (tapnet_amd.online.OnlineTracker) has the workspace pointers baked in:
  // freeing one under it would make every later replay read and write freed memory.
  if (c->pinned)
    return fail(c, TAPIR_ERR_INVALID,
                "workspace growth while pinned (a captured hipGraph uses the current buffers): "
                "tapir_reserve() the largest shape before capturing, or use a separate context");
  if (b.p) HIP_TRY(c, hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  const size_t want = (bytes + 255) / 256 * 256;
  HIP_TRY(c, hipMalloc(&b.p, want));
  b.cap = want;
  return TAPIR_OK;
}

size_t esize(int dtype) { return dtype == TAPIR_BF16 ? 2 : 4; }

inline uint16_t host_f2bf(float f) {
  uint32_t u; ::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// uploads a host f32 matrix [rows, cols] as operand type with row stride ld (zero padded)
int upload_matrix(tapir_ctx* c, const float* src, int rows, int cols, int ld, void** out) {
  const size_t es = esize(c->cfg.dtype);
  std::vector<uint8_t> tmp((size_t)rows * ld * es, 0);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < cols; ++k) {
      const float v = src[(size_t)r * cols + k];
      if (c->cfg.dtype == TAPIR_BF16) ((uint16_t*)tmp.data())[(size_t)r * ld + k] = host_f2bf(v);
      else ((float*)tmp.data())[(size_t)r * ld + k] = v;
    }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, tmp.size()));
  c->owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
  *out = d;
  return TAPIR_OK;
}

int upload_f32(tapir_ctx* c, const float* src, size_t n, float** out) {
  float* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, n * sizeof(float)));
  c->owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice));
  *out = d;
  return TAPIR_OK;
}

int get_w(tapir_ctx* c, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
  auto it = c->host_w.find(name);
  if (it == c->host_w.end()) return fail(c, TAPIR_ERR_WEIGHTS, "missing weight: " + name);
  if (it->second.shape != shape) return fail(c, TAPIR_ERR_WEIGHTS, "wrong shape for weight: " + name);
  *out = &it->second;
  return TAPIR_OK;
}

#define TRY(expr) do { int rc_ = (expr); if (rc_ != TAPIR_OK) return rc_; } while (0)

// uploads one cost-volume head (TAPIR: tapir_model.py:342-361, n_out = 2; TAP-Net:
// tapnet_model.py:64-107, n_out = 1) from the host tensors named prefix + {hid1..occ_out}
int upload_cv_head(tapir_ctx* c, const std::string& cv, int n_out, CvHeadWeights* out, int heads = 1) {
  const HostTensor* t;
  float* tmp;
  TRY(get_w(c, cv + "hid1.weight", {16, heads, 3, 3}, &t)); TRY(upload_f32(c, t->data.data(), (size_t)144 * heads, &tmp)); out->w1 = tmp;
  TRY(get_w(c, cv + "hid1.bias", {16}, &t)); TRY(upload_f32(c, t->data.data(), 16, &tmp)); out->b1 = tmp;
  TRY(get_w(c, cv + "hid2.weight", {1, 16, 3, 3}, &t)); TRY(upload_f32(c, t->data.data(), 144, &tmp)); out->w2 = tmp;
  TRY(get_w(c, cv + "hid2.bias", {1}, &t)); TRY(upload_f32(c, t->data.data(), 1, &tmp)); out->b2 = tmp;
  TRY(get_w(c, cv + "hid3.weight", {32, 16, 3, 3}, &t));
  {
    std::vector<float> r(144 * 32);
    for (int co = 0; co < 32; ++co)
      for (int ci = 0; ci < 16; ++ci)
        for (int k = 0; k < 9; ++k) r[(ci * 9 + k) * 32 + co] = t->data[(co * 16 + ci) * 9 + k];
    TRY(upload_f32(c, r.data(), r.size(), &tmp)); out->w3 = tmp;
    // bf16 build: the same weights as MFMA 16x16x32 B fragments, k = tap*16 + ci padded to 160:
    // fragment (k-step s, n-tile nt), lane l: column n = l & 15, k = 32 s + 8 (l >> 4) + j
    std::vector<uint16_t> fb((size_t)5 * 2 * 64 * 8, 0);
    for (int s5 = 0; s5 < 5; ++s5)
      for (int nt = 0; nt < 2; ++nt)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * s5 + 8 * (l >> 4) + j, tap = k / 16, ci = k % 16, co = nt * 16 + (l & 15);
            if (tap < 9) fb[(((size_t)s5 * 2 + nt) * 64 + l) * 8 + j] = host_f2bf(t->data[(co * 16 + ci) * 9 + tap]);
          }
    void* dfb = nullptr;
    HIP_TRY(c, hipMalloc(&dfb, fb.size() * 2));
    c->owned.push_back(dfb);
    HIP_TRY(c, hipMemcpy(dfb, fb.data(), fb.size() * 2, hipMemcpyHostToDevice));
    out->w3b = (const uint4*)dfb;
  }
  TRY(get_w(c, cv + "hid3.bias", {32}, &t)); TRY(upload_f32(c, t->data.data(), 32, &tmp)); out->b3 = tmp;
  TRY(get_w(c, cv + "hid4.weight", {16, 32}, &t)); TRY(upload_f32(c, t->data.data(), 512, &tmp)); out->w4 = tmp;
  TRY(get_w(c, cv + "hid4.bias", {16}, &t)); TRY(upload_f32(c, t->data.data(), 16, &tmp)); out->b4 = tmp;
  TRY(get_w(c, cv + "occ_out.weight", {n_out, 16}, &t)); TRY(upload_f32(c, t->data.data(), (size_t)n_out * 16, &tmp)); out->w5 = tmp;
  TRY(get_w(c, cv + "occ_out.bias", {n_out}, &t)); TRY(upload_f32(c, t->data.data(), (size_t)n_out, &tmp)); out->b5 = tmp;
  return TAPIR_OK;
}

// ---- weights of the fused mixer kernel (mixer_fused.hpp): every wave reads ONE linear stream of
// 1-KiB MFMA A fragments, in the order it multiplies them.  Fragment (row0, k0) of matrix W [rows,
// cols]: lane l holds W[row0 + (l & 15)][k0 + (l >> 4) * EPC + j], j < EPC (zero outside W).
template <typename TA>
void pack_fragment(uint8_t* dst, const float* W, int rows, int cols, int row0, int k0) {
  constexpr int EPC = 16 / (int)sizeof(TA);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < EPC; ++j) {
      const int r = row0 + (l & 15), k = k0 + (l >> 4) * EPC + j;
      const float v = (r < rows && k < cols) ? W[(size_t)r * cols + k] : 0.f;
      if (sizeof(TA) == 2) ((uint16_t*)dst)[l * EPC + j] = host_f2bf(v);
      else ((float*)dst)[l * EPC + j] = v;
    }
}

template <typename TA>
int build_fused_weights(tapir_ctx* c) {
  using CF = FusedCfg<TA>;
  const int nb = c->cfg.num_mixer_blocks;
  const long fpw = fused_frags_per_wave<TA>(c->k0_pad, nb);
  std::vector<uint8_t> host((size_t)FM_WAVES * fpw * 1024, 0);
  const std::string mx = "torch_pips_mixer.";
  const HostTensor *w0, *wout;
  TRY(get_w(c, mx + "linear.weight", {kHidden, c->in_dim}, &w0));
  TRY(get_w(c, mx + "linear_1.weight", {kMixOut, kHidden}, &wout));
  std::vector<const HostTensor*> wup(nb), wdn(nb);
  for (int b = 0; b < nb; ++b) {
    const std::string p = mx + "blocks." + std::to_string(b) + ".conv_channels_mixer.";
    TRY(get_w(c, p + "mlp2_up.weight", {kHidden4, kHidden}, &wup[b]));
    TRY(get_w(c, p + "mlp2_down.weight", {kHidden, kHidden4}, &wdn[b]));
  }
  constexpr int RAU = CF::HC / 8 / 16, NC = kHidden4 / CF::HC;
  for (int w = 0; w < FM_WAVES; ++w) {
    uint8_t* q = host.data() + (size_t)w * fpw * 1024;
    auto put = [&](const HostTensor* t, int rows, int cols, int row0, int k0) {
      pack_fragment<TA>(q, t->data.data(), rows, cols, row0, k0);
      q += 1024;
    };
    for (int ks = 0; ks < c->k0_pad / CF::KS; ++ks)
      for (int a = 0; a < 4; ++a) put(w0, kHidden, c->in_dim, 64 * w + 16 * a, ks * CF::KS);
    // per block, the order the pipelined chunk loop consumes them in: U0 U1 D0 U2 D1 ... D(NC-1)
    auto put_up = [&](int b, int hc) {
      for (int ks = 0; ks < kHidden / CF::KS; ++ks)
        for (int a = 0; a < RAU; ++a)
          put(wup[b], kHidden4, kHidden, hc * CF::HC + w * (CF::HC / 8) + 16 * a, ks * CF::KS);
    };
    auto put_dn = [&](int b, int hc) {
      for (int ks = 0; ks < CF::HC / CF::KS; ++ks)
        for (int a = 0; a < 4; ++a)
          put(wdn[b], kHidden, kHidden4, 64 * w + 16 * a, hc * CF::HC + ks * CF::KS);
    };
    for (int b = 0; b < nb; ++b) {
      put_up(b, 0);
      for (int hc = 1; hc < NC; ++hc) { put_up(b, hc); put_dn(b, hc - 1); }
      put_dn(b, NC - 1);
    }
    for (int ks = 0; ks < kHidden / CF::KS; ++ks)
      for (int a = 0; a < 4; ++a) put(wout, kMixOut, kHidden, 64 * w + 16 * a, ks * CF::KS);
    if (q + (size_t)FM_RING * 1024 != host.data() + (size_t)(w + 1) * fpw * 1024)
      return fail(c, TAPIR_ERR_WEIGHTS, "fused stream layout mismatch");
  }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  c->fused_stream = (uint4*)d;
  c->fused_fpw = fpw;
  // per-channel temporal-convolution parameters (LN1 scale folded into the first convolution)
  std::vector<FusedBlockParams> bps(nb);
  for (int b = 0; b < nb; ++b) {
    const std::string p = mx + "blocks." + std::to_string(b) + ".";
    const HostTensor *ln1, *w1, *b1, *w2, *b2;
    TRY(get_w(c, p + "layer_norm.weight", {kHidden}, &ln1));
    TRY(get_w(c, p + "mlp1_up.weight", {kHidden4, 1, 3}, &w1));
    TRY(get_w(c, p + "mlp1_up.bias", {kHidden4}, &b1));
    TRY(get_w(c, p + "mlp1_up_1.weight", {kHidden4, 1, 3}, &w2));
    TRY(get_w(c, p + "mlp1_up_1.bias", {kHidden4}, &b2));
    std::vector<float> mw((size_t)kHidden * FM_MIXW, 0.f);
    for (int ch = 0; ch < kHidden; ++ch) {
      // slot j (0-2: w1[m][k] * ln1, 3: b1[m], 4-6: w2[m][k], 7: sum of b2 for m == 0) of multiplier m
      // of channel ch sits at [(ch / 2) * FM_MIXW + m * 8 + j][ch & 1] (channel pairs interleaved)
      float* o = mw.data() + (size_t)(ch >> 1) * FM_MIXW * 2 + (ch & 1);
      float bsum = 0.f;
      for (int m = 0; m < 4; ++m) {
        const int oc = 4 * ch + m;
        for (int k = 0; k < 3; ++k) {
          o[2 * (m * 8 + k)] = w1->data[oc * 3 + k] * ln1->data[ch];
          o[2 * (m * 8 + 4 + k)] = w2->data[oc * 3 + k];
        }
        o[2 * (m * 8 + 3)] = b1->data[oc];
        bsum += b2->data[oc];
      }
      o[2 * 7] = bsum;
    }
    float* dm = nullptr;
    TRY(upload_f32(c, mw.data(), mw.size(), &dm));
    bps[b].mixw = dm;
    bps[b].ln2 = c->blocks[b].ln2; bps[b].bup = c->blocks[b].bup; bps[b].bdn = c->blocks[b].bdn;
  }
  c->fused_blocks = bps;
  return TAPIR_OK;
}

// the same matrices in the order / chunking of the wide kernel (mixer_fused_wide.hpp): chunks of 256
// hidden units, sequential U0 D0 U1 D1 ..., output Linear in two passes of two row tiles
int build_fused_wide_weights(tapir_ctx* c) {
  typedef bf16_t TA;
  const int nb = c->cfg.num_mixer_blocks;
  const long fpw = fused_wide_frags_per_wave(c->k0_pad, nb);
  std::vector<uint8_t> host((size_t)FM_WAVES * fpw * 1024, 0);
  const std::string mx = "torch_pips_mixer.";
  const HostTensor *w0, *wout;
  TRY(get_w(c, mx + "linear.weight", {kHidden, c->in_dim}, &w0));
  TRY(get_w(c, mx + "linear_1.weight", {kMixOut, kHidden}, &wout));
  constexpr int HC = FMW_HC, RAU = HC / 8 / 16, NC = kHidden4 / HC;
  for (int w = 0; w < FM_WAVES; ++w) {
    uint8_t* q = host.data() + (size_t)w * fpw * 1024;
    auto put = [&](const HostTensor* t, int rows, int cols, int row0, int k0) {
      pack_fragment<TA>(q, t->data.data(), rows, cols, row0, k0);
      q += 1024;
    };
    for (int ks = 0; ks < c->k0_pad / 32; ++ks)
      for (int a = 0; a < 4; ++a) put(w0, kHidden, c->in_dim, 64 * w + 16 * a, ks * 32);
    for (int b = 0; b < nb; ++b) {
      const std::string p = mx + "blocks." + std::to_string(b) + ".conv_channels_mixer.";
      const HostTensor *wup, *wdn;
      TRY(get_w(c, p + "mlp2_up.weight", {kHidden4, kHidden}, &wup));
      TRY(get_w(c, p + "mlp2_down.weight", {kHidden, kHidden4}, &wdn));
      for (int hc = 0; hc < NC; ++hc) {
        for (int ks = 0; ks < kHidden / 32; ++ks)
          for (int a = 0; a < RAU; ++a) put(wup, kHidden4, kHidden, hc * HC + w * (HC / 8) + 16 * a, ks * 32);
        for (int ks = 0; ks < HC / 32; ++ks)
          for (int a = 0; a < 4; ++a) put(wdn, kHidden, kHidden4, 64 * w + 16 * a, hc * HC + ks * 32);
      }
    }
    for (int half = 0; half < 2; ++half)
      for (int ks = 0; ks < kHidden / 32; ++ks)
        for (int a = 0; a < 2; ++a) put(wout, kMixOut, kHidden, 64 * w + 16 * (2 * half + a), ks * 32);
    if (q + (size_t)FMW_RING * 1024 != host.data() + (size_t)(w + 1) * fpw * 1024)
      return fail(c, TAPIR_ERR_WEIGHTS, "wide fused stream layout mismatch");
  }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  c->fused_wide_stream = (uint4*)d;
  c->fused_wide_fpw = fpw;
  return TAPIR_OK;
}

// ----------------------------------------------------------------------------
// small helper kernels
// ----------------------------------------------------------------------------
struct InitArgs {
  const float* qpts_video;  // [BQ,3] or null
  float* qpts_init;         // [BQ,3] (t,y,x) scaled to initial_resolution
  long BQ;
  float sy, sx;             // initial / video
};
__global__ void scale_qpts_kernel(InitArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.BQ) return;
  a.qpts_init[i * 3 + 0] = a.qpts_video[i * 3 + 0];
  a.qpts_init[i * 3 + 1] = a.qpts_video[i * 3 + 1] * a.sy;
  a.qpts_init[i * 3 + 2] = a.qpts_video[i * 3 + 2] * a.sx;
}

struct Iter0Args {
  const float* pos; const float* occ; const float* expd;   // state after the cost-volume stage
  float* occ0; float* expd0;
  float* out_tracks; float* out_occ; float* out_expd;
  long R; float vx, vy;
};
__global__ void iter0_kernel(Iter0Args a) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.R) return;
  a.out_tracks[r * 2 + 0] = a.pos[r * 2 + 0] * a.vx;
  a.out_tracks[r * 2 + 1] = a.pos[r * 2 + 1] * a.vy;
  a.out_occ[r] = a.occ[r]; a.out_expd[r] = a.expd[r];
  a.occ0[r] = a.occ[r]; a.expd0[r] = a.expd[r];
}

// ----------------------------------------------------------------------------
// stage drivers (templated on the operand type)
// ----------------------------------------------------------------------------
// operand-type copy registered for this f32 grid (tapir_set_staged_grid), or null
inline const tapir_ctx::Staged* find_staged(const tapir_ctx* c, const float* grid) {
  for (const auto& st : c->staged) if (st.f32 == grid) return &st;
  return nullptr;
}

// bytes of the tiled bf16 copy of a [frames, h*w, 256] grid
inline size_t tiled_bytes(long frames, int h, int w) { return (size_t)frames * ((h * w + 15) / 16) * 16 * kLowresDim * 2; }

// grows like ensure(); new memory is zeroed once (cells past the end of a frame's last tile are never written)
int ensure_zeroed(tapir_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return TAPIR_OK;
  TRY(ensure(c, b, bytes));
  HIP_TRY(c, hipMemset(b.p, 0, b.cap));
  return TAPIR_OK;
}

template <typename TA>
int cast_or_pool(tapir_ctx* c, const float* src, long frames, int h, int w, int C, int pool,
                 DevBuf& dst, hipStream_t s, DevBuf* tiled = nullptr) {
  const int oh = pool ? h / 2 : h, ow = pool ? w / 2 : w;
  TRY(ensure(c, dst, (size_t)frames * oh * ow * C * sizeof(TA)));
  if (tiled != nullptr) TRY(ensure_zeroed(c, *tiled, tiled_bytes(frames, h, w)));
  PoolArgs pa{src, dst.p, frames, h, w, C, pool, tiled != nullptr ? tiled->p : nullptr};
  const long total = frames * oh * ow * (C / 4);
  const int nb = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL((pool_cast_kernel<TA>), dim3(nb), dim3(256), 0, s, pa);
  return TAPIR_OK;
}

template <typename TA>
int cost_volume_gemm(tapir_ctx* c, const void* qf, const void* grid, int Q, int T, int hw, int C,
                     float* vol, hipStream_t s) {
  GemmArgs g{};
  g.A = qf; g.lda = C;
  g.W = grid; g.ldw = C;
  g.bias = nullptr; g.resid = nullptr; g.ldr = 0;
  // T*hw not a multiple of 4 (odd grids x odd frame counts): the GEMM stores 4 columns at a time, so it runs over the
  // next multiple with W's rows clamped to the real ones; the volume's rows carry the padding (launch_cv_heads: ld)
  const int n = T * hw, n4 = (n + 3) & ~3;
  g.C = vol; g.ldc = n4;
  g.M = Q; g.N = n4; g.K = C; g.w_rows = n == n4 ? 0 : n;
  g.stream_out = c->cv_stream_out;
  { ProfScope ps(c, TAPIR_PROF_CV_GEMM, s); launch_gemm<TA, float, EPI_BIAS>(g, s); }
  return TAPIR_OK;
}

int launch_cv_heads(tapir_ctx* c, const float* cv, const float* qpts_init, long maps, int T,
                    int h, int w, float* points, float* occ, float* expd, hipStream_t s) {
  CvHeadArgs a{};
  a.cv = cv; a.wt = c->cvw; a.qpts = qpts_init;
  a.points = points; a.occ = occ; a.expd = expd;
  a.T = T; a.h = h; a.w = w; a.maps = maps;
  a.ld = ((long)T * h * w) % 4 ? (((long)T * h * w + 3) & ~3L) : 0;
  a.temperature = c->cfg.softmax_temperature;
  a.img_h = (float)c->cfg.initial_h; a.img_w = (float)c->cfg.initial_w;
  a.dbg_times = (long long*)c->dbg_times;
  const int pn = (h + 2) * (w + 2), hw = h * w;
  ProfScope ps(c, TAPIR_PROF_CV_HEADS, s);
  if (c->cfg.dtype == TAPIR_BF16 && pn <= CV_SMALL_PAD && hw <= CV_SMALL_PPT * CV_THREADS) {
    // bf16 build: occlusion convolution on the matrix cores
    if (a.dbg_times != nullptr)
      TAPIR_LAUNCH((cv_heads_mfma_kernel<CV_SMALL_PAD, CV_SMALL_PPT, CV_THREADS, true>),
                   dim3((unsigned)std::min<long>(maps, 512)), dim3(CV_THREADS), s, a);
    else
      TAPIR_LAUNCH((cv_heads_mfma_kernel<CV_SMALL_PAD, CV_SMALL_PPT, CV_THREADS>),
                   dim3((unsigned)std::min<long>(maps, 512)), dim3(CV_THREADS), s, a);
  } else if (pn <= CV_SMALL_PAD && hw <= CV_SMALL_PPT * CV_THREADS) {
    TAPIR_LAUNCH((cv_heads_kernel<CV_SMALL_PAD, CV_SMALL_PPT>), dim3((unsigned)maps), dim3(CV_THREADS), s, a);
  } else if (pn <= CV_LARGE_PAD && hw <= CV_LARGE_PPT * CV_THREADS) {
    TAPIR_LAUNCH((cv_heads_kernel<CV_LARGE_PAD, CV_LARGE_PPT>), dim3((unsigned)maps), dim3(CV_THREADS), s, a);
  } else {
    return fail(c, TAPIR_ERR_UNSUPPORTED, "cost-volume grid larger than 1600 cells");
  }
  return TAPIR_OK;
}

// cost volume -> tracks for B clips.  qpts_init: device [B*Q,3] or null.
template <typename TA>
int cost_volume_stage(tapir_ctx* c, const float* qfeat, const float* grid, const float* qpts_init,
                      int B, int Q, int T, int h, int w, float* points, float* occ, float* expd,
                      hipStream_t s, bool tapnet = false, const Iter0Args* i0 = nullptr, bool* i0_done = nullptr) {
  // i0 != null: the caller's iter0_kernel arguments; the row-streamed kernel writes those copies itself (*i0_done = true)
  const int C = kLowresDim, hw = h * w;
  const void* qf_op = qfeat;
  const void* grid_op = grid;
  // row-streamed kernel, bf16: the grid also in the operand order of its contraction (one cast kernel writes both)
  const bool rows = cv_rows_supported(h, w) && c->cv_mode == 0 &&
                    !(cv_rows_wide(h, w) && tapnet && c->tapnet_heads != 1);   // (rows of > 32 cells: one head only)
  const bool tiled = rows && sizeof(TA) == 2 && c->cv_tiled;
  const void* tiled_op = nullptr;
  const tapir_ctx::Staged* stg = sizeof(TA) == 2 ? find_staged(c, grid) : nullptr;
  if (stg != nullptr && (!tiled || stg->tiled != nullptr)) {
    // the backbone's L2-normalise kernel already wrote the bf16 copies (row-major and, for this kernel, tile order)
    TRY(cast_or_pool<TA>(c, qfeat, 1, 1, B * Q, C, 0, c->qf_cast, s));
    qf_op = c->qf_cast.p; grid_op = stg->op; tiled_op = stg->tiled;
  } else if (sizeof(TA) == 2) {   // stage bf16 copies of both operands
    TRY(cast_or_pool<TA>(c, qfeat, 1, 1, B * Q, C, 0, c->qf_cast, s));
    if (c->cast_src[1] != grid || (tiled && c->tiled_src != grid)) {
      TRY(cast_or_pool<TA>(c, grid, (long)B * T, h, w, C, 0, c->grid_cast[1], s, tiled ? &c->grid_tiled : nullptr));
      c->cast_src[1] = grid;    // the same layout as the low-res pyramid level: prepare_level reuses it
      if (tiled) c->tiled_src = grid;
    }
    qf_op = c->qf_cast.p; grid_op = c->grid_cast[1].p;
    if (tiled) tiled_op = c->grid_tiled.p;
  }
  if ((rows || cv_fused_supported(h, w)) && c->cv_mode != 1) {
    // one kernel: contraction on the matrix cores into LDS + heads; no volume in HBM
    CvFusedArgs fa{};
    fa.qfeat = qf_op; fa.grid = grid_op; fa.wt = tapnet ? c->tapnet_cvw : c->cvw; fa.qpts = qpts_init;
    fa.grid_tiled = tiled ? tiled_op : nullptr;
    fa.tapnet = tapnet ? 1 : 0;
    fa.points = points; fa.occ = occ; fa.expd = expd;
    fa.B = B; fa.Q = Q; fa.T = T; fa.h = h; fa.w = w;
    fa.temperature = tapnet ? 10.0f : c->cfg.softmax_temperature;   // tapnet_model.py:61
    fa.img_h = (float)c->cfg.initial_h; fa.img_w = (float)c->cfg.initial_w;
    fa.dbg_times = (long long*)c->dbg_times;
    ProfScope ps(c, TAPIR_PROF_CV_HEADS, s);
    // row-streamed form (every wave owns whole maps, costvol_rows.hpp) for rows of up to 32 cells; the
    // pixel-tiled form (costvol_fused.hpp) for the other shapes it covers, and on request (cv_mode 2, A/B)
    if (rows && i0 != nullptr && i0_done != nullptr && !tapnet && c->fuse_iter0) {
      fa.occ0 = i0->occ0; fa.expd0 = i0->expd0; fa.out_tracks = i0->out_tracks; fa.out_occ = i0->out_occ; fa.out_expd = i0->out_expd;
      fa.vx = i0->vx; fa.vy = i0->vy;
      *i0_done = true;
    }
    if (rows) launch_cv_rows<TA>(fa, s, tapnet ? c->tapnet_heads : 1, c->cv_form);
    else launch_cv_fused<TA>(fa, s, tapnet ? c->tapnet_heads : 1);
    return TAPIR_OK;
  }
  if (tapnet) return fail(c, TAPIR_ERR_UNSUPPORTED, "TAP-Net head: grids of up to 32 x 32 cells");
  // grids beyond 32 x 32 cells (or cv_mode 1, tools): einsum into a workspace of <= 256 MiB per
  // launch, then the heads kernel
  const long row = ((long)T * hw + 3) & ~3L;            // volume row of a query, padded to 16 bytes
  long qc = (256L << 20) / (row * 4);
  qc = std::max<long>(1, std::min<long>(qc, Q));
  TRY(ensure(c, c->cv, (size_t)qc * row * sizeof(float)));
  for (int b = 0; b < B; ++b) {
    for (long q0 = 0; q0 < Q; q0 += qc) {
      const int nq = (int)std::min<long>(qc, Q - q0);
      const long bq = (long)b * Q + q0;
      const TA* qa = reinterpret_cast<const TA*>(qf_op) + bq * C;
      const TA* ga = reinterpret_cast<const TA*>(grid_op) + (long)b * T * hw * C;
      TRY(cost_volume_gemm<TA>(c, qa, ga, nq, T, hw, C, (float*)c->cv.p, s));
      TRY(launch_cv_heads(c, (const float*)c->cv.p, qpts_init ? qpts_init + bq * 3 : nullptr,
                          (long)nq * T, T, h, w, points + bq * T * 2, occ + bq * T,
                          expd + bq * T, s));
    }
  }
  return TAPIR_OK;
}

// ---- forward-backward cycle-consistency tracker (tapnet/training/supervised_point_prediction.py:443-546) ----------
struct CycPtsArgs { const float* tracks; const float* qpts; float* pts3; int* frame_map; long BQ; int Q, T; };
// (t, y, x) of the tracked point of every (query, frame), and the grid frame each query came from (:489-507)
__global__ void cycle_points_kernel(CycPtsArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.BQ * a.T) return;
  const long bq = i / a.T;
  const int t = (int)(i - bq * a.T);
  a.pts3[i * 3 + 0] = (float)t;
  a.pts3[i * 3 + 1] = a.tracks[i * 2 + 1];
  a.pts3[i * 3 + 2] = a.tracks[i * 2 + 0];
  if (t == 0) {   // round-half-even like jnp.round; frame b * T + round(t_query) of the [B*T] grid frames
    const int qf = min(max((int)rintf(a.qpts[bq * 3 + 0]), 0), a.T - 1);
    a.frame_map[bq] = (int)(bq / a.Q) * a.T + qf;
  }
}
struct CycOccArgs { const float* inv; const float* qpts; float* occ; long BQ; int T; float thr2; };
// occluded when the backward point misses the query by more than the threshold: logit +10 / -10 (:533-539)
__global__ void cycle_occlusion_kernel(CycOccArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.BQ * a.T) return;
  const long bq = i / a.T;
  const float dx = a.inv[i * 2 + 0] - a.qpts[bq * 3 + 2], dy = a.inv[i * 2 + 1] - a.qpts[bq * 3 + 1];
  a.occ[i] = (dx * dx + dy * dy > a.thr2) ? 10.0f : -10.0f;
}

// one raw pass: points = soft arg max of softmax(temperature * <qfeat, grid cell>) per (query, frame)
template <typename TA>
void cycle_pass(tapir_ctx* c, const void* qf_op, const void* grid_op, const void* tiled_op, const float* qpts,
                const int* frame_map, int B, int Q, int T, int h, int w, float img_h, float img_w, float temperature,
                float* points, hipStream_t s) {
  CvFusedArgs fa{};   // (the head weights stay null: raw mode does not read them)
  fa.qfeat = qf_op; fa.grid = grid_op; fa.grid_tiled = tiled_op; fa.qpts = qpts;
  fa.points = points;
  fa.B = B; fa.Q = Q; fa.T = T; fa.h = h; fa.w = w;
  fa.temperature = temperature; fa.img_h = img_h; fa.img_w = img_w;
  fa.raw = 1; fa.frame_map = frame_map;
  launch_cv_rows<TA>(fa, s, 1, c->cv_form);
}

template <typename TA>
int do_cycle_consistency(tapir_ctx* c, const float* qfeat, const float* grid, const float* qpts, int B, int Q, int T,
                         int h, int w, int img_h, int img_w, float temperature, float threshold, float* tracks,
                         float* occlusion, float* inverse_tracks, hipStream_t s) {
  const int C = kLowresDim;
  if (!cv_rows_supported(h, w))
    return fail(c, TAPIR_ERR_UNSUPPORTED,
                "cycle-consistency tracker: the row-streamed cost-volume kernel covers rows of up to 32 cells on up to "
                "32 (w <= 16: 64) rows, and rows of 33..64 cells on up to 64 rows");
  const long BQ = (long)B * Q, R = BQ * T;
  // The sampled vectors (1 KiB per query and frame) are held for a chunk of queries at a time, like the reference's
  // eval_chunk_size loop over queries (supervised_point_prediction.py:444-452): at most 256 MiB whatever Q is.
  const long qc = std::max<long>(1, std::min<long>(Q, (256L << 20) / ((long)T * C * 4)));
  TRY(ensure(c, c->cyc_pts, (size_t)R * 12)); TRY(ensure(c, c->cyc_feat, (size_t)qc * T * C * 4));
  TRY(ensure(c, c->cyc_map, (size_t)BQ * 4));
  if (inverse_tracks == nullptr) { TRY(ensure(c, c->cyc_inv, (size_t)R * 8)); inverse_tracks = (float*)c->cyc_inv.p; }
  const void* qf_op = qfeat; const void* grid_op = grid; const void* tiled_op = nullptr;
  if (sizeof(TA) == 2) {
    TRY(cast_or_pool<TA>(c, qfeat, 1, 1, B * Q, C, 0, c->qf_cast, s));
    TRY(cast_or_pool<TA>(c, grid, (long)B * T, h, w, C, 0, c->grid_cast[1], s, c->cv_tiled ? &c->grid_tiled : nullptr));
    c->cast_src[1] = nullptr; c->tiled_src = nullptr;
    qf_op = c->qf_cast.p; grid_op = c->grid_cast[1].p; tiled_op = c->cv_tiled ? c->grid_tiled.p : nullptr;
  }
  // forward (:453-469): every query against every frame, the query's own frame overridden by the query point
  cycle_pass<TA>(c, qf_op, grid_op, tiled_op, qpts, nullptr, B, Q, T, h, w, (float)img_h, (float)img_w, temperature, tracks, s);
  // the tracked points as (t, y, x) and the frames the queries came from (:489-514)
  CycPtsArgs pa{tracks, qpts, (float*)c->cyc_pts.p, (int*)c->cyc_map.p, BQ, Q, T};
  hipLaunchKernelGGL(cycle_points_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, pa);
  for (int b = 0; b < B; ++b) {
    for (long q0 = 0; q0 < Q; q0 += qc) {
      const int nq = (int)std::min<long>(qc, Q - q0);
      const long bq = (long)b * Q + q0;
      // features at the tracked points (:473-496): nq * T samples of clip b
      SampleArgs sa{grid + (size_t)b * T * h * w * C, (const float*)c->cyc_pts.p + bq * T * 3, (float*)c->cyc_feat.p,
                    1, nq * T, T, h, w, C, (float)img_h, (float)img_w};
      hipLaunchKernelGGL(query_feature_kernel, dim3((unsigned)(nq * T)), dim3(128), 0, s, sa);
      // backward (:516-531): the T sampled vectors of a query against the ONE frame the query came from -- nq "clips"
      // of one frame with T "queries" each; frame_map picks the grid frame (an index into all B * T frames)
      const void* f_op = c->cyc_feat.p;
      if (sizeof(TA) == 2) {
        TRY(cast_or_pool<TA>(c, (const float*)c->cyc_feat.p, 1, 1, nq * T, C, 0, c->qf_cast, s));
        f_op = c->qf_cast.p;
      }
      cycle_pass<TA>(c, f_op, grid_op, tiled_op, nullptr, (const int*)c->cyc_map.p + bq, nq, T, 1, h, w, (float)img_h,
                     (float)img_w, temperature, inverse_tracks + bq * T * 2, s);
    }
  }
  CycOccArgs oa{inverse_tracks, qpts, occlusion, BQ, T, threshold * threshold};
  hipLaunchKernelGGL(cycle_occlusion_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, oa);
  return TAPIR_OK;
}

template <typename TA>
int do_debug_contraction(tapir_ctx* c, const float* qfeat, const float* grid, int B, int Q, int T, int h, int w,
                         float* scratch, hipStream_t s) {
  const int C = kLowresDim;
  const void* qf_op = qfeat; const void* grid_op = grid; const void* tiled_op = nullptr;
  if (sizeof(TA) == 2) {
    TRY(cast_or_pool<TA>(c, qfeat, 1, 1, B * Q, C, 0, c->qf_cast, s));
    if (c->cast_src[1] != grid || c->tiled_src != grid) {
      TRY(cast_or_pool<TA>(c, grid, (long)B * T, h, w, C, 0, c->grid_cast[1], s, &c->grid_tiled));
      c->cast_src[1] = grid; c->tiled_src = grid;
    }
    qf_op = c->qf_cast.p; grid_op = c->grid_cast[1].p; tiled_op = c->cv_tiled ? c->grid_tiled.p : nullptr;
  }
  CvFusedArgs fa{};
  fa.qfeat = qf_op; fa.grid = grid_op; fa.grid_tiled = tiled_op; fa.points = scratch;
  fa.B = B; fa.Q = Q; fa.T = T; fa.h = h; fa.w = w; fa.temperature = 1.f; fa.img_h = 8.f * h; fa.img_w = 8.f * w;
  fa.raw = 2;
  ProfScope ps(c, TAPIR_PROF_CV_HEADS, s);
  launch_cv_rows<TA>(fa, s, 1, c->cv_form);
  return TAPIR_OK;
}

int pick_time_chunk(int N, int T) {
  // enough workgroups to fill 256 CUs x 4, chunks of at most MIX_MAX_TC frames, at least ~6
  // frames per chunk so the halo recompute stays small
  int nch = (T + MIX_MAX_TC - 1) / MIX_MAX_TC;
  const int want = std::min((1024 + N - 1) / N, (T + 5) / 6);
  nch = std::max(nch, std::max(1, want));
  return (T + nch - 1) / nch;
}

// mixer GEMM: split-K for few rows (online model), the tiled persistent kernel otherwise
template <typename TA, typename TO, int EPI>
int mixer_gemm(tapir_ctx* c, const GemmArgs& g, hipStream_t s) {
  // few rows (the online model): the whole-K small-tile kernel, one launch (gemm.hpp); small_gemm 0 pins the
  // round-2 split-K pair for A/B measurements (tapir_debug_set_gemm_mode)
  if (c->small_gemm && gemm_small_supported<TA>(g.M, g.N, g.K)) {
    launch_gemm_small<TA, TO, EPI>(g, s);
    return TAPIR_OK;
  }
  const int splits = gemm_splits<TA>(g.M, g.K);
  if (splits > 1 && g.N % 4 == 0) {
    TRY(ensure(c, c->splitk, (size_t)splits * g.M * g.N * sizeof(float)));
    launch_gemm_splitk<TA, TO, EPI>(g, splits, (float*)c->splitk.p, s);
  } else {
    launch_gemm<TA, TO, EPI>(g, s);
  }
  return TAPIR_OK;
}

// PIPSMLPMixer on R = N*T token rows already staged in c->mlp_in -> c->res [R,388]
// upd (nullable): the state update that follows the mixer in refine_pips; the track-resident kernels apply it in
// their output stage and set *upd_done (the caller then skips update_kernel)
template <typename TA>
void launch_patch_args(tapir_ctx* c, const PatchArgs& pa, hipStream_t s);

// Reads a buffer once and throws the values away.  Every workgroup of the track-resident mixer streams the SAME 51 MB of
// weight fragments; between two clips the backbone moves ~2.6 GB through HBM, so the first refinement iteration of a
// clip finds them neither in an L2 nor in the memory-side cache and -- every workgroup walking the stream in lock step
// behind the one that misses -- runs 907 instead of 722 us (rocprofv3 timeline of bench.py, profiles/r04_mixer_cold_start.txt).
// One pass over the stream in front of it (13-17 us at HBM speed) leaves it in the 256-MB memory-side cache.
struct WarmArgs { const uint4* p; long n16; unsigned* sink; };
__global__ __launch_bounds__(256) void warm_stream_kernel(WarmArgs a) {
  constexpr int U = 8;
  unsigned acc = 0;
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < a.n16; i0 += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const long i = i0 + k * stride; v[k] = a.p[i < a.n16 ? i : a.n16 - 1]; }
#pragma unroll
    for (int k = 0; k < U; ++k) acc += (v[k].x ^ v[k].y) + (v[k].z ^ v[k].w);
  }
  if (acc == 0x5bd1e995u) *a.sink = acc;   // keeps the loads alive; sink is a word of the context nobody reads
}
inline int warm_stream(tapir_ctx* c, const void* p, size_t bytes, hipStream_t s) {
  if (p == nullptr || bytes < 16) return TAPIR_OK;
  TRY(ensure(c, c->warm_sink, 4));
  WarmArgs a{reinterpret_cast<const uint4*>(p), (long)(bytes / 16), (unsigned*)c->warm_sink.p};
  const long per = 256L * 8;
  const unsigned grid = (unsigned)std::min<long>((a.n16 + per - 1) / per, 2048);
  hipLaunchKernelGGL(warm_stream_kernel, dim3(grid), dim3(256), 0, s, a);
  return TAPIR_OK;
}

// Which form of the mixer a call takes (run_mixer, and do_estimate's early weight warm-up): the track-resident fused kernel, its
// wide form, or (both false) separate launches.
template <typename TA>
int mixer_form(tapir_ctx* c, int N, int T, bool has_ctx, bool* fused_out, bool* wide_out) {
  const long R = (long)N * T;
  const bool causal = c->cfg.use_causal_conv != 0;
  bool fused = c->fused_stream != nullptr && fused_mixer_supported<TA>(T, c->k0_pad, causal, has_ctx);
  // wide form (bf16): two tracks of 17..48 frames per workgroup, or one track of 49..96 frames
  bool wide = sizeof(TA) == 2 && c->fused_wide_stream != nullptr && T > 16 &&
              fused_wide_supported(T, c->k0_pad, causal, has_ctx);
  if (c->mixer_mode == 2 && !fused)
    return fail(c, TAPIR_ERR_UNSUPPORTED, "fused mixer forced, but it does not cover this shape");
  if (c->mixer_mode == 3 && !wide)
    return fail(c, TAPIR_ERR_UNSUPPORTED, "wide fused mixer forced, but it does not cover this shape");
  if (c->mixer_mode == 1) fused = wide = false;
  if (c->mixer_mode == 2) wide = false;
  if (c->mixer_mode == 3 || c->mixer_mode == 4) fused = false;
  if (c->mixer_mode == 4 && !wide) return fail(c, TAPIR_ERR_UNSUPPORTED, "pair simulation: wide shapes only");
  if (c->mixer_mode == 0) {
    // one workgroup per track fills the chip up to 256 tracks; beyond that two tracks per workgroup
    // share the weight stream (MFMA-bound instead of L2-fill-bound); below fused_min_tracks (48) the tiled /
    // few-row GEMMs on all rows are faster than a mostly idle chip
    if (fused) fused = N >= c->fused_min_tracks && R >= (long)c->fused_min_tracks * 32;
    if (wide) wide = T > 48 ? N >= 64 : N > 256;
    if (wide) fused = false;
  }
  *fused_out = fused; *wide_out = wide;
  return TAPIR_OK;
}

template <typename TA>
int run_mixer(tapir_ctx* c, int N, int T, const float* ctx1_in, const float* ctx2_in,
              float* ctx1_out, float* ctx2_out, hipStream_t s, const UpdateArgs* upd = nullptr,
              bool* upd_done = nullptr, const PatchArgs* patch = nullptr) {
  // patch != null: the mixer input rows are still to be built (refine_pips's front half).  The track-resident kernel
  // builds them in its prologue (fuse_patch); every other form gets them from patch_corr_kernel, launched here.
  const long R = (long)N * T;
  const int nb = c->cfg.num_mixer_blocks;
  // track-resident fused kernel (mixer_fused.hpp) for whole non-causal clips; separate launches for
  // the online model, long clips, and few tracks (one workgroup per track: below ~128 tracks the
  // chip is mostly idle and the split-K / tiled GEMMs on all rows are faster)
  {
    const bool has_ctx = ctx1_in || ctx2_in || ctx1_out || ctx2_out;
    bool fused = false, wide = false;
    TRY(mixer_form<TA>(c, N, T, has_ctx, &fused, &wide));
    if ((fused || wide) && c->warm_weights && upd != nullptr && upd->first_of_level) {
      if (c->warm_pending) {       // do_estimate started the pass on the side stream, under the cost volume: join it
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
        c->warm_pending = false;
      } else {
        TRY(warm_stream(c, wide ? c->fused_wide_stream : c->fused_stream,
                        (size_t)FM_WAVES * (size_t)(wide ? c->fused_wide_fpw : c->fused_fpw) * 1024, s));
      }
    }
    const bool in_prologue = patch != nullptr && fused && !wide && c->fuse_patch && c->mixer_mode != 4;
    if (patch != nullptr && !in_prologue) launch_patch_args<TA>(c, *patch, s);
    patch = in_prologue ? patch : nullptr;
    if (fused || wide) {
      TRY(ensure(c, c->res, (size_t)R * kMixOut * 4));
      FusedArgs fa{};
      if (patch != nullptr) { fa.fuse_patch = 1; fa.patch = *patch; }
      fa.mlp_in = c->mlp_in.p; fa.ld_in = c->k0_pad;
      fa.stream = wide ? c->fused_wide_stream : c->fused_stream;
      fa.frags_per_wave = wide ? c->fused_wide_fpw : c->fused_fpw;
      fa.b0 = c->b0; fa.nblocks = nb;
      for (int i = 0; i < nb; ++i) fa.blocks[i] = c->fused_blocks[i];
      fa.dbg_times = (long long*)c->dbg_times;
      fa.lnF = c->lnF; fa.bout = c->bout; fa.res = (float*)c->res.p;
      fa.N = N; fa.T = T;
      fa.pair_sim = c->mixer_mode == 4 ? 1 : 0;
      if (upd != nullptr && upd_done != nullptr && c->fuse_update && !fa.pair_sim) {
        fa.fuse_update = 1; fa.upd = *upd; *upd_done = true;
      }
      ProfScope ps(c, TAPIR_PROF_MIXER, s);
      if (wide) launch_mixer_fused_wide(fa, s);
      else launch_mixer_fused<TA>(fa, s);
      return TAPIR_OK;
    }
  }
  TRY(ensure(c, c->xa, (size_t)R * kHidden * 4));
  TRY(ensure(c, c->xb, (size_t)R * kHidden * 4));
  TRY(ensure(c, c->xn, (size_t)R * kHidden * sizeof(TA)));
  TRY(ensure(c, c->hid, (size_t)R * kHidden4 * sizeof(TA)));
  TRY(ensure(c, c->res, (size_t)R * kMixOut * 4));
  {
    GemmArgs g{};
    g.A = c->mlp_in.p; g.lda = c->k0_pad; g.W = c->W0; g.ldw = c->k0_pad; g.bias = c->b0;
    g.C = c->xa.p; g.ldc = kHidden; g.M = (int)R; g.N = kHidden; g.K = c->k0_pad;
    TRY((mixer_gemm<TA, float, EPI_BIAS>(c, g, s)));
  }
  int stage = 0;                                   // (tools/probe_two_process.py: stop behind a launch group and dump the workspaces)
  auto stop_here = [&]() { return c->dbg_mixer_stop > 0 && ++stage >= c->dbg_mixer_stop; };
  if (stop_here()) return TAPIR_OK;
  const int TC = pick_time_chunk(N, T);
  const int nch = (T + TC - 1) / TC;
  // few rows (the online model, small query shards): the channel MLP of a block as ONE launch that leaves 2048 / 256 partial
  // outputs; the next consumer of x (the next block's mix_kernel, the final LayerNorm) adds them, the bias and the residual
  // while it stages its rows (gemm.hpp mlp_small_kernel).  mix_stream_kernel (plain clips of >= 12 frames) does not read pieces.
  const bool mix_general = c->cfg.use_causal_conv || ctx1_in || ctx2_in || ctx1_out || ctx2_out || T < 12;
  const bool mlp1 = c->small_gemm >= 2 && mlp_small_supported((int)R) && mix_general && c->dbg_mixer_stop == 0;
  const int nparts = kHidden4 / MLP_HS;
  if (mlp1) TRY(ensure(c, c->splitk, (size_t)nparts * R * kHidden * 4));
  float* res_prev = nullptr;      // mlp1: the residual stream the previous block's MLP read (that block's x_out)
  const float* bdn_prev = nullptr;
  // the online model (one frame, causal, <= 256 rows): every block in ONE persistent launch (mixer_online.hpp), the final
  // LayerNorm included
  bool persistent = false;
#ifndef TAPIR_HIPEMU
  persistent = c->small_gemm >= 3 && mlp1 && mixer_online_supported(N, T, c->cfg.use_causal_conv != 0) &&
               c->n_cus >= ONL_CLUSTERS * ONL_MEMBERS && (ctx1_out == nullptr) == (ctx2_out == nullptr);
  if (persistent) {
    TRY(ensure_zeroed(c, c->online_sync, (size_t)ONL_SYNC_WORDS * 4));   // (every launch leaves the counters zero: mixer_online.hpp)
    MixerOnlineArgs oa{};
    oa.x_in = (const float*)c->xa.p; oa.xn = c->xn.p; oa.part = (float*)c->splitk.p;
    oa.blocks = c->online_blocks; oa.lnF = c->lnF;
    oa.ctx1_in = ctx1_in; oa.ctx2_in = ctx2_in; oa.ctx1_out = ctx1_out; oa.ctx2_out = ctx2_out;
    oa.sync = (unsigned*)c->online_sync.p; oa.M = N; oa.nb = nb;
    oa.by_xcd = (c->online_form & 1) ? 0 : 1;
    oa.drop_member = (c->online_form & 4) ? 1 : 0;
    oa.spin_limit = oa.drop_member ? (1u << 14) : ONL_SPIN_LIMIT;
    oa.dbg_times = (long long*)c->dbg_times;
    ProfScope ps(c, TAPIR_PROF_MIX, s);
    if (sizeof(TA) == 2) launch_mixer_online<bf16_t>(oa, s, (c->online_form & 2) != 0);
    else launch_mixer_online<float>(oa, s, (c->online_form & 2) != 0);
  }
#endif
  for (int i = 0; i < nb && !persistent; ++i) {
    const BlockW& bw = c->blocks[i];
    MixArgs m{};
    m.x_in = (const float*)c->xa.p; m.x_out = (float*)c->xb.p; m.xn2 = c->xn.p;
    if (mlp1) {
      // residual buffers alternate (x_out may not alias what this launch reads: block 0 reads xa, later blocks read the
      // previous x_out as `presid`)
      m.x_out = (float*)((i & 1) ? c->xa.p : c->xb.p);
      if (i > 0) { m.parts = (const float*)c->splitk.p; m.nparts = nparts; m.pbias = bdn_prev; m.presid = res_prev; }
    }
    m.ln1 = bw.ln1; m.w1 = bw.w1; m.b1 = bw.b1; m.w2 = bw.w2; m.b2 = bw.b2; m.ln2 = bw.ln2;
    m.ctx1_in = ctx1_in ? ctx1_in + (size_t)i * N * 2 * kHidden : nullptr;
    m.ctx2_in = ctx2_in ? ctx2_in + (size_t)i * N * 2 * kHidden4 : nullptr;
    m.ctx1_out = ctx1_out ? ctx1_out + (size_t)i * N * 2 * kHidden : nullptr;
    m.ctx2_out = ctx2_out ? ctx2_out + (size_t)i * N * 2 * kHidden4 : nullptr;
    m.T = T; m.TC = TC; m.causal = c->cfg.use_causal_conv;
    { ProfScope ps(c, TAPIR_PROF_MIX, s);
      launch_mix<TA>(m, N, s); }
    if (stop_here()) return TAPIR_OK;
    if (mlp1) {
      MlpSmallArgs ma{c->xn.p, bw.Wup, bw.bup, bw.Wdn, (float*)c->splitk.p, (int)R};
      { ProfScope ps(c, TAPIR_PROF_GEMM_UP, s);
        if (sizeof(TA) == 2) launch_mlp_small<bf16_t>(ma, s); else launch_mlp_small<float>(ma, s); }
      res_prev = m.x_out; bdn_prev = bw.bdn;
      continue;
    }
    GemmArgs g1{};
    g1.A = c->xn.p; g1.lda = kHidden; g1.W = bw.Wup; g1.ldw = kHidden; g1.bias = bw.bup;
    g1.C = c->hid.p; g1.ldc = kHidden4; g1.M = (int)R; g1.N = kHidden4; g1.K = kHidden;
    { ProfScope ps(c, TAPIR_PROF_GEMM_UP, s, (c->small_gemm && gemm_small_supported<TA>(g1.M, g1.N, g1.K)) || gemm_splits<TA>(g1.M, g1.K) <= 1);
      TRY((mixer_gemm<TA, TA, EPI_BIAS_GELU>(c, g1, s))); }
    if (stop_here()) return TAPIR_OK;
    GemmArgs g2{};
    g2.A = c->hid.p; g2.lda = kHidden4; g2.W = bw.Wdn; g2.ldw = kHidden4; g2.bias = bw.bdn;
    g2.resid = (const float*)c->xb.p; g2.ldr = kHidden;
    g2.C = c->xa.p; g2.ldc = kHidden; g2.M = (int)R; g2.N = kHidden; g2.K = kHidden4;
    { ProfScope ps(c, TAPIR_PROF_GEMM_DOWN, s, (c->small_gemm && gemm_small_supported<TA>(g2.M, g2.N, g2.K)) || gemm_splits<TA>(g2.M, g2.K) <= 1);
      TRY((mixer_gemm<TA, float, EPI_BIAS_RESID>(c, g2, s))); }
    if (stop_here()) return TAPIR_OK;
  }
  LnArgs la{(const float*)c->xa.p, c->lnF, c->xn.p, R};
  if (mlp1 && nb > 0) { la.parts = (const float*)c->splitk.p; la.nparts = nparts; la.pbias = bdn_prev; la.presid = res_prev; }
  if (!persistent) hipLaunchKernelGGL((layernorm_kernel<TA>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, la);
  GemmArgs g{};
  g.A = c->xn.p; g.lda = kHidden; g.W = c->Wout; g.ldw = kHidden; g.bias = c->bout;
  g.C = c->res.p; g.ldc = kMixOut; g.M = (int)R; g.N = kMixOut; g.K = kHidden;
  // few rows inside refine_pips: the output Linear applies the state update itself (gemm.hpp EPI_BIAS_UPDATE: update_kernel's
  // operations in its order; the [R, 388] output never exists) -- one dependent launch less per refinement iteration
  if (upd != nullptr && upd_done != nullptr && c->fuse_update && c->small_gemm && gemm_small_supported<TA>(g.M, g.N, g.K) &&
      c->dbg_mixer_stop == 0) {
    g.upd = *upd;
    launch_gemm_small<TA, float, EPI_BIAS_UPDATE>(g, s);
    *upd_done = true;
    return TAPIR_OK;
  }
  TRY((mixer_gemm<TA, float, EPI_BIAS>(c, g, s)));
  return TAPIR_OK;
}

struct LevelGrids {   // operand-type pyramid of one feature level
  const void* grid[kMaxLevels];
  int h[kMaxLevels], w[kMaxLevels], C[kMaxLevels];
  const float* query[kMaxLevels];
  int n;
};

template <typename TA>
int make_patch_args(tapir_ctx* c, const LevelGrids& lg, int B, int Q, int T, const float* pos,
                    const float* occ, const float* expd, const float* feats, int orig_h, int orig_w,
                    PatchArgs* out) {
  const long R = (long)B * Q * T;
  TRY(ensure(c, c->mlp_in, (size_t)R * c->k0_pad * sizeof(TA)));
  PatchArgs pa{};
  pa.n_levels = lg.n;
  for (int l = 0; l < lg.n; ++l) {
    pa.lvl[l].grid = lg.grid[l]; pa.lvl[l].query = lg.query[l];
    pa.lvl[l].h = lg.h[l]; pa.lvl[l].w = lg.w[l]; pa.lvl[l].C = lg.C[l];
    pa.lvl[l].feat_off = (l == 0) ? 0 : kHiresDim;
  }
  pa.pos = pos; pa.occ = occ; pa.expd = expd; pa.feats = feats;
  pa.mlp_in = c->mlp_in.p; pa.ld = c->k0_pad;
  pa.B = B; pa.Q = Q; pa.T = T;
  pa.orig_h = (float)orig_h; pa.orig_w = (float)orig_w;
  *out = pa;
  return TAPIR_OK;
}

template <typename TA>
void launch_patch_args(tapir_ctx* c, const PatchArgs& pa, hipStream_t s) {
  ProfScope ps(c, TAPIR_PROF_PATCH, s);
  TAPIR_LAUNCH((patch_corr_kernel<TA, TA>), dim3(patch_corr_grid(pa.B, pa.Q, pa.T)), dim3(256), s, pa);
}

template <typename TA>
int launch_patch(tapir_ctx* c, const LevelGrids& lg, int B, int Q, int T, const float* pos,
                 const float* occ, const float* expd, const float* feats, int orig_h, int orig_w,
                 hipStream_t s) {
  PatchArgs pa{};
  TRY(make_patch_args<TA>(c, lg, B, Q, T, pos, occ, expd, feats, orig_h, orig_w, &pa));
  launch_patch_args<TA>(c, pa, s);
  return TAPIR_OK;
}

int check_pyramid(tapir_ctx* c, const LevelGrids& lg) {
  if (lg.n != 2 + c->cfg.pyramid_level) return fail(c, TAPIR_ERR_INVALID, "pyramid must have 2 + pyramid_level levels");
  if (lg.C[0] != kHiresDim) return fail(c, TAPIR_ERR_INVALID, "pyramid level 0 must have 128 channels");
  for (int l = 1; l < lg.n; ++l)
    if (lg.C[l] != kLowresDim) return fail(c, TAPIR_ERR_INVALID, "pyramid levels >= 1 must have 256 channels");
  return TAPIR_OK;
}

template <typename TA>
int do_build_cost_volume(tapir_ctx* c, const float* qfeat, const float* grid, int B, int Q, int T,
                         int h, int w, int C, float* volume, hipStream_t s) {
  const int hw = h * w;
  const void* qf_op = qfeat; const void* grid_op = grid;
  if (sizeof(TA) == 2) {
    TRY(cast_or_pool<TA>(c, qfeat, 1, 1, B * Q, C, 0, c->qf_cast, s));
    TRY(cast_or_pool<TA>(c, grid, (long)B * T, h, w, C, 0, c->grid_cast[1], s));
    c->cast_src[1] = nullptr; c->tiled_src = nullptr;
    qf_op = c->qf_cast.p; grid_op = c->grid_cast[1].p;
  }
  for (int b = 0; b < B; ++b) {
    const TA* qa = reinterpret_cast<const TA*>(qf_op) + (long)b * Q * C;
    const TA* ga = reinterpret_cast<const TA*>(grid_op) + (long)b * T * hw * C;
    TRY(cost_volume_gemm<TA>(c, qa, ga, Q, T, hw, C, volume + (long)b * Q * T * hw, s));
  }
  return TAPIR_OK;
}

struct StageArgs { const float* src; void* dst; long R; int cols, ld; };
template <typename TO>
__global__ void stage_rows_kernel(StageArgs a) {
  const long r = blockIdx.x;
  TO* d = reinterpret_cast<TO*>(a.dst) + r * a.ld;
  for (int k = threadIdx.x; k < a.ld; k += blockDim.x)
    Elem<TO>::st(d + k, k < a.cols ? a.src[r * a.cols + k] : 0.f);
}

template <typename TA>
int do_pips_mixer(tapir_ctx* c, const float* x, int N, int T, float* out, const float* c1i,
                  const float* c2i, float* c1o, float* c2o, hipStream_t s) {
  const long R = (long)N * T;
  TRY(ensure(c, c->mlp_in, (size_t)R * c->k0_pad * sizeof(TA)));
  StageArgs sa{x, c->mlp_in.p, R, c->in_dim, c->k0_pad};
  hipLaunchKernelGGL((stage_rows_kernel<TA>), dim3((unsigned)R), dim3(256), 0, s, sa);
  TRY(run_mixer<TA>(c, N, T, c1i, c2i, c1o, c2o, s));
  HIP_TRY(c, hipMemcpyAsync(out, c->res.p, (size_t)R * kMixOut * 4, hipMemcpyDeviceToDevice, s));
  return TAPIR_OK;
}

// Builds the operand-type pyramid for one feature level: casts (bf16 build) and
// average-pools as needed.  slot 0 = hires, 1 = lowres, 2 = pooled lowres.
template <typename TA>
int prepare_level(tapir_ctx* c, const float* hires, int hh, int hw_, const float* lowres, int lh,
                  int lw, const float* q_hires, const float* q_lowres, int B, int T,
                  LevelGrids* lg, hipStream_t s) {
  const long frames = (long)B * T;
  lg->n = 2 + c->cfg.pyramid_level;
  lg->h[0] = hh; lg->w[0] = hw_; lg->C[0] = kHiresDim; lg->query[0] = q_hires;
  lg->h[1] = lh; lg->w[1] = lw; lg->C[1] = kLowresDim; lg->query[1] = q_lowres;
  if (sizeof(TA) == 4) {
    lg->grid[0] = hires; lg->grid[1] = lowres;
  } else {
    const tapir_ctx::Staged* sh = find_staged(c, hires);
    const tapir_ctx::Staged* sl = find_staged(c, lowres);
    if (sh == nullptr && c->cast_src[0] != hires) {
      TRY(cast_or_pool<TA>(c, hires, frames, hh, hw_, kHiresDim, 0, c->grid_cast[0], s));
      c->cast_src[0] = hires;
    }
    if (sl == nullptr && c->cast_src[1] != lowres) {
      TRY(cast_or_pool<TA>(c, lowres, frames, lh, lw, kLowresDim, 0, c->grid_cast[1], s));
      c->cast_src[1] = lowres;
    }
    lg->grid[0] = sh != nullptr ? sh->op : c->grid_cast[0].p;
    lg->grid[1] = sl != nullptr ? sl->op : c->grid_cast[1].p;
  }
  if (c->cfg.pyramid_level >= 1) {
    if (c->cast_src[2] != lowres) {
      TRY(cast_or_pool<TA>(c, lowres, frames, lh, lw, kLowresDim, 1, c->pooled, s));
      c->cast_src[2] = lowres;
    }
    lg->grid[2] = c->pooled.p; lg->h[2] = lh / 2; lg->w[2] = lw / 2; lg->C[2] = kLowresDim;
    lg->query[2] = q_lowres;
  }
  return TAPIR_OK;
}

// one refinement iteration on state buffers (pos/occ/expd/feats are in/out)
template <typename TA>
int refine_iter(tapir_ctx* c, const LevelGrids& lg, int B, int Q, int T, float* pos, float* occ,
                float* expd, float* feats, bool first_of_level, bool last_of_level,
                const float* occ0, const float* expd0, int orig_h, int orig_w, int res_h, int res_w,
                float vx, float vy, float* out_tracks, float* out_occ, float* out_expd,
                const float* c1i, const float* c2i, float* c1o, float* c2o, hipStream_t s) {
  const long R = (long)B * Q * T;
  PatchArgs pa{};
  TRY(make_patch_args<TA>(c, lg, B, Q, T, pos, occ, expd, first_of_level ? nullptr : feats, orig_h, orig_w, &pa));
  UpdateArgs u{};
  u.pos = pos; u.occ = occ; u.expd = expd; u.feats = feats;
  u.q_hires = lg.query[0]; u.q_lowres = lg.query[1];
  u.out_tracks = out_tracks; u.out_occ = out_occ; u.out_expd = out_expd;
  u.occ0 = occ0; u.expd0 = expd0; u.R = R; u.T = T;
  u.sx = (float)orig_w / (float)res_w; u.sy = (float)orig_h / (float)res_h;
  u.vx = vx; u.vy = vy;
  u.first_of_level = first_of_level ? 1 : 0;
  u.last_of_level = last_of_level ? 1 : 0;
  bool upd_done = false;
  TRY(run_mixer<TA>(c, B * Q, T, c1i, c2i, c1o, c2o, s, &u, &upd_done, &pa));
  if (!upd_done) {   // separate-launch mixer: res [R,388] in the workspace
    u.res = (const float*)c->res.p;
    hipLaunchKernelGGL(update_kernel, dim3((unsigned)R), dim3(128), 0, s, u);
  }
  return TAPIR_OK;
}

template <typename TA>
int do_refine_pips(tapir_ctx* c, const tapir_pyramid* pyr, int B, int Q, int T, const float* pos,
                   const float* occ, const float* expd, const float* last_iter, int orig_h,
                   int orig_w, int res_h, int res_w, float* pos_out, float* occ_out,
                   float* expd_out, float* feats_out, const float* c1i, const float* c2i,
                   float* c1o, float* c2o, hipStream_t s) {
  const long R = (long)B * Q * T;
  LevelGrids lg{};
  lg.n = pyr->n_levels;
  if (lg.n < 2 || lg.n > kMaxLevels) return fail(c, TAPIR_ERR_INVALID, "bad pyramid size");
  for (int l = 0; l < lg.n; ++l) {
    lg.h[l] = pyr->h[l]; lg.w[l] = pyr->w[l]; lg.C[l] = pyr->C[l]; lg.query[l] = pyr->query[l];
    if (sizeof(TA) == 4) {
      lg.grid[l] = pyr->grid[l];
    } else {
      DevBuf& dst = (l == 2) ? c->pooled : c->grid_cast[l];
      TRY(cast_or_pool<TA>(c, pyr->grid[l], (long)B * T, pyr->h[l], pyr->w[l], pyr->C[l], 0, dst, s));
      c->cast_src[l] = nullptr; c->tiled_src = nullptr;
      lg.grid[l] = dst.p;
    }
  }
  TRY(check_pyramid(c, lg));
  // copy the inputs into the output buffers and update those in place
  if (pos_out != pos) HIP_TRY(c, hipMemcpyAsync(pos_out, pos, R * 2 * 4, hipMemcpyDeviceToDevice, s));
  if (occ_out != occ) HIP_TRY(c, hipMemcpyAsync(occ_out, occ, R * 4, hipMemcpyDeviceToDevice, s));
  if (expd_out != expd) HIP_TRY(c, hipMemcpyAsync(expd_out, expd, R * 4, hipMemcpyDeviceToDevice, s));
  if (last_iter != nullptr && feats_out != last_iter)
    HIP_TRY(c, hipMemcpyAsync(feats_out, last_iter, R * kFeatDim * 4, hipMemcpyDeviceToDevice, s));
  TRY(ensure(c, c->pos, (size_t)R * 2 * 4));   // scratch for the per-iteration outputs
  TRY(ensure(c, c->occ0, (size_t)R * 4));
  TRY(ensure(c, c->expd0, (size_t)R * 4));
  return refine_iter<TA>(c, lg, B, Q, T, pos_out, occ_out, expd_out, feats_out,
                         last_iter == nullptr, false, nullptr, nullptr, orig_h, orig_w, res_h,
                         res_w, 1.0f, 1.0f, (float*)c->pos.p, (float*)c->occ0.p,
                         (float*)c->expd0.p, c1i, c2i, c1o, c2o, s);
}

template <typename TA>
int do_estimate(tapir_ctx* c, const tapir_traj_args* a, hipStream_t s) {
  const int B = a->B, Q = a->Q, T = a->T;
  const long BQ = (long)B * Q, R = BQ * T;
  const int P = c->cfg.num_pips_iter;
  const int num_iters = P * (a->n_levels - 1);
  const int nb = c->cfg.num_mixer_blocks;
  const int ih = c->cfg.initial_h, iw = c->cfg.initial_w;
  for (int l = 0; l < kMaxLevels; ++l) c->cast_src[l] = nullptr;
  c->tiled_src = nullptr;
  TRY(ensure(c, c->pos, (size_t)R * 2 * 4));
  TRY(ensure(c, c->occ, (size_t)R * 4));
  TRY(ensure(c, c->expd, (size_t)R * 4));
  TRY(ensure(c, c->occ0, (size_t)R * 4));
  TRY(ensure(c, c->expd0, (size_t)R * 4));
  TRY(ensure(c, c->feats, (size_t)R * kFeatDim * 4));
  float* pos = (float*)c->pos.p; float* occ = (float*)c->occ.p; float* expd = (float*)c->expd.p;
  float* occ0 = (float*)c->occ0.p; float* expd0 = (float*)c->expd0.p;
  float* feats = (float*)c->feats.p;

  // query points: video -> initial_resolution coordinates (tapir_model.py:959-969)
  const float* qinit = nullptr;
  if (a->query_points != nullptr) {
    TRY(ensure(c, c->qpts, (size_t)BQ * 3 * 4));
    InitArgs ia{a->query_points, (float*)c->qpts.p, BQ, (float)ih / (float)a->video_h,
                (float)iw / (float)a->video_w};
    hipLaunchKernelGGL(scale_qpts_kernel, dim3((unsigned)((BQ + 255) / 256)), dim3(256), 0, s, ia);
    qinit = (const float*)c->qpts.p;
  }
  // The first refinement iteration's mixer finds its weight stream cold (the backbone's traffic has pushed it out: 185 us on that
  // launch); one pass over it (warm_stream_kernel, ~15 us + a dependent dispatch) sat between the cost volume and the first patch
  // correlation.  It depends on nothing: forked onto a side stream here, it runs under the cost volume and run_mixer joins it.
  c->warm_pending = false;
#ifndef TAPIR_HIPEMU
  if (c->warm_weights && c->warm_side && num_iters > 0 && a->ctx1_in == nullptr && a->ctx2_in == nullptr && a->ctx1_out == nullptr &&
      a->ctx2_out == nullptr && c->dbg_mixer_stop == 0) {
    bool fused = false, wide = false;
    if (c->side != nullptr && mixer_form<TA>(c, (int)BQ, T, false, &fused, &wide) == TAPIR_OK && (fused || wide)) {
      HIP_TRY(c, hipEventRecord(c->ev_fork, s));
      HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
      TRY(warm_stream(c, wide ? c->fused_wide_stream : c->fused_stream,
                      (size_t)FM_WAVES * (size_t)(wide ? c->fused_wide_fpw : c->fused_fpw) * 1024, c->side));
      HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
      c->warm_pending = true;
    }
  }
#endif
  const float vx = (float)a->video_w / (float)iw, vy = (float)a->video_h / (float)ih;
  Iter0Args i0{pos, occ, expd, occ0, expd0, a->tracks, a->occlusion, a->expected_dist, R, vx, vy};
  bool i0_done = false;   // (the row-streamed cost-volume kernel writes iter0's copies itself: one dependent launch less)
  TRY(cost_volume_stage<TA>(c, a->q_lowres[0], a->lowres[0], qinit, B, Q, T, a->lowres_h[0],
                            a->lowres_w[0], pos, occ, expd, s, false, &i0, &i0_done));
  if (!i0_done) hipLaunchKernelGGL(iter0_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, i0);

  for (int i = 0; i < num_iters; ++i) {
    const int lvl = i / P + 1;
    LevelGrids lg{};
    TRY(prepare_level<TA>(c, a->hires[lvl], a->hires_h[lvl], a->hires_w[lvl], a->lowres[lvl],
                          a->lowres_h[lvl], a->lowres_w[lvl], a->q_hires[lvl], a->q_lowres[lvl], B,
                          T, &lg, s));
    const size_t s1 = (size_t)nb * BQ * 2 * kHidden, s2 = (size_t)nb * BQ * 2 * kHidden4;
    TRY(refine_iter<TA>(c, lg, B, Q, T, pos, occ, expd, feats, i % P == 0, (i + 1) % P == 0, occ0,
                        expd0, ih, iw, a->res_h[lvl], a->res_w[lvl], vx, vy,
                        a->tracks + (size_t)(i + 1) * R * 2, a->occlusion + (size_t)(i + 1) * R,
                        a->expected_dist + (size_t)(i + 1) * R,
                        a->ctx1_in ? a->ctx1_in + i * s1 : nullptr,
                        a->ctx2_in ? a->ctx2_in + i * s2 : nullptr,
                        a->ctx1_out ? a->ctx1_out + i * s1 : nullptr,
                        a->ctx2_out ? a->ctx2_out + i * s2 : nullptr, s));
  }
  if (c->warm_pending) {   // (nobody joined the side stream: a capture must not end with it open)
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
    c->warm_pending = false;
  }
  return TAPIR_OK;
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
#define DISPATCH(ctx, fn, ...) \
  ((ctx)->cfg.dtype == TAPIR_BF16 ? fn<bf16_t>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

#define REQUIRE_FINALIZED(ctx)                                                     \
  do {                                                                             \
    if (!(ctx)) return TAPIR_ERR_INVALID;                                          \
    if (!(ctx)->finalized) return fail((ctx), TAPIR_ERR_WEIGHTS, "weights not finalized"); \
    HIP_TRY((ctx), hipSetDevice((ctx)->device));                                   \
  } while (0)
#define REQUIRE_READY(ctx)                                                         \
  do {                                                                             \
    REQUIRE_FINALIZED(ctx);                                                        \
    if (!(ctx)->tapir_ready) return fail((ctx), TAPIR_ERR_WEIGHTS, "TAPIR weights not set (TAP-Net head only)"); \
  } while (0)

extern "C" {

const char* tapir_version(void) { return "tapir_hip 0.1 (gfx950)"; }

int tapir_create(tapir_ctx** out, const tapir_cfg* cfg, int device) {
  if (!out || !cfg) return TAPIR_ERR_INVALID;
  if (cfg->pyramid_level < 0 || cfg->pyramid_level > 1) return TAPIR_ERR_UNSUPPORTED;
  if (cfg->num_pips_iter < 1 || cfg->num_mixer_blocks < 1) return TAPIR_ERR_INVALID;
  if (cfg->dtype != TAPIR_F32 && cfg->dtype != TAPIR_BF16) return TAPIR_ERR_INVALID;
  if (cfg->initial_h % 8 || cfg->initial_w % 8 || cfg->initial_h <= 0 || cfg->initial_w <= 0)
    return TAPIR_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return TAPIR_ERR_HIP;
  tapir_ctx* c = new tapir_ctx();
  c->cfg = *cfg;
  c->device = device;
  c->in_dim = kMixOut + kPatch * (2 + cfg->pyramid_level);
  // K of the input Linear, zero padded: whole GEMM k-steps, and rows of a multiple of 256 bytes for
  // the LDS image of the fused mixer kernel (128 bf16 / 64 f32 elements)
  const int kq = cfg->dtype == TAPIR_BF16 ? 128 : 64;
  c->k0_pad = (c->in_dim + kq - 1) / kq * kq;
  // (same-box A/B of builds from outside the process: environment switches)
  if (const char* e = getenv("TAPIR_FUSE_UPDATE")) c->fuse_update = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_ONLINE_FORM")) c->online_form = atoi(e) & 3;
  if (const char* e = getenv("TAPIR_SMALL_GEMM")) c->small_gemm = std::min(3, std::max(0, atoi(e)));
#ifndef TAPIR_HIPEMU
  { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) c->n_cus = n; }
  // the side stream of do_estimate's early weight warm-up (created here: never inside somebody's stream capture)
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
    c->side = nullptr;
#endif
  if (const char* e = getenv("TAPIR_CV_FORM")) c->cv_form = atoi(e);
  if (const char* e = getenv("TAPIR_FUSE_PATCH")) c->fuse_patch = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_WARM_WEIGHTS")) c->warm_weights = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_WARM_SIDE")) c->warm_side = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_FUSE_ITER0")) c->fuse_iter0 = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_CV_TILED")) c->cv_tiled = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_CV_STREAM_OUT")) c->cv_stream_out = atoi(e) != 0;
  if (const char* e = getenv("TAPIR_FUSED_MIN_TRACKS")) c->fused_min_tracks = std::max(1, atoi(e));
  if (const char* e = getenv("TAPIR_CONV_FLAT")) c->conv_flat = atoi(e) < 0 ? -1 : (atoi(e) != 0);
  if (const char* e = getenv("TAPIR_CONV_FLAT_MIN_SLABS")) c->conv_flat_min_slabs = std::max(1, atoi(e));
  if (const char* e = getenv("TAPIR_XCONV_NT")) { const int v = atoi(e); c->xconv_nt = (v == XC_NT || v == XC_NT_WIDE) ? v : 0; }
  *out = c;
  return TAPIR_OK;
}

void tapir_destroy(tapir_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (void* p : c->owned) (void)hipFree(p);
  for (void* p : c->conv_owned) (void)hipFree(p);
  DevBuf* bufs[] = {&c->cv, &c->mlp_in, &c->xa, &c->xb, &c->xn, &c->hid, &c->res, &c->pos, &c->occ,
                    &c->expd, &c->occ0, &c->expd0, &c->feats, &c->qpts, &c->qf_cast,
                    &c->grid_cast[0], &c->grid_cast[1], &c->grid_cast[2], &c->pooled, &c->splitk,
                    &c->cyc_pts, &c->cyc_feat, &c->cyc_map, &c->cyc_inv, &c->grid_tiled, &c->warm_sink, &c->online_sync};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (int k = 0; k < TAPIR_PROF_KINDS; ++k)
    for (auto& ev : c->prof_ev[k]) c->prof_free.push_back(ev);
  for (auto& ev : c->prof_free) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
}

const char* tapir_last_error(const tapir_ctx* c) { return c ? c->err.c_str() : "null context"; }

int tapir_set_weight(tapir_ctx* c, const char* name, const float* data, const int64_t* shape,
                     int ndim) {
  if (!c || !name || !data || !shape || ndim < 1 || ndim > 4) return TAPIR_ERR_INVALID;
  const std::string n(name);
  if (n.rfind("resnet_torch.", 0) == 0 || n.rfind("extra_convs.", 0) == 0) return TAPIR_OK;
  HostTensor t;
  size_t cnt = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); cnt *= (size_t)shape[i]; }
  t.data.assign(data, data + cnt);
  c->host_w[n] = std::move(t);
  c->finalized = false;
  return TAPIR_OK;
}

static int finalize_tapir(tapir_ctx* c);

int tapir_finalize_weights(tapir_ctx* c) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  for (void* p : c->owned) (void)hipFree(p);
  c->owned.clear();
  c->blocks.clear();
  c->tapnet_ready = false; c->tapir_ready = false;
  c->fused_stream = nullptr; c->fused_blocks.clear(); c->fused_fpw = 0;
  c->fused_wide_stream = nullptr; c->fused_wide_fpw = 0;
  const bool has_tapnet = c->host_w.count("tapnet_cost_volume_track_mods.hid1.weight") != 0;
  bool has_tapir = !has_tapnet;   // a context without TAP-Net head weights must be a complete TAPIR
  for (const auto& kv : c->host_w)
    if (kv.first.rfind("torch_", 0) == 0) has_tapir = true;
  if (has_tapnet) {
    // TAP-Net's num_heads = input channels of hid1 (tapnet_model.py:145-152): 1, 2 or 4
    const auto& s1 = c->host_w["tapnet_cost_volume_track_mods.hid1.weight"].shape;
    const int heads = s1.size() == 4 ? (int)s1[1] : 0;
    if (heads != 1 && heads != 2 && heads != 4)
      return fail(c, TAPIR_ERR_UNSUPPORTED, "TAP-Net head: num_heads (input channels of hid1) must be 1, 2 or 4");
    TRY(upload_cv_head(c, "tapnet_cost_volume_track_mods.", 1, &c->tapnet_cvw, heads));
    c->tapnet_heads = heads;
    c->tapnet_ready = true;
  }
  if (has_tapir) {
    TRY(finalize_tapir(c));
    c->tapir_ready = true;
  }
  c->host_w.clear();
  c->finalized = true;
  return TAPIR_OK;
}

static int finalize_tapir(tapir_ctx* c) {
  const HostTensor* t;
  TRY(upload_cv_head(c, "torch_cost_volume_track_mods.", 2, &c->cvw));
  const std::string mx = "torch_pips_mixer.";
  TRY(get_w(c, mx + "linear.weight", {kHidden, c->in_dim}, &t));
  TRY(upload_matrix(c, t->data.data(), kHidden, c->in_dim, c->k0_pad, &c->W0));
  TRY(get_w(c, mx + "linear.bias", {kHidden}, &t)); TRY(upload_f32(c, t->data.data(), kHidden, &c->b0));
  TRY(get_w(c, mx + "layer_norm.weight", {kHidden}, &t)); TRY(upload_f32(c, t->data.data(), kHidden, &c->lnF));
  TRY(get_w(c, mx + "linear_1.weight", {kMixOut, kHidden}, &t));
  TRY(upload_matrix(c, t->data.data(), kMixOut, kHidden, kHidden, &c->Wout));
  TRY(get_w(c, mx + "linear_1.bias", {kMixOut}, &t)); TRY(upload_f32(c, t->data.data(), kMixOut, &c->bout));
  for (int i = 0; i < c->cfg.num_mixer_blocks; ++i) {
    const std::string p = mx + "blocks." + std::to_string(i) + ".";
    BlockW b{};
    TRY(get_w(c, p + "layer_norm.weight", {kHidden}, &t)); TRY(upload_f32(c, t->data.data(), kHidden, &b.ln1));
    TRY(get_w(c, p + "mlp1_up.weight", {kHidden4, 1, 3}, &t)); TRY(upload_f32(c, t->data.data(), kHidden4 * 3, &b.w1));
    TRY(get_w(c, p + "mlp1_up.bias", {kHidden4}, &t)); TRY(upload_f32(c, t->data.data(), kHidden4, &b.b1));
    TRY(get_w(c, p + "mlp1_up_1.weight", {kHidden4, 1, 3}, &t)); TRY(upload_f32(c, t->data.data(), kHidden4 * 3, &b.w2));
    TRY(get_w(c, p + "mlp1_up_1.bias", {kHidden4}, &t)); TRY(upload_f32(c, t->data.data(), kHidden4, &b.b2));
    TRY(get_w(c, p + "layer_norm_1.weight", {kHidden}, &t)); TRY(upload_f32(c, t->data.data(), kHidden, &b.ln2));
    TRY(get_w(c, p + "conv_channels_mixer.mlp2_up.weight", {kHidden4, kHidden}, &t));
    TRY(upload_matrix(c, t->data.data(), kHidden4, kHidden, kHidden, &b.Wup));
    TRY(get_w(c, p + "conv_channels_mixer.mlp2_up.bias", {kHidden4}, &t)); TRY(upload_f32(c, t->data.data(), kHidden4, &b.bup));
    TRY(get_w(c, p + "conv_channels_mixer.mlp2_down.weight", {kHidden, kHidden4}, &t));
    TRY(upload_matrix(c, t->data.data(), kHidden, kHidden4, kHidden4, &b.Wdn));
    TRY(get_w(c, p + "conv_channels_mixer.mlp2_down.bias", {kHidden}, &t)); TRY(upload_f32(c, t->data.data(), kHidden, &b.bdn));
    c->blocks.push_back(b);
  }
  {
    std::vector<OnlineBlockW> tab;
    for (const BlockW& b : c->blocks) tab.push_back(OnlineBlockW{b.ln1, b.w1, b.b1, b.w2, b.b2, b.ln2, b.Wup, b.bup, b.Wdn, b.bdn});
    float* d = nullptr;
    static_assert(sizeof(OnlineBlockW) % 4 == 0, "table uploaded as words");
    TRY(upload_f32(c, reinterpret_cast<const float*>(tab.data()), tab.size() * sizeof(OnlineBlockW) / 4, &d));
    c->online_blocks = reinterpret_cast<OnlineBlockW*>(d);
  }
  if (c->cfg.num_mixer_blocks <= FM_MAX_BLOCKS) {
    if (c->cfg.dtype == TAPIR_BF16) {
      TRY(build_fused_weights<bf16_t>(c)); TRY(build_fused_wide_weights(c));
    }
    else TRY(build_fused_weights<float>(c));
  }
  return TAPIR_OK;
}

int tapir_reserve(tapir_ctx* c, int B, int Q, int T, int mh, int mw) {
  REQUIRE_READY(c);
  if (B < 1 || Q < 1 || T < 1 || mh < 1 || mw < 1) return fail(c, TAPIR_ERR_INVALID, "bad sizes");
  const size_t es = esize(c->cfg.dtype);
  const size_t R = (size_t)B * Q * T, BQ = (size_t)B * Q, frames = (size_t)B * T;
  TRY(ensure(c, c->mlp_in, R * c->k0_pad * es));
  TRY(ensure(c, c->xa, R * kHidden * 4)); TRY(ensure(c, c->xb, R * kHidden * 4));
  TRY(ensure(c, c->xn, R * kHidden * es)); TRY(ensure(c, c->hid, R * kHidden4 * es));
  TRY(ensure(c, c->res, R * kMixOut * 4));
  TRY(ensure(c, c->pos, R * 8)); TRY(ensure(c, c->occ, R * 4)); TRY(ensure(c, c->expd, R * 4));
  TRY(ensure(c, c->occ0, R * 4)); TRY(ensure(c, c->expd0, R * 4));
  TRY(ensure(c, c->feats, R * kFeatDim * 4)); TRY(ensure(c, c->qpts, BQ * 12));
  if (R <= 512) TRY(ensure(c, c->splitk, (size_t)8 * R * kHidden4 * 4));
  long qc = (256L << 20) / ((long)T * mh * mw * 4);
  qc = std::max<long>(1, std::min<long>(qc, Q));
  TRY(ensure(c, c->cv, (size_t)qc * T * mh * mw * 4));
  if (c->cfg.dtype == TAPIR_BF16) {
    TRY(ensure(c, c->qf_cast, BQ * kLowresDim * es));
    TRY(ensure(c, c->grid_cast[0], frames * 4 * mh * mw * kHiresDim * es));
    TRY(ensure(c, c->grid_cast[1], frames * mh * mw * kLowresDim * es));
    TRY(ensure_zeroed(c, c->grid_tiled, tiled_bytes((long)frames, (int)mh, (int)mw)));
  }
  if (c->cfg.pyramid_level >= 1) TRY(ensure(c, c->pooled, frames * (mh / 2) * (mw / 2) * kLowresDim * es));
  TRY(ensure_zeroed(c, c->online_sync, (size_t)ONL_SYNC_WORDS * 4));   // (the online mixer's counters: allocated and zeroed before anything is captured)
  TRY(ensure(c, c->warm_sink, 4));   // (warm_stream_kernel's sink: the first fused-mixer iteration of a level must not allocate while pinned)
  return TAPIR_OK;
}

int tapir_pin_workspaces(tapir_ctx* c, int on) {
  if (!c) return TAPIR_ERR_INVALID;
  if (on) ++c->pinned;
  else if (c->pinned > 0) --c->pinned;
  return TAPIR_OK;
}

int tapir_build_cost_volume(tapir_ctx* c, const float* qfeat, const float* grid, int B, int Q, int T,
                            int h, int w, int C, float* volume, void* stream) {
  REQUIRE_READY(c);
  if (!qfeat || !grid || !volume || B < 1 || Q < 1 || T < 1 || h < 1 || w < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (C % 64 != 0) return fail(c, TAPIR_ERR_UNSUPPORTED, "channels must be a multiple of 64");
  if (((long)T * h * w) % 4 != 0) return fail(c, TAPIR_ERR_UNSUPPORTED, "T*h*w must be a multiple of 4");
  return DISPATCH(c, do_build_cost_volume, c, qfeat, grid, B, Q, T, h, w, C, volume, (hipStream_t)stream);
}

int tapir_tracks_from_cost_volume(tapir_ctx* c, const float* qfeat, const float* grid,
                                  const float* query_points, int B, int Q, int T, int h, int w,
                                  float* points, float* occlusion, float* expected_dist,
                                  void* stream) {
  REQUIRE_READY(c);
  if (!qfeat || !grid || !points || !occlusion || !expected_dist || B < 1 || Q < 1 || T < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  c->cast_src[1] = nullptr; c->tiled_src = nullptr;   // (the cast cache is only valid within one API call: the caller may have rewritten the grid)
  return DISPATCH(c, cost_volume_stage, c, qfeat, grid, query_points, B, Q, T, h, w, points,
                  occlusion, expected_dist, (hipStream_t)stream);
}

int tapir_tapnet_tracks_from_cost_volume(tapir_ctx* c, const float* qfeat, const float* grid,
                                         const float* query_points, int B, int Q, int T, int h, int w,
                                         float* points, float* occlusion, void* stream) {
  REQUIRE_FINALIZED(c);
  if (!c->tapnet_ready) return fail(c, TAPIR_ERR_WEIGHTS, "TAP-Net head weights (tapnet_cost_volume_track_mods.*) not set");
  if (!qfeat || !grid || !points || !occlusion || B < 1 || Q < 1 || T < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (c->cv_mode == 1 || !(cv_fused_supported(h, w) || (c->tapnet_heads == 1 && c->cv_mode == 0 && cv_rows_supported(h, w))))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "TAP-Net head: grids of up to 32 x 32 cells (64 x 64 with one head), fused kernels only");
  c->cast_src[1] = nullptr; c->tiled_src = nullptr;
  return DISPATCH(c, cost_volume_stage, c, qfeat, grid, query_points, B, Q, T, h, w, points, occlusion,
                  nullptr, (hipStream_t)stream, true);
}

int tapir_cycle_consistency_tracks(tapir_ctx* c, const float* query_feats, const float* feature_grid,
                                    const float* query_points, int B, int Q, int T, int h, int w, int img_h, int img_w,
                                    float softmax_temperature, float dist_threshold, float* tracks, float* occlusion,
                                    float* inverse_tracks, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!query_feats || !feature_grid || !query_points || !tracks || !occlusion || B < 1 || Q < 1 || T < 1 || h < 1 ||
      w < 1 || img_h < 1 || img_w < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  return DISPATCH(c, do_cycle_consistency, c, query_feats, feature_grid, query_points, B, Q, T, h, w, img_h, img_w,
                  softmax_temperature, dist_threshold, tracks, occlusion, inverse_tracks, (hipStream_t)stream);
}

int tapir_get_query_features(tapir_ctx* c, const float* grid, const float* query_points, int B,
                             int Q, int T, int h, int w, int C, int video_h, int video_w,
                             float* out, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!grid || !query_points || !out || B < 1 || Q < 1 || T < 1 || h < 1 || w < 1 || C < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  SampleArgs a{grid, query_points, out, B, Q, T, h, w, C, (float)video_h, (float)video_w};
  hipLaunchKernelGGL(query_feature_kernel, dim3((unsigned)((long)B * Q)), dim3(128), 0,
                     (hipStream_t)stream, a);
  return TAPIR_OK;
}

int tapir_pips_mixer(tapir_ctx* c, const float* x, int N, int T, float* out, const float* ctx1_in,
                     const float* ctx2_in, float* ctx1_out, float* ctx2_out, void* stream) {
  REQUIRE_READY(c);
  if (!x || !out || N < 1 || T < 1) return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if ((ctx1_in || ctx2_in || ctx1_out || ctx2_out) && !c->cfg.use_causal_conv)
    return fail(c, TAPIR_ERR_INVALID, "causal context needs use_causal_conv");
  if ((ctx1_in == nullptr) != (ctx2_in == nullptr) || (ctx1_out == nullptr) != (ctx2_out == nullptr))
    return fail(c, TAPIR_ERR_INVALID, "causal context pointers must come in pairs");
  return DISPATCH(c, do_pips_mixer, c, x, N, T, out, ctx1_in, ctx2_in, ctx1_out, ctx2_out,
                  (hipStream_t)stream);
}

int tapir_refine_pips(tapir_ctx* c, const tapir_pyramid* pyr, int B, int Q, int T, const float* pos,
                      const float* occ, const float* expd, const float* last_iter, int orig_h,
                      int orig_w, int resized_h, int resized_w, float* pos_out, float* occ_out,
                      float* expd_out, float* feats_out, const float* ctx1_in, const float* ctx2_in,
                      float* ctx1_out, float* ctx2_out, void* stream) {
  REQUIRE_READY(c);
  if (!pyr || !pos || !occ || !expd || !pos_out || !occ_out || !expd_out || !feats_out || B < 1 ||
      Q < 1 || T < 1 || resized_h < 1 || resized_w < 1)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if ((ctx1_in || ctx1_out) && !c->cfg.use_causal_conv)
    return fail(c, TAPIR_ERR_INVALID, "causal context needs use_causal_conv");
  return DISPATCH(c, do_refine_pips, c, pyr, B, Q, T, pos, occ, expd, last_iter, orig_h, orig_w,
                  resized_h, resized_w, pos_out, occ_out, expd_out, feats_out, ctx1_in, ctx2_in,
                  ctx1_out, ctx2_out, (hipStream_t)stream);
}

int tapir_profile_enable(tapir_ctx* c, int on) {
  if (!c) return TAPIR_ERR_INVALID;
  c->prof = on < 0 ? ~0u : (unsigned)on;
  return TAPIR_OK;
}

int tapir_profile_stride(tapir_ctx* c, int stride) {
  if (!c || stride < 1) return TAPIR_ERR_INVALID;
  c->prof_stride = stride;
  for (int k = 0; k < TAPIR_PROF_KINDS; ++k) c->prof_count[k] = 0;
  return TAPIR_OK;
}

int tapir_profile_read(tapir_ctx* c, int kind, double* total_ms, int64_t* launches) {
  if (!c || kind < 0 || kind >= TAPIR_PROF_KINDS || !total_ms || !launches) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  double tot = 0.0;
  for (auto& ev : c->prof_ev[kind]) {
    HIP_TRY(c, hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, ev.a, ev.b));
    tot += ms;
    c->prof_free.push_back(ev);
  }
  *total_ms = tot;
  *launches = (int64_t)c->prof_ev[kind].size();
  c->prof_ev[kind].clear();
  return TAPIR_OK;
}

// ---- backbone glue kernels (between the PyTorch-ROCm convolutions)
static bool norm_channels_ok(const tapir_ctx* c, int C, int max_group) {
  const int ept = c->cfg.dtype == TAPIR_BF16 ? 8 : 4;
  if (C < ept || C % ept) return false;
  const int G = C / ept;
  return G <= max_group && (G & (G - 1)) == 0;
}

int tapir_inorm_stats(tapir_ctx* c, const void* a, const void* b, void* sum_out, float* part, int N,
                      int HW, int C, int slabs, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!a || !part || (b && !sum_out) || N < 1 || HW < 1 || slabs < 1 || slabs > HW)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!norm_channels_ok(c, C, NORM_THREADS)) return fail(c, TAPIR_ERR_UNSUPPORTED, "channel count");
  NormStatsArgs na{a, b, sum_out, part, HW, C, slabs};
  if (c->cfg.dtype == TAPIR_BF16)
    hipLaunchKernelGGL((inorm_stats_kernel<bf16_t>), dim3(slabs, N), dim3(NORM_THREADS), 0, (hipStream_t)stream, na);
  else
    hipLaunchKernelGGL((inorm_stats_kernel<float>), dim3(slabs, N), dim3(NORM_THREADS), 0, (hipStream_t)stream, na);
  return TAPIR_OK;
}

int tapir_inorm_relu(tapir_ctx* c, const void* x, const float* part, const float* gamma,
                     const float* beta, float* ss, void* y, void* y_sub, int N, int H, int W, int C,
                     int slabs, int per_s, int out_h, int out_w, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!x || !part || !gamma || !beta || !ss || !y || N < 1 || H < 1 || W < 1 || slabs < 1 || per_s < 0 || out_h < H ||
      out_w < W || (y_sub && ((H | W) & 1)))
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!norm_channels_ok(c, C, NORM_THREADS)) return fail(c, TAPIR_ERR_UNSUPPORTED, "channel count");
  NormFinalizeArgs nf{part, gamma, beta, ss, H * W, C, slabs, per_s, 0};
  hipLaunchKernelGGL(inorm_finalize_kernel, dim3(N, (nf.C + 63) / 64), dim3(NORM_THREADS), 0, (hipStream_t)stream, nf);
  NormApplyArgs na{};
  na.x = x; na.ss = ss; na.y = y; na.y_sub = y_sub;
  na.H = H; na.W = W; na.C = C; na.oh = out_h; na.ow = out_w;
  na.pix_slabs = std::max(1, std::min(H * W / 64, (2048 + N - 1) / N));
  if (c->cfg.dtype == TAPIR_BF16)
    hipLaunchKernelGGL((inorm_relu_kernel<bf16_t>), dim3(na.pix_slabs, N), dim3(NORM_THREADS), 0, (hipStream_t)stream, na);
  else
    hipLaunchKernelGGL((inorm_relu_kernel<float>), dim3(na.pix_slabs, N), dim3(NORM_THREADS), 0, (hipStream_t)stream, na);
  return TAPIR_OK;
}

int tapir_l2_normalize(tapir_ctx* c, const void* x, float* out, long pixels, int C, void* stream) {
  return tapir_l2_normalize_staged(c, x, out, nullptr, nullptr, pixels, C, 0, stream);
}

int tapir_set_staged_grid(tapir_ctx* c, const float* grid_f32, const void* grid_op, const void* grid_tiled) {
  if (!c || !grid_f32 || !grid_op) return TAPIR_ERR_INVALID;
  if (c->cfg.dtype != TAPIR_BF16) return fail(c, TAPIR_ERR_UNSUPPORTED, "staged grids: bf16 build only");
  for (auto& st : c->staged) if (st.f32 == grid_f32) { st.op = grid_op; st.tiled = grid_tiled; return TAPIR_OK; }
  c->staged.push_back({grid_f32, grid_op, grid_tiled});
  return TAPIR_OK;
}

int tapir_clear_staged_grids(tapir_ctx* c) {
  if (!c) return TAPIR_ERR_INVALID;
  c->staged.clear();
  return TAPIR_OK;
}

int tapir_l2_normalize_staged(tapir_ctx* c, const void* x, float* out, void* out_op, void* out_tiled, long pixels, int C,
                              int cells_per_frame, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!x || !out || pixels < 1) return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!norm_channels_ok(c, C, 64)) return fail(c, TAPIR_ERR_UNSUPPORTED, "channel count");
  if ((out_op || out_tiled) && c->cfg.dtype != TAPIR_BF16) return fail(c, TAPIR_ERR_UNSUPPORTED, "operand-type copies: bf16 build only");
  if (out_tiled && (!out_op || C != kLowresDim || cells_per_frame < 1 || pixels % cells_per_frame != 0))
    return fail(c, TAPIR_ERR_INVALID, "tile-order copy: 256 channels, whole frames, together with the row-major copy");
  const int ept = c->cfg.dtype == TAPIR_BF16 ? 8 : 4;
  const int PP = NORM_THREADS / (C / ept);
  L2Args la{x, out, pixels, C, out_op, out_tiled, cells_per_frame};
  const unsigned grid = (unsigned)std::min<long>((pixels + PP - 1) / PP, 8192);
  ProfScope ps(c, TAPIR_PROF_L2NORM, (hipStream_t)stream);
  if (c->cfg.dtype == TAPIR_BF16)
    TAPIR_LAUNCH((l2norm_kernel<bf16_t>), dim3(grid), dim3(NORM_THREADS), (hipStream_t)stream, la);
  else
    TAPIR_LAUNCH((l2norm_kernel<float>), dim3(grid), dim3(NORM_THREADS), (hipStream_t)stream, la);
  return TAPIR_OK;
}

// ---- backbone convolutions (conv_fused.hpp): resnet.py:185-257, conv_0 / conv_1 / proj_conv of BlockV2
int tapir_conv_plan(tapir_ctx* c, int H, int W, int cin, int cout, int ks, int stride, int* rows, int* tiles) {
  if (!c || !rows || !tiles) return TAPIR_ERR_INVALID;
  const int es = c->cfg.dtype == TAPIR_BF16 ? 2 : 4;
  int nt = 0;
  if (c->conv_small && es == 2 && conv_small_plan(H, W, cin, cout, ks, stride, rows, tiles, &nt)) return TAPIR_OK;   // (few-frame form)
  if (!conv3_plan(H, W, cin, cout, ks, stride, es, rows, tiles)) return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused: shape");
  return TAPIR_OK;
}

int tapir_conv_set_small(tapir_ctx* c, int on) {
  if (!c || on < 0 || on > 1) return TAPIR_ERR_INVALID;
  c->conv_small = on;
  return TAPIR_OK;
}

// Fragments of one convolution weight [cout, cin, ks, ks] for channel group cg, appended at q in the order the kernel's
// k loop consumes them: for tap, k-step, row tile r: fragment row m = l & 15 holds output channel
// cg*64 + 16 (m >> 2) + 4 r + (m & 3) -- so that lane group g = m >> 2 of the accumulator layout
// (rows 4 g + e of tile r) owns the 16 consecutive channels 16 g + 4 r + e of its pixel --
// input channels KSTEP kstep + EPC (l >> 4) + j (bf16: 32 per k-step, 8 per lane; f32: 16 / 4)
static uint8_t* conv_pack_frags(uint8_t* q, const float* w, int cg, int cin, int ks, bool bf) {
  const int kstep_n = bf ? 32 : 16, epc = bf ? 8 : 4, taps = ks * ks;
  for (int tap = 0; tap < taps; ++tap)
    for (int kstep = 0; kstep < cin / kstep_n; ++kstep)
      for (int r = 0; r < 4; ++r, q += 1024)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < epc; ++j) {
            const int m = l & 15;
            const int co = cg * 64 + 16 * (m >> 2) + 4 * r + (m & 3), ci = kstep_n * kstep + epc * (l >> 4) + j;
            const float v = w[((size_t)co * cin + ci) * taps + tap];
            if (bf) ((uint16_t*)q)[l * epc + j] = host_f2bf(v);
            else ((float*)q)[l * epc + j] = v;
          }
  return q;
}

int tapir_conv_pack(tapir_ctx* c, const float* w, int cout, int cin, int ks, void** wstream) {
  if (!c || !w || !wstream) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!conv3_supported(cin, cout, ks, 1) && !conv3_supported(cin, cout, ks, 2))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused: channel counts / kernel size");
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  const long fpc = conv3_frags_per_cg(cin, ks, bf ? 32 : 16);
  std::vector<uint8_t> host((size_t)(cout / 64) * fpc * 1024, 0);
  for (int cg = 0; cg < cout / 64; ++cg) conv_pack_frags(host.data() + (size_t)cg * fpc * 1024, w, cg, cin, ks, bf);
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->conv_owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  *wstream = d;
  return TAPIR_OK;
}

// conv_0 (3x3) + proj_conv (1x1) of a block as ONE stream per channel group: the projection's k-steps first, padded with
// zero fragments to whole ring turns (conv3_proj_ksteps), then the 3x3 fragments (conv_fused.hpp, DUAL)
int tapir_conv_pack_dual(tapir_ctx* c, const float* w3, const float* w1, int cout, int cin, int stride, void** wstream) {
  if (!c || !w3 || !w1 || !wstream) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!conv3_dual_supported(cin, cout, stride))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused dual: channel counts / stride");
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  const int kstep_n = bf ? 32 : 16;
  const long fpc = conv3_dual_frags_per_cg(cin, kstep_n);
  std::vector<uint8_t> host((size_t)(cout / 64) * fpc * 1024, 0);
  for (int cg = 0; cg < cout / 64; ++cg) {
    uint8_t* q = host.data() + (size_t)cg * fpc * 1024;
    conv_pack_frags(q, w1, cg, cin, 1, bf);
    conv_pack_frags(q + (size_t)conv3_proj_ksteps(cin, kstep_n) * 4 * 1024, w3, cg, cin, 3, bf);
  }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->conv_owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  *wstream = d;
  return TAPIR_OK;
}

int tapir_conv_free(tapir_ctx* c, void* wstream) {
  if (!c) return TAPIR_ERR_INVALID;
  if (!wstream) return TAPIR_OK;
  auto it = std::find(c->conv_owned.begin(), c->conv_owned.end(), wstream);
  if (it == c->conv_owned.end()) return fail(c, TAPIR_ERR_INVALID, "tapir_conv_free: not a pack of this context");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipFree(wstream));
  c->conv_owned.erase(it);
  c->xconv_cch.erase(wstream);
  return TAPIR_OK;
}

int tapir_conv_fused(tapir_ctx* c, const void* x, const float* part_in, int slabs_in, int per_s_in,
                     const float* gamma, const float* beta, float* ss, const void* wstream,
                     const void* shortcut, void* y, float* part_out, int N, int H, int W, int cin,
                     int cout, int ks, int stride, void* stream) {
  return tapir_conv_fused_nn(c, x, part_in, slabs_in, per_s_in, gamma, beta, ss, wstream, shortcut, y, part_out, N, H, W,
                             cin, cout, ks, stride, nullptr, stream);
}

static bool next_norm_ok(const tapir_next_norm* nx, const float* part_out) {
  return nx == nullptr || (nx->gamma && nx->beta && nx->ss && nx->arrive && part_out);
}

static int conv_fused_impl(tapir_ctx* c, const void* x, const float* part_in, int slabs_in, int per_s_in,
                           const float* gamma, const float* beta, float* ss, const void* wstream,
                           const void* shortcut, void* y, void* y_proj, float* part_out, int N, int H, int W, int cin,
                           int cout, int ks, int stride, const tapir_next_norm* next, void* stream);

int tapir_conv_fused_nn(tapir_ctx* c, const void* x, const float* part_in, int slabs_in, int per_s_in,
                        const float* gamma, const float* beta, float* ss, const void* wstream,
                        const void* shortcut, void* y, float* part_out, int N, int H, int W, int cin,
                        int cout, int ks, int stride, const tapir_next_norm* next, void* stream) {
  return conv_fused_impl(c, x, part_in, slabs_in, per_s_in, gamma, beta, ss, wstream, shortcut, y, nullptr, part_out, N, H, W,
                         cin, cout, ks, stride, next, stream);
}

int tapir_conv_fused_dual_nn(tapir_ctx* c, const void* x, const float* part_in, int slabs_in, int per_s_in,
                             const float* gamma, const float* beta, float* ss, const void* wstream_dual, void* y,
                             void* y_proj, float* part_out, int N, int H, int W, int cin, int cout, int stride,
                             const tapir_next_norm* next, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  if (!y_proj) return fail(c, TAPIR_ERR_INVALID, "conv_fused dual: y_proj");
  if (!conv3_dual_supported(cin, cout, stride)) return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused dual: channel counts / stride");
  return conv_fused_impl(c, x, part_in, slabs_in, per_s_in, gamma, beta, ss, wstream_dual, nullptr, y, y_proj, part_out, N, H, W,
                         cin, cout, 3, stride, next, stream);
}

// true: this launch takes the flat tiling (conv_flat.hpp) under the context's current mode
static bool conv_use_flat(tapir_ctx* c, int N, int H, int W, int cin, int cout, int ks, int stride) {
  if (c->cfg.dtype != TAPIR_BF16 || c->conv_flat == 0 || !conv_flat_supported(H, W, cin, cout, ks, stride, 2)) return false;
  int rows = 0, tiles = 0, waves = 0;
  conv3_plan(H, W, cin, cout, ks, stride, 2, &rows, &tiles, &waves);
  return c->conv_flat == 1 || N * tiles >= c->conv_flat_min_slabs;
}

int tapir_conv_flat_plan(tapir_ctx* c, int N, int H, int W, int cin, int cout, int ks, int stride, int* workgroups) {
  if (!c || !workgroups || N < 1) return TAPIR_ERR_INVALID;
  if (!conv_use_flat(c, N, H, W, cin, cout, ks, stride)) return TAPIR_ERR_UNSUPPORTED;
  int rows = 0, tiles = 0, waves = 0;
  conv3_plan(H, W, cin, cout, ks, stride, 2, &rows, &tiles, &waves);
  *workgroups = (N * tiles + CVL_SLABS - 1) / CVL_SLABS;
  return TAPIR_OK;
}

static int conv_fused_impl(tapir_ctx* c, const void* x, const float* part_in, int slabs_in, int per_s_in,
                           const float* gamma, const float* beta, float* ss, const void* wstream,
                           const void* shortcut, void* y, void* y_proj, float* part_out, int N, int H, int W, int cin,
                           int cout, int ks, int stride, const tapir_next_norm* next, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  if (!next_norm_ok(next, part_out)) return fail(c, TAPIR_ERR_INVALID, "next norm: gamma, beta, ss, arrive and part_out are all needed");
  HIP_TRY(c, hipSetDevice(c->device));
  // part_in NULL: ss already holds the merged (a, b) pairs of this input and norm (a previous call: conv_0 and
  // proj_conv of a block read the same normalised tensor) -- no second inorm_finalize launch
  if (!x || !ss || !wstream || !y || N < 1 || (part_in && (!gamma || !beta || slabs_in < 1 || per_s_in < 0)))
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (shortcut && !(ks == 3 && stride == 1)) return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused: shortcut on a 3x3 stride-1 convolution only");
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  int rows = 0, tiles = 0, waves = 0, nt_small = 0;
  // few-frame clips (tapir_conv_set_small): the K-split form of conv_small.hpp, with ITS tile geometry (tapir_conv_plan
  // answers accordingly while the mode is on)
  const bool small = bf && c->conv_small && conv_small_plan(H, W, cin, cout, ks, stride, &rows, &tiles, &nt_small);
  if (small && y_proj) return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused: the dual launch has no few-frame form (launch conv_0 and proj_conv separately)");
  if (!small && !conv3_plan(H, W, cin, cout, ks, stride, bf ? 2 : 4, &rows, &tiles, &waves))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "conv_fused: shape");
  // few-frame form: the consuming workgroups merge the summaries themselves (conv_small.hpp cvs_merge_pairs) -- no launch
  const bool merge_in_kernel = small && part_in != nullptr && slabs_in <= (256 / cin) * CVS_MERGE_MAXS;
  if (part_in != nullptr && !merge_in_kernel) {
    NormFinalizeArgs nf{part_in, gamma, beta, ss, H * W, cin, slabs_in, per_s_in, bf ? 8 : 4};
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3(N, (nf.C + 63) / 64), dim3(NORM_THREADS), 0, (hipStream_t)stream, nf);
  }
  Conv3Args ca{};
  ca.x = x; ca.ss = ss; ca.wstream = (const uint4*)wstream;
  ca.frags_per_cg = y_proj ? conv3_dual_frags_per_cg(cin, bf ? 32 : 16) : conv3_frags_per_cg(cin, ks, bf ? 32 : 16);
  ca.shortcut = shortcut; ca.y = y; ca.y_proj = y_proj; ca.part = part_out;
  ca.N = N; ca.H = H; ca.W = W; ca.Ho = (H + stride - 1) / stride; ca.Wo = (W + stride - 1) / stride;
  ca.pad_y = conv3_pad_lo(H, ks, stride); ca.pad_x = conv3_pad_lo(W, ks, stride);
  ca.TH = rows; ca.tiles = tiles; ca.waves = waves;
  ca.dbg_times = (long long*)c->dbg_times;
  if (next != nullptr) ca.fin = FinArgs{next->gamma, next->beta, next->ss, next->arrive, bf ? 8 : 4};
  if (merge_in_kernel) { ca.part_in = part_in; ca.gamma_in = gamma; ca.beta_in = beta; ca.slabs_in = slabs_in; ca.per_s_in = per_s_in; }
  {
    const int kind = (ks == 3 && stride == 1) ? (cin == 64 ? TAPIR_PROF_CONV3_C64 : cin == 128 ? TAPIR_PROF_CONV3_C128 : TAPIR_PROF_CONV3_C256)
                                              : TAPIR_PROF_CONV_OTHER;
    ProfScope ps(c, kind, (hipStream_t)stream);
    // the flat tiling of the whole launch (conv_flat.hpp) where it applies: same bits, a third of the weight traffic
    const bool flat = !small && conv_use_flat(c, N, H, W, cin, cout, ks, stride);
    if (small) launch_conv_small(ca, cin, cout, ks, stride, nt_small, (hipStream_t)stream);
    else if (flat) launch_conv_flat(ca, (hipStream_t)stream);
    else if (bf) launch_conv_fused<bf16_t>(ca, cin, cout, ks, stride, (hipStream_t)stream);
    else launch_conv_fused<float>(ca, cin, cout, ks, stride, (hipStream_t)stream);
  }
  HIP_TRY(c, hipGetLastError());
  return TAPIR_OK;
}

// ---- BootsTAPIR's ExtraConvs (tapir_model.py:159-186; extra_convs.hpp)
int tapir_layernorm_affine(tapir_ctx* c, const void* x, const float* gamma, const float* beta, void* y,
                           long pixels, int C, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!x || !gamma || !beta || !y || pixels < 1) return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!norm_channels_ok(c, C, 64)) return fail(c, TAPIR_ERR_UNSUPPORTED, "channel count");
  const int ept = c->cfg.dtype == TAPIR_BF16 ? 8 : 4;
  const int PP = NORM_THREADS / (C / ept);
  LnAffineArgs la{x, gamma, beta, y, pixels, C};
  const unsigned grid = (unsigned)std::min<long>((pixels + PP - 1) / PP, 8192);
  if (c->cfg.dtype == TAPIR_BF16)
    hipLaunchKernelGGL((ln_affine_kernel<bf16_t>), dim3(grid), dim3(NORM_THREADS), 0, (hipStream_t)stream, la);
  else
    hipLaunchKernelGGL((ln_affine_kernel<float>), dim3(grid), dim3(NORM_THREADS), 0, (hipStream_t)stream, la);
  return TAPIR_OK;
}

int tapir_xconv_plan_frames(tapir_ctx* c, int frames, int H, int W, int cin, int cout, int* rows, int* tiles, int* cch, int* nt) {
  if (!c || !rows || !tiles || !cch || !nt || frames < 0) return TAPIR_ERR_INVALID;
  if (!xconv_plan(H, W, cin, cout, c->cfg.dtype == TAPIR_BF16 ? 2 : 4, rows, tiles, cch, nt, c->xconv_nt, frames))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "xconv: shape");
  return TAPIR_OK;
}

int tapir_xconv_plan(tapir_ctx* c, int H, int W, int cin, int cout, int* rows, int* tiles, int* cch) {
  int nt = 0;
  if (!c || !rows || !tiles || !cch) return TAPIR_ERR_INVALID;
  return tapir_xconv_plan_frames(c, 0, H, W, cin, cout, rows, tiles, cch, &nt);
}

int tapir_xconv_pack(tapir_ctx* c, const float* w, int cout, int cin, int cch, void** wstream) {
  if (!c || !w || !wstream) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  const int kstep_n = bf ? 32 : 16, epc = bf ? 8 : 4;
  if (cout % 256 || cin % 256 || cch < 64 || cch > 256 || (cch & (cch - 1)) || cin % cch)
    return fail(c, TAPIR_ERR_UNSUPPORTED, "xconv: channel counts");
  // stream of output-channel group cg (64 channels), in the order a wave multiplies: for input-channel
  // chunk, tap, k-step of the chunk, row tile r -- fragment row m = l & 15 holds output channel
  // cg*64 + 16 (m >> 2) + 4 r + (m & 3) (the row permutation of the epilogue, as tapir_conv_pack),
  // input channels chunk*cch + KSTEP kstep + EPC (l >> 4) + j.  w = torch OIHW [cout, cin, 3, 3].
  const long fpc = xconv_frags_per_cg(cin, kstep_n);
  std::vector<uint8_t> host((size_t)(cout / 64) * fpc * 1024, 0);
  for (int cg = 0; cg < cout / 64; ++cg) {
    uint8_t* q = host.data() + (size_t)cg * fpc * 1024;
    for (int chn = 0; chn < cin / cch; ++chn)
      for (int tap = 0; tap < 9; ++tap)
        for (int kstep = 0; kstep < cch / kstep_n; ++kstep)
          for (int r = 0; r < 4; ++r, q += 1024)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < epc; ++j) {
                const int m = l & 15;
                const int co = cg * 64 + 16 * (m >> 2) + 4 * r + (m & 3);
                const int ci = chn * cch + kstep_n * kstep + epc * (l >> 4) + j;
                const float v = w[((size_t)co * cin + ci) * 9 + tap];
                if (bf) ((uint16_t*)q)[l * epc + j] = host_f2bf(v);
                else ((float*)q)[l * epc + j] = v;
              }
  }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->conv_owned.push_back(d);
  c->xconv_cch[d] = cch;
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  *wstream = d;
  return TAPIR_OK;
}

int tapir_xconv(tapir_ctx* c, const void* x, const void* wstream, const float* bias, const void* skip, void* y,
                int N, int H, int W, int cin, int cout, int gelu, void* stream) {
  return tapir_xconv_nt(c, x, wstream, bias, skip, y, N, H, W, cin, cout, gelu, 0, stream);
}

int tapir_xconv_nt(tapir_ctx* c, const void* x, const void* wstream, const float* bias, const void* skip, void* y,
                   int N, int H, int W, int cin, int cout, int gelu, int form, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  if (form != 0 && form != XC_NT && form != XC_NT_WIDE) return fail(c, TAPIR_ERR_INVALID, "xconv: form is 0, 4 or 8 pixel tiles per wave");
  HIP_TRY(c, hipSetDevice(c->device));
  if (!x || !wstream || !bias || !y || N < 1) return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (gelu && skip) return fail(c, TAPIR_ERR_UNSUPPORTED, "xconv: gelu and skip together");
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  int rows = 0, tiles = 0, cch = 0, nt = 0;
  if (!xconv_plan(H, W, cin, cout, bf ? 2 : 4, &rows, &tiles, &cch, &nt, form ? form : c->xconv_nt)) return fail(c, TAPIR_ERR_UNSUPPORTED, "xconv: shape");
  // few-frame clips (tapir_conv_set_small): both convolutions of a block in the K-split form of conv_small.hpp -- a workgroup
  // per (row tile, 16 output channels): one frame gives xconv_kernel 64 / 16 workgroups that each stream 1.18 / 4.7 MB of cold
  // weights through a 12-fragment ring, 28 / 93 us per launch (profiles/r06_online_timeline_v1.txt).  Needs packs for
  // 256-channel chunks (the fragment order is then the block convolutions': [chunk] tap, k-step, row tile)
  if (bf && c->conv_small && cch == 256 && ((gelu && !skip && cin == 256 && cout == 1024) || (!gelu && skip && cin == 1024 && cout == 256))) {
    auto it = c->xconv_cch.find(wstream);
    int srows = 0, stiles = 0, snt = 0;
    if (it != c->xconv_cch.end() && it->second == 256 && W <= 128 && H >= 1) {
      // conv_small_plan's geometry for a 3x3 / stride-1 map (it checks the channel counts of the BLOCK convolutions: restated)
      int th = W <= 64 ? 64 / W : 1;
      if (th > H) th = H;
      while (th >= 1 && (long)(th + 2) * (W + 2) * 256 * 2 > CVS_LDS_BYTES) --th;
      if (th >= 1) { srows = th; stiles = (H + th - 1) / th; snt = th * W <= 64 ? 4 : 8; }
    }
    if (srows > 0) {
      Conv3Args ca{};
      ca.x = x; ca.ss = bias; ca.wstream = (const uint4*)wstream; ca.frags_per_cg = xconv_frags_per_cg(cin, 32);
      ca.y = y; ca.N = N; ca.H = H; ca.W = W; ca.Ho = H; ca.Wo = W; ca.pad_y = 1; ca.pad_x = 1; ca.TH = srows; ca.tiles = stiles;
      ca.shortcut = skip; ca.cin_total = cin;
      if (launch_xconv_small(ca, cin, cout, snt, (hipStream_t)stream)) {
        HIP_TRY(c, hipGetLastError());
        return TAPIR_OK;
      }
    }
  }
  {
    // the stream is packed for ONE chunk width ([chunk][tap][k-step][row tile]): a pack made for another map
    // width would be multiplied in the wrong order
    auto it = c->xconv_cch.find(wstream);
    if (it == c->xconv_cch.end()) return fail(c, TAPIR_ERR_INVALID, "xconv: wstream is not a tapir_xconv_pack of this context");
    if (it->second != cch)
      return fail(c, TAPIR_ERR_INVALID, "xconv: the pack was built for chunks of " + std::to_string(it->second) +
                                           " input channels, this map needs " + std::to_string(cch) + " (tapir_xconv_plan)");
  }
  XConvArgs xa{};
  xa.x = x; xa.wstream = (const uint4*)wstream; xa.frags_per_cg = xconv_frags_per_cg(cin, bf ? 32 : 16);
  xa.bias = bias; xa.skip = skip; xa.y = y;
  xa.N = N; xa.H = H; xa.W = W; xa.cin = cin; xa.cout = cout;
  xa.TH = rows; xa.tiles = tiles; xa.passes = cout / 256; xa.nt = nt;
  const bool ok = bf ? launch_xconv<bf16_t>(xa, cch, gelu != 0, (hipStream_t)stream)
                     : launch_xconv<float>(xa, cch, gelu != 0, (hipStream_t)stream);
  if (!ok) return fail(c, TAPIR_ERR_UNSUPPORTED, "xconv: no kernel for this chunking");
  HIP_TRY(c, hipGetLastError());
  return TAPIR_OK;
}

// ---- the 7x7 / stride-2 stem (resnet.py:356-364)
int tapir_stem_plan(tapir_ctx* c, int H, int W, int* rows, int* tiles) {
  if (!c || !rows || !tiles) return TAPIR_ERR_INVALID;
  if (!stem_plan(H, W, c->cfg.dtype == TAPIR_BF16 ? 2 : 4, rows, tiles)) return fail(c, TAPIR_ERR_UNSUPPORTED, "stem_conv: shape");
  return TAPIR_OK;
}

int tapir_stem_pack(tapir_ctx* c, const float* w, void** wstream) {
  if (!c || !w || !wstream) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  // w [64, 3, 7, 7] (OIHW).  Kernel row ky, k-step sub, row tile r: fragment row m holds output channel
  // 16 (m >> 2) + 4 r + (m & 3) (the permutation of the epilogue); k = KSTEP sub + EPC (l >> 4) + j = 3 kx + ci
  // for k < 21, zero beyond
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  const int kstep_n = bf ? 32 : 16, epc = bf ? 8 : 4, ksub = 32 / kstep_n;
  const long nfr = 7 * ksub * 4 + 8;
  std::vector<uint8_t> host((size_t)nfr * 1024, 0);
  uint8_t* q = host.data();
  for (int ky = 0; ky < 7; ++ky)
    for (int sub = 0; sub < ksub; ++sub)
      for (int r = 0; r < 4; ++r, q += 1024)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < epc; ++j) {
            const int m = l & 15, co = 16 * (m >> 2) + 4 * r + (m & 3), k = kstep_n * sub + epc * (l >> 4) + j;
            if (k >= 21) continue;
            const float v = w[(((size_t)co * 3 + k % 3) * 7 + ky) * 7 + k / 3];
            if (bf) ((uint16_t*)q)[l * epc + j] = host_f2bf(v);
            else ((float*)q)[l * epc + j] = v;
          }
  void* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, host.size()));
  c->conv_owned.push_back(d);
  HIP_TRY(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  *wstream = d;
  return TAPIR_OK;
}

int tapir_stem_conv(tapir_ctx* c, const float* x, const void* wstream, void* y, float* part_out, int N, int H,
                    int W, void* stream) {
  return tapir_stem_conv_nn(c, x, wstream, y, part_out, N, H, W, nullptr, stream);
}

int tapir_stem_conv_nn(tapir_ctx* c, const float* x, const void* wstream, void* y, float* part_out, int N, int H,
                       int W, const tapir_next_norm* next, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!x || !wstream || !y || N < 1) return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!next_norm_ok(next, part_out)) return fail(c, TAPIR_ERR_INVALID, "next norm: gamma, beta, ss, arrive and part_out are all needed");
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  int rows = 0, tiles = 0;
  if (!stem_plan(H, W, bf ? 2 : 4, &rows, &tiles)) return fail(c, TAPIR_ERR_UNSUPPORTED, "stem_conv: shape");
  StemArgs sa{};
  sa.x = x; sa.wstream = (const uint4*)wstream; sa.y = y; sa.part = part_out;
  sa.N = N; sa.H = H; sa.W = W; sa.Ho = (H + 1) / 2; sa.Wo = (W + 1) / 2;
  sa.pad_y = conv3_pad_lo(H, 7, 2); sa.pad_x = conv3_pad_lo(W, 7, 2);
  sa.TH = rows; sa.tiles = tiles;
  if (next != nullptr) sa.fin = FinArgs{next->gamma, next->beta, next->ss, next->arrive, bf ? 8 : 4};
  {
    ProfScope ps(c, TAPIR_PROF_STEM, (hipStream_t)stream);
    if (bf) launch_stem_conv<bf16_t>(sa, (hipStream_t)stream);
    else launch_stem_conv<float>(sa, (hipStream_t)stream);
  }
  HIP_TRY(c, hipGetLastError());
  return TAPIR_OK;
}

int tapir_debug_gemm(tapir_ctx* c, const void* A, long lda, const void* W, long ldw,
                     const float* bias, const float* resid, long ldr, void* C, long ldc, int M,
                     int N, int K, int epi, int tile, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  const int kstep = c->cfg.dtype == TAPIR_BF16 ? 64 : 32;
  if (!A || !W || !C || M < 1 || N < 4 || K < kstep || K % kstep || N % 4 || ldc % 4 || epi < 0 ||
      epi > 2 || tile < 0 || (tile & 0xff) >= GEMM_TILE_COUNT || (epi == 2 && !resid))
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.resid = resid; g.ldr = ldr;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.dbg_times = (long long*)c->dbg_times;
  g.stream_out = getenv("TAPIR_DEBUG_GEMM_NT") != nullptr;   // (tools/kbench.py: the cost-volume GEMM's non-temporal stores, A/B)
  hipStream_t s = (hipStream_t)stream;
  const bool bf = c->cfg.dtype == TAPIR_BF16;
  const int mg = (tile >> 8) & 0xfff;   // test hook: cap the persistent grid (several tiles per workgroup)
  const bool traced = (tile >> 20) & 1;  // bf16 build: per-k-step cycle trace into the trace buffer
  tile &= 0xff;
  if (!gemm_tile_available(tile))
    return fail(c, TAPIR_ERR_UNSUPPORTED, "GEMM tile not in this build (experiment tiles need -DTAPIR_EXPERIMENTS)");
  if (traced) {
#ifdef TAPIR_EXPERIMENTS
    if (!bf || epi == 0 || !c->dbg_times) return fail(c, TAPIR_ERR_INVALID, "traced GEMM: bf16, epi 1|2, trace buffer set");
    if (epi == 1) launch_gemm_traced<bf16_t, bf16_t, EPI_BIAS_GELU>(g, s, tile, mg);
    else launch_gemm_traced<bf16_t, float, EPI_BIAS_RESID>(g, s, tile, mg);
    return TAPIR_OK;
#else
    return fail(c, TAPIR_ERR_UNSUPPORTED, "traced GEMM needs a -DTAPIR_EXPERIMENTS build");
#endif
  }
  if (epi == 0) { if (bf) launch_gemm<bf16_t, float, EPI_BIAS>(g, s, tile, mg); else launch_gemm<float, float, EPI_BIAS>(g, s, tile, mg); }
  else if (epi == 1) { if (bf) launch_gemm<bf16_t, bf16_t, EPI_BIAS_GELU>(g, s, tile, mg); else launch_gemm<float, float, EPI_BIAS_GELU>(g, s, tile, mg); }
  else { if (bf) launch_gemm<bf16_t, float, EPI_BIAS_RESID>(g, s, tile, mg); else launch_gemm<float, float, EPI_BIAS_RESID>(g, s, tile, mg); }
  return TAPIR_OK;
}

int tapir_debug_set_mixer_mode(tapir_ctx* c, int mode) {
  bool ok = mode >= 0 && mode <= 3;
#ifdef TAPIR_EXPERIMENTS
  ok = ok || mode == 4;   // timing-only pair simulation of the wide kernel
#endif
  if (!c || !ok) return TAPIR_ERR_INVALID;
  c->mixer_mode = mode;
  return TAPIR_OK;
}

int tapir_debug_set_conv_flat(tapir_ctx* c, int mode) {
  if (!c || mode < -1 || mode > 1) return TAPIR_ERR_INVALID;
  c->conv_flat = mode;
  return TAPIR_OK;
}

int tapir_debug_mixer_stop(tapir_ctx* c, int stages) {
  if (!c || stages < 0) return TAPIR_ERR_INVALID;
  c->dbg_mixer_stop = stages;
  return TAPIR_OK;
}

int tapir_debug_workspace(tapir_ctx* c, int which, void** p, unsigned long long* bytes) {
  if (!c || !p || !bytes) return TAPIR_ERR_INVALID;
  DevBuf* b[] = {&c->mlp_in, &c->xa, &c->xb, &c->xn, &c->hid, &c->res, &c->splitk, &c->online_sync};
  if (which < 0 || which >= (int)(sizeof(b) / sizeof(b[0]))) return TAPIR_ERR_INVALID;
  *p = b[which]->p; *bytes = b[which]->cap;
  return TAPIR_OK;
}

// Fills the LDS of every CU with `pattern` (workgroups of the full 160 KiB, several per CU one after the other): what a
// kernel reads from LDS before writing it is then the pattern, not whatever the previous workgroup on that CU left.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
  __shared__ unsigned s_all[160 * 1024 / 4];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) s_all[i] = pattern;
  __syncthreads();
  if (s_all[(threadIdx.x * 97 + blockIdx.x) % (160 * 1024 / 4)] != pattern) *sink = 1;   // keeps the stores alive
}
int tapir_debug_poison_lds(tapir_ctx* c, unsigned pattern, void* stream) {
  if (!c) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(ensure(c, c->warm_sink, 4));
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, pattern, (unsigned*)c->warm_sink.p);
  return TAPIR_OK;
}

int tapir_online_sync_error(tapir_ctx* c, unsigned* word) {
  if (!c || !word) return TAPIR_ERR_INVALID;
  *word = 0;
  if (c->online_sync.p == nullptr) return TAPIR_OK;      // the persistent launch never ran
  HIP_TRY(c, hipMemcpy(word, (const unsigned*)c->online_sync.p + 16 * ONL_CLUSTERS, 4, hipMemcpyDeviceToHost));
  if (*word != 0) HIP_TRY(c, hipMemset(c->online_sync.p, 0, c->online_sync.cap));   // a launch that gave up leaves its counters behind
  return TAPIR_OK;
}

int tapir_debug_set_gemm_mode(tapir_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 3 + 4 * 7) return TAPIR_ERR_INVALID;   // (tests) + 4 x the persistent launch's form bits
  c->small_gemm = mode & 3;
  c->online_form = mode >> 2;
  return TAPIR_OK;
}

int tapir_debug_set_patch_mode(tapir_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 1) return TAPIR_ERR_INVALID;
  c->fuse_patch = mode != 0;
  return TAPIR_OK;
}

int tapir_debug_set_update_mode(tapir_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 1) return TAPIR_ERR_INVALID;
  c->fuse_update = mode;
  return TAPIR_OK;
}

// tools/kbench.py --what contraction: the einsum('bnc,bthwc->tbnhw') phase of the row-streamed cost-volume kernel alone
// (cost maps into LDS, nothing else); scratch: B*T*ceil(Q/8) floats.  The cast of the grid is done once per grid pointer.
int tapir_debug_contraction(tapir_ctx* c, const float* qfeat, const float* grid, int B, int Q, int T, int h, int w,
                            float* scratch, void* stream) {
  if (!c || !qfeat || !grid || !scratch) return TAPIR_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!cv_rows_supported(h, w)) return fail(c, TAPIR_ERR_UNSUPPORTED, "rows of up to 64 cells");
  return DISPATCH(c, do_debug_contraction, c, qfeat, grid, B, Q, T, h, w, scratch, (hipStream_t)stream);
}

int tapir_debug_set_cv_mode(tapir_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return TAPIR_ERR_INVALID;
  c->cv_mode = mode;
  return TAPIR_OK;
}

int tapir_debug_set_trace(tapir_ctx* c, void* device_buffer) {
  if (!c) return TAPIR_ERR_INVALID;
  c->dbg_times = device_buffer;
  return TAPIR_OK;
}

int tapir_debug_mix(tapir_ctx* c, int block, const float* x_in, float* x_out, void* xn, int N,
                    int T, int tc, void* stream) {
  REQUIRE_READY(c);
  if (block < 0 || block >= (int)c->blocks.size() || !x_in || !x_out || !xn || N < 1 || T < 1 ||
      x_in == x_out || tc < 0)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  const BlockW& bw = c->blocks[block];
  MixArgs m{};
  m.x_in = x_in; m.x_out = x_out; m.xn2 = xn;
  m.ln1 = bw.ln1; m.w1 = bw.w1; m.b1 = bw.b1; m.w2 = bw.w2; m.b2 = bw.b2; m.ln2 = bw.ln2;
  m.T = T; m.TC = pick_time_chunk(N, T); m.causal = c->cfg.use_causal_conv;
  m.dbg_times = (long long*)c->dbg_times;
  if (c->cfg.dtype == TAPIR_BF16) launch_mix<bf16_t>(m, N, (hipStream_t)stream, tc);
  else launch_mix<float>(m, N, (hipStream_t)stream, tc);
  return TAPIR_OK;
}

int tapir_estimate_trajectories(tapir_ctx* c, const tapir_traj_args* a, void* stream) {
  REQUIRE_READY(c);
  if (!a || a->B < 1 || a->Q < 1 || a->T < 1 || a->n_levels < 2 || a->n_levels > TAPIR_MAX_LEVELS)
    return fail(c, TAPIR_ERR_INVALID, "bad argument");
  if (!a->tracks || !a->occlusion || !a->expected_dist) return fail(c, TAPIR_ERR_INVALID, "null output");
  for (int l = 0; l < a->n_levels; ++l) {
    if (!a->lowres[l] || !a->hires[l] || !a->q_lowres[l] || !a->q_hires[l])
      return fail(c, TAPIR_ERR_INVALID, "null feature level");
    if (a->res_h[l] < 1 || a->res_w[l] < 1) return fail(c, TAPIR_ERR_INVALID, "bad resolution");
  }
  if ((a->ctx1_in || a->ctx1_out) && !c->cfg.use_causal_conv)
    return fail(c, TAPIR_ERR_INVALID, "causal context needs use_causal_conv");
  if ((a->ctx1_in == nullptr) != (a->ctx2_in == nullptr) ||
      (a->ctx1_out == nullptr) != (a->ctx2_out == nullptr))
    return fail(c, TAPIR_ERR_INVALID, "causal context pointers must come in pairs");
  return DISPATCH(c, do_estimate, c, a, (hipStream_t)stream);
}

}  // extern "C"
