// The convolutions of the ResNet blocks of the feature backbone (tapnet/models/resnet.py:185-257:
// conv_0 / conv_1 / proj_conv of BlockV2; 3x3 or 1x1, stride 1 or 2, XLA "SAME" padding) as ONE kernel
// with everything the block does around a convolution folded in:
//   operand load : relu(instance_norm(x)) (resnet.py:241-242, 248-249) -- the raw tensor is read,
//                  normalised with the per-(image, channel) pair (a, b) of inorm_finalize_kernel,
//                  rectified and rounded to bf16 on its way into LDS; the zero padding of the
//                  convolution is applied AFTER that (out-of-image pixels are literal zeros);
//   epilogue     : + shortcut (resnet.py:256), rounding to bf16, and the per-(image, channel)
//                  (mean, M2) summary of the stored tensor for the NEXT InstanceNorm.
// The normalised activations, the pre-add convolution result and the separate statistics pass never
// exist in HBM (they were five of the seven memory passes of a block).
//
// Implicit GEMM on the matrix cores, the layout of the track-resident mixer (mixer_fused.hpp):
//   A = weights: per-wave packed stream of 1-KiB fragments (16 output channels x 32 input channels of
//       one tap), global -> register ring, never through LDS;
//   B = pixels : the input tile [in_rows x in_cols pixels][C_in] bf16 in LDS (halo / stride included),
//       16-byte chunks XOR-swizzled by the pixel index; the taps are constant offsets into it.
// Workgroup = 4 or 8 waves = (C_out / 64) output-channel groups x pixel groups; a wave owns 64 output
// channels x NT * 16 pixels (4 x NT accumulator fragments); the tile of a workgroup is `rows` full
// rows of the OUTPUT image (rows * W_out <= pixel groups * NT * 16).  Element type bf16 (16x16x32 MFMA) or
// f32 (the parity build: exact-f32 16x16x4 MFMA, 16 input channels per k-step, tiles half as tall).
#pragma once
#include "backbone.hpp"
#include "gemm.hpp"

namespace tapir {

constexpr int CV3_NT = 4;                    // pixel tiles (16 pixels) per wave
// A fragments in flight per wave: 12 (3 k-steps) for the 3x3 kernels, 8 for the 1x1 (2, 4 or 8 k-steps in all).
// (Deeper rings for C >= 128 -- 16 or 24 fragments, 236-256 VGPRs -- measured equal on the GPU with four streams in
// flight, profiles/r04_ab_backbone_knobs.txt: the weight stream's latency is covered by the other resident workgroup.)
constexpr int cv3_ring(int ks) { return ks == 3 ? 12 : 8; }
// Two workgroup sizes: 4 waves with a 72-KiB tile -- two workgroups per CU, so that the VALU- and
// memory-bound phases of one (staging, epilogue) run under the matrix phase of the other -- and
// 8 waves with a 144-KiB tile for the maps whose rows are too long for that.
constexpr int cv3_lds_bytes(int waves) { return waves * 18 * 1024; }

// element type T = bf16 (the bench build) or float (the parity build: exact-f32 MFMA, same structure)
template <typename T> struct CvT;
template <> struct CvT<bf16_t> { static constexpr int EPC = 8, KSTEP = 32; };
template <> struct CvT<float> { static constexpr int EPC = 4, KSTEP = 16; };

// The NEXT InstanceNorm's (a, b) pairs, merged inside this launch: every workgroup publishes its tile summary
// write-through and takes a ticket from the image's arrival counter; the workgroup that draws the last ticket merges
// all tile summaries of the image exactly as inorm_finalize_kernel does (same operations in the same order:
// bit-identical pairs) and resets the counter.  No waiting anywhere: nothing can hang.  ss == null: off (the consumer
// launches inorm_finalize_kernel on `part`).
struct FinArgs {
  const float* gamma;     // [C_out] of the norm that reads this output
  const float* beta;
  float* ss;              // [N, C_out, 2] in the planar layout of NormFinalizeArgs
  int* arrive;            // [N] arrival counters, zero before the launch and after it
  int planar;             // 8 | 4 (elements per 16-byte chunk of the context's type)
};

struct Conv3Args {
  const void* x;          // [N, H, W, C_in] raw input of the norm (T)
  const float* ss;        // [N, C_in / EPC, 2, EPC] (a of a 16-byte chunk of channels, then its b): operand = relu(a * x + b)
  const uint4* wstream;   // [C_out / 64][frags_per_cg][64 lanes] packed A fragments (tapir_conv_pack)
  long frags_per_cg;
  const void* shortcut;   // null, or [N, Ho, Wo, C_out] added before rounding (T)
  void* y;                // [N, Ho, Wo, C_out] (T)
  void* y_proj;           // DUAL launches: [N, Ho, Wo, C_out] (T), the block's 1x1 projection of the same operand
  float* part;            // null, or [N, tiles, C_out, 2]: (mean, M2) of the stored values of each tile
  int N, H, W;            // input image
  int Ho, Wo;             // output image = ceil(H / stride), ceil(W / stride)
  int pad_y, pad_x;       // SAME padding on the low side
  int TH, tiles;          // output rows per tile, tiles per image = ceil(Ho / TH)
  int waves;              // 4 or 8 (conv3_plan)
  int cin_total;          // conv_small.hpp MODE 2 only: input channels in memory (chunks of CIN); 0 elsewhere
  // conv_small.hpp MODE 0 only: the input norm's pairs merged by the CONSUMER, in its prologue, from the producer's tile
  // summaries (part_in != null: ss is not read) -- takes the producer's write-through + ticket + merge off the launch chain
  const float* part_in;   // [N, slabs_in, C_in, 2]
  const float* gamma_in;  // [C_in]
  const float* beta_in;
  int slabs_in, per_s_in;
  long long* dbg_times;   // TRACE build: [workgroups][waves][8] shader-cycle totals per phase
  FinArgs fin;
};

inline long conv3_frags_per_cg(int cin, int ks, int kstep) { return (long)ks * ks * (cin / kstep) * 4 + cv3_ring(ks); }
// DUAL streams (conv_0 + proj_conv of a block in one launch): the projection's k-steps come first, padded with zero
// fragments to whole ring turns (3 k-steps), so that the 3x3 loop behind them starts at ring phase 0
constexpr int conv3_proj_ksteps_c(int cin, int kstep) { return (cin / kstep + cv3_ring(3) / 4 - 1) / (cv3_ring(3) / 4) * (cv3_ring(3) / 4); }
inline int conv3_proj_ksteps(int cin, int kstep) { return conv3_proj_ksteps_c(cin, kstep); }
inline long conv3_dual_frags_per_cg(int cin, int kstep) { return conv3_frags_per_cg(cin, 3, kstep) + 4L * conv3_proj_ksteps(cin, kstep); }

// XLA SAME: total = max((ceil(n / s) - 1) * s + k - n, 0), low = total / 2
inline int conv3_pad_lo(int n, int k, int s) {
  const int total = ((n + s - 1) / s - 1) * s + k - n;
  return total > 0 ? total / 2 : 0;
}

inline bool conv3_supported(int cin, int cout, int ks, int stride) {
  const bool c_ok = (cin == 64 || cin == 128 || cin == 256) && (cout == 64 || cout == 128 || cout == 256);
  if (!c_ok || (ks != 1 && ks != 3) || (stride != 1 && stride != 2)) return false;
  if (stride == 1) return cin == cout && (ks == 3 || cin != 128);   // conv_0 / conv_1 / proj_conv of a stride-1 block
  return cout == 2 * cin;                                    // conv_0 / proj_conv of the first block of a stride-2 group
}

// blocks whose conv_0 (3x3) and proj_conv (1x1, same stride) run as ONE dual launch
inline bool conv3_dual_supported(int cin, int cout, int stride) {
  return stride == 1 ? (cin == cout && (cin == 64 || cin == 256)) : ((cin == 64 || cin == 128) && cout == 2 * cin);
}

// output rows per tile / tiles per image / waves per workgroup for an [H, W, C_in] input; false if the
// shape does not fit the kernel
inline bool conv3_plan(int H, int W, int cin, int cout, int ks, int stride, int esize, int* rows, int* tiles,
                       int* waves = nullptr) {
  if (!conv3_supported(cin, cout, ks, stride) || H < 1 || W < 1) return false;
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long in_cols = (long)stride * (Wo - 1) + ks;
  for (int wv = 4; wv <= 8; wv += 4) {
    if (wv * 64 < cout) continue;                              // a wave owns 64 output channels
    const int px = (wv * 64 / cout) * CV3_NT * 16;             // pixels per workgroup
    int th = px / Wo;
    if (th > Ho) th = Ho;
    while (th >= 1 && ((long)stride * (th - 1) + ks) * in_cols * cin * esize > cv3_lds_bytes(wv)) --th;
    // (a one-row 3x3 tile reads three rows per row of output: take the larger workgroup if it does better)
    if (th < 1 || (th < 2 && wv == 4 && Ho > 1 && ks == 3)) continue;
    *rows = th;
    *tiles = (Ho + th - 1) / th;
    if (waves) *waves = wv;
    return true;
  }
  return false;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
// relu on two packed bf16: as 16-bit integers the negative values (sign bit) are below zero
__device__ __forceinline__ unsigned relu_bf16x2(unsigned p) {
  const s16x2 v = __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0});
  return __builtin_bit_cast(unsigned, v);
}

// Epilogue shared by the block convolutions and the stem: the wave's 4 x NT accumulator fragments
// (lane group g of pixel column c holds channels cg * 64 + 16 g + 4 r + e of pixel qpix[i]) are rounded
// to T and stored as 16-byte pieces (2 for bf16, 4 for f32) per pixel and lane; the (mean, M2) summary of the STORED values of
// the workgroup's TP pixels goes to part[COUT][2] (Chan merge of the per-wave two-pass summaries).
// `scratch` = LDS nobody reads any more (WAVES x 64 float2).
// Last arriver of image n: inorm_finalize_kernel's arithmetic (64 channels x 4 slab lanes per pass, the four partial
// sums of a channel meeting in LDS) over the image's tile summaries, read past the non-coherent caches.
template <int COUT, int THREADS>
__device__ __forceinline__ void fin_merge(const FinArgs& fin, const float* part_img, int n, int slabs, int per_s,
                                          int HW, float* s_red /* [4][64] */) {
  const int tid = threadIdx.x;
  const int ch = tid & 63, q = (tid >> 6) & 3;
  const bool act = tid < 256;
  auto slab_n = [&](int s) { return s < slabs ? (float)max(0, min(HW, (s + 1) * per_s) - s * per_s) : 0.f; };
  const float inv_hw = 1.0f / (float)HW;
  const bool one_round = slabs <= NORM_FIN_LANES * NORM_FIN_MAXS;
  for (int c0 = 0; c0 < COUT; c0 += 64) {
    const int c = c0 + ch;
    const float* ps = part_img + (long)c * 2;
    float2 v[NORM_FIN_MAXS];
    float s1 = 0.f;
    for (int s0 = q; s0 < slabs; s0 += NORM_FIN_LANES * NORM_FIN_MAXS) {
#pragma unroll
      for (int k = 0; k < NORM_FIN_MAXS; ++k) v[k] = agent_load_f2(ps + (long)min(s0 + k * NORM_FIN_LANES, slabs - 1) * COUT * 2);
#pragma unroll
      for (int k = 0; k < NORM_FIN_MAXS; ++k) s1 = fmaf(slab_n(s0 + k * NORM_FIN_LANES), v[k].x, s1);
    }
    if (act) s_red[q * 64 + ch] = s1;
    lds_barrier();
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_FIN_LANES; ++k) mean += s_red[k * 64 + ch];
    mean *= inv_hw;
    float m2 = 0.f;
    for (int s0 = q; s0 < slabs; s0 += NORM_FIN_LANES * NORM_FIN_MAXS) {
      if (!one_round) {
#pragma unroll
        for (int k = 0; k < NORM_FIN_MAXS; ++k) v[k] = agent_load_f2(ps + (long)min(s0 + k * NORM_FIN_LANES, slabs - 1) * COUT * 2);
      }
#pragma unroll
      for (int k = 0; k < NORM_FIN_MAXS; ++k) {
        const float nk = slab_n(s0 + k * NORM_FIN_LANES);
        const float d = v[k].x - mean;
        m2 += nk > 0.f ? fmaf(nk * d, d, v[k].y) : 0.f;
      }
    }
    lds_barrier();        // every lane has read the first partial sums
    if (act) s_red[q * 64 + ch] = m2;
    lds_barrier();
    if (act && q == 0) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < NORM_FIN_LANES; ++k) tot += s_red[k * 64 + ch];
      const float rstd = 1.0f / sqrtf(tot * inv_hw + kInEps);
      const float sc = rstd * fin.gamma[c];
      const int P = fin.planar;
      const long i0 = P ? ((long)n * COUT + (c & ~(P - 1))) * 2 + (c & (P - 1)) : ((long)n * COUT + c) * 2;
      fin.ss[i0] = sc;
      fin.ss[i0 + (P ? P : 1)] = fin.beta[c] - mean * sc;
    }
    lds_barrier();        // s_red is reused by the next 64 channels
  }
}

// part = this tile's summary slot; with fin.ss != null also: part_img = the image's [tiles][COUT][2] summaries,
// n = image, tiles / per_s / HW = the slab geometry inorm_finalize_kernel would get for them.
template <typename T, int COUT, int NT, int WAVES>
__device__ __forceinline__ void cv3_epilogue(const f32x4 (&acc)[4][NT], const int (&qpix)[NT], int TP,
                                             T* ytile, float* part, char* scratch, const FinArgs& fin = FinArgs{},
                                             const float* part_img = nullptr, int n = 0, int tiles = 0,
                                             int per_s = 0, int HW = 0) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int CG = COUT / 64, PG = WAVES / CG;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int cg = wave % CG, pg = wave / CG;
  float2 (*const s_stat)[64] = reinterpret_cast<float2 (*)[64]>(scratch);   // [wave][channel of the wave]
  const int cnt_w = max(0, min(NT * 16, TP - pg * NT * 16));
  const float inv_cnt = cnt_w > 0 ? 1.0f / (float)cnt_w : 0.f;
  uint2 pk[4][NT];                                 // (bf16 only)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (BF) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk[r][i].x = pack_bf16x2(acc[r][i][0], acc[r][i][1]);
        pk[r][i].y = pack_bf16x2(acc[r][i][2], acc[r][i][3]);
      }
    }
    if (qpix[i] < TP) {
      T* yq = ytile + (long)qpix[i] * COUT + cg * 64 + 16 * g;
      if (BF) {
        uint4* yp = reinterpret_cast<uint4*>(yq);
        yp[0] = make_uint4(pk[0][i].x, pk[0][i].y, pk[1][i].x, pk[1][i].y);
        yp[1] = make_uint4(pk[2][i].x, pk[2][i].y, pk[3][i].x, pk[3][i].y);
      } else {
        f32x4* yp = reinterpret_cast<f32x4*>(yq);
#pragma unroll
        for (int r = 0; r < 4; ++r) yp[r] = acc[r][i];
      }
    }
  }
  if (part == nullptr) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool ok = qpix[i] < TP;
      if (BF) {
        const uint2 p = pk[r][i];
        v[i][0] = ok ? __uint_as_float(p.x << 16) : 0.f;
        v[i][1] = ok ? __uint_as_float(p.x & 0xffff0000u) : 0.f;
        v[i][2] = ok ? __uint_as_float(p.y << 16) : 0.f;
        v[i][3] = ok ? __uint_as_float(p.y & 0xffff0000u) : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = ok ? acc[r][i][e] : 0.f;
      }
    }
    float mean[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i) s += v[i][e];
      mean[e] = s;
    }
    row_sum_n<4>(mean);
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] *= inv_cnt;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const float d = qpix[i] < TP ? v[i][e] - mean[e] : 0.f;
        s = fmaf(d, d, s);
      }
      m2[e] = s;
    }
    row_sum_n<4>(m2);
    if (c == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s_stat[wave][16 * g + 4 * r + e] = make_float2(mean[e], m2[e]);
    }
  }
  lds_barrier();
  if (tid < COUT) {
    const int cgi = tid / 64, ch = tid % 64;
    float cn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int p = 0; p < PG; ++p) {
      const float nb = (float)max(0, min(NT * 16, TP - p * NT * 16));
      const float2 sv = s_stat[p * CG + cgi][ch];
      merge_stats(cn, mean, m2, nb, sv.x, sv.y);
    }
    if (fin.ss != nullptr) agent_store_f2(part + tid * 2, mean, m2);      // write-through: read by another workgroup
    else *reinterpret_cast<float2*>(part + tid * 2) = make_float2(mean, m2);
  }
  if (fin.ss == nullptr) return;
  dma_wait<0>();          // every wave: its summary stores have been written through
  lds_barrier();          // (also: the reads of s_stat above are done, the scratch is reused below)
  int* const s_flag = reinterpret_cast<int*>(scratch);
  if (tid == 0) {
    // RELAXED on purpose.  An acq_rel ticket (or a __threadfence before it) is not free here: the release half is a
    // write-back of the XCD's whole L2 (buffer_wbl2), which holds up to 100 MB of the activations this launch just
    // stored -- measured 1.6 -> 6.1 ms per clip.  The ordering the hand-off needs is built from what
    // the hardware guarantees instead: the summaries are stored write-through (sc1) and the wave's vector-memory
    // queue is drained (dma_wait<0>, an asm volatile with a memory clobber the compiler cannot move the atomic
    // across) before the ticket; the last arriver reads them with sc1 loads that bypass L1 and the non-coherent L2s.
    const int last = agent_fetch_add(fin.arrive + n, 1) == tiles - 1;
    if (last) agent_store_int(fin.arrive + n, 0);
    *s_flag = last;
  }
  lds_barrier();
  if (*s_flag == 0) return;
  fin_merge<COUT, WAVES * 64>(fin, part_img, n, tiles, per_s, HW, reinterpret_cast<float*>(scratch) + 16);
}

// DUAL (3x3 only): the launch also computes the block's 1x1 projection (resnet.py:232-240 proj_conv; same stride) of the
// SAME normalised operand -- the centre tap of the staged tile -- into y_proj, in front of the 3x3 loop: one staging of
// the input instead of two launches that each stage it (the weight stream carries the projection's fragments first).
template <typename T, int CIN, int COUT, int KS, int STRIDE, int NT, int WAVES, bool HAS_SC, bool TRACE = false,
          bool DUAL = false>
__global__ __launch_bounds__(WAVES * 64, 2) void conv_fused_kernel(Conv3Args a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EPC = CvT<T>::EPC;
  constexpr int THREADS = WAVES * 64;
  constexpr int CG = COUT / 64, PG = WAVES / CG;
  static_assert(PG >= 1, "a wave owns 64 output channels");
  constexpr int CB = CIN * (int)sizeof(T);         // bytes per input pixel
  constexpr int CPP = CIN / EPC;                   // 16-byte chunks per input pixel
  constexpr int SWZ = (CPP < 16 ? CPP : 16) - 1;
  constexpr int TAPS = KS * KS;
  constexpr int KPT = CIN / CvT<T>::KSTEP;         // k-steps per tap
  constexpr int RING = cv3_ring(KS), G = RING / 4; // k-steps per ring turn
  constexpr int UNR = KS == 3 ? 2 * G : G;         // k-steps per loop trip (even: the B buffers alternate)
  static_assert(UNR % 2 == 0 && UNR % G == 0 && (TAPS * KPT) % UNR == 0, "whole loop trips");
  static_assert(!DUAL || (KS == 3 && !HAS_SC), "the projection is fused into conv_0 (3x3, no shortcut)");
  constexpr int KP = DUAL ? (KPT + G - 1) / G * G : 0;   // k-steps of the projection, padded to whole ring turns (conv3_proj_ksteps)
  __shared__ uint4 s_tile[cv3_lds_bytes(WAVES) / 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int cg = wave % CG, pg = wave / CG;
  // Consecutive workgroup ids go round the 8 XCDs (each with its own L2): XCD x takes the x-th
  // CONTIGUOUS eighth of the tiles, so that the halo rows two neighbouring tiles share are fetched into
  // one L2 once instead of into two.
  const int total = a.N * a.tiles;
  const int per_xcd = (total + 7) >> 3;
  const int bid = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (bid >= total) return;
  const int n = bid / a.tiles, t = bid - n * a.tiles;
  const int H = a.H, W = a.W, Wo = a.Wo;
  const int PW = STRIDE * (Wo - 1) + KS;           // columns of the input tile
  const int r0 = t * a.TH;                         // first output row
  const int rows = min(a.TH, a.Ho - r0);
  const int HP = (STRIDE * (rows - 1) + KS) * PW;  // pixels of the input tile
  const int TP = rows * Wo;                        // output pixels of this tile
  const int y0 = STRIDE * r0 - a.pad_y, x0 = -a.pad_x;   // input coordinates of tile pixel (0, 0)
  char* const tile = reinterpret_cast<char*>(s_tile);

  // TRACE (tools/kbench.py --what convtrace): shader cycles per phase, per wave
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int k) {
#ifndef TAPIR_HIPEMU
    if (TRACE) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
      if (k >= 0) tph[k] += t_ - tlast;
      tlast = t_;
    }
#endif
  };
  tick(-1);

  // ---- the wave's weight stream: the first ring of fragments is in flight during the staging
  const uint4* wp = a.wstream + ((long)cg * a.frags_per_cg) * 64 + lane;
  uint4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = *wp; wp += 64; }

  // ---- this lane's output pixel of each of the wave's pixel tiles: its first tap in the input tile
  int Pc[NT], qpix[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int q = (pg * NT + i) * 16 + c;
    qpix[i] = q;
    const int qq = q < TP ? q : 0;
    const int yy = qq / Wo, xx = qq - yy * Wo;
    Pc[i] = STRIDE * (yy * PW + xx);               // tile pixel of tap (0, 0)
  }
  // the accumulators start from the shortcut (resnet.py:256): its loads complete under the staging
  const long img = ((long)n * a.Ho + r0) * Wo;     // first output pixel of the tile
  // (lane g of a pixel column holds the 16 CONSECUTIVE channels cg * 64 + 16 g + 4 r + e of its pixel:
  // the host packing permutes the rows of the A fragments accordingly, see tapir_conv3x3_pack)
  f32x4 acc[4][NT];
  // (the raw 32 bytes are parked in acc[0] / acc[1] and converted after the staging: a conversion
  // right here would make the wave wait out the load latency once per pixel tile)
  auto bf4 = [](unsigned p, unsigned q) {
    return f32x4{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u),
                 __uint_as_float(q << 16), __uint_as_float(q & 0xffff0000u)};
  };
  // All loads of the prologue are unconditional (clamped addresses, the mask is applied on use): a load
  // under a lane condition costs a branch and, in hipcc's hands, a wait for it right behind the branch.
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_SC) {
      const int qq = qpix[i] < TP ? qpix[i] : 0;
      const f32x4* sp = reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.shortcut) +
                                                       (img + qq) * COUT + cg * 64 + 16 * g);
      acc[0][i] = sp[0];
      acc[1][i] = sp[1];
      if (!BF) { acc[2][i] = sp[2]; acc[3][i] = sp[3]; }   // f32: the 16 channels as they are
    }
  }

  // ---- stage relu(a x + b) of the input tile; a thread keeps one channel chunk (8 channels).
  // (A row-by-row walk with loop-invariant column math has half the VALU instructions and was 15 %
  // SLOWER on the GPU: the phase is bound by the memory system)
  {
    constexpr int PPS = THREADS / CPP;             // pixels per sweep
    constexpr int U = WAVES == 4 ? 18 : 16;        // loads in flight per thread (one trip covers the usual tile)
    const int chunk = tid % CPP, pl = tid / CPP;
    constexpr int NSS = 2 * EPC / 4;               // bf16: a[0..3], a[4..7], b[0..3], b[4..7]; f32: a[0..3], b[0..3]
    f32x4 ssv[NSS];
    {
      const f32x4* sp = reinterpret_cast<const f32x4*>(a.ss + ((long)n * CIN + EPC * chunk) * 2);
#pragma unroll
      for (int k = 0; k < NSS; ++k) ssv[k] = sp[k];
    }
    const T* xin = reinterpret_cast<const T*>(a.x) + (long)n * H * W * CIN + EPC * chunk;
    // tile pixel P = hy * PW + hx walks in steps of PPS without a division per element
    const int dq = PPS / PW, dr = PPS - dq * PW;
    int hy = pl / PW, hx = pl - hy * PW;
    for (int P0 = pl; P0 < HP; P0 += U * PPS) {
      uint4 v[U];
      int off[U];                                  // LDS byte offset, -1: past the tile; bit 30: outside the image
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int P = P0 + u * PPS;
        const int y = y0 + hy, x = x0 + hx;
        const bool in = P < HP && y >= 0 && y < H && x >= 0 && x < W;
        const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        v[u] = *reinterpret_cast<const uint4*>(xin + (yc * W + xc) * CIN);
        off[u] = P < HP ? ((P * CB + ((chunk ^ (P & SWZ)) << 4)) | (in ? 0 : (1 << 30))) : -1;
        hx += dr; hy += dq;
        if (hx >= PW) { hx -= PW; ++hy; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const unsigned m = (off[u] >> 30) ? 0u : 0xffffffffu;
        unsigned r4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (BF) {
            const f32x2 xv = f32x2{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xffff0000u)};
            const f32x2 sa = f32x2{ssv[k >> 1][2 * (k & 1)], ssv[k >> 1][2 * (k & 1) + 1]};
            const f32x2 sb = f32x2{ssv[NSS / 2 + (k >> 1)][2 * (k & 1)], ssv[NSS / 2 + (k >> 1)][2 * (k & 1) + 1]};
            const f32x2 yv = __builtin_elementwise_fma(xv, sa, sb);
            r4[k] = relu_bf16x2(pack_bf16x2(yv.x, yv.y)) & m;
          } else {
            r4[k] = __float_as_uint(fmaxf(fmaf(__uint_as_float(w4[k]), ssv[0][k], ssv[1][k]), 0.f)) & m;
          }
        }
        if (off[u] >= 0) *reinterpret_cast<uint4*>(tile + (off[u] & 0x3fffffff)) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
      }
    }
  }

  if (HAS_SC && BF) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const uint4 s0 = __builtin_bit_cast(uint4, acc[0][i]), s1 = __builtin_bit_cast(uint4, acc[1][i]);
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool ok = qpix[i] < TP;
      acc[0][i] = ok ? bf4(s0.x, s0.y) : zero; acc[1][i] = ok ? bf4(s0.z, s0.w) : zero;
      acc[2][i] = ok ? bf4(s1.x, s1.y) : zero; acc[3][i] = ok ? bf4(s1.z, s1.w) : zero;
    }
  }
  if (HAS_SC && !BF) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (qpix[i] >= TP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  tick(0);
  lds_barrier();
  tick(1);

  if constexpr (DUAL) {
    // ---- proj_conv: KPT k-steps on the tap that reads input pixel (stride y, stride x) of output pixel (y, x) -- the
    // tile's tap (pad_y, pad_x); the zero fragments that pad the stream to whole ring turns multiply whatever the
    // clamped k-step reads.  Stores only: the shortcut it produces is not normalised (no statistics).
    const int toffp = a.pad_y * PW + a.pad_x;
    auto read_p = [&](int ks, uint4 (&fb)[NT]) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int P = Pc[i] + toffp;
        fb[i] = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
      }
    };
    uint4 fb0[NT], fb1[NT];
    read_p(0, fb0);
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      uint4 (&nxt)[NT] = (kk & 1) ? fb0 : fb1;
      uint4 (&cur)[NT] = (kk & 1) ? fb1 : fb0;
      read_p(kk + 1 < KPT ? kk + 1 : KPT - 1, nxt);
      sched_fence();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint4 fa = ring[(kk % G) * 4 + r];
#pragma unroll
        for (int i = 0; i < NT; ++i) MfmaStep<T>::run(fa, cur[i], acc[r][i]);
        ring[(kk % G) * 4 + r] = *wp;
        wp += 64;
        sched_fence();
      }
    }
    cv3_epilogue<T, COUT, NT, WAVES>(acc, qpix, TP, reinterpret_cast<T*>(a.y_proj) + img * COUT, nullptr, tile);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- TAPS x KPT k-steps; B fragments one k-step ahead, A fragments refilled after their last MFMA
  auto read_b = [&](int tap, int ks, uint4 (&fb)[NT]) {
    const int dy = (tap * 11) >> 5;                // tap / 3 for tap < 9
    const int toff = KS == 3 ? dy * PW + (tap - 3 * dy) : 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int P = Pc[i] + toff;
      fb[i] = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
    }
  };
  {
    uint4 fb0[NT], fb1[NT];
    read_b(0, 0, fb0);
    int tap = 0, ks = 0;
    for (int grp = 0; grp < TAPS * KPT / UNR; ++grp) {
#pragma unroll
      for (int kk = 0; kk < UNR; ++kk) {
        int ks1 = ks + 1, tap1 = tap;
        if (ks1 == KPT) { ks1 = 0; tap1 = tap + 1; }
        if (tap1 == TAPS) tap1 = 0;                // past the end: any valid address (not used)
        uint4 (&nxt)[NT] = (kk & 1) ? fb0 : fb1;
        uint4 (&cur)[NT] = (kk & 1) ? fb1 : fb0;
        read_b(tap1, ks1, nxt);
        sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint4 fa = ring[(kk % G) * 4 + r];
#pragma unroll
          for (int i = 0; i < NT; ++i) MfmaStep<T>::run(fa, cur[i], acc[r][i]);
          ring[(kk % G) * 4 + r] = *wp;
          wp += 64;
          sched_fence();
        }
        tap = tap1; ks = ks1;
      }
    }
  }
  tick(2);
  lds_barrier();   // every wave is done with the tile: the region is reused for the summaries
  tick(3);

  // ---- epilogue: round, store, per-channel (mean, M2) of what was stored
  cv3_epilogue<T, COUT, NT, WAVES>(acc, qpix, TP, reinterpret_cast<T*>(a.y) + img * COUT,
                                   a.part ? a.part + ((long)n * a.tiles + t) * COUT * 2 : nullptr, tile, a.fin,
                                   a.part ? a.part + (long)n * a.tiles * COUT * 2 : nullptr, n, a.tiles, a.TH * a.Wo,
                                   a.Ho * a.Wo);
  tick(4);
  if (TRACE && a.dbg_times != nullptr && lane == 0) {
    tick(5);
    long long* o = a.dbg_times + ((long)bid * WAVES + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (long long)tph[k];
  }
}

template <typename T>
inline void launch_conv_fused(const Conv3Args& a, int cin, int cout, int ks, int stride, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((a.N * a.tiles + 7) / 8))), block((unsigned)(a.waves * 64));
#ifdef TAPIR_EXPERIMENTS
  constexpr bool kTrace = true;                    // phase trace (tools/kbench.py --what convtrace)
  const bool trace = a.dbg_times != nullptr;
#else
  constexpr bool kTrace = false;
  const bool trace = false;
#endif
#define TAPIR_CV3_SC(CI_, CO_, K_, S_, W_, SC_)                                                                          \
  do {                                                                                                                  \
    if (trace) hipLaunchKernelGGL((conv_fused_kernel<T, CI_, CO_, K_, S_, CV3_NT, W_, SC_, kTrace>), grid, block, 0, s, a); \
    else TAPIR_LAUNCH((conv_fused_kernel<T, CI_, CO_, K_, S_, CV3_NT, W_, SC_>), grid, block, s, a);                      \
  } while (0)
  // the shortcut is only ever added by conv_1 (3x3, stride 1, C -> C)
#define TAPIR_CV3(CI_, CO_, K_, S_)                                                    \
  do {                                                                                 \
    if (a.waves == 4) {                                                                \
      if ((K_) == 3 && (S_) == 1 && a.shortcut) TAPIR_CV3_SC(CI_, CO_, K_, S_, 4, (K_ == 3 && S_ == 1)); \
      else TAPIR_CV3_SC(CI_, CO_, K_, S_, 4, false);                                   \
    } else {                                                                           \
      if ((K_) == 3 && (S_) == 1 && a.shortcut) TAPIR_CV3_SC(CI_, CO_, K_, S_, 8, (K_ == 3 && S_ == 1)); \
      else TAPIR_CV3_SC(CI_, CO_, K_, S_, 8, false);                                   \
    }                                                                                  \
  } while (0)
  if (a.y_proj != nullptr) {   // conv_0 + proj_conv of a block (launch_conv_fused is only handed shapes conv3_dual_supported takes)
#define TAPIR_CV3_DUAL(CI_, CO_, S_)                                                                                     \
  do {                                                                                                                  \
    if (a.waves == 4) TAPIR_LAUNCH((conv_fused_kernel<T, CI_, CO_, 3, S_, CV3_NT, 4, false, false, true>), grid, block, s, a); \
    else TAPIR_LAUNCH((conv_fused_kernel<T, CI_, CO_, 3, S_, CV3_NT, 8, false, false, true>), grid, block, s, a);       \
  } while (0)
    if (stride == 1) { if (cin == 64) TAPIR_CV3_DUAL(64, 64, 1); else TAPIR_CV3_DUAL(256, 256, 1); }
    else { if (cin == 64) TAPIR_CV3_DUAL(64, 128, 2); else TAPIR_CV3_DUAL(128, 256, 2); }
#undef TAPIR_CV3_DUAL
    return;
  }
  if (stride == 1) {
    if (ks == 3) {
      if (cin == 64) TAPIR_CV3(64, 64, 3, 1);
      else if (cin == 128) TAPIR_CV3(128, 128, 3, 1);
      else TAPIR_CV3(256, 256, 3, 1);
    } else {
      if (cin == 64) TAPIR_CV3(64, 64, 1, 1);
      else TAPIR_CV3(256, 256, 1, 1);
    }
  } else {
    if (ks == 3) {
      if (cin == 64) TAPIR_CV3(64, 128, 3, 2);
      else TAPIR_CV3(128, 256, 3, 2);
    } else {
      if (cin == 64) TAPIR_CV3(64, 128, 1, 2);
      else TAPIR_CV3(128, 256, 1, 2);
    }
  }
#undef TAPIR_CV3
#undef TAPIR_CV3_SC
  (void)cout;
}

// ---------------------------------------------------------------------------------------------------
// The stem: 7x7 / stride 2 / SAME convolution 3 -> 64 channels on the frames themselves
// (resnet.py:356-364 initial_conv; its output feeds the first InstanceNorm, whose statistics the
// epilogue emits).  K = 7 rows x (7 columns x 3 channels = 21 contiguous input values, padded to 32):
// one k-step per kernel row.  The input rows [2 rows_out + 5][W x 3] sit in LDS as bf16, unswizzled: the
// B fragment of (pixel x, kernel row ky, lane group g) is the 8 values at row 2 y + ky, element
// 6 x + 8 g -- a 4-byte aligned address, read as four ds_read_b32; the values past the 21st multiply
// zero weights.  Frames are f32 (the model's input), rounded to bf16 on the way into LDS like the
// library path's cast (f32 contexts: an f32 image, two k-steps of 16 per kernel row); 4 waves x
// (64 output channels x 64 pixels), two workgroups per CU.
constexpr int STEM_WAVES = 4;
constexpr int STEM_LDS_BYTES = 64 * 1024;

struct StemArgs {
  const float* x;         // [N, H, W, 3] f32
  const uint4* wstream;   // [7 * ksub + ring][4][64 lanes] packed A fragments (tapir_stem_pack)
  void* y;                // [N, Ho, Wo, 64] (the context's element type)
  float* part;            // null, or [N, tiles, 64, 2]
  int N, H, W, Ho, Wo, pad_y, pad_x, TH, tiles;
  FinArgs fin;
};

// bytes per LDS row of the input image: (2 Wo + 5) pixels x 3 channels + 64 bytes of slack for the padded k
inline long stem_row_bytes(int Wo, int esize) { return ((long)(2 * (Wo - 1) + 7) * 3 * esize + 64 + 15) / 16 * 16; }

inline bool stem_plan(int H, int W, int esize, int* rows, int* tiles) {
  if (H < 1 || W < 2 || (W & 1)) return false;     // (pairs of f32 are loaded: even rows of 3 W values)
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int th = (STEM_WAVES * CV3_NT * 16) / Wo;
  if (th > Ho) th = Ho;
  while (th >= 1 && (2 * (th - 1) + 7) * stem_row_bytes(Wo, esize) > STEM_LDS_BYTES) --th;
  if (th < 1) return false;
  *rows = th;
  *tiles = (Ho + th - 1) / th;
  return true;
}

// T = bf16: the frames are rounded to bf16 into the LDS image, one k-step of 32 per kernel row;
// T = float (parity build): f32 image, two k-steps of 16 per kernel row, exact-f32 MFMA.
template <typename T>
__global__ __launch_bounds__(STEM_WAVES * 64, 2) void stem_conv_kernel(StemArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int ES = (int)sizeof(T), EPC = CvT<T>::EPC, KSTEP = CvT<T>::KSTEP;
  constexpr int KSUB = 32 / KSTEP;                 // k-steps per kernel row
  constexpr int NK = 7 * KSUB;
  constexpr int NT = CV3_NT, WAVES = STEM_WAVES, THREADS = WAVES * 64, RING = 8;
  __shared__ uint4 s_tile[STEM_LDS_BYTES / 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int total = a.N * a.tiles;
  const int per_xcd = (total + 7) >> 3;
  const int bid = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (bid >= total) return;
  const int n = bid / a.tiles, t = bid - n * a.tiles;
  const int H = a.H, W = a.W, Wo = a.Wo;
  const int PW = 2 * (Wo - 1) + 7;                 // input columns of the tile
  const int RS = (PW * 3 * ES + 64 + 15) / 16 * 16;   // bytes per LDS row (stem_row_bytes)
  const int r0 = t * a.TH;
  const int rows = min(a.TH, a.Ho - r0);
  const int in_rows = 2 * (rows - 1) + 7;
  const int TP = rows * Wo;
  const int y0 = 2 * r0 - a.pad_y, e0 = -3 * a.pad_x;   // first input row / first input ELEMENT of a tile row
  char* const tile = reinterpret_cast<char*>(s_tile);

  const uint4* wp = a.wstream + lane;
  uint4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = *wp; wp += 64; }

  int Pb[NT], qpix[NT];                            // byte offset of tap (0, 0, channel 0) of this lane's pixels
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int q = (wave * NT + i) * 16 + c;
    qpix[i] = q;
    const int qq = q < TP ? q : 0;
    const int yy = qq / Wo, xx = qq - yy * Wo;
    Pb[i] = 2 * yy * RS + 6 * ES * xx + 16 * g;
  }

  // ---- stage the input rows, pairs of f32 (zero outside the image and in the slack).  All (row, pair) slots of the
  // tile form one index space walked in batches of SB unconditional loads per thread from clamped coordinates, masked
  // on use: a load under the in-image lane condition is waited for on the spot, which made the staging a chain of
  // 2 x in_rows (18 at 256 x 256) dependent 8-byte HBM round trips per workgroup (rounds 2-3; found in the ISA in
  // round 4: one global_load / s_waitcnt vmcnt(0) pair per trip) -- now one or two.
  {
    constexpr int SB = 16;
    const int pairs = RS / (2 * ES);
    const int slots = in_rows * pairs;
    const float* xin = a.x + (long)n * H * W * 3;
    for (int base = tid; base < slots; base += SB * THREADS) {
      float vx[SB], vy[SB];
      int dst[SB];           // byte offset in the LDS image, -1 past its end
      bool ok[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int id = base + k * THREADS;
        const int idc = id < slots ? id : slots - 1;
        const int hy = idc / pairs, p = idc - hy * pairs;
        const int y = y0 + hy;
        const int ge = e0 + 2 * p;                 // even: a pair never straddles the image edge (3 W is even)
        ok[k] = y >= 0 && y < H && ge >= 0 && ge + 1 < 3 * W && 2 * p < PW * 3 + 1;
        const float2 v = *reinterpret_cast<const float2*>(xin + (long)min(max(y, 0), H - 1) * W * 3 + min(max(ge, 0), 3 * W - 2));
        vx[k] = v.x; vy[k] = v.y;
        dst[k] = id < slots ? hy * RS + 2 * ES * p : -1;
      }
#pragma unroll
      for (int k = 0; k < SB; ++k) { pin(vx[k]); pin(vy[k]); }
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        if (dst[k] < 0) continue;
        const float wx = ok[k] ? vx[k] : 0.f, wy = ok[k] ? vy[k] : 0.f;
        if (BF) *reinterpret_cast<unsigned*>(tile + dst[k]) = pack_bf16x2(wx, wy);
        else *reinterpret_cast<float2*>(tile + dst[k]) = make_float2(wx, wy);
      }
    }
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();

  // ---- 7 kernel rows x KSUB k-steps; B fragments one step ahead (the address 6 ES x + 16 g is only
  // 4-byte aligned for bf16: four ds_read_b32)
  auto read_b = [&](int ks, uint4 (&fb)[NT]) {
    const int off = (ks / KSUB) * RS + (ks % KSUB) * KSTEP * ES;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const unsigned* p = reinterpret_cast<const unsigned*>(tile + Pb[i] + off);
      fb[i] = make_uint4(p[0], p[1], p[2], p[3]);
    }
  };
  {
    uint4 fb0[NT], fb1[NT];
    read_b(0, fb0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      uint4 (&nxt)[NT] = (ks & 1) ? fb0 : fb1;
      uint4 (&cur)[NT] = (ks & 1) ? fb1 : fb0;
      read_b(ks + 1 < NK ? ks + 1 : 0, nxt);
      sched_fence();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint4 fa = ring[(ks & 1) * 4 + r];
#pragma unroll
        for (int i = 0; i < NT; ++i) MfmaStep<T>::run(fa, cur[i], acc[r][i]);
        ring[(ks & 1) * 4 + r] = *wp;
        wp += 64;
        sched_fence();
      }
    }
  }
  lds_barrier();
  const long img = ((long)n * a.Ho + r0) * Wo;
  cv3_epilogue<T, 64, NT, WAVES>(acc, qpix, TP, reinterpret_cast<T*>(a.y) + img * 64,
                                 a.part ? a.part + ((long)n * a.tiles + t) * 64 * 2 : nullptr, tile, a.fin,
                                 a.part ? a.part + (long)n * a.tiles * 64 * 2 : nullptr, n, a.tiles, a.TH * Wo,
                                 a.Ho * Wo);
}

template <typename T>
inline void launch_stem_conv(const StemArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((a.N * a.tiles + 7) / 8))), block(STEM_WAVES * 64);
  TAPIR_LAUNCH((stem_conv_kernel<T>), grid, block, s, a);
}

}  // namespace tapir
