// Memory-bound half of the feature backbone (TAPIR.get_feature_grids,
// tapnet/models/tapir_model.py:626-729; ResNet-v2 blocks with InstanceNorm,
// tapnet/models/resnet.py:152-257): everything between the convolutions.
//
//   inorm_stats_kernel : per-(image, channel) mean / M2 of x, or of x = a + b
//                        (the residual add of the previous block, resnet.py:256,
//                        fused with the statistics of the next block's norm :241)
//   inorm_relu_kernel  : relu((x - mean) / sqrt(var + 1e-5) * scale + offset)
//                        (hk.InstanceNorm, resnet.py:177-181, + jax.nn.relu :242,249),
//                        optionally into a zero-bordered buffer (XLA "SAME" padding of a
//                        stride-2 3x3 convolution pads one row / column on the HIGH side
//                        only) and a 2x2-subsampled copy (input of the strided 1x1 projection)
//   l2norm_kernel      : per-pixel L2 normalisation of the feature maps, f32 out (:709-720)
//
// Layout: NHWC, C in {64, 128, 256} contiguous, element type T = bf16 or f32; a
// thread always moves 16 bytes (8 bf16 / 4 f32 channels of one pixel), a wave 1 KiB of
// consecutive addresses.  All arithmetic is f32.  These kernels are HBM-bound: the
// roofline is bytes moved / 8 TB/s.
#pragma once
#include "common.hpp"

namespace tapir {

constexpr int NORM_THREADS = 256;
constexpr float kInEps = 1e-5f;   // hk.InstanceNorm eps (resnet.py:180)

template <typename T> struct Vec16;   // 16 bytes of T <-> EPT floats
template <> struct Vec16<float> {
  static constexpr int EPT = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ void unpack(const uint4& t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
  }
  static __device__ __forceinline__ float round(float x) { return x; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int EPT = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void unpack(const uint4& t, float (&v)[8]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
  static __device__ __forceinline__ float round(float x) { return bf2f(f2bf(x)); }
};

// Chan et al. merge of two (count, mean, M2) summaries.
__device__ __forceinline__ void merge_stats(float& n, float& mean, float& m2, float nb, float meanb,
                                            float m2b) {
  const float nt = n + nb;
  if (nt <= 0.f) return;
  const float d = meanb - mean;
  const float fb = nb / nt;
  mean += d * fb;
  m2 += m2b + d * d * n * fb;
  n = nt;
}

struct NormStatsArgs {
  const void* a;        // [N, HW, C]
  const void* b;        // null, or second addend [N, HW, C]
  void* sum_out;        // b != null: a + b is written here (may alias a or b)
  float* part;          // [N, slabs, C, 2]  (mean, M2) of each slab; count = slab length
  int HW, C, slabs;
};

// grid (slabs, N).  Thread = (pixel lane pl, channel group cg); it owns EPT channels.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void inorm_stats_kernel(NormStatsArgs a) {
  constexpr int EPT = Vec16<T>::EPT;
  __shared__ float s_mean[NORM_THREADS][EPT + 1];
  __shared__ float s_m2[NORM_THREADS][EPT + 1];
  __shared__ float s_cnt[NORM_THREADS];
  const int tid = threadIdx.x;
  const int G = a.C / EPT;                  // threads per pixel (8, 16, 32 or 64)
  const int PP = NORM_THREADS / G;          // pixels per sweep
  const int cg = tid % G, pl = tid / G;
  const int n = blockIdx.y;
  const int per = (a.HW + a.slabs - 1) / a.slabs;
  const int p0 = blockIdx.x * per, p1 = min(a.HW, p0 + per);
  const T* pa = reinterpret_cast<const T*>(a.a) + (long)n * a.HW * a.C + cg * EPT;
  const T* pb = a.b ? reinterpret_cast<const T*>(a.b) + (long)n * a.HW * a.C + cg * EPT : nullptr;
  T* po = a.b ? reinterpret_cast<T*>(a.sum_out) + (long)n * a.HW * a.C + cg * EPT : nullptr;

  // shifted sums around the first value K seen by this thread: s1 = sum(x-K), s2 = sum((x-K)^2)
  float K[EPT], s1[EPT], s2[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { K[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
  int cnt = 0;
  auto accumulate = [&](float (&v)[EPT]) {
    if (cnt == 0) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) K[e] = v[e];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const float d = v[e] - K[e];
      s1[e] += d;
      s2[e] = fmaf(d, d, s2[e]);
    }
    ++cnt;
  };
  // four pixels per thread and trip: four (eight with the residual operand) independent 16-byte
  // loads in flight per thread -- one load per trip leaves HBM latency exposed (4 KiB per workgroup)
  constexpr int U = 4;
  int p = p0 + pl;
  for (; p + (U - 1) * PP < p1; p += U * PP) {
    float v[U][EPT];
#pragma unroll
    for (int u = 0; u < U; ++u) Vec16<T>::load(pa + (long)(p + u * PP) * a.C, v[u]);
    if (pb != nullptr) {
      float w[U][EPT];
#pragma unroll
      for (int u = 0; u < U; ++u) Vec16<T>::load(pb + (long)(p + u * PP) * a.C, w[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) v[u][e] = Vec16<T>::round(v[u][e] + w[u][e]);
        Vec16<T>::store(po + (long)(p + u * PP) * a.C, v[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) accumulate(v[u]);
  }
  for (; p < p1; p += PP) {
    float v[EPT];
    Vec16<T>::load(pa + (long)p * a.C, v);
    if (pb != nullptr) {
      float w[EPT];
      Vec16<T>::load(pb + (long)p * a.C, w);
#pragma unroll
      for (int e = 0; e < EPT; ++e) v[e] = Vec16<T>::round(v[e] + w[e]);
      Vec16<T>::store(po + (long)p * a.C, v);
    }
    accumulate(v);
  }
  const float fc = (float)cnt;
  const float inv = cnt > 0 ? 1.0f / fc : 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    s_mean[tid][e] = K[e] + s1[e] * inv;
    s_m2[tid][e] = fmaxf(s2[e] - s1[e] * s1[e] * inv, 0.f);
  }
  s_cnt[tid] = fc;
  __syncthreads();
  // tree merge over the PP pixel lanes of each channel group (PP is a power of two): in round
  // `half` the lanes pl < half fold in the summary of lane pl + half
  for (int half = PP >> 1; half >= 1; half >>= 1) {
    if (pl < half) {
      const int o = tid + half * G;
      const float nb = s_cnt[o];
      float ncur = s_cnt[tid];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        ncur = s_cnt[tid];
        float mean = s_mean[tid][e], m2 = s_m2[tid][e];
        merge_stats(ncur, mean, m2, nb, s_mean[o][e], s_m2[o][e]);
        s_mean[tid][e] = mean; s_m2[tid][e] = m2;
      }
      s_cnt[tid] = ncur;
    }
    __syncthreads();
  }
  if (pl == 0) {
    float* out = a.part + (((long)n * a.slabs + blockIdx.x) * a.C + cg * EPT) * 2;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { out[2 * e] = s_mean[cg][e]; out[2 * e + 1] = s_m2[cg][e]; }
  }
}

struct NormFinalizeArgs {
  const float* part;    // [N, slabs, C, 2] from inorm_stats_kernel
  const float* gamma;   // [C] scale
  const float* beta;    // [C] offset
  float* ss;            // [N, C, 2]: (rstd * gamma, beta - mean * rstd * gamma)
  int HW, C, slabs;
  int per_s;            // pixels per slab (the last one may be shorter); 0 = ceil(HW / slabs)
  int planar;           // P = 8 | 4: ss is [N, C / P, 2, P] (the P scales of a channel chunk, then its P shifts); 0: [N, C, 2]
};

// grid (N, C / 64), 256 threads = 64 channels x 4 slab lanes: lane q of a channel takes the slabs
// q, q + 4, ...  Two passes over the summaries (all loads of a pass issued before the first use): the
// weighted mean, then M2 about it -- Chan's formula for all slabs at once, no serial chain of pairwise
// merges with a division each (that chain was 9 us per launch, 20 launches on every stream's dependency
// chain per clip); the four partial sums of a channel meet in LDS.
constexpr int NORM_FIN_LANES = 4;
constexpr int NORM_FIN_MAXS = 16;    // slabs per lane held in registers per round
__global__ __launch_bounds__(NORM_THREADS) void inorm_finalize_kernel(NormFinalizeArgs a) {
  __shared__ float s_red[NORM_FIN_LANES][64];
  const int n = blockIdx.x;
  const int ch = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = min((int)blockIdx.y * 64 + ch, a.C - 1);
  const bool live = (int)blockIdx.y * 64 + ch < a.C;
  const int per_s = a.per_s > 0 ? a.per_s : (a.HW + a.slabs - 1) / a.slabs;
  const float2* ps = reinterpret_cast<const float2*>(a.part) + (long)n * a.slabs * a.C + c;
  auto slab_n = [&](int s) { return s < a.slabs ? (float)max(0, min(a.HW, (s + 1) * per_s) - s * per_s) : 0.f; };
  const float inv_hw = 1.0f / (float)a.HW;
  const bool one_round = a.slabs <= NORM_FIN_LANES * NORM_FIN_MAXS;
  float2 v[NORM_FIN_MAXS];
  float s1 = 0.f;
  for (int s0 = q; s0 < a.slabs; s0 += NORM_FIN_LANES * NORM_FIN_MAXS) {
#pragma unroll
    for (int k = 0; k < NORM_FIN_MAXS; ++k) v[k] = ps[(long)min(s0 + k * NORM_FIN_LANES, a.slabs - 1) * a.C];
#pragma unroll
    for (int k = 0; k < NORM_FIN_MAXS; ++k) s1 = fmaf(slab_n(s0 + k * NORM_FIN_LANES), v[k].x, s1);
  }
  s_red[q][ch] = s1;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < NORM_FIN_LANES; ++k) mean += s_red[k][ch];
  mean *= inv_hw;
  float m2 = 0.f;
  for (int s0 = q; s0 < a.slabs; s0 += NORM_FIN_LANES * NORM_FIN_MAXS) {
    if (!one_round) {   // (more than 64 slabs: the registers hold the last round only)
#pragma unroll
      for (int k = 0; k < NORM_FIN_MAXS; ++k) v[k] = ps[(long)min(s0 + k * NORM_FIN_LANES, a.slabs - 1) * a.C];
    }
#pragma unroll
    for (int k = 0; k < NORM_FIN_MAXS; ++k) {
      const float nk = slab_n(s0 + k * NORM_FIN_LANES);
      const float d = v[k].x - mean;
      m2 += nk > 0.f ? fmaf(nk * d, d, v[k].y) : 0.f;
    }
  }
  __syncthreads();        // every lane has read the first partial sums
  s_red[q][ch] = m2;
  __syncthreads();
  if (q == 0 && live) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_FIN_LANES; ++k) tot += s_red[k][ch];
    const float rstd = 1.0f / sqrtf(tot * inv_hw + kInEps);
    const float sc = rstd * a.gamma[c];
    const int P = a.planar;
    const long i0 = P ? ((long)n * a.C + (c & ~(P - 1))) * 2 + (c & (P - 1)) : ((long)n * a.C + c) * 2;
    a.ss[i0] = sc;
    a.ss[i0 + (P ? P : 1)] = a.beta[c] - mean * sc;
  }
}

struct NormApplyArgs {
  const void* x;        // [N, H, W, C]
  const float* ss;      // [N, C, 2] from inorm_finalize_kernel
  void* y;              // [N, oh, ow, C]: pixel (h, w) -> (h, w); rows/cols >= H / W are never written
  void* y_sub;          // null, or [N, H/2, W/2, C]: the pixels with even h and w
  int H, W, C, oh, ow;
  int pix_slabs;        // gridDim.x
};

// grid (pix_slabs, N)
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void inorm_relu_kernel(NormApplyArgs a) {
  constexpr int EPT = Vec16<T>::EPT;
  const int tid = threadIdx.x;
  const int G = a.C / EPT, PP = NORM_THREADS / G;
  const int cg = tid % G, pl = tid / G;
  const int n = blockIdx.y;
  const int HW = a.H * a.W;
  float scale[EPT], shift[EPT];
  {
    const float4* in = reinterpret_cast<const float4*>(a.ss + ((long)n * a.C + cg * EPT) * 2);
#pragma unroll
    for (int e = 0; e < EPT; e += 2) {
      const float4 t = in[e / 2];
      scale[e] = t.x; shift[e] = t.y; scale[e + 1] = t.z; shift[e + 1] = t.w;
    }
  }
  const int per = (HW + a.pix_slabs - 1) / a.pix_slabs;
  const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  const T* px = reinterpret_cast<const T*>(a.x) + (long)n * HW * a.C + cg * EPT;
  T* py = reinterpret_cast<T*>(a.y) + (long)n * a.oh * a.ow * a.C + cg * EPT;
  T* ps = a.y_sub ? reinterpret_cast<T*>(a.y_sub) + (long)n * (a.H / 2) * (a.W / 2) * a.C + cg * EPT
                  : nullptr;
  auto emit = [&](int p, float (&v)[EPT]) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) v[e] = fmaxf(fmaf(v[e], scale[e], shift[e]), 0.f);
    const int h = p / a.W, w = p - h * a.W;
    Vec16<T>::store(py + ((long)h * a.ow + w) * a.C, v);
    if (ps != nullptr && !(h & 1) && !(w & 1))
      Vec16<T>::store(ps + ((long)(h >> 1) * (a.W / 2) + (w >> 1)) * a.C, v);
  };
  constexpr int U = 4;   // independent loads in flight per thread
  int p = p0 + pl;
  for (; p + (U - 1) * PP < p1; p += U * PP) {
    float v[U][EPT];
#pragma unroll
    for (int u = 0; u < U; ++u) Vec16<T>::load(px + (long)(p + u * PP) * a.C, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) emit(p + u * PP, v[u]);
  }
  for (; p < p1; p += PP) {
    float v[EPT];
    Vec16<T>::load(px + (long)p * a.C, v);
    emit(p, v);
  }
}

struct L2Args {
  const void* x; float* out; long pixels; int C;
  // bf16 build, optional: the same values rounded to bf16 row-major [pixels, C] (the hot path's pyramid level) and,
  // for the 256-channel map, in the cost-volume kernel's tile order (pips.hpp PoolArgs::tiled) -- what
  // pool_cast_kernel would otherwise produce by re-reading the f32 grids (151 MB per 48-frame clip)
  void* out_op; void* out_tiled; int cells;   // cells = h * w of one frame (tile order is per frame)
};

// x / sqrt(max(sum_c x^2, 1e-12)) per pixel (tapir_model.py:709-720), f32 out.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void l2norm_kernel(L2Args a) {
  constexpr int EPT = Vec16<T>::EPT;
  const int G = a.C / EPT;                     // lanes per pixel: 16, 32 or 64 (power of two)
  const int PP = NORM_THREADS / G;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  for (long p = (long)blockIdx.x * PP + pl; p < a.pixels; p += (long)gridDim.x * PP) {
    float v[EPT];
    Vec16<T>::load(reinterpret_cast<const T*>(a.x) + p * a.C + cg * EPT, v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) s = fmaf(v[e], v[e], s);
    for (int m = G >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float r = 1.0f / sqrtf(fmaxf(s, 1e-12f));
    float* o = a.out + p * a.C + cg * EPT;
#pragma unroll
    for (int e = 0; e < EPT; ++e) v[e] *= r;
#pragma unroll
    for (int e = 0; e < EPT; e += 4)
      *reinterpret_cast<float4*>(o + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
    if (sizeof(T) == 2 && a.out_op != nullptr) {       // EPT = 8: one 16-byte chunk per lane
      uint4 q;
      q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]);
      q.z = pack_bf16x2(v[4 % EPT], v[5 % EPT]); q.w = pack_bf16x2(v[6 % EPT], v[7 % EPT]);
      reinterpret_cast<uint4*>(a.out_op)[p * G + cg] = q;
      if (a.out_tiled != nullptr) {                    // C = 256: chunk cg of cell (p % cells) of frame p / cells
        const long f = p / a.cells;
        const int cell = (int)(p - f * a.cells), ntile = (a.cells + 15) >> 4;
        reinterpret_cast<uint4*>(a.out_tiled)[((f * ntile + (cell >> 4)) * 32 + cg) * 16 + (cell & 15)] = q;
      }
    }
  }
}

}  // namespace tapir
