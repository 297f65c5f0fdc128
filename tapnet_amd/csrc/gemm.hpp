// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)
//
// Both operands are K-contiguous ("NT"): A is an activation matrix [M, lda],
// W is a weight matrix in its natural [out, in] layout (hk.Linear /
// torch.nn.Linear) or, for the cost volume, the feature grid [T*h*w, C].
// Two element types share one tile structure:
//   bf16 : v_mfma_f32_16x16x32_bf16, f32 accumulate        (speed path)
//   f32  : v_mfma_f32_16x16x4_f32, exact f32 FMA chain      (parity path)
//
// Structure (one template, three tile shapes):
//   * 256-thread workgroup = 4 waves as 2(M) x 2(N); a wave owns FM x FN MFMA
//     16x16 fragments, so the tile is (32 FM) x (32 FN): 192x128, 128x128, 192x64.
//   * K-step = 128 bytes per row (64 bf16 / 32 f32).  Tiles are staged with the
//     LDS DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into two
//     SEPARATE __shared__ arrays used alternately (loop unrolled by two).  Two
//     objects, not one indexed array: the compiler's waitcnt pass only lets a
//     ds_read of the current tile go ahead of the in-flight DMA of the next one
//     when alias analysis proves them disjoint; with one array it drains vmcnt
//     before every fragment read and the copy no longer overlaps the MFMAs.
//   * LDS rows are 128 B = eight 16-B chunks; physical chunk p of row r holds
//     logical chunk p ^ f(r).  The DMA writes LDS lane-linearly, so the swizzle
//     is applied to the per-lane SOURCE address and again on the ds_read_b128
//     fragment reads (cdna_hip_programming.md rule 21 / T2); with f = the low 3
//     bits of the reading lane's fragment row, every ds_read_b128 lane group
//     covers all 64 banks.
//   * Operand roles are swapped in the MFMA (W rows feed the "A" port,
//     activation rows the "B" port), so a lane ends up with 4 consecutive output
//     columns of one output row per fragment.  The W rows a lane feeds to
//     fragment j are PERMUTED (row = 4 FN (q>>2) + 4 j + (q&3) for port row q) so
//     that its FN fragments hold 4 FN CONSECUTIVE columns: the epilogue stores
//     32 B (bf16) / 64 B (f32) per lane, 128 / 256 contiguous bytes per row.
//   * Persistent workgroups: grid = min(tiles, 2 per CU); a workgroup walks its
//     tiles and issues the first DMA of the next tile before the epilogue of the
//     current one.  Hardware block b runs on XCD b % 8: every XCD gets a
//     contiguous range of tiles (N fastest), so tiles that share an A row-panel
//     meet in one XCD's private L2 (T1).
//   * Two workgroups per CU (80 KiB LDS each for the 192x128 tile): one's
//     epilogue / DMA wait overlaps the other's MFMAs.
#pragma once
#include <algorithm>

#include "common.hpp"

namespace tapir {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2 };

constexpr int GEMM_THREADS = 256;

struct GemmArgs {
  const void* A; long lda;                 // strides in elements
  const void* W; long ldw;
  const float* bias;                       // [N] or null
  const float* resid; long ldr;            // EPI_BIAS_RESID: [M, ldr] f32
  void* C; long ldc;
  int M, N, K;                             // K multiple of (128 / sizeof(T)); N, ldc multiples of 4
};

template <typename T> struct MfmaStep;

template <> struct MfmaStep<bf16_t> {
  // one 16-byte chunk = 8 bf16 = one 16x16x32 MFMA k-slice per lane group
  static __device__ __forceinline__ void run(const uint4& w, const uint4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
        __builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
};
template <> struct MfmaStep<float> {
  // one 16-byte chunk = 4 f32: lane group g holds k = 4g..4g+3; MFMA j consumes
  // component j of every lane, so the four MFMAs together cover k = 0..15.
  static __device__ __forceinline__ void run(const uint4& w, const uint4& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};

template <typename TO> struct Store4;
template <> struct Store4<float> {
  static __device__ __forceinline__ void run(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};
template <> struct Store4<bf16_t> {
  static __device__ __forceinline__ void run(bf16_t* p, float a, float b, float c, float d) {
    uint2 v;
    v.x = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
    v.y = (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16);
    *reinterpret_cast<uint2*>(p) = v;
  }
};
// 8 consecutive outputs (two fragments of one lane)
template <typename TO> struct Store8;
template <> struct Store8<float> {
  static __device__ __forceinline__ void run(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Store8<bf16_t> {
  static __device__ __forceinline__ void run(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    o.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
    o.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

template <int FM, int FN> struct GemmTile {
  static constexpr int BM = 32 * FM;             // 2 waves x FM fragments x 16 rows
  static constexpr int BN = 32 * FN;
  static constexpr int CH_A = BM * 8 / GEMM_THREADS;   // 16-byte chunks per thread and k-step
  static constexpr int CH_W = BN * 8 / GEMM_THREADS;
  static constexpr int LDS_CHUNKS = (BM + BN) * 8;     // one stage
  static constexpr int LDS_BYTES = 2 * LDS_CHUNKS * 16;
  // swizzle of W row R (tile-local): the low 3 bits of the fragment row q of the lane that reads
  // it, q = 4 (R / (4 FN) % 4) + R % 4  ->  (R & 3) | bit (R / (4 FN)) & 1
  static __device__ __forceinline__ int swz_w(int R) { return (R & 3) | (((R / (4 * FN)) & 1) << 2); }
};

// Per-thread source offsets (in elements, relative to A / W) of the chunks this thread copies in
// every k-step of one tile.  Rows past M / N are clamped to the last valid row (their products
// are never stored).
template <typename TA, int FM, int FN>
struct GemmStager {
  using TL = GemmTile<FM, FN>;
  static constexpr int EPC = 16 / (int)sizeof(TA);
  long offA[TL::CH_A], offW[TL::CH_W];
  __device__ __forceinline__ void set_tile(const GemmArgs& g, int m0, int n0, int tid) {
#pragma unroll
    for (int s = 0; s < TL::CH_A; ++s) {
      const int id = tid + GEMM_THREADS * s;
      const int row = id >> 3, p = id & 7;
      offA[s] = (long)min(m0 + row, g.M - 1) * g.lda + (p ^ (row & 7)) * EPC;
    }
#pragma unroll
    for (int s = 0; s < TL::CH_W; ++s) {
      const int id = tid + GEMM_THREADS * s;
      const int row = id >> 3, p = id & 7;
      offW[s] = (long)min(n0 + row, g.N - 1) * g.ldw + (p ^ TL::swz_w(row)) * EPC;
    }
  }
  // LDS stage layout: A rows [0, BM), W rows [BM, BM+BN); chunk id = row*8 + p (lane-linear)
  __device__ __forceinline__ void issue(const TA* __restrict__ A, const TA* __restrict__ W, int k0,
                                        int tid, uint4* lds) const {
#pragma unroll
    for (int s = 0; s < TL::CH_A; ++s) glds16(A + offA[s] + k0, lds + tid + GEMM_THREADS * s);
#pragma unroll
    for (int s = 0; s < TL::CH_W; ++s)
      glds16(W + offW[s] + k0, lds + TL::BM * 8 + tid + GEMM_THREADS * s);
  }
};

// MFMAs of one k-step from one staged tile.
template <typename TA, int FM, int FN>
__device__ __forceinline__ void gemm_compute_tile(const uint4* lds, int a_row, int w_row, int fr,
                                                  int fg, f32x4 (&acc)[FM][FN]) {
  using TL = GemmTile<FM, FN>;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int c = (kk * 4 + fg) ^ (fr & 7);
    uint4 fa[FM], fw[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[i] = lds[(a_row + i * 16) * 8 + c];
#pragma unroll
    for (int j = 0; j < FN; ++j) fw[j] = lds[(TL::BM + w_row + j * 4) * 8 + c];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) MfmaStep<TA>::run(fw[j], fa[i], acc[i][j]);
  }
}

// TA: operand element type (bf16_t or float); TO: output element type.
template <typename TA, typename TO, int EPI, int FM, int FN>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(GemmArgs g) {
  using TL = GemmTile<FM, FN>;
  constexpr int EPC = 16 / (int)sizeof(TA);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                 // elements per 128-byte k-step
  __shared__ uint4 lds0[TL::LDS_CHUNKS];
  __shared__ uint4 lds1[TL::LDS_CHUNKS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15;   // fragment row handled by this lane
  const int fg = lane >> 4;   // k lane-group (operand reads) / column group (results)
  const int a_row = wm * FM * 16 + fr;
  const int w_row = wn * FN * 16 + (fr >> 2) * (4 * FN) + (fr & 3);

  // XCD-aware persistent schedule: XCD x owns logical tiles [start, start + len), N fastest.
  const int tiles_n = (g.N + TL::BN - 1) / TL::BN;
  const int ntiles = tiles_n * ((g.M + TL::BM - 1) / TL::BM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;   // the launcher makes gridDim.x a multiple of 8
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int len = q + (xcd < r ? 1 : 0);
  if (slot >= len) return;

  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const TA* __restrict__ W = reinterpret_cast<const TA*>(g.W);
  TO* __restrict__ C = reinterpret_cast<TO*>(g.C);
  const int nk = g.K / BK;
  const bool wide = (g.ldc * (long)sizeof(TO)) % 16 == 0;   // rows keep 16-byte alignment

  GemmStager<TA, FM, FN> st;
  int local = slot;
  int m0 = ((start + local) / tiles_n) * TL::BM;
  int n0 = ((start + local) % tiles_n) * TL::BN;
  st.set_tile(g, m0, n0, tid);
  st.issue(A, W, 0, tid, lds0);

  for (;;) {
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();   // k-step 0 of this tile has landed in lds0 (drains the DMA: vmcnt)
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
      st.issue(A, W, (kt + 1) * BK, tid, lds1);   // in flight during the MFMAs below
      gemm_compute_tile<TA, FM, FN>(lds0, a_row, w_row, fr, fg, acc);
      __syncthreads();                            // lds1 landed; everyone is done with lds0
      if (kt + 2 < nk) st.issue(A, W, (kt + 2) * BK, tid, lds0);
      gemm_compute_tile<TA, FM, FN>(lds1, a_row, w_row, fr, fg, acc);
      __syncthreads();
    }
    if (kt < nk) {   // odd number of k-steps: the last one is in lds0
      gemm_compute_tile<TA, FM, FN>(lds0, a_row, w_row, fr, fg, acc);
      __syncthreads();
    }

    // next tile: start its first copy now, it lands while the epilogue below runs
    const int cm0 = m0, cn0 = n0;
    local += per_xcd;
    const bool more = local < len;
    if (more) {
      m0 = ((start + local) / tiles_n) * TL::BM;
      n0 = ((start + local) % tiles_n) * TL::BN;
      st.set_tile(g, m0, n0, tid);
      st.issue(A, W, 0, tid, lds0);
    }

    // epilogue: lane holds C[m = .. + 16 i + fr][n = .. + 4 FN fg + 4 j + 0..3]
    const int mb = cm0 + wm * FM * 16 + fr;
    const int nb = cn0 + wn * FN * 16 + fg * (4 * FN);
    float4 bias4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias != nullptr)
        bias4[j] = *reinterpret_cast<const float4*>(g.bias + min(nb + j * 4, g.N - 4));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = mb + i * 16;
      const int mc = min(m, g.M - 1);
      float4 res4[FN];
      if (EPI == EPI_BIAS_RESID) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          res4[j] = *reinterpret_cast<const float4*>(g.resid + (long)mc * g.ldr + min(nb + j * 4, g.N - 4));
      }
#pragma unroll
      for (int j = 0; j < FN; j += 2) {
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 b = bias4[j + h];
          float v0 = acc[i][j + h][0] + b.x, v1 = acc[i][j + h][1] + b.y;
          float v2 = acc[i][j + h][2] + b.z, v3 = acc[i][j + h][3] + b.w;
          if (EPI == EPI_BIAS_GELU) {
            v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
          }
          if (EPI == EPI_BIAS_RESID) {
            v0 += res4[j + h].x; v1 += res4[j + h].y; v2 += res4[j + h].z; v3 += res4[j + h].w;
          }
          v[4 * h + 0] = v0; v[4 * h + 1] = v1; v[4 * h + 2] = v2; v[4 * h + 3] = v3;
        }
        const int n = nb + j * 4;
        TO* dst = C + (long)m * g.ldc + n;
        if (m < g.M) {
          if (wide && n + 8 <= g.N) {
            Store8<TO>::run(dst, v);
          } else {
            if (n + 4 <= g.N) Store4<TO>::run(dst, v[0], v[1], v[2], v[3]);
            if (n + 8 <= g.N) Store4<TO>::run(dst + 4, v[4], v[5], v[6], v[7]);
          }
        }
      }
    }
    if (!more) break;
  }
}

// Shape-driven tile choice.  Slots = 2 workgroups per CU x 256 CUs.
enum { GEMM_TILE_AUTO = 0, GEMM_TILE_192x128 = 1, GEMM_TILE_128x128 = 2, GEMM_TILE_192x64 = 3 };

inline int gemm_pick_tile(int M, int N) {
  // few wide tiles when they still fill the chip, narrower ones otherwise
  const long t192 = (long)((M + 191) / 192) * ((N + 127) / 128);
  if (M % 192 == 0 || M >= 192 * 64) {
    if (t192 >= 256 || (N % 128 != 0 && N > 64)) return GEMM_TILE_192x128;
    return GEMM_TILE_192x64;
  }
  return GEMM_TILE_128x128;
}

template <typename TA, typename TO, int EPI, int FM, int FN>
inline void launch_gemm_tile(const GemmArgs& g, hipStream_t stream, int max_grid) {
  using TL = GemmTile<FM, FN>;
  const int ntiles = ((g.N + TL::BN - 1) / TL::BN) * ((g.M + TL::BM - 1) / TL::BM);
  const int per_cu = std::max(1, std::min(2, (160 * 1024) / TL::LDS_BYTES));
  int grid = std::min((ntiles + 7) / 8 * 8, 256 * per_cu);
  if (max_grid > 0) grid = std::min(grid, (max_grid + 7) / 8 * 8);
  hipLaunchKernelGGL((gemm_nt_kernel<TA, TO, EPI, FM, FN>), dim3(grid), dim3(GEMM_THREADS), 0,
                     stream, g);
}

template <typename TA, typename TO, int EPI>
inline void launch_gemm(const GemmArgs& g, hipStream_t stream, int tile = GEMM_TILE_AUTO,
                        int max_grid = 0) {
  if (tile == GEMM_TILE_AUTO) tile = gemm_pick_tile(g.M, g.N);
  switch (tile) {
    case GEMM_TILE_192x128: launch_gemm_tile<TA, TO, EPI, 6, 4>(g, stream, max_grid); break;
    case GEMM_TILE_192x64: launch_gemm_tile<TA, TO, EPI, 6, 2>(g, stream, max_grid); break;
    default: launch_gemm_tile<TA, TO, EPI, 4, 4>(g, stream, max_grid); break;
  }
}

}  // namespace tapir
