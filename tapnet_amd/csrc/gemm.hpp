// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)
//
// Both operands are K-contiguous ("NT"): A is an activation matrix [M, lda],
// W is a weight matrix in its natural [out, in] layout (hk.Linear /
// torch.nn.Linear) or, for the cost volume, the feature grid [T*h*w, C].
// Two element types share one tile structure:
//   bf16 : v_mfma_f32_16x16x32_bf16, f32 accumulate        (speed path)
//   f32  : v_mfma_f32_16x16x4_f32, exact f32 FMA chain      (parity path)
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA 16x16 fragments, 64 accumulator VGPRs).  K-step = 128 bytes per row
// (64 bf16 / 32 f32), LDS double-buffered (2 x 32 KiB).  Tiles are staged with
// the LDS DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write): the
// copy of tile k+1 is issued before the MFMAs of tile k and drained by the
// barrier that ends the step.  LDS rows are 128 B = eight 16-B chunks; physical
// chunk p of row r holds logical chunk p ^ (r & 7) -- the DMA writes LDS
// lane-linearly, so the swizzle is applied to the per-lane SOURCE address and
// again on the ds_read_b128 fragment reads (cdna_hip_programming.md rule 21 /
// T2), which then spread over all banks.
//
// Workgroups are numbered so that the ones sharing an A row-panel are
// consecutive on ONE XCD (block b runs on XCD b % 8): the panel is fetched into
// that XCD's private L2 once instead of up to eight times (T1).
//
// Operand roles are swapped in the MFMA (W rows feed the "A" port, activation
// rows the "B" port), so a lane ends up with 4 CONSECUTIVE output columns of
// one output row (D[n=4*(l>>4)+r][m=l&15]) and the epilogue issues one 16-byte
// (f32) / 8-byte (bf16) store per fragment instead of four scalar ones.
#pragma once
#include "common.hpp"

namespace tapir {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2 };

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
  const void* A; long lda; long strideA;   // strides in elements; stride* = per blockIdx.z
  const void* W; long ldw; long strideW;
  const float* bias;                       // [N] or null
  const float* resid; long ldr;            // EPI_BIAS_RESID: [M, ldr] f32
  void* C; long ldc; long strideC;
  int M, N, K;                             // K multiple of (128 / sizeof(T))
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

// Issues the LDS-DMA copies of one k-step: rows 0..127 = A rows m0.., rows 128..255 = W rows
// n0..; 2048 16-byte chunks, 8 per thread, chunk id = s*256 + tid (lane-linear in LDS).
// Rows past M / N are clamped to the last valid row (their products are never stored).
template <typename TA>
__device__ __forceinline__ void gemm_stage_tile(const TA* __restrict__ A, const TA* __restrict__ W,
                                                long lda, long ldw, int M, int N, int m0, int n0,
                                                int k0, int tid, uint4* lds) {
  constexpr int EPC = 16 / (int)sizeof(TA);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int id = tid + GEMM_THREADS * s;
    const int row = id >> 3, p = id & 7;
    const int c = p ^ (row & 7);
    const TA* src;
    if (s < 4) src = A + (long)min(m0 + row, M - 1) * lda + k0 + c * EPC;
    else src = W + (long)min(n0 + row - GEMM_BM, N - 1) * ldw + k0 + c * EPC;
    glds16(src, lds + id);
  }
}

// TA: operand element type (bf16_t or float); TO: output element type.
template <typename TA, typename TO, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmArgs g) {
  constexpr int EPC = 16 / (int)sizeof(TA);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                 // elements per 128-byte k-step
  __shared__ uint4 lds[2][(GEMM_BM + GEMM_BN) * 8];   // 2 x 32 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware numbering: hardware block b -> XCD b % 8; give every XCD a contiguous range of
  // logical tiles, N fastest, so tiles sharing an A panel follow each other on one XCD.
  const int tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int m0 = (logical / tiles_n) * GEMM_BM;
  const int n0 = (logical % tiles_n) * GEMM_BN;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A) + (long)blockIdx.z * g.strideA;
  const TA* __restrict__ W = reinterpret_cast<const TA*>(g.W) + (long)blockIdx.z * g.strideW;
  TO* __restrict__ C = reinterpret_cast<TO*>(g.C) + (long)blockIdx.z * g.strideC;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  gemm_stage_tile<TA>(A, W, g.lda, g.ldw, g.M, g.N, m0, n0, 0, tid, lds[0]);
  __syncthreads();   // drains the DMA (vmcnt) and makes the tile visible to every wave

  const int fr = lane & 15;   // fragment row handled by this lane
  const int fg = lane >> 4;   // k lane-group
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk)   // DMA of the next tile into the other buffer, in flight during the MFMAs
      gemm_stage_tile<TA>(A, W, g.lda, g.ldw, g.M, g.N, m0, n0, (kt + 1) * BK, tid, lds[buf ^ 1]);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fg;
      uint4 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        fa[i] = lds[buf][row * 8 + (c ^ (row & 7))];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        fw[j] = lds[buf][(GEMM_BM + row) * 8 + (c ^ (row & 7))];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) MfmaStep<TA>::run(fw[j], fa[i], acc[i][j]);
    }
    __syncthreads();   // next tile landed; everyone is done reading `buf`
  }

  // epilogue: lane holds C[m = .. + (l&15)][n = .. + 4*(l>>4) + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      if (n >= g.N) continue;   // N is a multiple of 4 for every call site
      float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
      if (g.bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
      }
      if (EPI == EPI_BIAS_GELU) {
        v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
      }
      if (EPI == EPI_BIAS_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(g.resid + (long)m * g.ldr + n);
        v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
      }
      Store4<TO>::run(C + (long)m * g.ldc + n, v0, v1, v2, v3);
    }
  }
}

template <typename TA, typename TO, int EPI>
inline void launch_gemm(const GemmArgs& g, int batch, hipStream_t stream) {
  dim3 grid(((g.N + GEMM_BN - 1) / GEMM_BN) * ((g.M + GEMM_BM - 1) / GEMM_BM), 1, batch);
  hipLaunchKernelGGL((gemm_nt_kernel<TA, TO, EPI>), grid, dim3(GEMM_THREADS), 0, stream, g);
}

}  // namespace tapir
