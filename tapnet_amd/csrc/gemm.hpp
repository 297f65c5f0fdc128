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
//     columns of one output row per fragment.  Which W row feeds which port row
//     is a free permutation; it is chosen per output type so that every store
//     instruction (16 B per lane) writes 64 contiguous bytes per output row --
//     full 32-byte sectors (GemmTile::col).
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

// Applies one mixer output to the running estimate (tapir_model.py:613-623,
// 1026-1039):  pos += d_xy * (orig/resized), occ += d, expd += d, feats += d.
struct UpdateArgs {
  const float* res;        // [R, 388]
  float* pos;              // [R, 2]  (x, y) in initial_resolution pixels
  float* occ; float* expd; // [R]
  float* feats;            // [R, 384] in/out
  const float* q_hires;    // [B*Q, 128] used when first_of_level
  const float* q_lowres;   // [B*Q, 256]
  float* out_tracks;       // [R, 2] this iteration's slice, video pixels
  float* out_occ; float* out_expd;
  const float* occ0; const float* expd0;   // cost-volume values (reset after a level)
  long R; int T;
  float sx, sy;            // orig / resized  (x, y)
  float vx, vy;            // video / initial_resolution (train2orig)
  int first_of_level;      // feats input was the tiled query feature
  int last_of_level;       // reset occ/expd to the cost-volume values afterwards
};
// EPI_BIAS_UPDATE (gemm_small_kernel only): the [M, 388] output of the mixer's last Linear is not stored; it is applied to the
// running estimate as update_kernel (mixer.hpp) would: the same operations in the same order
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_BIAS_UPDATE = 3 };

struct GemmArgs {
  const void* A; long lda;                 // strides in elements
  const void* W; long ldw;
  const float* bias;                       // [N] or null
  const float* resid; long ldr;            // EPI_BIAS_RESID: [M, ldr] f32
  UpdateArgs upd;                          // EPI_BIAS_UPDATE
  void* C; long ldc;
  int M, N, K;                             // K multiple of (128 / sizeof(T)); N, ldc multiples of 4
  int w_rows;                              // 0, or the rows W really has (< N: N is rounded up to a multiple of 4 for the
                                           // vector stores; the extra columns repeat W's last row and land in C's row padding)
  long long* dbg_times;                    // null, or [workgroups][16] wall-clock stamps (tools/kbench.py --what gemmtrace)
  int stream_out;                          // f32 outputs nobody re-reads soon (the cost-volume workspace, 268 MB per launch): non-temporal stores
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
    v.x = pack_bf16x2(a, b);
    v.y = pack_bf16x2(c, d);
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
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

// Tile configuration: WM x WN waves, each FM x FN MFMA 16x16 fragments, NS LDS stages.
template <int WM_, int WN_, int FM_, int FN_, int NS_, bool PREFETCH_ = false> struct GemmTile {
  static constexpr int WM = WM_, WN = WN_, FM = FM_, FN = FN_, NS = NS_;
  static constexpr bool PREFETCH = PREFETCH_ && NS_ >= 3;   // fragment prefetch across the barrier
  static constexpr int THREADS = WM * WN * 64;
  static constexpr int BM = WM * FM * 16;
  static constexpr int BN = WN * FN * 16;
  static constexpr int CH_A = BM * 8 / THREADS;   // 16-byte chunks per thread and k-step
  static constexpr int CH_W = BN * 8 / THREADS;
  static constexpr int CH = CH_A + CH_W;           // LDS-DMA instructions per thread and stage
  static constexpr int STAGE_CHUNKS = (BM + BN) * 8;
  static constexpr int LDS_BYTES = NS * STAGE_CHUNKS * 16;
  static_assert(BM * 8 % THREADS == 0 && BN * 8 % THREADS == 0, "stage must split evenly");
  static_assert(FN % 2 == 0 && NS >= 2 && NS <= 4, "unsupported tile");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  // Output-column maps.  The lane (fr = l & 15, fg = l >> 4) feeds MFMA port row fr of fragment j
  // with W row w_row(j, fr) and receives the 4 columns col(j, fg) .. +3 of output row fr.  One
  // store INSTRUCTION writes 16 bytes per lane, so the map is chosen per output type such that
  // the four lanes fg = 0..3 of a row write 64 CONTIGUOUS bytes (two full 32-byte sectors):
  //   f32  (PAIR = false): col = 16 j + 4 fg             (one float4 per fragment)
  //   bf16 (PAIR = true) : col = 32 (j/2) + 8 fg + 4 (j%2) (one 8 x bf16 store per fragment pair)
  // swz_w(R) must return the low 3 bits of the fragment row of the lane that READS LDS row R.
  template <bool PAIR> static __device__ __forceinline__ int w_row(int j, int fr) {
    return PAIR ? (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3) : j * 16 + fr;
  }
  template <bool PAIR> static __device__ __forceinline__ int col(int j, int fg) {
    return PAIR ? (j >> 1) * 32 + fg * 8 + (j & 1) * 4 : j * 16 + fg * 4;
  }
  template <bool PAIR> static __device__ __forceinline__ int swz_w(int R) {
    return PAIR ? (R & 3) | (((R >> 3) & 1) << 2) : (R & 7);
  }
};

// Walks the (tile, k-step) sequence of one persistent workgroup for the DMA side, which runs
// NS-1 k-steps ahead of the MFMA side (and therefore crosses tile boundaries on its own).
// Holds the per-thread source offsets (in elements, relative to A / W) of the chunks this thread
// copies in every k-step of the current tile.  Rows past M / N are clamped to the last valid row
// (their products are never stored).
template <typename TA, typename TL, bool PAIR>
struct GemmStager {
  static constexpr int EPC = 16 / (int)sizeof(TA);
  long offA[TL::CH_A], offW[TL::CH_W];
  int local, kt;        // tile (index within this XCD's range) and k-step of the next copy
  __device__ __forceinline__ void set_tile(const GemmArgs& g, int m0, int n0, int tid) {
    tid = opaque(tid);   // recompute the per-thread row / chunk indices here instead of keeping
                         // them in (spilled) registers for the whole kernel
#pragma unroll
    for (int s = 0; s < TL::CH_A; ++s) {
      const int id = tid + TL::THREADS * s;
      const int row = id >> 3, p = id & 7;
      offA[s] = (long)min(m0 + row, g.M - 1) * g.lda + (p ^ (row & 7)) * EPC;
    }
#pragma unroll
    for (int s = 0; s < TL::CH_W; ++s) {
      const int id = tid + TL::THREADS * s;
      const int row = id >> 3, p = id & 7;
      offW[s] = (long)min(n0 + row, (g.w_rows ? g.w_rows : g.N) - 1) * g.ldw + (p ^ TL::template swz_w<PAIR>(row)) * EPC;
    }
  }
  // LDS stage layout: A rows [0, BM), W rows [BM, BM+BN); chunk id = row*8 + p (lane-linear)
  // wave_u: the wave index as a wave-uniform (scalar) value: the LDS destination of the DMA is
  // M0 = the wave's base address; the hardware adds lane * 16 itself.
  __device__ __forceinline__ void issue(const TA* __restrict__ A, const TA* __restrict__ W, int k0,
                                        int wave_u, uint4* lds) const {
#pragma unroll
    for (int s = 0; s < TL::CH_A; ++s) glds16(A + offA[s] + k0, lds + wave_u * 64 + TL::THREADS * s);
#pragma unroll
    for (int s = 0; s < TL::CH_W; ++s)
      glds16(W + offW[s] + k0, lds + TL::BM * 8 + wave_u * 64 + TL::THREADS * s);
  }
};

// Fragment reads of one half k-step (kk = 0 / 1: 16-byte chunks kk*4 .. kk*4+3 of every row).
template <typename TL, bool PAIR>
__device__ __forceinline__ void gemm_load_frags(const uint4* lds, int a_row, int w_base, int fr, int fg,
                                                int kk, uint4 (&fa)[TL::FM], uint4 (&fw)[TL::FN]) {
  const int c = (kk * 4 + fg) ^ (fr & 7);
  // in the order the MFMAs consume them (LDS reads return in order: the first MFMA then waits
  // for FN + 1 reads only, see the fence in the k-step)
#pragma unroll
  for (int j = 0; j < TL::FN; ++j)
    fw[j] = lds[(TL::BM + w_base + TL::template w_row<PAIR>(j, fr)) * 8 + c];
#pragma unroll
  for (int i = 0; i < TL::FM; ++i) fa[i] = lds[(a_row + i * 16) * 8 + c];
}
template <typename TA, typename TL>
__device__ __forceinline__ void gemm_mfma_frags(const uint4 (&fa)[TL::FM], const uint4 (&fw)[TL::FN],
                                                f32x4 (&acc)[TL::FM][TL::FN]) {
#pragma unroll
  for (int i = 0; i < TL::FM; ++i)
#pragma unroll
    for (int j = 0; j < TL::FN; ++j) MfmaStep<TA>::run(fw[j], fa[i], acc[i][j]);
}

// Accumulators of a new tile: zero, or -- EPI_BIAS_RESID -- the residual tile itself (it has the
// accumulator's layout, so "+ skip" costs no epilogue loads and no registers; the loads are waited for
// together with the first stage of copies).
template <typename TO, int EPI, typename TL>
__device__ __forceinline__ void gemm_init_acc(const GemmArgs& g, f32x4 (&acc)[TL::FM][TL::FN], int mb,
                                              int nb, int fg) {
  constexpr bool PAIR = sizeof(TO) == 2;
#pragma unroll
  for (int i = 0; i < TL::FM; ++i)
#pragma unroll
    for (int j = 0; j < TL::FN; ++j) {
      if (EPI == EPI_BIAS_RESID) {
        const int m = min(mb + i * 16, g.M - 1), n = min(nb + TL::template col<PAIR>(j, fg), g.N - 4);
        acc[i][j] = *reinterpret_cast<const f32x4*>(g.resid + (long)m * g.ldr + n);
      } else {
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
}

// Epilogue of one tile: lane holds C[m = mb + 16 i][n = nb + col(j, fg) + 0..3] (GemmTile::col).
template <typename TO, int EPI, typename TL, bool INTERIOR>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, TO* __restrict__ C,
                                              const f32x4 (&acc)[TL::FM][TL::FN],
                                              const f32x4 (&bias4)[TL::FN], int mb, int nb, int fg) {
  constexpr int FM = TL::FM, FN = TL::FN;
  constexpr bool PAIR = sizeof(TO) == 2;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = mb + i * 16;
    // (EPI_BIAS_RESID: the accumulators were INITIALISED with the residual, see gemm_init_acc)
    float v[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f32x4 b = bias4[j];
      float v0 = acc[i][j][0] + b[0], v1 = acc[i][j][1] + b[1];
      float v2 = acc[i][j][2] + b[2], v3 = acc[i][j][3] + b[3];
      if (EPI == EPI_BIAS_GELU) {
        v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
      }
      v[j][0] = v0; v[j][1] = v1; v[j][2] = v2; v[j][3] = v3;
    }
    TO* row = C + (long)m * g.ldc;
    if (PAIR) {
#pragma unroll
      for (int j = 0; j < FN; j += 2) {   // fragments j, j+1 are 8 consecutive columns
        const int n = nb + TL::template col<PAIR>(j, fg);
        const float w8[8] = {v[j][0], v[j][1], v[j][2], v[j][3], v[j + 1][0], v[j + 1][1], v[j + 1][2], v[j + 1][3]};
        if (INTERIOR) {
          Store8<TO>::run(row + n, w8);
        } else if (m < g.M) {
          if (n + 4 <= g.N) Store4<TO>::run(row + n, w8[0], w8[1], w8[2], w8[3]);
          if (n + 8 <= g.N) Store4<TO>::run(row + n + 4, w8[4], w8[5], w8[6], w8[7]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = nb + TL::template col<PAIR>(j, fg);
        if (INTERIOR || (m < g.M && n + 4 <= g.N)) {
          if (sizeof(TO) == 4 && g.stream_out)
            __builtin_nontemporal_store(f32x4{v[j][0], v[j][1], v[j][2], v[j][3]}, reinterpret_cast<f32x4*>(row + n));
          else
            Store4<TO>::run(row + n, v[j][0], v[j][1], v[j][2], v[j][3]);
        }
      }
    }
  }
}

template <int S> struct StageTag { static constexpr int value = S; };

// TA: operand element type (bf16_t or float); TO: output element type.
// TRACE (tools/kbench.py --what gemmsteps): every wave accumulates, over all its k-steps, the shader
// cycles (s_memtime) spent in the four parts of a k-step -- issuing the copies, fragment reads +
// MFMAs (until the last LDS read has returned), waiting for its own copies (vmcnt), waiting at the
// barrier -- plus the epilogues, and lane 0 writes the six totals to g.dbg_times[(wg*16+wave)*8..].
template <typename TA, typename TO, int EPI, typename TL, bool TRACE = false>
__global__ __launch_bounds__(TL::THREADS) void gemm_nt_kernel(GemmArgs g) {
  constexpr int EPC = 16 / (int)sizeof(TA);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                 // elements per 128-byte k-step
  constexpr int NS = TL::NS, FM = TL::FM, FN = TL::FN;
  // One __shared__ object PER STAGE (see the header comment): the waitcnt pass then waits, before
  // the fragment reads of stage s, only for the copies into stage s (a counted vmcnt).
  __shared__ uint4 lds0[TL::STAGE_CHUNKS];
  __shared__ uint4 lds1[TL::STAGE_CHUNKS];
  __shared__ uint4 lds2[NS > 2 ? TL::STAGE_CHUNKS : 1];
  __shared__ uint4 lds3[NS > 3 ? TL::STAGE_CHUNKS : 1];
  uint4* const bufs[4] = {lds0, lds1, lds2, lds3};

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / TL::WN, wn = wave % TL::WN;
  const int fr = lane & 15;   // fragment row handled by this lane
  const int fg = lane >> 4;   // k lane-group (operand reads) / column group (results)
  const int a_row = wm * FM * 16 + fr;
  constexpr bool PAIR = sizeof(TO) == 2;   // output-column map, see GemmTile
  const int w_row = wn * FN * 16;          // first W row of this wave

  // XCD-aware persistent schedule: XCD x owns logical tiles [start, start + len), N fastest.
  const int tiles_n = (g.N + TL::BN - 1) / TL::BN;
  const int ntiles = tiles_n * ((g.M + TL::BM - 1) / TL::BM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;   // the launcher makes gridDim.x a multiple of 8
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int len = q + (xcd < r ? 1 : 0);
  if (slot >= len) return;

  // split-K: slice blockIdx.y multiplies columns [y K, (y+1) K) of A and W into its own partial
  // output [M, ldc] (g.K is the slice length; splitk_reduce_kernel sums the slices)
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A) + (long)blockIdx.y * g.K;
  const TA* __restrict__ W = reinterpret_cast<const TA*>(g.W) + (long)blockIdx.y * g.K;
  TO* __restrict__ C = reinterpret_cast<TO*>(g.C) + (long)blockIdx.y * g.M * g.ldc;
  const int nk = g.K / BK;
  const bool wide = (g.ldc * (long)sizeof(TO)) % 16 == 0;   // rows keep 16-byte alignment
  const bool has_bias = g.bias != nullptr;

  // ---- DMA side: issues k-step copies in (tile, k) order, NS-1 steps ahead of the MFMAs
  int stamp_i = 0;
  auto stamp = [&]() {
#ifndef TAPIR_NO_STAMPS
    if (!TRACE && g.dbg_times != nullptr && tid == 0 && stamp_i < 16)
      g.dbg_times[(long)blockIdx.x * 16 + stamp_i] = wall_clock64();
    ++stamp_i;
#endif
  };
  unsigned long long tk[5] = {0, 0, 0, 0, 0}, tsum[5] = {0, 0, 0, 0, 0}, tstart = 0;
  auto tick = [&](int k) {   // the value arrives asynchronously (SMEM): read it after tick_sync()
#ifndef TAPIR_HIPEMU
    if (TRACE) asm volatile("s_memtime %0" : "=s"(tk[k]) :: "memory");
#endif
  };
  auto tick_sync = [&]() {
#ifndef TAPIR_HIPEMU
    if (TRACE)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tk[0]), "+s"(tk[1]), "+s"(tk[2]), "+s"(tk[3]), "+s"(tk[4]) :: "memory");
#endif
  };
  if (TRACE) { tick(0); tick_sync(); tstart = tk[0]; }
  stamp();   // 0: start
  GemmStager<TA, TL, PAIR> st;
  st.local = slot; st.kt = 0;
  st.set_tile(g, ((start + slot) / tiles_n) * TL::BM, ((start + slot) % tiles_n) * TL::BN, tid);
  auto issue_next = [&](uint4* dst) -> bool {   // false once every k-step of this workgroup is issued
    if (st.local >= len) return false;
    st.issue(A, W, st.kt * BK, wave_u, dst);
    if (++st.kt == nk) {
      st.kt = 0;
      st.local += per_xcd;
      if (st.local < len)
        st.set_tile(g, ((start + st.local) / tiles_n) * TL::BM,
                    ((start + st.local) % tiles_n) * TL::BN, tid);
    }
    return true;
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_next(bufs[s]);
  // bias of the first tile (requested after the first copies are under way)
  int bias_n0 = ((start + slot) % tiles_n) * TL::BN;
  f32x4 bias4[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_bias)
      bias4[j] = *reinterpret_cast<const f32x4*>(
          g.bias + min(bias_n0 + wn * FN * 16 + TL::template col<PAIR>(j, fg), g.N - 4));
  }

  // Pipeline depth.  NS >= 3: the fragments of the first half of k-step t+1 are read (into
  // registers) during k-step t, BEFORE the barrier that ends it, so the MFMAs restart right after
  // the barrier instead of waiting for an LDS round trip of all eight waves at once.  That needs
  // stage t+1 visible during step t: each step ends by waiting for the copies of stage t+2
  // (AHEAD = 2), which leaves NS-3 stages in flight across the barrier.  NS == 2: plain double
  // buffer (AHEAD = 1, nothing in flight across the barrier).
  constexpr bool PREFETCH = TL::PREFETCH;
  constexpr int AHEAD = PREFETCH ? 2 : 1;
  constexpr int INFLIGHT = (NS - 1 - AHEAD) * TL::CH;
  // accumulators of the first tile (EPI_BIAS_RESID: residual loads, in flight with the first copies)
  f32x4 acc[FM][FN];
  gemm_init_acc<TO, EPI, TL>(g, acc, ((start + slot) / tiles_n) * TL::BM + wm * FM * 16 + fr,
                             ((start + slot) % tiles_n) * TL::BN + wn * FN * 16, fg);
  if (st.local < len) dma_wait<INFLIGHT>(); else dma_wait<0>();
  // "use" the bias here so that the compiler's wait for it sits here and not in every epilogue
#pragma unroll
  for (int j = 0; j < FN; ++j) consume(bias4[j]);
  block_barrier();
  stamp();   // 1: first stage landed

  // ---- MFMA side.  One k-step: refill the stage read in the previous step, multiply from stage
  // S, wait until this wave's copies of stage t+AHEAD have landed, barrier.
  uint4 fa0[FM], fw0[FN];   // first-half fragments of the current stage (NS >= 3: prefetched)
  if (PREFETCH) gemm_load_frags<TL, PAIR>(bufs[0], a_row, w_row, fr, fg, 0, fa0, fw0);
  auto step = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    uint4 fa1[FM], fw1[FN];
    bool issued;
    if (PREFETCH) {
      issued = issue_next(bufs[(S + NS - 1) % NS]);
      tick(1);
      sched_fence();   // keep the copy's address arithmetic out of the fragment live ranges
      gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 1, fa1, fw1);
      gemm_mfma_frags<TA, TL>(fa0, fw0, acc);
      gemm_load_frags<TL, PAIR>(bufs[(S + 1) % NS], a_row, w_row, fr, fg, 0, fa0, fw0);
      gemm_mfma_frags<TA, TL>(fa1, fw1, acc);
    } else {
      // Order of a k-step (every sched_fence pins it): first-half fragment reads; the copies of a
      // later stage (their issue takes hundreds of cycles, the LDS round trip hides under it);
      // arrival of the first half; second-half reads; first-half MFMAs (second half in flight);
      // arrival of the second half; second-half MFMAs.  Without the arrival points the compiler
      // interleaves pairs of reads with pairs of MFMAs and, because it answers LDS-DMA in flight
      // with s_waitcnt lgkmcnt(0) only, exposes a full LDS round trip four times per k-step.
      gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 0, fa0, fw0);
      sched_fence();
      issued = issue_next(bufs[(S + NS - 1) % NS]);
      tick(1);
      sched_fence();
#pragma unroll
      for (int j = 0; j < FN; ++j) arrive(fw0[j]);
#pragma unroll
      for (int i = 0; i < FM; ++i) arrive(fa0[i]);
      sched_fence();
      // tiles with 128 accumulator registers (256x256) cannot also hold both halves' fragments:
      // their second-half reads follow the first-half MFMAs (32 MFMAs of the SIMD's other wave cover them)
      constexpr bool SEQ = FM * FN >= 32;
      if (!SEQ) gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 1, fa1, fw1);
      sched_fence();
      gemm_mfma_frags<TA, TL>(fa0, fw0, acc);
      sched_fence();
      if (SEQ) { gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 1, fa1, fw1); sched_fence(); }
#pragma unroll
      for (int j = 0; j < FN; ++j) arrive(fw1[j]);
#pragma unroll
      for (int i = 0; i < FM; ++i) arrive(fa1[i]);
      gemm_mfma_frags<TA, TL>(fa1, fw1, acc);
    }
    sched_fence();
    // every LDS read above has returned (lgkmcnt) before another wave may refill what it read
    if (TRACE) { tick_sync(); tick(2); }
    if (issued) dma_lds_wait<INFLIGHT>(); else dma_lds_wait<0>();
    if (TRACE) { tick(3); tick_sync(); }
    block_barrier();
    if (TRACE) {
      tick(4); tick_sync();
      tsum[0] += tk[1] - tk[0]; tsum[1] += tk[2] - tk[1]; tsum[2] += tk[3] - tk[2]; tsum[3] += tk[4] - tk[3];
      tk[0] = tk[4];
    }
  };

  int phase = 0;   // stage that holds the current k-step
  for (int local = slot; local < len; local += per_xcd) {
    // this lane's output coordinates: C[m = mb + 16 i][n = nb + col(j, fg) + 0..3]
    const int m0 = ((start + local) / tiles_n) * TL::BM;
    const int n0 = ((start + local) % tiles_n) * TL::BN;
    const int mb = m0 + wm * FM * 16 + fr;
    const int nb = n0 + wn * FN * 16;   // first column of this wave
    int kt = 0;
    if (TRACE) { tick(0); tick_sync(); if (local != slot) tsum[4] += tk[0] - tk[4]; }
    while (kt < nk) {
      switch (phase) {
        case 0:
          step(StageTag<0>{});
          if (++kt == nk) { phase = 1 % NS; break; }
          [[fallthrough]];
        case 1:
          step(StageTag<1>{});
          if (NS == 2) { ++kt; phase = 0; break; }
          if (++kt == nk) { phase = 2 % NS; break; }
          [[fallthrough]];
        case 2:
          if constexpr (NS > 2) {
            step(StageTag<2>{});
            if (NS == 3) { ++kt; phase = 0; break; }
            if (++kt == nk) { phase = 3 % NS; break; }
          }
          [[fallthrough]];
        default:
          if constexpr (NS > 3) {
            step(StageTag<3>{});
            ++kt; phase = 0;
          }
          break;
      }
    }

    stamp();   // 2 + 2 i: k-loop of tile i done
    // epilogue
    // Bias: normally still in registers from the previous tile (the persistent schedule gives a
    // workgroup tiles of ONE column block whenever tiles_n divides the per-XCD stride).  A reload
    // is a VGPR-destination load whose first use makes the compiler drain vmcnt, i.e. also the
    // copies in flight for the next tile -- correct, just slower, and rare.
    if (n0 != bias_n0) {
      bias_n0 = n0;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        if (has_bias)
          bias4[j] = *reinterpret_cast<const f32x4*>(g.bias + min(nb + TL::template col<PAIR>(j, fg), g.N - 4));
#pragma unroll
      for (int j = 0; j < FN; ++j) consume(bias4[j]);
    }
    // interior tiles with 16-byte aligned rows take a branch-free path (no per-lane predicates:
    // the compiler otherwise sinks the GELU arithmetic into a maze of masked store blocks)
    const bool interior = wide && m0 + TL::BM <= g.M && n0 + TL::BN <= g.N;
    if (interior) gemm_epilogue<TO, EPI, TL, true>(g, C, acc, bias4, mb, nb, fg);
    else gemm_epilogue<TO, EPI, TL, false>(g, C, acc, bias4, mb, nb, fg);
    stamp();   // 3 + 2 i: epilogue of tile i issued
    if (local + per_xcd < len) {   // accumulators of the next tile
      const int nl = local + per_xcd;
      gemm_init_acc<TO, EPI, TL>(g, acc, ((start + nl) / tiles_n) * TL::BM + wm * FM * 16 + fr,
                                 ((start + nl) % tiles_n) * TL::BN + wn * FN * 16, fg);
    }
  }
  if (TRACE && g.dbg_times != nullptr) {
    tick(0); tick_sync();
    tsum[4] += tk[0] - tk[4];
    if (lane == 0) {
      long long* o = g.dbg_times + ((long)blockIdx.x * 16 + wave) * 8;
      for (int k = 0; k < 5; ++k) o[k] = (long long)tsum[k];
      o[5] = (long long)(tk[0] - tstart);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant: PW producer waves issue every LDS-DMA copy, the WM x WN consumer
// waves only read fragments and issue MFMAs.  tools/mfma_probe.hip puts the price of the copies
// at 20 % of the k-loop although the DMA itself runs at L2 speed: a global_load_lds costs the
// issuing wave 60-180 cycles, during which it feeds no MFMA.  With the roles split, the consumer
// stream holds no vector-memory instruction at all (so the stores of its epilogue cannot perturb
// any vmcnt accounting either), and the producers run up to NS-1 stages ahead, through tile
// boundaries and under the consumers' epilogue.  Every wave executes one s_barrier per k-step:
//   producer, step t: copies of stage t+NS-1 (into the buffer read in step t-1), then wait until its
//                     copies of stage t+1 have landed (vmcnt), barrier
//   consumer, step t: fragments + MFMAs of stage t, LDS reads returned (lgkmcnt), barrier
#ifdef TAPIR_EXPERIMENTS
template <typename TA, typename TO, int EPI, typename TL, int PW>
__global__ __launch_bounds__(TL::THREADS + PW * 64) void gemm_ws_kernel(GemmArgs g) {
  constexpr int EPC = 16 / (int)sizeof(TA);
  constexpr int BK = 8 * EPC;
  constexpr int NS = TL::NS, FM = TL::FM, FN = TL::FN;
  constexpr int PT = PW * 64;                       // producer threads
  constexpr int CH_A = TL::BM * 8 / PT, CH_W = TL::BN * 8 / PT;
  static_assert(TL::BM * 8 % PT == 0 && TL::BN * 8 % PT == 0 && NS >= 3, "producer split");
  constexpr bool PAIR = sizeof(TO) == 2;
  __shared__ uint4 lds0[TL::STAGE_CHUNKS];
  __shared__ uint4 lds1[TL::STAGE_CHUNKS];
  __shared__ uint4 lds2[TL::STAGE_CHUNKS];
  __shared__ uint4 lds3[NS > 3 ? TL::STAGE_CHUNKS : 1];
  uint4* const bufs[4] = {lds0, lds1, lds2, lds3};

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave_u >= TL::WM * TL::WN;

  const int tiles_n = (g.N + TL::BN - 1) / TL::BN;
  const int ntiles = tiles_n * ((g.M + TL::BM - 1) / TL::BM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int len = q + (xcd < r ? 1 : 0);
  if (slot >= len) return;
  const int nk = g.K / BK;

  if (producer) {
    // ------------------------------------------------------------------ producer waves
    const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
    const TA* __restrict__ W = reinterpret_cast<const TA*>(g.W);
    const int pt = tid - TL::THREADS;                 // 0 .. PT-1
    const int pw = wave_u - TL::WM * TL::WN;          // producer wave index
    long offA[CH_A], offW[CH_W];
    auto set_tile = [&](int local) {
      const int m0 = ((start + local) / tiles_n) * TL::BM, n0 = ((start + local) % tiles_n) * TL::BN;
      const int t = opaque(pt);
#pragma unroll
      for (int s = 0; s < CH_A; ++s) {
        const int id = t + PT * s;
        const int row = id >> 3, p = id & 7;
        offA[s] = (long)min(m0 + row, g.M - 1) * g.lda + (p ^ (row & 7)) * EPC;
      }
#pragma unroll
      for (int s = 0; s < CH_W; ++s) {
        const int id = t + PT * s;
        const int row = id >> 3, p = id & 7;
        offW[s] = (long)min(n0 + row, (g.w_rows ? g.w_rows : g.N) - 1) * g.ldw + (p ^ TL::template swz_w<PAIR>(row)) * EPC;
      }
    };
    int i_local = slot, i_kt = 0;
    set_tile(i_local);
    auto issue_next = [&](uint4* dst) -> bool {
      if (i_local >= len) return false;
      const int k0 = i_kt * BK;
#pragma unroll
      for (int s = 0; s < CH_A; ++s) glds16(A + offA[s] + k0, dst + pw * 64 + PT * s);
#pragma unroll
      for (int s = 0; s < CH_W; ++s) glds16(W + offW[s] + k0, dst + TL::BM * 8 + pw * 64 + PT * s);
      if (++i_kt == nk) {
        i_kt = 0;
        i_local += per_xcd;
        if (i_local < len) set_tile(i_local);
      }
      return true;
    };
    constexpr int INFLIGHT = (NS - 2) * (CH_A + CH_W);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_next(bufs[s]);
    if (i_local < len) dma_wait<INFLIGHT>(); else dma_wait<0>();
    block_barrier();
    auto pstep = [&](auto tag) {
      constexpr int S = decltype(tag)::value;
      const bool issued = issue_next(bufs[(S + NS - 1) % NS]);
      if (issued) dma_wait<INFLIGHT>(); else dma_wait<0>();
      block_barrier();
    };
    long steps = 0;   // k-steps of this workgroup
    for (int local = slot; local < len; local += per_xcd) steps += nk;
    int phase = 0;
    for (long t = 0; t < steps; ++t) {
      switch (phase) {
        case 0: pstep(StageTag<0>{}); break;
        case 1: pstep(StageTag<1>{}); break;
        case 2: pstep(StageTag<2>{}); break;
        default: if constexpr (NS > 3) pstep(StageTag<3>{}); break;
      }
      phase = phase + 1 == NS ? 0 : phase + 1;
    }
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  TO* __restrict__ C = reinterpret_cast<TO*>(g.C);
  const int wm = wave_u / TL::WN, wn = wave_u % TL::WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_row = wm * FM * 16 + fr;
  const int w_row = wn * FN * 16;
  const bool wide = (g.ldc * (long)sizeof(TO)) % 16 == 0;
  const bool has_bias = g.bias != nullptr;
  f32x4 acc[FM][FN];
  block_barrier();   // stage 0 has landed (pairs with the producers' first barrier)
  auto cstep = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    uint4 fa0[FM], fw0[FN], fa1[FM], fw1[FN];
    gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 0, fa0, fw0);
    gemm_load_frags<TL, PAIR>(bufs[S], a_row, w_row, fr, fg, 1, fa1, fw1);
    // ALL fragment reads of the k-step are issued before its first MFMA: left alone, the scheduler
    // interleaves pairs of reads with pairs of MFMAs to save registers and the wave then sits
    // through an LDS round trip (s_waitcnt lgkmcnt(0)) four times per k-step instead of once
    sched_fence();
    gemm_mfma_frags<TA, TL>(fa0, fw0, acc);
    gemm_mfma_frags<TA, TL>(fa1, fw1, acc);
    lds_barrier();   // reads returned before a producer may refill this stage
  };
  int phase = 0;
  for (int local = slot; local < len; local += per_xcd) {
    gemm_init_acc<TO, EPI, TL>(g, acc, ((start + local) / tiles_n) * TL::BM + wm * FM * 16 + fr,
                               ((start + local) % tiles_n) * TL::BN + wn * FN * 16, fg);
    for (int kt = 0; kt < nk; ++kt) {
      switch (phase) {
        case 0: cstep(StageTag<0>{}); break;
        case 1: cstep(StageTag<1>{}); break;
        case 2: cstep(StageTag<2>{}); break;
        default: if constexpr (NS > 3) cstep(StageTag<3>{}); break;
      }
      phase = phase + 1 == NS ? 0 : phase + 1;
    }
    const int m0 = ((start + local) / tiles_n) * TL::BM;
    const int n0 = ((start + local) % tiles_n) * TL::BN;
    const int mb = m0 + wm * FM * 16 + fr;
    const int nb = n0 + wn * FN * 16;
    f32x4 bias4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_bias)
        bias4[j] = *reinterpret_cast<const f32x4*>(g.bias + min(nb + TL::template col<PAIR>(j, fg), g.N - 4));
    }
    const bool interior = wide && m0 + TL::BM <= g.M && n0 + TL::BN <= g.N;
    if (interior) gemm_epilogue<TO, EPI, TL, true>(g, C, acc, bias4, mb, nb, fg);
    else gemm_epilogue<TO, EPI, TL, false>(g, C, acc, bias4, mb, nb, fg);
  }
}

#endif  // TAPIR_EXPERIMENTS

// Tile shapes.  Slots = workgroups per CU (LDS-limited) x 256 CUs.
enum { GEMM_TILE_AUTO = 0, GEMM_TILE_192x128 = 1, GEMM_TILE_128x128 = 2, GEMM_TILE_192x64 = 3,
       GEMM_TILE_192x128_S3 = 4, GEMM_TILE_192x128_WS = 5, GEMM_TILE_192x256 = 6, GEMM_TILE_256x128 = 7,
       GEMM_TILE_128x128_W8 = 8, GEMM_TILE_192x64_W8 = 9, GEMM_TILE_128x64_W8 = 10,
       GEMM_TILE_256x128_W16_PF = 11, GEMM_TILE_128x128_W8_PF = 12, GEMM_TILE_256x128_W16 = 13,
       GEMM_TILE_256x128_W16_S3 = 14, GEMM_TILE_128x128_W8_S3 = 15, GEMM_TILE_256x256 = 16,
       GEMM_TILE_COUNT = 17 };
typedef GemmTile<4, 2, 3, 4, 4> GemmTileBig;     // 192x128, 8 waves, 4 stages = 160 KiB: 1 per CU
typedef GemmTile<4, 2, 3, 4, 3> GemmTileBig3;    // same, 3 stages = 120 KiB
typedef GemmTile<2, 2, 4, 4, 2> GemmTileSquare;  // 128x128, 4 waves, 64 KiB: 2 per CU
typedef GemmTile<2, 2, 6, 2, 2> GemmTileTall;    // 192x64, 4 waves, 64 KiB: 2 per CU
typedef GemmTile<2, 4, 6, 4, 2> GemmTileWide;    // 192x256, 8 waves, 2 stages = 112 KiB: least LDS fill per flop
typedef GemmTile<4, 2, 4, 4, 3> GemmTileLong;    // 256x128, 8 waves, 3 stages = 144 KiB
typedef GemmTile<2, 4, 4, 2, 2> GemmTileSquare8; // 128x128, 8 waves (64x32 each), 64 KiB: 2 per CU = 16 waves
typedef GemmTile<4, 2, 3, 2, 2> GemmTileTall8;   // 192x64, 8 waves (48x32 each), 64 KiB: 2 per CU = 16 waves
typedef GemmTile<4, 2, 2, 2, 2> GemmTileSmall8;  // 128x64, 8 waves (32x32 each), 48 KiB: 3 per CU = 24 waves
// sixteen waves in ONE workgroup per CU (4 per SIMD), 3 stages, fragments of step t+1 read during step t
typedef GemmTile<4, 4, 4, 2, 3, true> GemmTileLong16P;    // 256x128, 144 KiB
typedef GemmTile<2, 4, 4, 2, 3, true> GemmTileSquare8P;   // 128x128, 8 waves, 96 KiB
typedef GemmTile<4, 4, 4, 2, 2> GemmTileLong16;           // 256x128, 2 stages = 96 KiB
typedef GemmTile<4, 4, 4, 2, 3> GemmTileLong16S3;         // 256x128, 3 stages (one in flight across the barrier)
typedef GemmTile<2, 4, 4, 2, 3> GemmTileSquare8S3;        // 128x128, 8 waves, 3 stages = 96 KiB: 1 per CU
typedef GemmTile<2, 4, 8, 4, 2> GemmTileHuge;             // 256x256, 8 waves (128x64 each), 2 stages = 128 KiB

// Measured on MI355X at the config-2 shapes (tools/kbench.py, profiles/r01_kbench_gemm.log,
// profiles/r01_gemm_step_trace.txt): every configuration with 16 waves per CU lands within 5 % of
// the others whatever its tile, stage count or DMA / MFMA role split (192x128 4-stage,
// wave-specialised, 192x256 with half the LDS fill per flop, 256x128 with 16 waves and 3 stages,
// fragment prefetch, copies interleaved with the MFMAs ...): the per-wave cycle trace shows a wave
// spending 17 % of its life in fragment reads + MFMAs and the rest issuing copies, waiting for
// them or at the barrier -- the k-loop runs at the rate the CU's vector-memory path delivers the
// operands, and re-arranging the instructions only moves the waiting from one phase to another.
// Narrow outputs with a long K (N <= 1024) run best on 192x64, wide GELU outputs on 128x128,
// both with EIGHT waves and two workgroups per CU.
// Long-K narrow outputs (mlp2_down) and everything at >= 24576 rows (BootsTAPIR-size query sets, the
// high-resolution config) run best on 256x128 with SIXTEEN waves in one workgroup per CU and three
// stages (one in flight across the barrier): same-box A/B, profiles/r01_ab_bigtile.log: mlp2_down
// 39.8 -> 37.9 us at 12288 rows and 176 -> 136 us at 49152; mlp2_up 156 -> 146 us at 49152 but
// 33.1 -> 33.8 us at 12288 rows, where it keeps the 128x128 tile.
inline int gemm_pick_tile(int M, int N, int K) {
  // wide outputs at >= 24576 rows: 256x256 (8 waves, 128x64 each: least copy volume and fewest LDS
  // reads per MFMA; 1536 tiles at config-3 size = 6 per CU): mlp2_up 158 -> 144 us standalone
  if (M >= 24576 && N >= 2048) return GEMM_TILE_256x256;
  if (N >= 512 && (M >= 24576 || (N <= 1024 && K >= 1024 && M >= 6144))) return GEMM_TILE_256x128_W16_S3;
  // short K, very wide N (the cost volume as a GEMM writing the f32 volume, K = 256, N = T h w): the launch is bound by what
  // it stores; the largest tiles re-read the grid least (profiles/r05_kbench_contraction.txt: 84 us at M = 682 against
  // 104 us with 128x128; 505 against 732 us at M = 4096)
  if (K <= 256 && N >= 16384 && M >= 128) return M >= 2048 ? GEMM_TILE_256x256 : GEMM_TILE_256x128_W16_S3;
  if ((M % 192 == 0 || M >= 192 * 32) && N <= 1024) return GEMM_TILE_192x64;
  return GEMM_TILE_128x128_W8;
}

template <typename TA, typename TO, int EPI, typename TL>
inline void launch_gemm_tile(const GemmArgs& g, hipStream_t stream, int max_grid, int splits = 1) {
  const int ntiles = ((g.N + TL::BN - 1) / TL::BN) * ((g.M + TL::BM - 1) / TL::BM);
  const int per_cu = std::max(1, std::min(3, (160 * 1024) / TL::LDS_BYTES));
  int grid = std::min((ntiles + 7) / 8 * 8, 256 * per_cu);
  if (max_grid > 0) grid = std::min(grid, (max_grid + 7) / 8 * 8);
  TAPIR_LAUNCH((gemm_nt_kernel<TA, TO, EPI, TL>), dim3(grid, splits), dim3(TL::THREADS), stream, g);
}

// the same launch with the per-k-step cycle trace compiled in (debug entry only)
#ifdef TAPIR_EXPERIMENTS
template <typename TA, typename TO, int EPI, typename TL>
inline void launch_gemm_tile_traced(const GemmArgs& g, hipStream_t stream, int max_grid) {
  const int ntiles = ((g.N + TL::BN - 1) / TL::BN) * ((g.M + TL::BM - 1) / TL::BM);
  const int per_cu = std::max(1, std::min(3, (160 * 1024) / TL::LDS_BYTES));
  int grid = std::min((ntiles + 7) / 8 * 8, 256 * per_cu);
  if (max_grid > 0) grid = std::min(grid, (max_grid + 7) / 8 * 8);
  hipLaunchKernelGGL((gemm_nt_kernel<TA, TO, EPI, TL, true>), dim3(grid), dim3(TL::THREADS), 0, stream, g);
}
#endif

// ---- split-K for few rows (the online model: M = points x 1 frame).  With M = 256 a mixer GEMM is
// 8-32 tiles of up to 32 DEPENDENT k-steps on a 256-CU chip; slicing K puts 64 workgroups of 4 k-steps
// each to work, and an element-wise pass adds the slices, the bias and the epilogue.
struct SplitKReduceArgs {
  const float* part;     // [splits, M, N]
  const float* bias;     // [N] or null
  const float* resid; long ldr;
  void* C; long ldc;
  int M, N, splits;
};
template <typename TO, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKReduceArgs a) {
  const long total = (long)a.M * (a.N / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / (a.N / 4)), n = (int)(i % (a.N / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias != nullptr) v = *reinterpret_cast<const float4*>(a.bias + n);
    for (int s = 0; s < a.splits; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(a.part + ((long)s * a.M + m) * a.N + n);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (EPI == EPI_BIAS_GELU) { v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w); }
    if (EPI == EPI_BIAS_RESID) {
      const float4 r = *reinterpret_cast<const float4*>(a.resid + (long)m * a.ldr + n);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    Store4<TO>::run(reinterpret_cast<TO*>(a.C) + (long)m * a.ldc + n, v.x, v.y, v.z, v.w);
  }
}

// number of K slices for a GEMM of M rows (1 = no split): slices of >= 4 k-steps, <= 8 slices
template <typename TA>
inline int gemm_splits(int M, int K) {
  const int kstep = 128 / (int)sizeof(TA);
  if (M > 512) return 1;
  int best = 1;
  for (int s = 2; s <= 8; ++s)
    if (K % (s * kstep) == 0 && K / (s * kstep) >= 3) best = s;
  return best;
}

// C = epi(A W^T + bias) through `splits` K slices; part: f32 workspace [splits, M, N]
template <typename TA, typename TO, int EPI>
inline void launch_gemm_splitk(const GemmArgs& g, int splits, float* part, hipStream_t stream) {
  GemmArgs p = g;
  p.bias = nullptr; p.resid = nullptr;
  p.C = part; p.ldc = g.N; p.K = g.K / splits;
  launch_gemm_tile<TA, float, EPI_BIAS, GemmTileSquare8>(p, stream, 0, splits);
  SplitKReduceArgs r{part, g.bias, g.resid, g.ldr, g.C, g.ldc, g.M, g.N, splits};
  const long total = (long)g.M * (g.N / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<TO, EPI>), dim3((unsigned)std::min<long>((total + 255) / 256, 2048)),
                     dim3(256), 0, stream, r);
}

// ---- few-row GEMM in ONE launch (the online model: M = tracked points x 1 frame = 256 rows).  The
// split-K pair above is two dependent launches (tiled kernel with LDS-DMA prologue + element-wise reduce:
// 8.8 + 5 us for 0.5 GFLOP) 26 times per refinement iteration, 4 iterations per frame.  At this size
// nothing is bandwidth- or MFMA-bound; what counts is the number of dependent launches and round trips.
// gemm_small_kernel: a workgroup of 4 waves owns a tile of FM*16 rows x FN*16 columns over the WHOLE K; the
// waves split K (wave w takes k-steps w, w + 4, ...), each keeps the whole tile's accumulators, operands go
// global -> registers straight in MFMA fragment layout (16 bytes per lane, no LDS staging: a fragment is
// read by exactly one wave), DEPTH k-steps of loads in flight; the four partial tiles meet in LDS and every
// wave finishes a quarter of the fragments (bias, GELU / residual, store).  Tiles are small (32 x 64 /
// 32 x 32) so that 100-256 workgroups are in flight for M = 256.  Measured on MI355X (rocprofv3, online
// step): 8.4 us (K = 512) / 11.1 us (K = 2048) per launch against 8.8 + 5 us for the pair; 3.20 -> 2.89 ms
// per frame.  (An 8-wave form with ALL operand loads of a wave issued up front -- one round trip instead
// of four -- measured SLOWER: 14.5 / 10.4 us; every kernel of this launch-bound step, however small, takes
// >= 4-5 us at the clocks the mostly idle chip runs at.)
template <typename TA, typename TO, int EPI, int FM, int FN>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
  constexpr int EPC = 16 / (int)sizeof(TA);          // elements per 16-byte chunk
  constexpr int KS = 4 * EPC;                        // k per MFMA step (one chunk per lane group)
  constexpr int NF = FM * FN;
  constexpr int DEPTH = (FM + FN) <= 4 ? 4 : 2;      // k-steps of operand loads in flight per wave
  __shared__ f32x4 s_part[4][NF][64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, gq = lane >> 4;
  const int tiles_n = (g.N + FN * 16 - 1) / (FN * 16);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * FM * 16, n0 = tn * FN * 16;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const TA* __restrict__ W = reinterpret_cast<const TA*>(g.W);
  // this lane's operand rows (clamped: rows past M / N are computed and never stored)
  const TA* pa[FM];
  const TA* pw[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) pa[i] = A + (long)min(m0 + 16 * i + c, g.M - 1) * g.lda + EPC * gq;
#pragma unroll
  for (int j = 0; j < FN; ++j) pw[j] = W + (long)min(n0 + 16 * j + c, (g.w_rows ? g.w_rows : g.N) - 1) * g.ldw + EPC * gq;
  f32x4 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ksteps = g.K / KS;                       // (K is a multiple of 4 KS)
  const int mine = (ksteps - wave + 3) / 4;          // k-steps of this wave: wave, wave + 4, ...
  uint4 fa[DEPTH][FM], fw[DEPTH][FN];
  auto load = [&](int slot, int it) {
    const int k = (wave + 4 * min(it, mine - 1)) * KS;   // past the end: any valid address (not used)
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[slot][i] = *reinterpret_cast<const uint4*>(pa[i] + k);
#pragma unroll
    for (int j = 0; j < FN; ++j) fw[slot][j] = *reinterpret_cast<const uint4*>(pw[j] + k);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, d);
  for (int it0 = 0; it0 < mine; it0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (it0 + d < mine) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int i = 0; i < FM; ++i) MfmaStep<TA>::run(fw[d][j], fa[d][i], acc[j][i]);
      }
      load(d, it0 + d + DEPTH);
    }
  }
  // ---- the four partial tiles meet in LDS; wave w finishes the fragments f = w, w + 4, ...
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) s_part[wave][j * FM + i][lane] = acc[j][i];
  __syncthreads();
#pragma unroll
  for (int f0 = 0; f0 < NF; f0 += 4) {
    const int f = f0 + wave;
    if (f >= NF) break;
    const int j = f / FM, i = f - j * FM;
    f32x4 v = s_part[0][f][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += s_part[w][f][lane];
    // lane (c, gq) holds output columns n .. n + 3 of row m (MFMA C/D layout: W rows on the A port)
    const int m = m0 + 16 * i + c, n = n0 + 16 * j + 4 * gq;
    if (m < g.M && n < g.N) {
      if (g.bias != nullptr) v += *reinterpret_cast<const f32x4*>(g.bias + n);
      if (EPI == EPI_BIAS_GELU) { v[0] = gelu_tanh(v[0]); v[1] = gelu_tanh(v[1]); v[2] = gelu_tanh(v[2]); v[3] = gelu_tanh(v[3]); }
      if (EPI == EPI_BIAS_RESID) v += *reinterpret_cast<const f32x4*>(g.resid + (long)m * g.ldr + n);
      if (EPI == EPI_BIAS_UPDATE) {
        // update_kernel on row r = m of the (never stored) [M, 388] output: columns 0..3 move the state, 4.. the features
        const UpdateArgs& a = g.upd;
        const long r = m;
        if (n == 0) {
          const float px = a.pos[r * 2 + 0] + v[0] * a.sx;
          const float py = a.pos[r * 2 + 1] + v[1] * a.sy;
          const float oc = a.occ[r] + v[2];
          const float ex = a.expd[r] + v[3];
          a.pos[r * 2 + 0] = px; a.pos[r * 2 + 1] = py;
          a.out_tracks[r * 2 + 0] = px * a.vx; a.out_tracks[r * 2 + 1] = py * a.vy;
          a.out_occ[r] = oc; a.out_expd[r] = ex;
          a.occ[r] = a.last_of_level ? a.occ0[r] : oc;
          a.expd[r] = a.last_of_level ? a.expd0[r] : ex;
        } else {
          const int c0 = n - 4;                      // (a multiple of 4: never across the hires | lowres boundary at 128)
          const long bq = r / a.T;
          f32x4 prev;
          if (a.first_of_level)
            prev = c0 < kHiresDim ? *reinterpret_cast<const f32x4*>(a.q_hires + bq * kHiresDim + c0)
                                  : *reinterpret_cast<const f32x4*>(a.q_lowres + bq * kLowresDim + (c0 - kHiresDim));
          else
            prev = *reinterpret_cast<const f32x4*>(a.feats + r * kFeatDim + c0);
          *reinterpret_cast<f32x4*>(a.feats + r * kFeatDim + c0) = v + prev;
        }
      } else {
        Store4<TO>::run(reinterpret_cast<TO*>(g.C) + (long)m * g.ldc + n, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// shapes gemm_small_kernel covers (K a multiple of four MFMA k-steps; N, ldc multiples of 4 like every GEMM here)
template <typename TA>
inline bool gemm_small_supported(int M, int N, int K) {
  const int ks = 4 * (16 / (int)sizeof(TA));
  return M >= 1 && M <= 512 && K >= 4 * ks && K % (4 * ks) == 0 && N % 4 == 0;
}

template <typename TA, typename TO, int EPI>
inline void launch_gemm_small(const GemmArgs& g, hipStream_t stream) {
  const int tm = (g.M + 31) / 32;
  if (g.N >= 1024) {   // 32 x 64 tiles: 256 workgroups for [256, 2048]
    TAPIR_LAUNCH((gemm_small_kernel<TA, TO, EPI, 2, 4>), dim3((unsigned)(tm * ((g.N + 63) / 64))), dim3(256), stream, g);
  } else {             // 32 x 32 tiles: 128 workgroups for [256, 512]
    TAPIR_LAUNCH((gemm_small_kernel<TA, TO, EPI, 2, 2>), dim3((unsigned)(tm * ((g.N + 31) / 32))), dim3(256), stream, g);
  }
}

// ---- the channel MLP of a PIPs block for FEW rows in ONE launch (the online model: M = tracked points x 1 frame).
// tapir_model.py:127-156: x += W_dn gelu(W_up LN2(x) + b_up) + b_dn.  As two gemm_small_kernel launches (up: 8.3 us, down:
// 11.0 us for 256 rows) the pair is 96 of the 148 dependent launches of an online frame's mixer
// (profiles/r06_online_timeline.txt).  What such a launch costs beyond the ~5.5 us of a dependent launch with nothing in it is
// the bytes ONE CU has to pull in cold (96 KB / 256 KB per workgroup for up / down; a first form of this kernel with 544 KB per
// workgroup and four dependent round trips per phase took 33.8 us, profiles/r06_ab_mlp_small.txt) -- so the decomposition
// keeps the bytes per workgroup low with every CU busy, and -- what the persistent form of mixer_online.hpp taught -- the bytes of
// PARTIAL SUMS low, which are written and read back once per block: a workgroup = (tile of 32 rows, group of MLP_HS = 128
// hidden units, one of MLP_CG = 2 halves of the output columns) pulls 128 KB of W_up + 32 KB of LN2(x) + 64 KB of W_dn, ALL of
// it requested before the first MFMA (one round trip), and leaves a [32, 256] piece: 16 partial sums per output element
// (8 MB per block at 256 rows; the first form, 64 hidden units x all 512 columns, left 32: 16 MB).  The two workgroups of a
// hidden group compute the same hidden tile (the first product twice: MFMA time nobody waits for).
//   phase 1: hid[32, 128] = gelu(xn_tile W_up[h0 .. h0 + 128]^T + b_up): the four waves split K = 512 (one quarter each, all
//            of a wave's fragments in flight), meet in LDS, add in wave order, round to the operand type into LDS;
//   phase 2: part[group][32, col0 .. col0 + 256] = hid W_dn[col0 .., h0 .. h0 + 128]^T: wave w takes output columns
//            col0 + 64 w .. + 63, its 16 weight fragments were requested at the top of the kernel.
// Nobody waits and nobody merges at the tail of the launch (the lesson of conv_small.hpp: a merge by the last arriver is ~6 us ON
// the chain of a launch this small): the CONSUMER -- the next block's mix_kernel, or the final layernorm_kernel -- adds the
// 2048 / 128 = 16 partials, b_dn and the residual in a fixed order while it stages its rows (mixer.hpp parts_sum2).  The hidden
// tensor never exists in HBM.  M <= 512 rows: 16 x 32 = 512 workgroups at most.
#ifndef TAPIR_MLP_HS                 // (A/B builds: -DTAPIR_MLP_HS=64 -DTAPIR_MLP_CG=1 is the first decomposition, 32 partial sums)
#define TAPIR_MLP_HS 128
#define TAPIR_MLP_CG 2
#endif
constexpr int MLP_HS = TAPIR_MLP_HS; // hidden units per workgroup
constexpr int MLP_PARTS = 2048 / MLP_HS;   // partial outputs per element (16)
constexpr int MLP_CG = TAPIR_MLP_CG; // column groups of the second product: a workgroup stores 512 / MLP_CG output columns
constexpr int MLP_UNITS = MLP_PARTS * MLP_CG;   // workgroups per row tile: unit u = (hidden group u % MLP_PARTS, column group u / MLP_PARTS)
struct MlpSmallArgs {
  const void* xn;      // [M, 512] LN2(x) in the operand type
  const void* Wup;     // [2048, 512]
  const float* bup;    // [2048]
  const void* Wdn;     // [512, 2048]
  float* part;         // [2048 / MLP_HS][M, 512] f32 partial outputs (no bias, no residual)
  int M;
};
// One (row tile, hidden slice) unit of the kernel above, split into "request the weights", "request the rows" and "compute +
// store", so that the persistent form (mixer_online.hpp) can request a block's weights before it waits for the rows.
template <typename TA>
struct MlpSmallTile {
  static constexpr int EPC = 16 / (int)sizeof(TA);   // elements per 16-byte chunk
  static constexpr int KS = 4 * EPC;                 // k per MFMA step
  static constexpr int K1 = 128 / KS;                // k-steps of a wave's quarter of K = 512 (bf16: 4, f32: 8)
  static constexpr int K2 = MLP_HS / KS;             // k-steps of the second product (bf16: 4, f32: 8)
  static constexpr int NJ1 = MLP_HS / 16;            // fragment rows of the hidden tile (8)
  static constexpr int NF1 = NJ1 * 2;                // its fragments (x 2 row halves)
  static constexpr int CW = 512 / MLP_CG / 4;        // output columns per wave (64)
  static constexpr int NJ2 = CW / 16;                // its fragment rows (4)
  static constexpr int LDH = MLP_HS + EPC;           // hidden row stride in LDS (+ one chunk: rows 16 lanes apart on different banks)
  static constexpr bool EARLY = sizeof(TA) == 2;     // W_dn's fragments requested with W_up's (f32: too many registers; parity build)
  uint4 fw[K1][NJ1], fa[K1][2], fd[K2][NJ2];
  f32x4 fb[NF1 / 4];                                 // the up-projection's bias of the fragments this wave finishes
  int wave, c, gq;
  __device__ __forceinline__ void init() {
    const int lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c = lane & 15; gq = lane >> 4;
  }
  // W_dn rows (output columns) col0 + CW wave + 16 j + c over the group's hidden units
  __device__ __forceinline__ void load_dn(const void* Wdn, int h0, int col0) {
    const TA* Wd = reinterpret_cast<const TA*>(Wdn);
#pragma unroll
    for (int ks = 0; ks < K2; ++ks)
#pragma unroll
      for (int j = 0; j < NJ2; ++j)
        fd[ks][j] = ldg16(Wd + (long)(col0 + CW * wave + 16 * j + c) * 2048 + h0 + ks * KS + EPC * gq);
  }
  // W_up rows h0 + 16 j + c over k in [128 wave, 128 wave + 128)
  __device__ __forceinline__ void load_up(const void* Wup, int h0) {
    const TA* Wu = reinterpret_cast<const TA*>(Wup);
#pragma unroll
    for (int ks = 0; ks < K1; ++ks)
#pragma unroll
      for (int j = 0; j < NJ1; ++j)
        fw[ks][j] = ldg16(Wu + (long)(h0 + 16 * j + c) * 512 + 128 * wave + ks * KS + EPC * gq);
  }
  // bias of the fragments f = wave, wave + 4, ... (requested with the rows: behind the next block's weights it would wait for them)
  __device__ __forceinline__ void load_bias(const float* bup, int h0) {
#pragma unroll
    for (int n = 0; n < NF1 / 4; ++n) fb[n] = ldg_f4(bup + h0 + 16 * ((4 * n + wave) >> 1) + 4 * gq);
  }
  // the tile's rows over the same k range (rows past M: clamped, computed and never stored)
  __device__ __forceinline__ void load_rows(const void* xn, int m0, int M) {
    const TA* A = reinterpret_cast<const TA*>(xn);
#pragma unroll
    for (int ks = 0; ks < K1; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[ks][i] = ldg16(A + (long)min(m0 + 16 * i + c, M - 1) * 512 + 128 * wave + ks * KS + EPC * gq);
  }
#ifndef TAPIR_HIPEMU
  // the same rows handed over by other workgroups of this launch (mixer_online.hpp): loads past L1 (sc1)
  __device__ __forceinline__ void load_rows_shared(__amdgpu_buffer_rsrc_t xn, int m0, int M) {
#pragma unroll
    for (int ks = 0; ks < K1; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = (min(m0 + 16 * i + c, M - 1) * 512 + 128 * wave + ks * KS + EPC * gq) * (int)sizeof(TA);
        fa[ks][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xn, off, 0, 16));
      }
  }
#endif
  // ---- phase 1: this wave's quarter of K into the LDS meeting area (W_up's registers are free afterwards)
  __device__ __forceinline__ void phase1(f32x4 (*s_part)[NF1][64]) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NJ1][2];
#pragma unroll
    for (int j = 0; j < NJ1; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K1; ++ks)
#pragma unroll
      for (int j = 0; j < NJ1; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) MfmaStep<TA>::run(fw[ks][j], fa[ks][i], acc[j][i]);
#pragma unroll
    for (int j = 0; j < NJ1; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) s_part[wave][j * 2 + i][lane] = acc[j][i];
  }
  // ---- the four partial tiles meet: wave w finishes fragments f = w, w + 4, ... (f = 2 j + i): lane (c, gq) holds hidden units
  // h0 + 16 j + 4 gq + e of row 16 i + c; bias, gelu, rounded to the operand type into the hidden tile
  __device__ __forceinline__ void mid(f32x4 (*s_part)[NF1][64], TA* s_hid) {
    const int lane = threadIdx.x & 63;
    lds_barrier();   // (LDS only: __syncthreads() would wait for every global load in flight -- the next block's weights)
#pragma unroll
    for (int f0 = 0; f0 < NF1; f0 += 4) {
      const int f = f0 + wave;
      const int j = f >> 1, i = f & 1;
      f32x4 v = s_part[0][f][lane];
      v = v + s_part[1][f][lane];
      v = v + s_part[2][f][lane];
      v = v + s_part[3][f][lane];
      const int hl = 16 * j + 4 * gq;
      v = v + fb[f0 / 4];
      Store4<TA>::run(&s_hid[(16 * i + c) * LDH + hl], gelu_tanh(v[0]), gelu_tanh(v[1]), gelu_tanh(v[2]), gelu_tanh(v[3]));
    }
    lds_barrier();
  }
  // ---- phase 2: acc = hid W_dn[col0 + CW wave .. + CW - 1, group]^T (W_dn's registers are free afterwards)
  __device__ __forceinline__ void phase2(const TA* s_hid, f32x4 (&acc)[NJ2][2]) {
#pragma unroll
    for (int j = 0; j < NJ2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < K2; ++ks) {
      uint4 fh[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fh[i] = *reinterpret_cast<const uint4*>(&s_hid[(16 * i + c) * LDH + ks * KS + EPC * gq]);
#pragma unroll
      for (int j = 0; j < NJ2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) MfmaStep<TA>::run(fd[ks][j], fh[i], acc[j][i]);
    }
  }
  // part[hg][rows, col0 + CW wave .. + CW - 1]; SHARED: for other workgroups of this launch: write-through (sc1) 16-byte stores
  template <bool SHARED = false>
  __device__ __forceinline__ void store(const f32x4 (&acc)[NJ2][2], float* part, int m0, int hg, int col0, int M
#ifndef TAPIR_HIPEMU
                                        , __amdgpu_buffer_rsrc_t prsrc = __amdgpu_buffer_rsrc_t()
#endif
                                        ) {
    float* out = part + (long)hg * M * 512 + col0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + 16 * i + c;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
#ifndef TAPIR_HIPEMU
          if (SHARED) {
            const int off = (((hg * M + m) * 512) + col0 + CW * wave + 16 * j + 4 * gq) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tapir_u32x4, acc[j][i]), prsrc, off, 0, 16);
            continue;
          }
#endif
          *reinterpret_cast<f32x4*>(out + (long)m * 512 + CW * wave + 16 * j + 4 * gq) = acc[j][i];
        }
      }
    }
  }
};
template <typename TA>
__global__ __launch_bounds__(256) void mlp_small_kernel(MlpSmallArgs g) {
  __shared__ f32x4 s_part[4][MlpSmallTile<TA>::NF1][64];   // the four waves' partial hidden tiles (fragment layout)
  __shared__ __attribute__((aligned(16))) TA s_hid[32 * MlpSmallTile<TA>::LDH];
  const int tm = blockIdx.x / MLP_UNITS, u = blockIdx.x - tm * MLP_UNITS;
  const int hg = u % MLP_PARTS, col0 = (u / MLP_PARTS) * (512 / MLP_CG);   // (workgroups of one hidden group on one XCD)
  MlpSmallTile<TA> t;
  t.init();
  const int h0 = hg * MLP_HS, m0 = tm * 32;
  // every operand of this wave, requested now
  t.load_up(g.Wup, h0);
  if (MlpSmallTile<TA>::EARLY) t.load_dn(g.Wdn, h0, col0);
  t.load_rows(g.xn, m0, g.M);
  t.load_bias(g.bup, h0);
  t.phase1(s_part);
  if (!MlpSmallTile<TA>::EARLY) t.load_dn(g.Wdn, h0, col0);
  t.mid(s_part, s_hid);
  f32x4 acc[MlpSmallTile<TA>::NJ2][2];
  t.phase2(s_hid, acc);
  t.store(acc, g.part, m0, hg, col0, g.M);
}
inline bool mlp_small_supported(int M) { return M >= 1 && M <= 512; }
template <typename TA>
inline void launch_mlp_small(const MlpSmallArgs& g, hipStream_t stream) {
  TAPIR_LAUNCH((mlp_small_kernel<TA>), dim3((unsigned)(((g.M + 31) / 32) * MLP_UNITS)), dim3(256), stream, g);
}

#ifdef TAPIR_EXPERIMENTS
template <typename TA, typename TO, int EPI>
inline void launch_gemm_traced(const GemmArgs& g, hipStream_t stream, int tile, int max_grid) {
  if (tile == GEMM_TILE_AUTO) tile = gemm_pick_tile(g.M, g.N, g.K);
  switch (tile) {
    case GEMM_TILE_192x128: launch_gemm_tile_traced<TA, TO, EPI, GemmTileBig>(g, stream, max_grid); break;
    case GEMM_TILE_192x64: launch_gemm_tile_traced<TA, TO, EPI, GemmTileTall>(g, stream, max_grid); break;
    case GEMM_TILE_192x256: launch_gemm_tile_traced<TA, TO, EPI, GemmTileWide>(g, stream, max_grid); break;
    default: launch_gemm_tile_traced<TA, TO, EPI, GemmTileSquare8>(g, stream, max_grid); break;
  }
}
#endif

// true when `tile` is compiled into this build: the product library carries the four tile shapes
// gemm_pick_tile selects; the other twelve (the tile-shape study of round 1: profiles/r01_kbench_gemm.log)
// and the wave-specialised kernel are built only with -DTAPIR_EXPERIMENTS (tools/kbench.py, the host
// emulator's tile tests).
inline bool gemm_tile_available(int tile) {
#ifdef TAPIR_EXPERIMENTS
  return tile >= 0 && tile < GEMM_TILE_COUNT;
#else
  return tile == GEMM_TILE_AUTO || tile == GEMM_TILE_192x64 || tile == GEMM_TILE_128x128_W8 ||
         tile == GEMM_TILE_256x128_W16_S3 || tile == GEMM_TILE_256x256;
#endif
}

template <typename TA, typename TO, int EPI>
inline void launch_gemm(const GemmArgs& g, hipStream_t stream, int tile = GEMM_TILE_AUTO,
                        int max_grid = 0) {
  if (tile == GEMM_TILE_AUTO) tile = gemm_pick_tile(g.M, g.N, g.K);
  switch (tile) {
    case GEMM_TILE_192x64: launch_gemm_tile<TA, TO, EPI, GemmTileTall>(g, stream, max_grid); break;
    case GEMM_TILE_256x128_W16_S3: launch_gemm_tile<TA, TO, EPI, GemmTileLong16S3>(g, stream, max_grid); break;
    case GEMM_TILE_256x256: launch_gemm_tile<TA, TO, EPI, GemmTileHuge>(g, stream, max_grid); break;
#ifdef TAPIR_EXPERIMENTS
    case GEMM_TILE_192x128: launch_gemm_tile<TA, TO, EPI, GemmTileBig>(g, stream, max_grid); break;
    case GEMM_TILE_192x128_S3: launch_gemm_tile<TA, TO, EPI, GemmTileBig3>(g, stream, max_grid); break;
    case GEMM_TILE_128x128: launch_gemm_tile<TA, TO, EPI, GemmTileSquare>(g, stream, max_grid); break;
    case GEMM_TILE_192x256: launch_gemm_tile<TA, TO, EPI, GemmTileWide>(g, stream, max_grid); break;
    case GEMM_TILE_256x128: launch_gemm_tile<TA, TO, EPI, GemmTileLong>(g, stream, max_grid); break;
    case GEMM_TILE_192x64_W8: launch_gemm_tile<TA, TO, EPI, GemmTileTall8>(g, stream, max_grid); break;
    case GEMM_TILE_128x64_W8: launch_gemm_tile<TA, TO, EPI, GemmTileSmall8>(g, stream, max_grid); break;
    case GEMM_TILE_256x128_W16_PF: launch_gemm_tile<TA, TO, EPI, GemmTileLong16P>(g, stream, max_grid); break;
    case GEMM_TILE_128x128_W8_PF: launch_gemm_tile<TA, TO, EPI, GemmTileSquare8P>(g, stream, max_grid); break;
    case GEMM_TILE_256x128_W16: launch_gemm_tile<TA, TO, EPI, GemmTileLong16>(g, stream, max_grid); break;
    case GEMM_TILE_128x128_W8_S3: launch_gemm_tile<TA, TO, EPI, GemmTileSquare8S3>(g, stream, max_grid); break;
    case GEMM_TILE_192x128_WS: {   // 8 consumer + 4 producer waves, 4 stages, one workgroup per CU
      using TL = GemmTileBig;
      const int ntiles = ((g.N + TL::BN - 1) / TL::BN) * ((g.M + TL::BM - 1) / TL::BM);
      int grid = std::min((ntiles + 7) / 8 * 8, 256);
      if (max_grid > 0) grid = std::min(grid, (max_grid + 7) / 8 * 8);
      TAPIR_LAUNCH((gemm_ws_kernel<TA, TO, EPI, TL, 4>), dim3(grid), dim3(TL::THREADS + 256), stream, g);
      break;
    }
#endif
    default: launch_gemm_tile<TA, TO, EPI, GemmTileSquare8>(g, stream, max_grid); break;
  }
}

}  // namespace tapir
