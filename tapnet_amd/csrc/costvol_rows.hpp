// TAPIR.tracks_from_cost_volume (tapnet/models/tapir_model.py:399-471) as ONE kernel, row-streamed: the
// einsum('bnc,bthwc->tbnhw') :433 is contracted on the matrix cores into LDS (as in costvol_fused.hpp), and
// then EVERY WAVE OWNS WHOLE MAPS: a wave walks its map row by row and keeps everything between the cost map
// and the outputs in its own registers and its own slice of LDS -- no workgroup barrier after the contraction.
//
// Why.  costvol_fused.hpp splits the pixels of ONE map over the eight waves and runs map after map: four
// workgroup barriers per map around phases of 2-5 k cycles each, phases of different maps never overlap, the
// matrix pipes sit at 25 % (profiles/r03_pmc_sq.txt); the exact-f32 MFMA work of a map is 3.6 k cycles of its
// 15.3 k.  Here the two waves of a SIMD are in different phases of different maps at any time, so the VALU /
// LDS work of one runs under the MFMAs of the other, and the only rendezvous of a unit is the one after the
// contraction.
//
// Work unit = (clip b, frame t, tile of QPW / HEADS queries); WAVES waves.
//   G.  cost maps: cells (A port, the frame's grid [h*w, 256] out of L2) x queries (B port, registers) ->
//       s_cm[QPW][(h+2)(w+2)] f32 with a zero halo; all waves; ONE barrier.
//   then wave v takes queries v, v + WAVES, ..., and for each walks the rows y = 0 .. h (one flush step):
//   M1. per 16-pixel tile of row y, two chained exact-f32 MFMA products (v_mfma_f32_16x16x4_f32):
//         D1[ch][px]      = b1 + sum_tap W1[ch][tap] * cm[px + tap]        (:443-444; K = 9 taps padded to 12)
//         P[dy, dx][px]   = sum_ch W2[ch][3 dy + dx] * relu(D1[ch][px])    (:446; K = 16 channels)
//       D1 lands with 4 channels of one pixel per lane = the B-operand layout of the second product.  The
//       rows of the second product are ordered so that LANE GROUP g holds kernel row dy = g and REGISTER r
//       holds kernel column dx = r (row 4 g + r of A2 = tap 3 g + r; g = 3 and r = 3: zero rows).  Then
//         s[x] = P[g,0][x-1] + P[g,1][x] + P[g,2][x+1]          two DPP row shifts, the tile edge from the
//                                                                neighbouring tile (row_ror) via `old`
//       is lane group g's share of logits row y + 1 - g, and the pending sums travel one lane group per row
//       step (ONE ds_bpermute):  A[g] <- s[g] + A[g-1].  After row y, group 2 holds the finished logits of
//       row y - 1 (:446), which overwrite row y - 1 of the cost map in place (dead by then: conv 1 of row
//       y + 1 reads rows y .. y + 2).  The 16 -> 1 convolution never touches LDS (costvol_fused.hpp: 9 LDS
//       writes per pixel + 9 reads per cell and a barrier).  relu(D1) goes to a 4-row ring of the wave
//       ([row][pixel][16 ch], operand type) for the occlusion head.
//   M3. every second row: one output row of conv 16 -> 32 3x3 stride 2 (XLA SAME) :459-461 as an implicit
//       GEMM from the ring (bf16 MFMA, or exact f32 in the parity build), ReLU, running sum.
//   M2. after the last row: softmax(temperature * logits) :454 and the soft arg max with radius 5 around the
//       FIRST maximum :455 (model_utils.py:209-314) from the in-place logits, 16 cells per lane in registers,
//       DPP reductions; query-frame override; mean, Linear 32 -> 16, ReLU, Linear 16 -> 2 :462-470.
// Everything that feeds the soft arg max is exact f32 in both builds.
// LDS (bf16 build): 16 cost maps 72 KiB + 8 rings 34 KiB = 107 KiB; f32 build: 8 maps 36 KiB + rings 68 KiB.
//
// Rows of 33 .. 64 cells run the wide instantiation (PW = 66, below); rows of <= 32 cells taller than 16 register rows
// allow stay on costvol_fused.hpp.
#pragma once
#include "common.hpp"
#include "costvol.hpp"         // CvHeadWeights
#include "costvol_fused.hpp"   // CvFusedArgs, CvFusedCfg, keep_head, mfma_f32
#include "gemm.hpp"            // MfmaStep

namespace tapir {

constexpr int CVR_WAVES = 8;
constexpr int CVR_THREADS = CVR_WAVES * 64;
constexpr int CVR_PW = 34;           // padded row: w <= 32
constexpr int CVR_RING = 4;          // rows of relu(hid1) a wave keeps (3 needed by the stride-2 window)
constexpr int CVR_ZREG = 16;         // logits per lane in the soft-arg-max pass

// DPP row shifts with the row edge taken from `edge` (the lane that has no source keeps `old`).
//   shr1_or(edge, v): lane c <- v[c - 1], lane 0 <- edge[0];   shl1_or(edge, v): lane c <- v[c + 1], lane 15 <- edge[15]
__device__ __forceinline__ float shr1_or(float edge, float v, int lane) {
#ifdef TAPIR_HIPEMU
  const float a = __shfl(v, (lane & 48) | ((lane - 1) & 15));
  return (lane & 15) ? a : edge;
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x111, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float shl1_or(float edge, float v, int lane) {
#ifdef TAPIR_HIPEMU
  const float a = __shfl(v, (lane & 48) | ((lane + 1) & 15));
  return ((lane & 15) != 15) ? a : edge;
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x101, 0xf, 0xf, false));
#endif
}
// rotations inside a 16-lane row: ror1: lane c <- v[(c - 1) & 15] (lane 0 <- lane 15); rol1: lane c <- v[(c + 1) & 15]
__device__ __forceinline__ float ror1(float v, int lane) {
#ifdef TAPIR_HIPEMU
  return __shfl(v, (lane & 48) | ((lane - 1) & 15));
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float rol1(float v, int lane) {
#ifdef TAPIR_HIPEMU
  return __shfl(v, (lane & 48) | ((lane + 1) & 15));
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12F, 0xf, 0xf, false));   // row_ror:15
#endif
}
// value the optimiser cannot see through (keeps a splat from being hoisted as a live vector)
__device__ __forceinline__ float opaque_f32(float v) {
#ifndef TAPIR_HIPEMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}
// lane l <- v[l - 16] (the same column of the previous lane group); lane group 0 <- 0
__device__ __forceinline__ float group_up(float v, int lane) {
#ifdef TAPIR_HIPEMU
  const float a = __shfl(v, (lane - 16) & 63);
#else
  const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane - 16) & 63) << 2, __float_as_int(v)));
#endif
  return lane >= 16 ? a : 0.f;
}

// max(x, m) as ONE instruction: fmaxf() is preceded by a canonicalising v_max_f32 x, x when its argument comes out
// of an MFMA (and the med3 builtin is folded back into that pair), 3 instructions per value in a loop that is bound
// by instruction issue.
__device__ __forceinline__ float max1_f32(float x, float m) {
#ifdef TAPIR_HIPEMU
  return x > m ? x : m;
#else
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
#endif
}
__device__ __forceinline__ float relu_f32(float x) {
#ifdef TAPIR_HIPEMU
  return x > 0.f ? x : 0.f;
#else
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
#endif
}

// NTX = 16-pixel tiles per row (1: w <= 16, 2: w <= 32).  QPW = cost maps per workgroup (<= 16: the contraction's
// B port holds 16 query columns): 16 = two maps per wave, 107 KiB of LDS, one workgroup per CU; 8 = one map per wave,
// 71 KiB, two workgroups per CU (the contraction of one runs under the rows of the other) at twice the grid reads.
// RAGW = false: w is exactly 16 NTX (the selects that zero the pixels past the row end are compiled out).
// WAVES = 8 or 16 waves per workgroup (16: one map per wave at QPW = 16, four waves per SIMD from ONE workgroup at
// half the grid reads of two QPW = 8 workgroups; needs <= 128 VGPRs).
// PW = padded row length of the LDS images: 34 (rows of <= 32 cells, NTX <= 2) or 66 (rows of <= 64 cells, NTX = 3 / 4:
// `initial_resolution` up to 512 x 512 -- 6 maps on 6 waves (4 on 4 in the f32 build) fill the 160 KiB of LDS, the
// logits of the soft-arg-max pass are streamed from LDS twice instead of held in registers, an occlusion row is two tiles).
template <typename TA, int QPW, int NTX, bool TRACE = false, int HEADS = 1, bool RAGW = true, int WAVES = CVR_WAVES,
          int PW = CVR_PW>
// (waves per SIMD asked of the register allocator: the f32 instantiations -- the parity build, nothing timed -- need more
// than 128 VGPRs for the exact-f32 fragments and get two)
__global__ __launch_bounds__(WAVES * 64, PW > CVR_PW ? 1 : ((QPW == 16 && WAVES == 8) || sizeof(TA) == 4) ? 2 : 4) void cv_rows_kernel(CvFusedArgs a) {
  constexpr int THREADS = WAVES * 64;
  // floats per cost map incl. the halo.  Rows of <= 32 cells: 34 x 34 = 1156 rounded up to 1163 = 11 (mod 32): the contraction
  // stores cell 4 g + r of map c from lane (c, g) -- with a map stride of 4 (mod 32) the 16 active lanes of a half wave hit 8
  // banks, with 11 they hit 16 (round 5; the wide form keeps 66 x 66: 6 maps need an even stride for the 16-byte zero fill)
  constexpr int PAD = PW == CVR_PW ? CVF_PAD + 7 : PW * PW;
  constexpr bool ZSTREAM = PW > CVR_PW;                     // soft arg max: two passes over the in-place logits
  constexpr int NOT = NTX > 2 ? 2 : 1;                      // 16-pixel tiles of an occlusion-convolution output row
  static_assert(NTX * 16 <= PW - 2, "tiles per row");
  constexpr int QPT = QPW / HEADS;                   // queries per tile
  static_assert(QPW % HEADS == 0 && (16 / (int)sizeof(TA)) % HEADS == 0, "heads");
  constexpr int EPC = 16 / (int)sizeof(TA);          // elements per 16-byte chunk
  constexpr int KCH = kLowresDim / EPC / 4;          // chunk-steps over K = 256 (4 chunks per step)
  constexpr bool BF = sizeof(TA) == 2;
  constexpr int RING_ROW = PW * 16 * (int)sizeof(TA) / 16;   // uint4 per ring row
  __shared__ __attribute__((aligned(16))) float s_cm[QPW][PAD];                // cost maps (zero halo), then logits in place
  __shared__ uint4 s_ring[WAVES][CVR_RING * RING_ROW];                     // relu(hid1) [row & 3][pixel][16 ch], zero halo
  __shared__ __attribute__((aligned(16))) float s_head[512 + 16 + 32 + 2 + 2];  // w4 [16][32], b4, w5 [2][16], b5
  __shared__ float s_vec[WAVES][32 + 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int h = a.h, w = a.w, hw = h * w;
  const int pw = w + 2;
  const int qtiles = (a.Q + QPT - 1) / QPT;
  // XCD x takes the x-th contiguous eighth of the (frame, query tile) units: a frame's grid comes into ONE L2
  const long units = (long)a.B * a.T * qtiles;
  const long per_xcd = (units + 7) >> 3;
  const long unit = (long)(blockIdx.x & 7u) * per_xcd + (long)(blockIdx.x >> 3);
  if (unit >= units) return;
  const int qt = (int)(unit % qtiles);
  const long frame = unit / qtiles;                  // b * T + t
  const int t = (int)(frame % a.T);
  const long b = frame / a.T;
  const int q0 = qt * QPT;
  const int nq = min(QPT, a.Q - q0);                 // valid queries of this tile

  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int k) {
#ifndef TAPIR_HIPEMU
    if (TRACE) {
      unsigned long long tt;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
      if (k >= 0) tph[k] += tt - tlast;
      tlast = tt;
    }
#endif
  };
  tick(-1);

  // ---- per-lane constants of the two small convolutions (exact-f32 MFMA operands), requested before the grid
  //   conv 1: A1[ch = c][k-slot g] of MFMA j = W1[c][tap 4 j + g]                        (taps >= 9: 0)
  //   conv 2: A2[row c][k-slot g] of MFMA j = W2[ch 4 g + j][tap 3 (c >> 2) + (c & 3)]   (rows with c >> 2 == 3 or c & 3 == 3: 0)
  float a1[HEADS][3], a2[4];     // W1 is [16][HEADS][3][3]
  int off1[3];                   // LDS offset (floats) of tap 4 j + g relative to the pixel's halo index
  f32x4 b1v = f32x4{0.f, 0.f, 0.f, 0.f};
  float b2 = 0.f, b3a = 0.f, b3b = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tc = min(4 * j + g, 8);
    off1[j] = (tc / 3 - 1) * pw + (tc % 3 - 1);
#pragma unroll
    for (int hd = 0; hd < HEADS; ++hd) a1[hd][j] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) a2[j] = 0.f;
  if (!a.raw) {                  // (raw mode has no heads: the weight pointers may be null)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tap = 4 * j + g;
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd) {
        const float v = a.wt.w1[(c * HEADS + hd) * 9 + min(tap, 8)];   // unconditional load, masked on use
        a1[hd][j] = tap < 9 ? v : 0.f;
      }
    }
    const int dy = c >> 2, dx = c & 3;
    const bool live = dy < 3 && dx < 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = a.wt.w2[(4 * g + j) * 9 + min(3 * min(dy, 2) + min(dx, 2), 8)];
      a2[j] = live ? v : 0.f;
    }
    b1v = *reinterpret_cast<const f32x4*>(a.wt.b1 + 4 * g);
    b2 = a.wt.b2[0];
    b3a = a.wt.b3[c]; b3b = a.wt.b3[16 + c];
  }
  // ---- G: cost maps.  B operand: lane (query c, chunk group g) holds chunks 4 s + g of its row.
  {
    const int qrow = min(q0 + (c < QPW ? c / HEADS : 0), a.Q - 1);
    const uint4* qsrc = reinterpret_cast<const uint4*>(
        reinterpret_cast<const TA*>(a.qfeat) + (b * a.Q + qrow) * kLowresDim);
    uint4 fq[KCH];
#pragma unroll
    for (int s = 0; s < KCH; ++s) fq[s] = keep_head<TA, HEADS>(qsrc[4 * s + g], c % HEADS);
    const long gframe = a.frame_map != nullptr ? (long)a.frame_map[frame] : frame;   // which grid this unit's frame is
    const TA* gbase = reinterpret_cast<const TA*>(a.grid) + gframe * (long)hw * kLowresDim;
    const int ntile = (hw + 15) / 16;
    // tile order (bf16 build; pips.hpp PoolArgs::tiled): chunk 4 s + g of cell c of tile `it` is 16-byte piece
    // (it * 32 + 4 s + g) * 16 + c of the frame: the 16 lanes of a lane group read 256 contiguous bytes, the wave 1 KiB
    const bool tiled = BF && a.grid_tiled != nullptr;
    const uint4* tbase = tiled ? reinterpret_cast<const uint4*>(a.grid_tiled) + (gframe * ntile * 32 + g) * 16 + c : nullptr;
    auto load_tile = [&](int it, uint4 (&f)[KCH]) {
      if (tiled) {
        const uint4* csrc = tbase + (long)it * 512;
#pragma unroll
        for (int s = 0; s < KCH; ++s) f[s] = csrc[64 * s];
      } else {
        const int cell = min(it * 16 + c, hw - 1);     // A row of this lane (clamped; masked at the store)
        const uint4* csrc = reinterpret_cast<const uint4*>(gbase + (long)cell * kLowresDim);
#pragma unroll
        for (int s = 0; s < KCH; ++s) f[s] = csrc[4 * s + g];
      }
    };
    // Two tiles in flight per wave.  Every load_tile is UNCONDITIONAL (a tile index past the end is clamped to the last
    // tile: a cache hit whose values are never multiplied): with the loads under `if (next tile exists)` the wait
    // counters merge over both paths at the loop head and hipcc drains the queue -- s_waitcnt vmcnt(1) / vmcnt(0) in
    // front of the MFMAs of the OLDER tile, i.e. no double buffering at all (found in the ISA, round 4).
    uint4 fa0[KCH], fa1[KCH];
    int it = wave;
    const int nmine = it < ntile ? (ntile - 1 - it) / WAVES + 1 : 0;   // tiles of this wave (wave-uniform)
    load_tile(min(it, ntile - 1), fa0);              // the first two tiles fly while the halos are zeroed
    load_tile(min(it + WAVES, ntile - 1), fa1);
    // zero: cost maps (halo cells are never written), rings (halo columns / out-of-image rows)
    static_assert((QPW * PAD) % 4 == 0, "16-byte zero fill");
    for (int i = tid; i < QPW * PAD / 4; i += THREADS) reinterpret_cast<uint4*>(&s_cm[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = lane; i < CVR_RING * RING_ROW; i += 64) s_ring[wave][i] = make_uint4(0u, 0u, 0u, 0u);
    lds_barrier();   // zero fill done before the first cost values land
    // D: lane holds cells itile*16 + 4 g + r of map c (= query c / HEADS, head c % HEADS).  The cell -> (row, column)
    // walk is incremental: one division per lane for the wave's first tile, then + WAVES * 16 cells per step.
    const int stepq = (WAVES * 16) / w, stepr = (WAVES * 16) - stepq * w;
    int cy, cx;
    { const int p0 = it * 16 + 4 * g; cy = p0 / w; cx = p0 - cy * w; }
    const bool row4 = (w & 3) == 0;                  // 4 consecutive cells never leave their row
    float* const mycm = s_cm[c < QPW ? c : 0];
    auto mul_tile = [&](int itile, const uint4 (&f)[KCH]) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KCH; ++s) MfmaStep<TA>::run(f[s], fq[s], acc);
      if (c < nq * HEADS) {
        const int p0 = itile * 16 + 4 * g;
        float* dst = mycm + (cy + 1) * pw + cx + 1;
        if (row4) {
          if (p0 < hw) { dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3]; }   // hw is a multiple of 4 too
        } else {
          int yy = cy, xx = cx;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (p0 + r < hw) mycm[(yy + 1) * pw + xx + 1] = acc[r];
            if (++xx == w) { xx = 0; ++yy; }
          }
        }
      }
      cy += stepq; cx += stepr;
      if (cx >= w) { cx -= w; ++cy; }
    };
    for (int k = 0; k < nmine; k += 2) {
      mul_tile(it, fa0);
      load_tile(min(it + 2 * WAVES, ntile - 1), fa0);
      it += WAVES;
      if (k + 1 < nmine) mul_tile(it, fa1);
      load_tile(min(it + 2 * WAVES, ntile - 1), fa1);
      it += WAVES;
    }
  }
  // head tail (Linear 32 -> 16, Linear 16 -> 2): 562 floats through LDS, all loads of a thread in flight together
  if (!a.raw) {
    const int n5 = a.tapnet ? 16 : 32, nb5 = a.tapnet ? 1 : 2;
    constexpr int PER = (512 + THREADS - 1) / THREADS;
    float hv[PER], hv1 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) hv[k] = tid + k * THREADS < 512 ? a.wt.w4[tid + k * THREADS] : 0.f;
    if (tid < 16) hv1 = a.wt.b4[tid];                  // 16 + n5 + nb5 <= 50 further values
    else if (tid < 16 + n5) hv1 = a.wt.w5[tid - 16];
    else if (tid < 16 + n5 + nb5) hv1 = a.wt.b5[tid - 16 - n5];
#pragma unroll
    for (int k = 0; k < PER; ++k) if (tid + k * THREADS < 512) s_head[tid + k * THREADS] = hv[k];
    if (tid < 16 + n5 + nb5) s_head[512 + (tid < 16 + n5 ? tid : 48 + (tid - 16 - n5))] = hv1;
  }
  // occlusion convolution: stride 2, XLA SAME (pad_lo = total / 2)
  const int oh = (h + 1) / 2, ow = (w + 1) / 2, opix = oh * ow;
  const int ply = max((oh - 1) * 2 + 3 - h, 0) / 2, plx = max((ow - 1) * 2 + 3 - w, 0) / 2;
  uint4 wb[BF ? 5 : 1][2];
#pragma unroll
  for (int s = 0; s < (BF ? 5 : 1); ++s) wb[s][0] = wb[s][1] = make_uint4(0u, 0u, 0u, 0u);
  if (BF && !a.raw) {
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int n = 0; n < 2; ++n) wb[s][n] = a.wt.w3b[(s * 2 + n) * 64 + lane];
  }
  // soft-arg-max pass: lane <-> (row within a group of RPS rows, column): cells of CVR_ZREG row groups
  const int wq = w <= 16 ? 16 : w <= 32 ? 32 : 64, rps = 64 / wq;      // rows per step
  const int zx = lane & (wq - 1), zy = lane / wq;
  const bool zcol = zx < w;
  const float zcx = (float)zx + 0.5f;
  const float zscale = a.raw ? a.temperature : 1.0f;   // raw: the cost map itself is the logit map (x temperature)
  // this lane's pixel column in tile tx, its ring / map offsets
  int px[NTX];
  bool pin[NTX];
#pragma unroll
  for (int tx = 0; tx < NTX; ++tx) { px[tx] = 16 * tx + c; pin[tx] = px[tx] < w; }
  constexpr bool ragw = RAGW;
  bool ovalid[NOT][4];           // output pixels 16 ot + 4 g + r of an occlusion row that exist
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
    for (int r = 0; r < 4; ++r) ovalid[ot][r] = 16 * ot + 4 * g + r < ow;
  constexpr bool full_ow = NTX == 2 && !RAGW;      // w = 32: all 16 output pixels of an occlusion row exist
  const float floor3 = a.tapnet ? -3.0e38f : 0.f;   // TAPIR: ReLU (tapir_model.py:461); TAP-Net: none
  const float b2t = b2 * a.temperature;
  // conv 1: index of tap 4 j + g of this lane's pixel of tile tx relative to halo row y
  int cbase[3][NTX];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) cbase[j][tx] = pw + min(16 * tx + c, w - 1) + 1 + off1[j];
  // Ring rows are stored with the EVEN pixels first and the odd pixels behind them (round 5): the stride-2 window of the
  // occlusion convolution reads pixels 2 c + k of 16 lanes c -- 64 bytes apart in a pixel-major row, 4 of 16 lanes per
  // bank group (a 2-way conflict on every ds_read_b128, the worst ratio of the tree in profiles/r04_pmc_sq.txt); in this
  // order they are 32 bytes apart, and with the two 16-byte halves of the odd pixels swapped the 16 lanes of a pass and
  // the 32 lanes of a ring store hit 64 distinct banks.
  constexpr int RHALF = PW / 2;
  auto rpos = [](int pix) { return (pix >> 1) + (pix & 1) * RHALF; };
  // occlusion convolution: this lane's A row of tile ot = output pixel min(16 ot + c, ow - 1); ring column of tap column 0
  int ocol[NOT];
  int otap_row[5], otap_off[NOT][5];   // k-step s: ring row (relative to 2 oy - ply) and uint4 offset inside the row
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot) {
    ocol[ot] = 2 * min(16 * ot + c, ow - 1) - plx + 1;
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      const int tap = min(2 * s5 + (g >> 1), 8);       // tap 9 has zero weights
      otap_row[s5] = tap / 3;
      const int pix = ocol[ot] + tap % 3;
      otap_off[ot][s5] = rpos(pix) * 2 + ((g & 1) ^ (pix & 1));
    }
  }
  uint4* const ring = s_ring[wave];
  float* const vec = s_vec[wave];

  tick(0);
  lds_barrier();   // cost maps complete; the waves part here
  tick(1);
  if (a.raw == 2) {   // tools/kbench.py --what contraction: time the contraction alone (keep one store so that it is not dead code)
    if (tid == 0 && a.points != nullptr) a.points[unit] = s_cm[0][pw + 1];
    return;
  }

  for (int m = wave; m < nq; m += WAVES) {
    float* const cm = s_cm[m * HEADS];               // head hd of this query: cm + hd * PAD
    float pend[NTX];                                 // pending logit sums (lane group g: row y + 1 - g)
    float osum[2] = {0.f, 0.f};                      // sum over this lane's output pixels of relu(conv + b3), channels c, 16 + c
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) pend[tx] = 0.f;
    if (!a.raw) {   // (raw: the soft arg max of the cost map itself, no heads -- TAP-Net's cycle-consistency tracker)
    // ring row of image row -1 (read by the stride-2 window when ply = 1) is all zero
    for (int i = lane; i < RING_ROW; i += 64) ring[((-1) & (CVR_RING - 1)) * RING_ROW + i] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();
    // Software pipeline over the rows (the MFMAs of a row are two dependent chains; a wave issues in order, so
    // everything that waits for a product is placed one stage later, in the shadow of the next product):
    //   iteration y:  move the pending sums one lane group up (bpermute, consumed below)
    //                 B(y)    conv 1 of row y                              -- 3 HEADS MFMAs per tile in flight
    //                 E(y-1)  logit chain of row y-1 from conv 2's result of the PREVIOUS iteration
    //                 G       ReLU + sum of the occlusion row multiplied in the previous iteration
    //                 A(y+1)  cost-map reads of row y+1
    //                 C(y)    ReLU, ring store  (waits for B)
    //                 D(y)    conv 2 of row y                              -- 4 MFMAs per tile in flight
    //                 F       occlusion-convolution row whose last input row is y  -- 10 (bf16) MFMAs in flight
    // The steady-state loop covers the rows inside the image; y = h (the zero row below the image: SAME padding of
    // the stride-2 window) and y = h + 1 drain the pipeline after it.  The kernel is bound by instruction issue as
    // much as by the matrix pipe (profiles/r04_cv_rows_*.txt), so the loop body is kept free of selects that only the
    // drain steps or ragged rows need.
    float cmv[HEADS][3][NTX];
    auto read_cm = [&](int y) {
      const float* row = cm + y * pw;
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int tx = 0; tx < NTX; ++tx)   // pixels past the row end read the halo / the next row: zeroed in C
            cmv[hd][j][tx] = row[hd * PAD + cbase[j][tx]];
    };
    f32x4 d2p[NTX];                                  // conv 2 of the previous row
    f32x4 oc0[NOT], oc1[NOT];                        // occlusion row multiplied in the previous iteration (bias inside)
    bool oc_pending = false;
    // E: s = P[g,0][x-1] + P[g,1][x] + P[g,2][x+1], then the lane-group chain; lane group 2 ends up with the logits of
    // row yl, stored in place of the cost map's row yl (yl = -1: the dead top halo row)
    auto chain = [&](const float (&moved)[NTX], int yl) {
      float zv[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        const float el = tx > 0 ? ror1(d2p[tx - 1][0], lane) : 0.f;          // lane 0 <- lane 15 of the tile to the left
        const float er = tx + 1 < NTX ? rol1(d2p[tx + 1][2], lane) : 0.f;    // lane 15 <- lane 0 of the tile to the right
        const float sv = shr1_or(el, d2p[tx][0], lane) + d2p[tx][1] + shl1_or(er, d2p[tx][2], lane);
        pend[tx] = sv + moved[tx];
        zv[tx] = fmaf(pend[tx], a.temperature, b2t);
      }
      if (g == 2) {
        float* zrow = cm + (yl + 1) * pw + 1;
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx)
          if (!ragw || pin[tx]) zrow[px[tx]] = zv[tx];
      }
    };
    auto occl_sum = [&]() {   // D of the occlusion row: lane holds channel c (and 16 + c), output pixels 16 ot + 4 g + r
#pragma unroll
      for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = max1_f32(oc0[ot][r], floor3), v1 = max1_f32(oc1[ot][r], floor3);
          osum[0] += (full_ow || ovalid[ot][r]) ? v0 : 0.f;
          osum[1] += (full_ow || ovalid[ot][r]) ? v1 : 0.f;
        }
    };
    // F: output row oy of the stride-2 convolution from ring rows 2 oy - ply .. + 2
    auto occl_row = [&](int oy) {
      wave_sync();
      // (the bias splats are rebuilt from ONE register each: hoisted out of the row loop as two 4-register vectors they
      // were spilled under the 128-VGPR cap and reloaded from scratch, with a vmcnt(0), in front of every occlusion row)
      const float ba = opaque_f32(b3a), bb = opaque_f32(b3b);
      const int r0 = 2 * oy - ply;
#pragma unroll
      for (int ot = 0; ot < NOT; ++ot) {
        if (ot * 16 >= ow) { oc0[ot] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; oc1[ot] = oc0[ot]; continue; }
        oc0[ot] = f32x4{ba, ba, ba, ba}; oc1[ot] = f32x4{bb, bb, bb, bb};
        if (BF) {
          // k = tap * 16 + ci (padded to 160): lane group g of k-step s reads channels 8 (g & 1) .. +7 of
          // tap 2 s + (g >> 1): one 16-byte read of the pixel-major bf16 ring
#pragma unroll
          for (int s5 = 0; s5 < 5; ++s5) {
            const int rr = (r0 + otap_row[s5]) & (CVR_RING - 1);
            const uint4 af = ring[rr * RING_ROW + otap_off[ot][s5]];
            oc0[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                              __builtin_bit_cast(bf16x8, wb[s5][0]), oc0[ot], 0, 0, 0);
            oc1[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                              __builtin_bit_cast(bf16x8, wb[s5][1]), oc1[ot], 0, 0, 0);
          }
        } else {
          // exact f32: 36 k-slices of 4: slice j = (tap = j / 4, channels 4 (j % 4) + g)
          const float* h1 = reinterpret_cast<const float*>(ring);
          for (int tap = 0; tap < 9; ++tap) {
            const int rr = (r0 + tap / 3) & (CVR_RING - 1);
            const int pp = rr * PW + rpos(ocol[ot] + tap % 3);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int ci = 4 * jj + g;
              const float av = h1[pp * 16 + ci];
              const float* wr = a.wt.w3 + (ci * 9 + tap) * 32;
              oc0[ot] = mfma_f32(av, wr[c], oc0[ot]);
              oc1[ot] = mfma_f32(av, wr[16 + c], oc1[ot]);
            }
          }
        }
      }
      oc_pending = true;
    };
    // hid1 -> ring: lane holds channels 4 g .. 4 g + 3 of pixel px
    auto ring_store = [&](int y, const f32x4 (&d1)[NTX]) {
      const int rrow = (y & (CVR_RING - 1)) * PW;
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        if (!ragw || pin[tx]) {
          const int pix = 1 + px[tx];                  // (column 0 is the halo)
          const int ridx = rrow + rpos(pix);
          if (BF) {
            uint2 o;
            o.x = pack_bf16x2(d1[tx][0], d1[tx][1]);
            o.y = pack_bf16x2(d1[tx][2], d1[tx][3]);
            reinterpret_cast<uint2*>(ring)[ridx * 4 + (g ^ ((pix & 1) << 1))] = o;   // odd pixels: 16-byte halves swapped
          } else {
            reinterpret_cast<f32x4*>(ring)[ridx * 4 + g] = d1[tx];
          }
        }
      }
    };
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) d2p[tx] = f32x4{0.f, 0.f, 0.f, 0.f};
    read_cm(0);
    for (int y = 0; y < h; ++y) {
      float moved[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) moved[tx] = group_up(pend[tx], lane);
      // ---- B(y)
      f32x4 d1[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) d1[tx] = b1v;
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int tx = 0; tx < NTX; ++tx) d1[tx] = mfma_f32(a1[hd][j], cmv[hd][j][tx], d1[tx]);
      sched_fence();
      // ---- E(y-1), G, A(y+1)
      if (y >= 1) chain(moved, y - 2);
      if (oc_pending) { occl_sum(); oc_pending = false; }
      tick(4);
      if (y + 1 < h) read_cm(y + 1);
      sched_fence();
      // ---- C(y), D(y)
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[tx][r] = relu_f32(d1[tx][r]);
        if (ragw) {   // (wave-uniform) pixels past the row end: zero, like the halo
#pragma unroll
          for (int r = 0; r < 4; ++r) d1[tx][r] = pin[tx] ? d1[tx][r] : 0.f;
        }
      }
      ring_store(y, d1);
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) d2p[tx] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx) d2p[tx] = mfma_f32(a2[j], d1[tx][j], d2p[tx]);
      tick(2);
      // ---- F: the output row whose last input row (2 oy - ply + 2) is y
      const int yy = y + ply - 2;
      if (yy >= 0 && (yy & 1) == 0) occl_row(yy >> 1);
      tick(5);
    }
    // ---- drain, y = h: the chain of row h - 1, the zero row below the image, the last occlusion row
    {
      float moved[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) moved[tx] = group_up(pend[tx], lane);
      chain(moved, h - 2);
      if (oc_pending) { occl_sum(); oc_pending = false; }
      f32x4 zero[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) { zero[tx] = f32x4{0.f, 0.f, 0.f, 0.f}; d2p[tx] = zero[tx]; }
      ring_store(h, zero);
      const int yy = h + ply - 2;
      if (yy >= 0 && (yy & 1) == 0 && (yy >> 1) < oh) occl_row(yy >> 1);
    }
    // ---- drain, y = h + 1: the chain of the zero row finishes the logits of row h - 1
    {
      float moved[NTX];
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) moved[tx] = group_up(pend[tx], lane);
      chain(moved, h - 1);
      if (oc_pending) { occl_sum(); oc_pending = false; }
    }

    }
    // ---- M2: arg max (FIRST maximum: jnp.argmax, model_utils.py:232), softmax window sums
    wave_sync();
    float best = -3.0e38f, bestf = 3.0e9f;   // cell indices are < 2^24: exact as floats
    float red[4] = {0.f, 0.f, 0.f, 0.f};     // sum e, sum x e, sum y e, sum e inside the window
    auto accumulate = [&](float zv, int yk, float ax, float ay, float zmax) {
      const float e = fast_exp(zv - zmax);
      const float ccy = (float)yk + 0.5f;
      red[0] += e;
      const float dd = (zcx - ax) * (zcx - ax) + (ccy - ay) * (ccy - ay);
      if (dd < 25.0f) { red[1] += zcx * e; red[2] += ccy * e; red[3] += e; }   // radius 5, strict (model_utils.py:236)
    };
    if constexpr (!ZSTREAM) {
      float z[CVR_ZREG];
#pragma unroll
      for (int k = 0; k < CVR_ZREG; ++k) {
        const int yk = k * rps + zy;
        const bool in = zcol && yk < h;
        z[k] = in ? cm[(min(yk, h - 1) + 1) * pw + min(zx, w - 1) + 1] * zscale : -3.0e38f;
        best = fmaxf(best, z[k]);
      }
      best = wave_max(best);
#pragma unroll
      for (int k = CVR_ZREG - 1; k >= 0; --k)
        if (z[k] == best) bestf = (float)((k * rps + zy) * w + zx);
      const int besti = (int)(-wave_max(-bestf));
      const float ax = (float)(besti % w) + 0.5f, ay = (float)(besti / w) + 0.5f;
#pragma unroll
      for (int k = 0; k < CVR_ZREG; ++k) {
        const int yk = k * rps + zy;
        if (zcol && yk < h) accumulate(z[k], yk, ax, ay, best);
      }
    } else {
      // rows of up to 64 cells: the logits do not fit the registers -- two passes over the in-place logits
      const float* zrow = cm + pw + min(zx, w - 1) + 1;
      for (int yk = zy; yk < h; yk += rps) {
        const float v = zcol ? zrow[yk * pw] * zscale : -3.0e38f;
        if (v > best) { best = v; bestf = (float)(yk * w + zx); }     // (increasing index: the lane's FIRST maximum)
      }
      const float bw = wave_max(best);
      const int besti = (int)(-wave_max(-(best == bw ? bestf : 3.0e9f)));
      best = bw;
      const float ax = (float)(besti % w) + 0.5f, ay = (float)(besti / w) + 0.5f;
      for (int yk = zy; yk < h; yk += rps)
        if (zcol) accumulate(zrow[yk * pw] * zscale, yk, ax, ay, best);
    }
    wave_sum_n<4>(red);
    const long map = (b * a.Q + q0 + m) * a.T + t;
    if (!a.raw) {
      // occlusion head tail: mean over the output pixels, Linear 32 -> 16 + ReLU, Linear 16 -> 2 (:462-470)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        osum[k] += __shfl_xor(osum[k], 16);
        osum[k] += __shfl_xor(osum[k], 32);
      }
      if (lane < 16) { vec[lane] = osum[0] / (float)opix; vec[16 + lane] = osum[1] / (float)opix; }
      wave_sync();
      if (lane < 16) {
        float acc = s_head[512 + lane];
        for (int k = 0; k < 32; ++k) acc = fmaf(s_head[lane * 32 + k], vec[k], acc);
        vec[32 + lane] = fmaxf(acc, 0.f);
      }
      wave_sync();
      if (lane < (a.tapnet ? 1 : 2)) {
        float acc = s_head[560 + lane];
        for (int k = 0; k < 16; ++k) acc = fmaf(s_head[528 + lane * 16 + k], vec[32 + k], acc);
        if (lane == 0) a.occ[map] = acc; else a.expd[map] = acc;
        if (a.out_tracks != nullptr) {      // iter0_kernel's copies
          if (lane == 0) { a.occ0[map] = acc; a.out_occ[map] = acc; } else { a.expd0[map] = acc; a.out_expd[map] = acc; }
        }
      }
    }
    if (lane == 0) {
      const float tot = red[0];
      const float fsx = red[1] / tot;
      const float fsy = red[2] / tot;
      const float fsw = fmaxf(red[3] / tot, 1e-12f);
      float outx = (fsx / fsw) * a.img_w / (float)w;
      float outy = (fsy / fsw) * a.img_h / (float)h;
      if (a.qpts != nullptr) {
        const float* q = a.qpts + (b * a.Q + q0 + m) * 3;
        if ((int)rintf(q[0]) == t) { outx = q[2]; outy = q[1]; }   // round-half-even like jnp.round
      }
      a.points[map * 2 + 0] = outx;
      a.points[map * 2 + 1] = outy;
      if (a.out_tracks != nullptr) {
        a.out_tracks[map * 2 + 0] = outx * a.vx;
        a.out_tracks[map * 2 + 1] = outy * a.vy;
      }
    }
    wave_sync();   // vec is reused by the next map
    tick(3);
  }
  if (TRACE && a.dbg_times != nullptr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.dbg_times[(unit * WAVES + wave) * 8 + k] = (long long)tph[k];
  }
}

// rows of up to 32 cells whose logits fit CVR_ZREG registers per lane in the soft-arg-max pass (the narrow instantiations),
// or rows of 33 .. 64 cells on up to 64 rows (the wide ones, PW = 66)
inline bool cv_rows_wide(int h, int w) { return w > 32 && w <= 64 && h >= 1 && h <= 64; }
inline bool cv_rows_supported(int h, int w) {
  if (cv_rows_wide(h, w)) return true;
  if (h < 1 || w < 1 || w > 32 || !cv_fused_supported(h, w)) return false;
  const int rps = w <= 16 ? 4 : 2;
  return (h + rps - 1) / rps <= CVR_ZREG;
}

constexpr int CVR_PW_WIDE = 66;

// rows of 33 .. 64 cells: 6 maps x 6 waves (bf16; 159 KiB of LDS) or 4 x 4 (f32), one map per wave, one workgroup per CU
template <typename TA>
inline void launch_cv_rows_wide(const CvFusedArgs& a, hipStream_t s) {
  constexpr int QW = sizeof(TA) == 2 ? 6 : 4;
  const int qtiles = (a.Q + QW - 1) / QW;
  const dim3 grid((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), block(QW * 64);
  if (a.w > 48) TAPIR_LAUNCH((cv_rows_kernel<TA, QW, 4, false, 1, true, QW, CVR_PW_WIDE>), grid, block, s, a);
  else TAPIR_LAUNCH((cv_rows_kernel<TA, QW, 3, false, 1, true, QW, CVR_PW_WIDE>), grid, block, s, a);
}

// Forms of the row-streamed kernel (bf16 build; the f32 parity build always runs 8 maps on 8 waves), measured at
// 256 / 1024 queries x 48 frames (cost-volume stage, profiles/r04_kbench_cv_forms.txt):
//   1 (default): 8 maps, 8 waves   one map per wave; two workgroups per CU: the contraction of one runs under the rows
//                                   of the other, at twice the grid reads                            210 / 652 us
//   0: 16 maps, 16 waves            one map per wave; ONE workgroup per CU, four waves per SIMD      225 / 752 us
//   2: 16 maps, 8 waves             two maps per wave; one workgroup per CU, two waves per SIMD       234 / 785 us
// (the pixel-tiled kernel of costvol_fused.hpp: 365 / 1343 us)
template <typename TA, int QPW, int WAVES>
inline void launch_cv_rows_q(const CvFusedArgs& a, hipStream_t s, int heads) {
  const int qpt = QPW / heads;
  const int qtiles = (a.Q + qpt - 1) / qpt;
  const dim3 grid((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), block(WAVES * 64);
  const bool wide = a.w > 16;
  const bool ragw = (a.w & 15) != 0;
#define CVR_GO(H, R)                                                                                          \
  do {                                                                                                        \
    if (wide) TAPIR_LAUNCH((cv_rows_kernel<TA, QPW, 2, false, H, R, WAVES>), grid, block, s, a);              \
    else TAPIR_LAUNCH((cv_rows_kernel<TA, QPW, 1, false, H, R, WAVES>), grid, block, s, a);                   \
  } while (0)
  if (heads == 2) { CVR_GO(2, true); return; }
  if (heads == 4) { CVR_GO(4, true); return; }
#ifdef TAPIR_EXPERIMENTS
  if (a.dbg_times != nullptr && wide && !ragw) {
    hipLaunchKernelGGL((cv_rows_kernel<TA, QPW, 2, true, 1, false, WAVES>), grid, block, 0, s, a);
    return;
  }
#endif
  if (ragw) CVR_GO(1, true); else CVR_GO(1, false);
#undef CVR_GO
}

template <typename TA>
inline void launch_cv_rows(const CvFusedArgs& a, hipStream_t s, int heads = 1, int form = 1) {
  if (cv_rows_wide(a.h, a.w) && heads == 1) { launch_cv_rows_wide<TA>(a, s); return; }
  if constexpr (sizeof(TA) == 2) {
    if (heads == 1 && form == 0) { launch_cv_rows_q<TA, 16, 16>(a, s, heads); return; }
    if (form == 2) { launch_cv_rows_q<TA, 16, 8>(a, s, heads); return; }
  }
  launch_cv_rows_q<TA, 8, 8>(a, s, heads);
}

}  // namespace tapir
