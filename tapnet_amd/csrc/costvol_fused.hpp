// TAPIR.tracks_from_cost_volume (tapnet/models/tapir_model.py:399-471) as ONE kernel: the
// einsum('bnc,bthwc->tbnhw') :433 is contracted on the matrix cores straight into LDS and the heads
// (:438-470) consume it there -- the [T,B,N,h,w] volume never exists in HBM (the reference chunks the
// queries only to bound that tensor, :880-881).
//
// Work unit = (clip b, frame t, tile of QPW queries).  512 threads = 8 waves.
//   G. cost maps: cells (A port, streamed from the frame's feature grid [h*w, 256]: 512 KiB per
//      workgroup out of L2, the only global read of any size) x queries (B port, the tile's query
//      features held in registers) -> s_cm[QPW][(h+2)(w+2)] f32 with a zero halo.  bf16 operands /
//      f32 accumulate (v_mfma_f32_16x16x32_bf16), or exact f32 (v_mfma_f32_16x16x4_f32) in the parity
//      build.
//   then, map by map:
//   M1. conv 1 -> 16 (3x3 SAME) + ReLU :443-444 and conv 16 -> 1 :446 as two chained MFMA products per
//      tile of 16 pixels, both exact f32 (v_mfma_f32_16x16x4_f32: everything that feeds the soft
//      arg max stays f32 in both builds):
//        D1[ch][px]   = sum_tap W1[ch][tap] * cm[px + tap]          (K = 9 taps, padded to 12)
//        P[tap][px]   = sum_ch  W2[ch][tap] * relu(D1[ch][px] + b1)  (K = 16 channels)
//      D1 lands with 4 channels of one pixel per lane, which is exactly the B-operand layout of the
//      second product (the channel <-> k-slot assignment is a free permutation applied to W2), so
//      the 16-channel map goes from accumulator to operand without leaving registers.  P holds, per
//      pixel, the 9 per-tap channel contractions; the convolution's spatial part is then
//        logit[cell] = b2 + sum_tap P[tap][cell + offset(tap)]       (9 LDS reads per cell)
//      -- the 16 -> 1 convolution read 36 x 16 bytes of LDS per cell before (38 % of the old kernel).
//      relu(D1) is also stored pixel-major for the occlusion head (bf16 in the bf16 build).
//   M2. softmax(temperature * logits) over the cells :454, soft arg max with radius 5 around the arg
//      max :455 (model_utils.py:209-314), query-frame override.
//   M3. occlusion head :459-470: conv 16 -> 32 3x3 stride 2 (XLA SAME) as an implicit GEMM on the
//      matrix cores (bf16, or exact f32 in the parity build), ReLU, mean, Linear 32 -> 16, ReLU,
//      Linear 16 -> 2.
// LDS: 16 (8) cost maps 72 KiB (36), hid1 36 KiB (72), P 41 KiB: 150 KiB, one workgroup per CU.
#pragma once
#include "common.hpp"
#include "costvol.hpp"   // CvHeadWeights
#include "gemm.hpp"      // MfmaStep

namespace tapir {

constexpr int CVF_THREADS = 512;
constexpr int CVF_WAVES = CVF_THREADS / 64;
constexpr int CVF_PAD = 1156;        // (h+2)(w+2) <= 34 x 34
constexpr int CVF_CPT = 2;           // cells per thread: h*w <= 1024

struct CvFusedArgs {
  const void* qfeat;      // [B*Q, 256] operand type
  const void* grid;       // [B*T, h*w, 256] operand type
  const void* grid_tiled; // costvol_rows.hpp, bf16: the same grid as [B*T][tiles of 16 cells][32 chunks][16 cells][8] (or null)
  CvHeadWeights wt;
  const float* qpts;      // [B*Q, 3] (t, y, x) in initial_resolution coordinates, or null
  float* points;          // [B*Q*T, 2]
  float* occ;             // [B*Q*T]
  float* expd;            // [B*Q*T]
  int B, Q, T, h, w;
  float temperature;
  float img_h, img_w;
  int tapnet;             // 1: TAP-Net head (tapnet_model.py:157-166): no ReLU after the stride-2
                          // convolution, ONE output logit (occlusion; expd is not written)
  long long* dbg_times;   // TRACE build: [workgroups][8] shader-cycle totals per phase (wave 0)
  // costvol_rows.hpp only:
  int raw;                // 1: no heads -- points = soft arg max of softmax(temperature * cost map) (occ / expd not written);
                          // 2: stop after the contraction (tools: tapir_debug_contraction)
  const int* frame_map;   // null, or [B*T]: unit frame -> index of the grid frame it correlates with
  // costvol_rows.hpp only, or null: what estimate_trajectories does with the stage's result before the first refinement (engine.hip
  // iter0_kernel: a launch of its own otherwise) -- the values the later levels reset to, and iteration 0 of the outputs
  float* occ0; float* expd0;                       // [B*Q*T] copies of occ / expd
  float* out_tracks; float* out_occ; float* out_expd;   // [B*Q*T, 2] points x (vx, vy) (video pixels), [B*Q*T], [B*Q*T]
  float vx, vy;
};

template <typename TA> struct CvFusedCfg;
template <> struct CvFusedCfg<bf16_t> { static constexpr int QPW = 16; };
template <> struct CvFusedCfg<float> { static constexpr int QPW = 8; };

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Channels of head hd among the EPC elements of a 16-byte chunk: channel = chunk * EPC + e, head = channel % HEADS =
// e % HEADS (EPC is a multiple of HEADS): keeps element e of every chunk where e % HEADS == hd.
template <typename TA, int HEADS>
__device__ __forceinline__ uint4 keep_head(uint4 v, int hd) {
  if (HEADS == 1) return v;
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (sizeof(TA) == 2) {                       // word k = elements 2 k (low half), 2 k + 1 (high half)
      const unsigned lo = ((2 * k) % HEADS == hd) ? 0x0000ffffu : 0u;
      const unsigned hi = ((2 * k + 1) % HEADS == hd) ? 0xffff0000u : 0u;
      w[k] &= lo | hi;
    } else {                                     // word k = element k
      w[k] = (k % HEADS == hd) ? w[k] : 0u;
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// HEADS > 1: TAP-Net's num_heads (tapnet_model.py:247-254, 145-149): channel c of the features belongs to head
// c % HEADS, every head's cost map is one input channel of hid1.  A tile holds QPW / HEADS queries x HEADS maps
// (map index = query * HEADS + head: the query features with the other heads' channels zeroed).
template <typename TA, bool TRACE = false, int HEADS = 1>
__global__ __launch_bounds__(CVF_THREADS) void cv_fused_kernel(CvFusedArgs a) {
  constexpr int QPW = CvFusedCfg<TA>::QPW;
  constexpr int QPT = QPW / HEADS;                   // queries per tile
  static_assert(QPW % HEADS == 0 && (16 / (int)sizeof(TA)) % HEADS == 0, "heads");
  constexpr int EPC = 16 / (int)sizeof(TA);          // elements per 16-byte chunk
  constexpr int KCH = kLowresDim / EPC / 4;          // chunk-steps over K = 256 (4 chunks per step)
  constexpr bool BF = sizeof(TA) == 2;
  __shared__ __attribute__((aligned(16))) float s_cm[QPW][CVF_PAD];   // cost maps, zero halo
  __shared__ uint4 s_h1[CVF_PAD * 16 * sizeof(TA) / 16];   // relu(hid1) [pixel][16 ch] operand type, zero halo
  __shared__ __attribute__((aligned(16))) float s_p[9][CVF_PAD];      // per-tap channel contractions, zero halo
  __shared__ float s_red[6][CVF_WAVES];
  __shared__ int s_redi[CVF_WAVES];
  __shared__ float s_occ[CVF_WAVES][32];
  __shared__ float s_vec[32 + 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int h = a.h, w = a.w, hw = h * w;
  const int pw = w + 2, pn = pw * (h + 2);
  const int qtiles = (a.Q + QPT - 1) / QPT;
  // Consecutive workgroup ids go round the 8 XCDs (each with its own L2): XCD x takes the x-th CONTIGUOUS
  // eighth of the (frame, query tile) units, i.e. all query tiles of its frames -- a frame's grid (512 KiB)
  // comes into ONE L2 once.  (In launch order the 16 tiles of a frame landed on all eight XCDs: 202 MB
  // fetched per launch for a 25 MB grid, profiles/r02_pmc_traffic.json.)
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

  // TRACE (tools/kbench.py --what cvtrace): shader cycles per phase, wave 0 of every workgroup
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

  // ---- zero halos / buffers (interior cells are rewritten per map, halo cells never)
  static_assert((QPW * CVF_PAD) % 4 == 0 && (9 * CVF_PAD) % 4 == 0, "16-byte zero fill");
  for (int i = tid; i < QPW * CVF_PAD / 4; i += CVF_THREADS) reinterpret_cast<uint4*>(&s_cm[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < (int)(sizeof(s_h1) / 16); i += CVF_THREADS) s_h1[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < 9 * CVF_PAD / 4; i += CVF_THREADS) reinterpret_cast<uint4*>(&s_p[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);

  // ---- G: cost maps.  B operand: lane (query c, chunk group g) holds chunks 4 s + g of its row.
  {
    const int qrow = min(q0 + (c < QPW ? c / HEADS : 0), a.Q - 1);
    const uint4* qsrc = reinterpret_cast<const uint4*>(
        reinterpret_cast<const TA*>(a.qfeat) + (b * a.Q + qrow) * kLowresDim);
    uint4 fq[KCH];
#pragma unroll
    for (int s = 0; s < KCH; ++s) fq[s] = keep_head<TA, HEADS>(qsrc[4 * s + g], c % HEADS);
    const TA* gbase = reinterpret_cast<const TA*>(a.grid) + frame * (long)hw * kLowresDim;
    const int ntile = (hw + 15) / 16;
    lds_barrier();   // zero fill done before the first cost values land
    // one tile ahead: the fragments of tile it + 8 are requested before tile it is multiplied
    auto load_tile = [&](int it, uint4 (&f)[KCH]) {
      const int cell = min(it * 16 + c, hw - 1);     // A row of this lane (clamped; masked at the store)
      const uint4* csrc = reinterpret_cast<const uint4*>(gbase + (long)cell * kLowresDim);
#pragma unroll
      for (int s = 0; s < KCH; ++s) f[s] = csrc[4 * s + g];
    };
    auto mul_tile = [&](int it, const uint4 (&f)[KCH]) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KCH; ++s) MfmaStep<TA>::run(f[s], fq[s], acc);
      // D: lane holds cells it*16 + 4 g + r of map c (= query c / HEADS, head c % HEADS)
      if (c < nq * HEADS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = it * 16 + 4 * g + r;
          if (p < hw) s_cm[c][(p / w + 1) * pw + (p % w) + 1] = acc[r];
        }
      }
    };
    uint4 fa0[KCH], fa1[KCH];
    int it = wave;
    if (it < ntile) load_tile(it, fa0);
    while (it < ntile) {
      if (it + CVF_WAVES < ntile) load_tile(it + CVF_WAVES, fa1);
      mul_tile(it, fa0);
      it += CVF_WAVES;
      if (it >= ntile) break;
      if (it + CVF_WAVES < ntile) load_tile(it + CVF_WAVES, fa0);
      mul_tile(it, fa1);
      it += CVF_WAVES;
    }
  }

  // ---- per-lane constants of the two small convolutions (exact-f32 MFMA operands)
  //   conv 1: A1[ch = c][k-slot g] of MFMA j = W1[c][tap 4 j + g]          (taps >= 9: 0)
  //   conv 2: A2[tap = c][k-slot g] of MFMA j = W2[ch 4 g + j][tap c]      (taps >= 9: 0)
  float a1[HEADS][3], a2[4];     // W1 is [16][HEADS][3][3]
  int off1[3];   // LDS offset of tap 4 j + g relative to the pixel's halo index
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 4 * j + g;
#pragma unroll
    for (int hd = 0; hd < HEADS; ++hd) a1[hd][j] = tap < 9 ? a.wt.w1[(c * HEADS + hd) * 9 + tap] : 0.f;
    const int tc = min(tap, 8);
    off1[j] = (tc / 3 - 1) * pw + (tc % 3 - 1);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) a2[j] = c < 9 ? a.wt.w2[(4 * g + j) * 9 + c] : 0.f;
  const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.wt.b1 + 4 * g);
  const float b2 = a.wt.b2[0];
  // occlusion convolution: stride 2, XLA SAME (pad_lo = total / 2)
  const int oh = (h + 1) / 2, ow = (w + 1) / 2, opix = oh * ow;
  const int ply = max((oh - 1) * 2 + 3 - h, 0) / 2, plx = max((ow - 1) * 2 + 3 - w, 0) / 2;
  const float b3a = a.wt.b3[c], b3b = a.wt.b3[16 + c];
  uint4 wb[BF ? 5 : 1][2];
  if (BF) {
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int n = 0; n < 2; ++n) wb[s][n] = a.wt.w3b[(s * 2 + n) * 64 + lane];
  }
  const int npt = (hw + 15) / 16;   // pixel tiles of a map
  // integer division by the run-time grid width costs ~30 instructions: every index that does not
  // depend on the map is computed once
  int cbase[CVF_CPT];      // halo index of the top-left cell of the 3x3 window of cell tid + s * 512
  float ccx[CVF_CPT], ccy[CVF_CPT];
#pragma unroll
  for (int s = 0; s < CVF_CPT; ++s) {
    const int p = min(tid + s * CVF_THREADS, hw - 1);
    cbase[s] = (p / w) * pw + (p % w);
    ccx[s] = (float)(p % w) + 0.5f; ccy[s] = (float)(p / w) + 0.5f;
  }
  constexpr int TPW = (CVF_CPT * CVF_THREADS / 16 + CVF_WAVES - 1) / CVF_WAVES;   // pixel tiles per wave (8)
  int thidx[TPW];          // halo index of pixel (tile wave + 8 k, lane column c)
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int p = min((wave + CVF_WAVES * k) * 16 + c, hw - 1);
    thidx[k] = (p / w + 1) * pw + (p % w) + 1;
  }
  tick(0);   // zero fill + cost maps + constants

  for (int m = 0; m < nq; ++m) {
    lds_barrier();   // cost maps complete (m = 0) / previous map's readers of s_h1, s_p done
    tick(1);
    const float* cm = s_cm[m * HEADS];                 // head hd of this query: cm + hd * CVF_PAD
    // ---- M1: hid1 = relu(conv1(cm) + b1) -> s_h1, P = per-tap contraction of hid1 with W2 -> s_p
    // two pixel tiles at a time: the MFMAs of a tile form dependent chains (40-cycle latency each)
#pragma unroll
    for (int k = 0; k < TPW; k += 2) {
      const int it0 = wave + CVF_WAVES * k, it1 = it0 + CVF_WAVES;
      if (it0 < npt) {
        const int h0 = thidx[k], h1 = thidx[k + 1 < TPW ? k + 1 : k];
        f32x4 d1a = b1v, d1b = b1v;
#pragma unroll
        for (int hd = 0; hd < HEADS; ++hd)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            d1a = mfma_f32(a1[hd][j], cm[hd * CVF_PAD + h0 + off1[j]], d1a);
            d1b = mfma_f32(a1[hd][j], cm[hd * CVF_PAD + h1 + off1[j]], d1b);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) { d1a[r] = fmaxf(d1a[r], 0.f); d1b[r] = fmaxf(d1b[r], 0.f); }
        f32x4 d2a = f32x4{0.f, 0.f, 0.f, 0.f}, d2b = d2a;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d2a = mfma_f32(a2[j], d1a[j], d2a);
          d2b = mfma_f32(a2[j], d1b[j], d2b);
        }
        auto put = [&](int it, int hidx, const f32x4& d1, const f32x4& d2) {
          if (it < npt && it * 16 + c < hw) {
            // hid1: lane holds channels 4 g .. 4 g + 3 of pixel c
            if (BF) {
              uint2 o;
              o.x = pack_bf16x2(d1[0], d1[1]);
              o.y = pack_bf16x2(d1[2], d1[3]);
              reinterpret_cast<uint2*>(s_h1)[hidx * 4 + g] = o;
            } else {
              reinterpret_cast<f32x4*>(s_h1)[hidx * 4 + g] = d1;
            }
            // P: lane holds taps 4 g .. 4 g + 3 of pixel c
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (4 * g + r < 9) s_p[4 * g + r][hidx] = d2[r];
          }
        };
        put(it0, h0, d1a, d2a);
        put(it1, h1, d1b, d2b);
      }
    }
    tick(2);
    lds_barrier();
    tick(3);

    // ---- M2: logits, arg max, softmax window sums
    float z[CVF_CPT];
    float best = -3.0e38f;
#pragma unroll
    for (int s = 0; s < CVF_CPT; ++s) {
      float acc = b2;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) acc += s_p[tap][cbase[s] + (tap / 3) * pw + (tap % 3)];
      z[s] = tid + s * CVF_THREADS < hw ? acc * a.temperature : -3.0e38f;
      best = fmaxf(best, z[s]);
    }
    // arg max = FIRST maximum (jnp.argmax, model_utils.py:232): wave maximum, then the smallest cell
    // index among the lanes that hold it (both on the DPP path), then the same across the waves
    best = wave_max(best);
    float bestf = 3.0e9f;   // cell indices are < 2^24: exact as floats
#pragma unroll
    for (int s = CVF_CPT - 1; s >= 0; --s)
      if (z[s] == best) bestf = (float)(tid + s * CVF_THREADS);
    int besti = (int)(-wave_max(-bestf));
    if (lane == 0) { s_red[0][wave] = best; s_redi[wave] = besti; }
    lds_barrier();
    best = s_red[0][0]; besti = s_redi[0];
#pragma unroll
    for (int k = 1; k < CVF_WAVES; ++k) {
      const float ob = s_red[0][k]; const int oi = s_redi[k];
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const float ax = (float)(besti % w) + 0.5f, ay = (float)(besti / w) + 0.5f;
    float esum = 0.f, sx = 0.f, sy = 0.f, sw = 0.f;
#pragma unroll
    for (int s = 0; s < CVF_CPT; ++s) {
      if (tid + s * CVF_THREADS < hw) {
        const float e = fast_exp(z[s] - best);
        esum += e;
        const float d2 = (ccx[s] - ax) * (ccx[s] - ax) + (ccy[s] - ay) * (ccy[s] - ay);
        if (d2 < 25.0f) { sx += ccx[s] * e; sy += ccy[s] * e; sw += e; }   // radius 5, strict (model_utils.py:236)
      }
    }
    esum = wave_sum(esum); sx = wave_sum(sx); sy = wave_sum(sy); sw = wave_sum(sw);
    if (lane == 0) { s_red[1][wave] = esum; s_red[3][wave] = sx; s_red[4][wave] = sy; s_red[5][wave] = sw; }
    tick(4);

    // ---- M3: occlusion head, conv 16 -> 32 stride 2 on the matrix cores
    float osum[2] = {0.f, 0.f};   // sum over this lane's pixels of relu(conv + b), channels c and 16 + c
    for (int mt = wave; mt * 16 < opix; mt += CVF_WAVES) {
      const int P = min(mt * 16 + c, opix - 1);        // this lane's A row (clamped: masked below)
      const int oy = P / ow, ox = P - oy * ow;
      const int base = (2 * oy - ply + 1) * pw + (2 * ox - plx + 1);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      if (BF) {
        // k = tap * 16 + ci (padded to 160): lane group g of k-step s reads channels 8 (g & 1) .. +7 of
        // tap 2 s + (g >> 1): one 16-byte read of the pixel-major bf16 map
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const int tap = min(2 * s + (g >> 1), 8);    // tap 9 has zero weights
          const int pp = base + (tap / 3) * pw + (tap % 3);
          const uint4 af = s_h1[pp * 2 + (g & 1)];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                         __builtin_bit_cast(bf16x8, wb[s][0]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                         __builtin_bit_cast(bf16x8, wb[s][1]), acc1, 0, 0, 0);
        }
      } else {
        // exact f32: 36 k-slices of 4: slice j = (tap = j / 4, channels 4 (j % 4) + g)
        const float* h1 = reinterpret_cast<const float*>(s_h1);
        for (int tap = 0; tap < 9; ++tap) {
          const int pp = base + (tap / 3) * pw + (tap % 3);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int ci = 4 * jj + g;
            const float av = h1[pp * 16 + ci];
            const float* wr = a.wt.w3 + (ci * 9 + tap) * 32;
            acc0 = mfma_f32(av, wr[c], acc0);
            acc1 = mfma_f32(av, wr[16 + c], acc1);
          }
        }
      }
      // D: lane holds channel c (and 16 + c), pixels mt*16 + 4 g + r
      const float floor3 = a.tapnet ? -3.0e38f : 0.f;   // TAPIR: ReLU (tapir_model.py:461); TAP-Net: none
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (mt * 16 + 4 * g + r < opix) {
          osum[0] += fmaxf(acc0[r] + b3a, floor3);
          osum[1] += fmaxf(acc1[r] + b3b, floor3);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      osum[k] += __shfl_xor(osum[k], 16);
      osum[k] += __shfl_xor(osum[k], 32);
    }
    if (lane < 16) { s_occ[wave][lane] = osum[0]; s_occ[wave][16 + lane] = osum[1]; }
    tick(5);
    lds_barrier();
    tick(6);

    // ---- tail in ONE wave (rotating): LDS operations of a wave execute in order, no further barriers;
    // the other waves go on to the next map (they next write s_red / s_occ only behind a barrier this
    // wave has to reach too)
    if (wave == (m & (CVF_WAVES - 1))) {
      if (lane < 32) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < CVF_WAVES; ++k) tsum += s_occ[k][lane];
        s_vec[lane] = tsum / (float)opix;
      }
      wave_sync();
      if (lane < 16) {
        float acc = a.wt.b4[lane];
        for (int k = 0; k < 32; ++k) acc = fmaf(a.wt.w4[lane * 32 + k], s_vec[k], acc);
        s_vec[32 + lane] = fmaxf(acc, 0.f);
      }
      wave_sync();
      const long map = (b * a.Q + q0 + m) * a.T + t;
      if (lane < (a.tapnet ? 1 : 2)) {
        float acc = a.wt.b5[lane];
        for (int k = 0; k < 16; ++k) acc = fmaf(a.wt.w5[lane * 16 + k], s_vec[32 + k], acc);
        if (lane == 0) a.occ[map] = acc; else a.expd[map] = acc;
      }
      if (lane == 0) {
        float tot = 0.f, tsx = 0.f, tsy = 0.f, tsw = 0.f;
#pragma unroll
        for (int k = 0; k < CVF_WAVES; ++k) { tot += s_red[1][k]; tsx += s_red[3][k]; tsy += s_red[4][k]; tsw += s_red[5][k]; }
        const float fsx = tsx / tot;
        const float fsy = tsy / tot;
        const float fsw = fmaxf(tsw / tot, 1e-12f);
        float outx = (fsx / fsw) * a.img_w / (float)w;
        float outy = (fsy / fsw) * a.img_h / (float)h;
        if (a.qpts != nullptr) {
          const float* q = a.qpts + (b * a.Q + q0 + m) * 3;
          if ((int)rintf(q[0]) == t) { outx = q[2]; outy = q[1]; }   // round-half-even like jnp.round
        }
        a.points[map * 2 + 0] = outx;
        a.points[map * 2 + 1] = outy;
      }
    }
    tick(7);
  }
  if (TRACE && a.dbg_times != nullptr && tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.dbg_times[unit * 8 + k] = (long long)tph[k];
  }
}

inline bool cv_fused_supported(int h, int w) {
  return (h + 2) * (w + 2) <= CVF_PAD && h * w <= CVF_CPT * CVF_THREADS && h >= 1 && w >= 1;
}

template <typename TA>
inline void launch_cv_fused(const CvFusedArgs& a, hipStream_t s, int heads = 1) {
  const int qpt = CvFusedCfg<TA>::QPW / heads;
  const int qtiles = (a.Q + qpt - 1) / qpt;
  if (heads == 2) {
    TAPIR_LAUNCH((cv_fused_kernel<TA, false, 2>), dim3((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), dim3(CVF_THREADS), s, a);
    return;
  }
  if (heads == 4) {
    TAPIR_LAUNCH((cv_fused_kernel<TA, false, 4>), dim3((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), dim3(CVF_THREADS), s, a);
    return;
  }
#ifdef TAPIR_EXPERIMENTS
  if (a.dbg_times != nullptr) {
    hipLaunchKernelGGL((cv_fused_kernel<TA, true>), dim3((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), dim3(CVF_THREADS), 0, s, a);
    return;
  }
#endif
  TAPIR_LAUNCH((cv_fused_kernel<TA>), dim3((unsigned)(8 * (((long)a.B * a.T * qtiles + 7) / 8))), dim3(CVF_THREADS), s, a);
}

}  // namespace tapir
