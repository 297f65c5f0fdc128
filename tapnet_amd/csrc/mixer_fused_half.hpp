// Track-resident PIPs MLP-mixer, HALF-CU form (bf16): one track per workgroup as in mixer_fused.hpp, but the
// workgroup is 4 waves and 80 KiB of LDS, so TWO workgroups -- two independent tracks -- are resident on a CU.
//
// Why.  At one track per CU the 8-wave kernel is bound by the L2 -> CU fill of the channel MLP's weights (64 B/clk:
// 65 k cycles per block) PLUS the token mixing, LayerNorms and first GELU of the same track (~38 k cycles of VALU
// work that nothing of that track can overlap): the fill path idles through every VALU phase and the matrix pipes
// through both.  The wide kernel (two tracks per workgroup, one weight stream for both) removes the fill bound but
// serialises the VALU phases of both tracks (206 k cycles per block for the pair, profiles/r03_mixer_phase_traces.txt).
// Two INDEPENDENT workgroups on one CU need no schedule of their own: while one is in its VALU phases the other
// streams weights, the hardware interleaves them instruction by instruction, and the CU's fill path stays busy --
// the pair is bound by 2 x 65 k cycles of fill per block.  It needs > 256 tracks (two per CU); at 256 tracks the
// 8-wave kernel with one track on every CU is the faster one.
//
// Layout (differences from mixer_fused.hpp).  256 threads = 4 waves, wave w owns output channels [128 w, 128 w + 128)
// of every 512-wide tensor: the residual is xr[8][NT] (96 VGPRs at 48 frames), a k-step of a 512-row GEMM is 8
// fragments = one revolution of the 8-deep register ring.  Hidden chunks of 128 units through two 12-KiB LDS buffers
// with the same software pipeline (up(0); for c: { up(c); down(c-1) with the GELU of chunk c between its MFMAs;
// barrier }; down(15)): 16 chunks per block.  LDS: LN2(x) 48 KiB + 2 x 12 KiB + up-projection bias 8 KiB = 80 KiB;
// the LayerNorm summaries live in the last 8 KiB of the activation region (beyond the 64 KiB of temporal-convolution
// parameters, inside the hidden-chunk buffers, which are dead whenever a LayerNorm runs).
// The weight stream is packed per wave for this chunking (tapir_finalize_weights: build_fused_half_weights).
#pragma once
#include "mixer_fused.hpp"

namespace tapir {

constexpr int FMH_WAVES = 4;
constexpr int FMH_THREADS = FMH_WAVES * 64;
constexpr int FMH_RING = 8;
constexpr int FMH_HC = 128;                               // hidden units per chunk
constexpr int FMH_QA = kHidden / FMH_WAVES / 16;          // 8 row tiles of a wave in a 512-row GEMM
constexpr int FMH_RAU = FMH_HC / FMH_WAVES / 16;          // 2 hidden-row tiles of a wave per chunk
constexpr int FMH_NC = kHidden4 / FMH_HC;                 // 16 chunks per block
constexpr int FMH_STAT_OFF = 64 * 1024;                   // LayerNorm summaries inside the activation region

inline long fused_half_frags_per_wave(int k0_pad, int nblocks) {
  const long in = (long)(k0_pad / 32) * FMH_QA;
  const long up = (long)FMH_RAU * (kHidden / 32);          // per chunk
  const long dn = (long)FMH_QA * (FMH_HC / 32);
  const long blk = (long)FMH_NC * (up + dn);
  const long out = (long)FMH_QA * (kHidden / 32);
  return in + nblocks * blk + out + FMH_RING;              // + one ring of padding (prefetched, never used)
}

// One GEMM phase with all 8 row tiles of the wave: acc[r][i] += W_frag(r, k) . act(token tile i, k) over KSTEPS
// (template, fully unrolled: `step` is a compile-time constant inside epi) or `ksteps` (run time) k-steps, both even.
// A k-step is one revolution of the ring; B fragments one k-step ahead, alternating between two register sets.
#ifndef TAPIR_FMH_SKEW
#define TAPIR_FMH_SKEW 8          // x s_sleep(127) = x 8 k cycles
#endif
#ifndef TAPIR_FMH_DB
#define TAPIR_FMH_DB 0
#endif
constexpr bool FMH_DB = TAPIR_FMH_DB != 0;   // B fragments double-buffered (24 VGPRs) or read at use (12)

template <int NT, int KSTEPS = 0, typename Epi = NoEpilogue>
__device__ __forceinline__ void half_gemm8(const uint4*& wp, uint4 (&ring)[FMH_RING], const char* bbase, int bstride,
                                           int ksteps, int c, int g, f32x4 (&acc)[FMH_QA][NT], Epi epi = Epi()) {
  static_assert(FMH_RING == FMH_QA, "a k-step is one revolution of the ring");
  const char* brow = bbase + c * bstride;
  auto read_b = [&](int ks, uint4 (&fb)[NT]) {
    const int chunk = (ks * 4 + g) ^ c;
#pragma unroll
    for (int i = 0; i < NT; ++i)
      fb[i] = *reinterpret_cast<const uint4*>(brow + 16 * i * bstride + (chunk << 4));
  };
  auto kstep = [&](int ks, const uint4 (&cur)[NT]) {
#pragma unroll
    for (int r = 0; r < FMH_QA; ++r) {
      const uint4 fa = ring[r];
#pragma unroll
      for (int i = 0; i < NT; ++i) MfmaStep<bf16_t>::run(fa, cur[i], acc[r][i]);
      ring[r] = *wp;
      wp += 64;
      epi(ks * FMH_QA + r);
      sched_fence();
    }
  };
  if (KSTEPS > 0) ksteps = KSTEPS;
  uint4 fb0[NT], fb1[NT];   // (fb1 is dead when !FMH_DB)
  if (FMH_DB) read_b(0, fb0);
  auto pair = [&](int ks) {
    if constexpr (FMH_DB) {
      read_b(ks + 1, fb1);
      sched_fence();
      kstep(ks, fb0);
      read_b(ks + 2 < ksteps ? ks + 2 : 0, fb0);   // past the end: any valid address (the values are not used)
      sched_fence();
      kstep(ks + 1, fb1);
    } else {
      // no registers for a second set of B fragments: the other wave of the SIMD covers the LDS round trip
      read_b(ks, fb0);
      sched_fence();
      kstep(ks, fb0);
      read_b(ks + 1, fb0);
      sched_fence();
      kstep(ks + 1, fb0);
    }
  };
  if constexpr (KSTEPS > 0) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) pair(ks);
  } else {
    for (int ks = 0; ks < ksteps; ks += 2) pair(ks);
  }
}

template <int NT, bool RAGGED>
__global__ __launch_bounds__(FMH_THREADS) TAPIR_WAVES_PER_EU(2, 2) void mixer_fused_half_kernel(FusedArgs a) {
  typedef bf16_t TA;
  constexpr int KS = 32, HC = FMH_HC, QA = FMH_QA, RAU = FMH_RAU, NC = FMH_NC;
  constexpr int ROWS = NT * 16;
  constexpr int XN_STRIDE = kHidden * 2, H_STRIDE = HC * 2;
  constexpr int XN_BYTES = ROWS * XN_STRIDE, H_BYTES = ROWS * H_STRIDE;
  constexpr int DN_KSTEPS = HC / KS;                                      // 4
  constexpr int PAR_BYTES = kHidden * FM_MIXW * 4;                       // 64 KiB
  constexpr int STAT_END = FMH_STAT_OFF + 2 * ROWS * FMH_WAVES * 8;
  constexpr int ACT_BYTES = XN_BYTES + 2 * H_BYTES > STAT_END ? XN_BYTES + 2 * H_BYTES : STAT_END;
  static_assert(XN_BYTES + 2 * H_BYTES <= ACT_BYTES && PAR_BYTES <= FMH_STAT_OFF, "activation region");
  static_assert(FMH_STAT_OFF >= XN_BYTES && FMH_STAT_OFF + 2 * ROWS * FMH_WAVES * 8 <= ACT_BYTES, "LayerNorm summaries");
  static_assert(RAU * NT * 4 <= DN_KSTEPS * QA, "one GELU per down-projection fragment step");
  static_assert(ACT_BYTES + kHidden4 * 4 <= 80 * 1024, "two workgroups per CU");
  __shared__ uint4 s_act[ACT_BYTES / 16];
  __shared__ float s_bup[kHidden4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n = blockIdx.x;
  const int T = a.T;
  char* const s_xn = reinterpret_cast<char*>(s_act);
  char* const s_h0 = s_xn + XN_BYTES;
  const int ch_lane = 128 * wave + 4 * g;    // channel of (q = 0, r = 0) of this lane

  // The two workgroups of a CU are dispatched together and execute the same instruction stream: left alone they
  // stay in lockstep -- both in their VALU phases, then both streaming weights -- and nothing overlaps.  The second
  // wave of workgroups (blockIdx >= number of CUs) starts about half a block late.
#ifndef TAPIR_HIPEMU
  if (a.skew_div > 0 && ((blockIdx.x / (unsigned)a.skew_div) & 1u)) {
#pragma unroll 1
    for (int k = 0; k < TAPIR_FMH_SKEW; ++k) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // ---- weight stream: fill the ring (the loads fly while the input rows are staged)
  const uint4* wp = a.stream + ((long)wave * a.frags_per_wave) * 64 + lane;
  uint4 ring[FMH_RING];
#pragma unroll
  for (int s = 0; s < FMH_RING; ++s) { ring[s] = *wp; wp += 64; }

  // ---- stage the mixer-input rows of this track: [ROWS][ld_in], rows >= T zero
  const int in_stride = a.ld_in * 2;
  {
    const int cpr = in_stride >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(
        reinterpret_cast<const char*>(a.mlp_in) + (long)n * T * in_stride);
    for (int id = tid; id < ROWS * cpr; id += FMH_THREADS) {
      const int row = id / cpr, q = id - row * cpr;
      const int tok = NT * (row & 15) + (row >> 4);   // LDS row 16 i + c holds token NT c + i
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (tok < T) v = src[tok * cpr + q];
      s_act[row * cpr + (q ^ (row & 15))] = v;
    }
  }
  lds_barrier();

  // ---- residual stream <- input Linear
  f32x4 xr[QA][NT];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const f32x4 b = gload4(a.b0 + ch_lane + 16 * q);
#pragma unroll
    for (int i = 0; i < NT; ++i) xr[q][i] = b;
  }
  half_gemm8<NT>(wp, ring, s_xn, in_stride, a.ld_in / KS, c, g, xr);

  float valid[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) valid[i] = (NT * c + i < T) ? 1.0f : 0.0f;

  // per-token LayerNorm statistics: every wave reduces its 128 channels to (sum, M2 about ITS mean), the four
  // summaries merge with Chan's formula (equal counts); two summary buffers alternate
  int ln_phase = 0;
  auto ln_stats = [&](float (&mean)[NT], float (&rstd)[NT], bool wait_params = false) {
    float2 (*stat)[FMH_WAVES] = reinterpret_cast<float2 (*)[FMH_WAVES]>(
        s_xn + FMH_STAT_OFF + ln_phase * (ROWS * FMH_WAVES * 8));
    ln_phase ^= 1;
    float s[NT], m2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < QA; ++q) t += (xr[q][i][0] + xr[q][i][1]) + (xr[q][i][2] + xr[q][i][3]);
      s[i] = t;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) s[i] += __shfl_xor(s[i], 16);
#pragma unroll
    for (int i = 0; i < NT; ++i) s[i] += __shfl_xor(s[i], 32);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float mw = s[i] * (1.0f / 128.0f);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < QA; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = xr[q][i][r] - mw; t = fmaf(d, d, t); }
      m2[i] = t;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) m2[i] += __shfl_xor(m2[i], 16);
#pragma unroll
    for (int i = 0; i < NT; ++i) m2[i] += __shfl_xor(m2[i], 32);
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) stat[16 * i + c][wave] = make_float2(s[i], m2[i]);
    }
    if (wait_params) dma_wait<0>();   // LN1: the parameter copies of this block have landed; the ring is idle here
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float2 p[FMH_WAVES];
#pragma unroll
      for (int k = 0; k < FMH_WAVES / 2; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(&stat[16 * i + c][2 * k]);
        p[2 * k] = make_float2(v.x, v.y); p[2 * k + 1] = make_float2(v.z, v.w);
      }
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < FMH_WAVES; ++k) tot += p[k].x;
      const float mu = tot * (1.0f / kHidden);
      float M2 = 0.f;
#pragma unroll
      for (int k = 0; k < FMH_WAVES; ++k) {
        const float d = p[k].x * (1.0f / 128.0f) - mu;
        M2 += p[k].y + 128.0f * d * d;
      }
      mean[i] = mu;
      rstd[i] = 1.0f / sqrtf(M2 * (1.0f / kHidden) + kLnEps);
    }
  };

  auto write_xn = [&](const float* scale, const float (&mean)[NT], const float (&rstd)[NT]) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const f32x4 sc = gload4(scale + ch_lane + 16 * q);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        store_act4<TA>(s_xn, XN_STRIDE, 16 * i + c, ch_lane + 16 * q, c,
                       (xr[q][i][0] - mean[i]) * rstd[i] * sc[0], (xr[q][i][1] - mean[i]) * rstd[i] * sc[1],
                       (xr[q][i][2] - mean[i]) * rstd[i] * sc[2], (xr[q][i][3] - mean[i]) * rstd[i] * sc[3]);
    }
  };

  const int hid_lane = wave * (HC / FMH_WAVES) + 4 * g;   // hidden unit (within a chunk) of (row tile 0, reg 0)
  // B-fragment read addresses of row c (token tile 0) for k-steps ks = j (mod 4), LN2(x) image and hidden chunk 0;
  // GELU store addresses of this lane's hidden units (row tile r), token tile 0, hidden chunk buffer 0
  const char* xb[4];
  const char* hb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int off = ((j ^ (c >> 2)) << 6) + ((g ^ (c & 3)) << 4);
    xb[j] = s_xn + c * XN_STRIDE + off;
    hb[j] = s_h0 + c * H_STRIDE + off;
  }
  char* hs[RAU];
#pragma unroll
  for (int r = 0; r < RAU; ++r) {
    const int ch0 = hid_lane + 16 * r;
    hs[r] = s_h0 + c * H_STRIDE + (((ch0 / 8) ^ c) << 4) + (ch0 % 8) * 2;
  }

  // temporal-convolution parameters of a block (64 KiB) by LDS-DMA into the activation region (see mixer_fused.hpp)
  constexpr int PARV = PAR_BYTES / 16 / FMH_THREADS;      // 16
  auto params_dma = [&](int blk) {
    const float* src = a.blocks[blk].mixw;
    char* dst = reinterpret_cast<char*>(s_act) + 1024 * wave;     // 1 KiB per wave and instruction
#pragma unroll
    for (int k = 0; k < PARV; ++k) glds16(src + (tid + k * FMH_THREADS) * 4, dst + 1024 * FMH_WAVES * k);
  };
  lds_barrier();   // every wave is done with the input rows: the region is reused from here on

  for (int b = 0; b < a.nblocks; ++b) {
    const FusedBlockParams& bp = a.blocks[b];
    float mean[NT], rstd[NT];
    params_dma(b);

    // ---- token mixing: LN1 -> depthwise conv k=3 (x4 channels) -> GELU -> depthwise conv k=3 -> sum of 4 -> + skip
    ln_stats(mean, rstd, true);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x4* pw = reinterpret_cast<const f32x4*>(s_act) +
                          opaque((ch_lane + 16 * q + 2 * rp) >> 1) * (2 * FM_MIXW / 4);
        f32x2 xc[NT], xp[NT], xq[NT], s0[NT], s1[NT], s2[NT];
        const f32x2 zero = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          xc[i] = (f32x2{xr[q][i][2 * rp], xr[q][i][2 * rp + 1]} - mean[i]) * rstd[i];
          if (RAGGED) xc[i] = xc[i] * valid[i];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          xp[i] = i > 0 ? xc[i - 1] : f32x2{lane_up(xc[NT - 1].x, lane), lane_up(xc[NT - 1].y, lane)};
          xq[i] = i + 1 < NT ? xc[i + 1] : f32x2{lane_dn(xc[0].x, lane), lane_dn(xc[0].y, lane)};
          s0[i] = zero; s1[i] = zero; s2[i] = zero;
        }
        f32x2 bsum = zero;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 v0 = pw[4 * m], v1 = pw[4 * m + 1], v2 = pw[4 * m + 2], v3 = pw[4 * m + 3];
          const f32x2 w10 = f32x2{v0[0], v0[1]}, w11 = f32x2{v0[2], v0[3]}, w12 = f32x2{v1[0], v1[1]},
                      b1m = f32x2{v1[2], v1[3]}, w20 = f32x2{v2[0], v2[1]}, w21 = f32x2{v2[2], v2[3]},
                      w22 = f32x2{v3[0], v3[1]};
          if (m == 0) bsum = f32x2{v3[2], v3[3]};
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            f32x2 u = b1m;
            u = __builtin_elementwise_fma(w10, xp[i], u);
            u = __builtin_elementwise_fma(w11, xc[i], u);
            u = __builtin_elementwise_fma(w12, xq[i], u);
            f32x2 gl = gelu_tanh2(u);
            if (RAGGED) gl = gl * valid[i];
            s0[i] = __builtin_elementwise_fma(w20, gl, s0[i]);
            s1[i] = __builtin_elementwise_fma(w21, gl, s1[i]);
            s2[i] = __builtin_elementwise_fma(w22, gl, s2[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const f32x2 pa = i > 0 ? s0[i - 1] : f32x2{lane_up(s0[NT - 1].x, lane), lane_up(s0[NT - 1].y, lane)};
          const f32x2 pb = i + 1 < NT ? s2[i + 1] : f32x2{lane_dn(s2[0].x, lane), lane_dn(s2[0].y, lane)};
          const f32x2 y = bsum + pa + s1[i] + pb;
          xr[q][i][2 * rp] += y.x;
          xr[q][i][2 * rp + 1] += y.y;
        }
        sched_fence();
      }
    }

    // ---- channel MLP: x += W_dn . gelu(W_up . LN2(x) + b_up) + b_dn
    ln_stats(mean, rstd);
    write_xn(bp.ln2, mean, rstd);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const f32x4 bd = gload4(bp.bdn + ch_lane + 16 * q);
#pragma unroll
      for (int i = 0; i < NT; ++i) xr[q][i] += bd;
    }
    // up-projection bias through LDS (a vector load inside the chunk loop would drain the ring when waited for)
    *reinterpret_cast<f32x4*>(&s_bup[tid * 8]) = gload4(bp.bup + tid * 8);
    *reinterpret_cast<f32x4*>(&s_bup[tid * 8 + 4]) = gload4(bp.bup + tid * 8 + 4);
    lds_barrier();   // LN2(x) (and the bias) visible to every wave

    // Hot loops with EXPLICIT LDS addresses.  The swizzled chunk of k-step ks is (4 ks + g) ^ c =
    // 4 (ks ^ (c >> 2)) + (g ^ (c & 3)): the lane-dependent part takes only four values (ks & 3), the rest is a
    // compile-time offset -- 4 address registers per image instead of one per k-step (left to the compiler, the
    // sixteen loop-invariant addresses are hoisted out of the chunk loop, spilled, and every reload inside the loop
    // is a vector-memory load that drains the weight ring when it is waited for).
    f32x4 ua[RAU][NT];
    auto up = [&](int hc) {
#pragma unroll
      for (int r = 0; r < RAU; ++r) {
        const f32x4 bu = *reinterpret_cast<const f32x4*>(&s_bup[hc * HC + hid_lane + 16 * r]);
#pragma unroll
        for (int i = 0; i < NT; ++i) ua[r][i] = bu;
      }
      uint4 fb[NT];
#pragma unroll
      for (int ks = 0; ks < kHidden / KS; ++ks) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          fb[i] = *reinterpret_cast<const uint4*>(xb[ks & 3] + (ks >> 2) * 256 + i * 16 * XN_STRIDE);
        sched_fence();
#pragma unroll
        for (int r = 0; r < RAU; ++r) {
          const int slot = (ks * RAU + r) % FMH_RING;
          const uint4 fa = ring[slot];
#pragma unroll
          for (int i = 0; i < NT; ++i) MfmaStep<TA>::run(fa, fb[i], ua[r][i]);
          ring[slot] = *wp;
          wp += 64;
          sched_fence();
        }
      }
    };
    // GELU of value v (0 .. RAU*NT*4-1) of the up accumulators; every fourth one stores its fragment
    auto gelu_step = [&](int v, int hoff) {
      if (v < RAU * NT * 4) {
        const int item = v >> 2, r = item / NT, i = item % NT, k = v & 3;
        ua[r][i][k] = gelu_tanh(ua[r][i][k]);
        if (k == 3) {
          uint2 o;
          o.x = pack_bf16x2(ua[r][i][0], ua[r][i][1]);
          o.y = pack_bf16x2(ua[r][i][2], ua[r][i][3]);
          *reinterpret_cast<uint2*>(hs[r] + hoff + i * 16 * H_STRIDE) = o;
        }
      }
    };
    // down-projection of the chunk in buffer `hoff_rd` into the residual, the GELU + store of the NEXT chunk
    // (-> buffer hoff_wr; < 0: none) between its MFMAs
    auto down = [&](int hoff_rd, int hoff_wr) {
      uint4 fb[NT];
#pragma unroll
      for (int ks = 0; ks < DN_KSTEPS; ++ks) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          fb[i] = *reinterpret_cast<const uint4*>(hb[ks] + hoff_rd + i * 16 * H_STRIDE);
        sched_fence();
#pragma unroll
        for (int r = 0; r < QA; ++r) {
          const uint4 fa = ring[r];
#pragma unroll
          for (int i = 0; i < NT; ++i) MfmaStep<TA>::run(fa, fb[i], xr[r][i]);
          ring[r] = *wp;
          wp += 64;
          if (hoff_wr >= 0) gelu_step(ks * QA + r, hoff_wr);
          sched_fence();
        }
      }
    };
    up(0);
#pragma unroll
    for (int v = 0; v < RAU * NT * 4; ++v) gelu_step(v, 0);   // chunk 0: nothing of this track to overlap with
    lds_barrier();
    for (int hc = 1; hc < NC; ++hc) {
      up(hc);
      down(((hc - 1) & 1) * H_BYTES, (hc & 1) * H_BYTES);
      lds_barrier();
    }
    down(((NC - 1) & 1) * H_BYTES, -1);
    lds_barrier();   // every wave is done with the activation images before the next block reuses them
  }

  // ---- final LayerNorm + output Linear: 388 outputs, rows padded to 512
  {
    float mean[NT], rstd[NT];
    ln_stats(mean, rstd);
    write_xn(a.lnF, mean, rstd);
    lds_barrier();
    // (the residual is dead after write_xn: its registers hold the output accumulators)
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int o0 = ch_lane + 16 * q;
      f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
      if (o0 < kMixOut) bo = gload4(a.bout + o0);
#pragma unroll
      for (int i = 0; i < NT; ++i) xr[q][i] = bo;
    }
    half_gemm8<NT>(wp, ring, s_xn, XN_STRIDE, kHidden / KS, c, g, xr);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int o0 = ch_lane + 16 * q;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int t = NT * c + i;
        if (o0 < kMixOut && t < T) fused_emit(a, (long)n * T + t, n, o0, xr[q][i]);
      }
    }
  }
}

// shapes the half-CU kernel covers: non-causal whole clips of 17..48 frames (bf16)
inline bool fused_half_supported(int T, int k0_pad, bool causal, bool has_ctx) {
  if (causal || has_ctx || T <= 16 || T > 48) return false;
  if ((k0_pad * 2) % 256 != 0 || (k0_pad / 32) % 2 != 0) return false;
  return 16L * ((T + 15) / 16) * k0_pad * 2 <= FMH_STAT_OFF;     // the staged input image fits below the summaries
}

inline void launch_mixer_fused_half(const FusedArgs& a, hipStream_t s) {
  const int nt = (a.T + 15) / 16;
  const bool ragged = a.T % 16 != 0;
  const dim3 grid((unsigned)a.N), block(FMH_THREADS);
  if (nt == 2) {
    if (ragged) TAPIR_LAUNCH((mixer_fused_half_kernel<2, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_half_kernel<2, false>), grid, block, s, a);
  } else {
    if (ragged) TAPIR_LAUNCH((mixer_fused_half_kernel<3, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_half_kernel<3, false>), grid, block, s, a);
  }
}

}  // namespace tapir
