// Track-resident mixer with an FP8 (e4m3) WEIGHT STREAM -- experiments builds only, NOT verified on hardware (written at
// the end of round 3 without GPU time left; the host emulator checks the index math, the packing and the scale folding).
//
// Why.  At one 48-frame track per CU the kernel is bound by the bytes of the channel MLP's weights on the L2 -> CU path
// (DESIGN 3.1); fp8 weights halve them.  Fragments are 512 bytes (8 bytes per lane), converted to bf16 in registers just
// before their MFMAs (v_cvt_scalef32_pk_bf16_fp8: 4 instructions per fragment and lane); activations, accumulation,
// LDS images and the chunk pipeline are mixer_fused.hpp's.  ONE scale per matrix, folded where it costs nothing:
// W_up's into LayerNorm 2's scale (host), W_down's into the GELU output (the B operand), the input / output Linear's
// into the accumulators after the sum.  Accuracy cost measured on the CPU: profiles/r03_fp8_weight_experiment.json.
//
// ASSUMED instruction semantics (to be probed on the GPU first thing): __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src,
// scale, hi) converts bytes {0, 1} (hi = false) or {2, 3} (hi = true) of src, OCP e4m3 (bias 7, no infinities), times
// scale, to two bf16 (byte 0 / 2 in the low half).
#pragma once
#include "../mixer_fused.hpp"

namespace tapir {

struct FusedQArgs {
  FusedArgs base;                        // stream = [8 waves][frags_per_wave][64 lanes] x 8 bytes
  float s_in, s_out;                     // scales of the input / output Linear
  float s_dn[FM_MAX_BLOCKS];             // scale of W_down per block
  const float* ln2s[FM_MAX_BLOCKS];      // LayerNorm-2 scale x scale of W_up, per block
};

__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned src, bool hi) {
#ifdef TAPIR_HIPEMU
  auto dec = [](unsigned b) {            // OCP e4m3fn
    const unsigned s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, (int)e - 7);
    return s ? -v : v;
  };
  const unsigned h = hi ? src >> 16 : src;
  return pack_bf16x2(dec(h & 255), dec((h >> 8) & 255));
#else
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t r = hi ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, 1.0f, true)
                        : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, 1.0f, false);
  return __builtin_bit_cast(unsigned, r);
#endif
}
// lane's 8 fp8 (k offsets 0..7 of its row) -> the bf16 A fragment of MfmaStep<bf16_t>
__device__ __forceinline__ uint4 fp8x8_to_bf16x8(uint2 v) {
  return make_uint4(fp8x2_to_bf16x2(v.x, false), fp8x2_to_bf16x2(v.x, true), fp8x2_to_bf16x2(v.y, false),
                    fp8x2_to_bf16x2(v.y, true));
}

// mixer_fused.hpp's fused_gemm with 8-byte fragments converted at use
template <int RA, int NT, int GROUPS = 0, typename Epi = NoEpilogue, int RING = FM_RING, bool DB = true>
__device__ __forceinline__ void fused_gemm_q(const uint2*& wp, uint2 (&ring)[RING],
                                           const char* bbase, int bstride, int groups, int c, int g,
                                           f32x4 (&acc)[RA][NT], Epi epi = Epi()) {
  constexpr int G = RING / RA;
  static_assert(RING % RA == 0, "ring must hold whole k-steps");
  const char* brow = bbase + c * bstride;
  auto read_b = [&](int ks, uint4 (&fb)[NT]) {
    const int chunk = (ks * 4 + g) ^ c;
#pragma unroll
    for (int i = 0; i < NT; ++i)
      fb[i] = *reinterpret_cast<const uint4*>(brow + 16 * i * bstride + (chunk << 4));
  };
  if (GROUPS > 0) groups = GROUPS;
  const int ksteps = groups * G;
  uint4 fb0[NT], fb1[NT];   // (fb1 is dead when !DB)
  if (DB) read_b(0, fb0);
  auto group = [&](int kg) {
#pragma unroll
    for (int kk = 0; kk < G; ++kk) {
      if constexpr (DB) {
        uint4 (&nxt)[NT] = (kk & 1) ? fb0 : fb1;
        int ks1 = kg * G + kk + 1;
        ks1 = ks1 < ksteps ? ks1 : 0;   // past the end: any valid address (the values are not used)
        read_b(ks1, nxt);
      } else {
        // (NT > 3: no registers for a second set of B fragments; the other wave of the SIMD covers
        // the LDS round trip)
        read_b(kg * G + kk, fb0);
      }
      uint4 (&cur)[NT] = (DB && (kk & 1)) ? fb1 : fb0;
      sched_fence();
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const uint4 fa = fp8x8_to_bf16x8(ring[kk * RA + r]);
#pragma unroll
        for (int i = 0; i < NT; ++i) MfmaStep<bf16_t>::run(fa, cur[i], acc[r][i]);
        ring[kk * RA + r] = *wp;
        wp += 64;
        epi((kg * G + kk) * RA + r);
        sched_fence();
      }
    }
  };
  if constexpr (GROUPS > 0) {
#pragma unroll
    for (int kg = 0; kg < GROUPS; ++kg) group(kg);
  } else {
    for (int kg = 0; kg < groups; ++kg) group(kg);
  }
}

// RAGGED: T is not a multiple of 16 -- the tokens past the end of the clip are masked out of both
// temporal convolutions (zero padding at the clip end).
template <int NT, bool RAGGED>
__global__ __launch_bounds__(FM_THREADS) void mixer_fused_fp8w_kernel(FusedQArgs qa) {
  typedef bf16_t TA;
  const FusedArgs& a = qa.base;
  constexpr bool TRACE = false;
  constexpr int KS = 32, HC = 512;
  constexpr int ROWS = NT * 16;
  constexpr int RAU = HC / 8 / 16;          // hidden-row tiles of a wave per chunk
  constexpr int NC = kHidden4 / HC;         // hidden chunks per block
  constexpr int XN_STRIDE = kHidden * (int)sizeof(TA);
  constexpr int H_STRIDE = HC * (int)sizeof(TA);
  constexpr int XN_BYTES = ROWS * XN_STRIDE, H_BYTES = ROWS * H_STRIDE;
  constexpr int DN_GROUPS = (HC / KS) / (FM_RING / 4);
  static_assert(NT >= 1 && NT <= 3, "token tiles");
  static_assert(RAU * NT * 4 <= DN_GROUPS * FM_RING, "one GELU per down-projection fragment step");
  // LN2(x) image [ROWS][512], then TWO hidden chunks [ROWS][HC] (chunk c+1 is written while chunk c
  // is multiplied); the mixer-input rows use the same region at the start, and so do the
  // temporal-convolution parameters of a block (64 KiB) while its token mixing runs
  constexpr int PAR_BYTES = kHidden * FM_MIXW * 4;
  constexpr int ACT_BYTES = XN_BYTES + 2 * H_BYTES > PAR_BYTES ? XN_BYTES + 2 * H_BYTES : PAR_BYTES;
  static_assert(ACT_BYTES + 2 * ROWS * 8 * 8 + kHidden4 * 4 <= 160 * 1024, "LDS budget");
  __shared__ uint4 s_act[ACT_BYTES / 16];
  __shared__ __attribute__((aligned(16))) float2 s_stat[2][ROWS][8];   // per-wave (sum, M2) LayerNorm summaries
  __shared__ float s_bup[kHidden4];         // up-projection bias of the current block (see below)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n = blockIdx.x;
  const int T = a.T;
  char* const s_xn = reinterpret_cast<char*>(s_act);
  char* const s_h0 = s_xn + XN_BYTES;
  const int ch_lane = 64 * wave + 4 * g;    // channel of (a = 0, r = 0) of this lane

  // TRACE (tools/kbench.py --what fusedtrace): shader cycles (s_memtime) per phase, summed per wave
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int k) {
#ifndef TAPIR_HIPEMU
    if (TRACE) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
      if (k >= 0) tph[k] += t - tlast;
      tlast = t;
    }
#endif
  };
  tick(-1);

  // ---- weight stream: fill the ring (the loads fly while the input rows are staged)
  const uint2* wp = reinterpret_cast<const uint2*>(a.stream) + ((long)wave * a.frags_per_wave) * 64 + lane;
  uint2 ring[FM_RING];
#pragma unroll
  for (int s = 0; s < FM_RING; ++s) { ring[s] = *wp; wp += 64; }

  // ---- stage the mixer-input rows of this track: [ROWS][ld_in], rows >= T zero
  const int in_stride = a.ld_in * (int)sizeof(TA);
  {
    const int cpr = in_stride >> 4;   // 16-byte chunks per row (a multiple of 16)
    const uint4* src = reinterpret_cast<const uint4*>(
        reinterpret_cast<const char*>(a.mlp_in) + (long)n * T * in_stride);
    for (int id = tid; id < ROWS * cpr; id += FM_THREADS) {
      const int row = id / cpr, q = id - row * cpr;
      const int tok = NT * (row & 15) + (row >> 4);   // LDS row 16 i + c holds token NT c + i
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (tok < T) v = src[tok * cpr + q];
      s_act[row * cpr + (q ^ (row & 15))] = v;
    }
  }
  lds_barrier();

  // ---- residual stream <- input Linear (tapir_model.py:139): x = mlp_in . W0^T + b0
  f32x4 xr[4][NT];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < NT; ++i) xr[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  fused_gemm_q<4, NT>(wp, ring, s_xn, in_stride, a.ld_in / KS / (FM_RING / 4), c, g, xr);
#pragma unroll
  for (int q = 0; q < 4; ++q) {     // x = s_in * (W_q . in) + b0: the matrix' scale comes out of the sum
    const f32x4 b = gload4(a.b0 + ch_lane + 16 * q);
#pragma unroll
    for (int i = 0; i < NT; ++i) xr[q][i] = xr[q][i] * qa.s_in + b;
  }
  tick(0);

  float valid[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) valid[i] = (NT * c + i < T) ? 1.0f : 0.0f;

  // per-token LayerNorm statistics over the 512 channels spread over lane groups and waves: every
  // wave reduces its 64 channels to (sum, M2 about ITS mean) -- two-pass inside the wave -- and the
  // eight summaries merge with Chan's formula (equal counts): ONE barrier per LayerNorm and the
  // numerics of the two-pass form.  Two summary buffers alternate between consecutive LayerNorms.
  int ln_phase = 0;
  auto ln_stats = [&](float (&mean)[NT], float (&rstd)[NT], bool wait_params = false) {
    float2 (*stat)[8] = s_stat[ln_phase];
    ln_phase ^= 1;
    float s[NT], m2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += (xr[q][i][0] + xr[q][i][1]) + (xr[q][i][2] + xr[q][i][3]);
      s[i] = t;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) s[i] += __shfl_xor(s[i], 16);
#pragma unroll
    for (int i = 0; i < NT; ++i) s[i] += __shfl_xor(s[i], 32);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float mw = s[i] * (1.0f / 64.0f);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
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
    if (wait_params) dma_wait<0>();   // LN1: the parameter copies of this block (params_dma) have landed; the ring is idle here
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float2 p[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(&stat[16 * i + c][2 * k]);
        p[2 * k] = make_float2(v.x, v.y); p[2 * k + 1] = make_float2(v.z, v.w);
      }
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) tot += p[k].x;
      const float mu = tot * (1.0f / kHidden);
      float M2 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = p[k].x * (1.0f / 64.0f) - mu;
        M2 += p[k].y + 64.0f * d * d;
      }
      mean[i] = mu;
      rstd[i] = 1.0f / sqrtf(M2 * (1.0f / kHidden) + kLnEps);
    }
  };

  // LN(x) * scale -> operand type -> LDS image at s_xn
  auto write_xn = [&](const float* scale, const float (&mean)[NT], const float (&rstd)[NT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 sc = gload4(scale + ch_lane + 16 * q);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        store_act4<TA>(s_xn, XN_STRIDE, 16 * i + c, ch_lane + 16 * q, c,
                       (xr[q][i][0] - mean[i]) * rstd[i] * sc[0], (xr[q][i][1] - mean[i]) * rstd[i] * sc[1],
                       (xr[q][i][2] - mean[i]) * rstd[i] * sc[2], (xr[q][i][3] - mean[i]) * rstd[i] * sc[3]);
    }
  };

  const int hid_lane = wave * (HC / 8) + 4 * g;   // hidden unit (within a chunk) of (row tile 0, reg 0)

  // Temporal-convolution parameters of a block (64 KiB) go through LDS: read straight from global
  // memory, channel by channel, each read is a dependent L2 round trip with nothing to hide it
  // behind (16 per lane and block).  They are copied by LDS-DMA (global_load_lds: no registers) into
  // the activation region -- dead from the barrier that ends a block until LN2 -- right after that
  // barrier; ln_stats waits for the copies just before ITS barrier, so the latency passes under the
  // LN1 statistics.  (The first version carried them across the barrier in 32 VGPRs: hipcc spilled all
  // of them to scratch -- load, wait, spill, eight times over, then eight reloads with vmcnt(0) in
  // front of each LDS write, every one of them draining the weight ring: 100 spilled VGPRs,
  // 240 MB of scratch writes per launch.)
  constexpr int PARV = PAR_BYTES / 16 / FM_THREADS;
  auto params_dma = [&](int blk) {
    const float* src = a.blocks[blk].mixw;
    char* dst = reinterpret_cast<char*>(s_act) + 1024 * wave;     // 1 KiB per wave and instruction
#pragma unroll
    for (int k = 0; k < PARV; ++k) glds16(src + (tid + k * FM_THREADS) * 4, dst + 8192 * k);
  };
  lds_barrier();   // every wave is done with the input rows: the region is reused from here on

  for (int b = 0; b < a.nblocks; ++b) {
    const FusedBlockParams& bp = a.blocks[b];
    float mean[NT], rstd[NT];
    params_dma(b);

    // ---- token mixing (tapir_model.py:39-89,111-119): LN1 -> depthwise conv k=3 (x4 channels) ->
    // GELU -> depthwise conv k=3 -> sum of each group of 4 -> + skip, per channel, along time.
    // Two adjacent channels of the lane at a time (registers 2 rp, 2 rp + 1 of a fragment), all
    // arithmetic on f32x2 -> packed f32 instructions; the parameters of a channel pair are
    // interleaved in LDS ([32][2] floats).
    ln_stats(mean, rstd, true);
    tick(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        // (opaque: the parameter addresses of a lane are otherwise all computed up front and spilled)
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
        // one multiplier m (of the x4 depthwise expansion) at a time: 8 parameter pairs live
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
        // y[t] = sum_m b2[m] + S0[t-1] + S1[t] + S2[t+1]   (shifting the three partial sums instead
        // of the four GELU outputs: 2 shifted values per token tile instead of 8)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const f32x2 pa = i > 0 ? s0[i - 1] : f32x2{lane_up(s0[NT - 1].x, lane), lane_up(s0[NT - 1].y, lane)};
          const f32x2 pb = i + 1 < NT ? s2[i + 1] : f32x2{lane_dn(s2[0].x, lane), lane_dn(s2[0].y, lane)};
          const f32x2 y = bsum + pa + s1[i] + pb;
          xr[q][i][2 * rp] += y.x;
          xr[q][i][2 * rp + 1] += y.y;
        }
        // one channel pair at a time: without the fence the scheduler hoists the parameter reads of
        // all 16 channels of the lane to the top and spills the residual
        sched_fence();
      }
    }

    tick(2);
    // ---- channel MLP (tapir_model.py:92-98,121-123): x += W_dn . gelu(W_up . LN2(x) + b_up) + b_dn
    ln_stats(mean, rstd);
    write_xn(qa.ln2s[b], mean, rstd);        // LN2's scale times W_up's scale
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bd = gload4(bp.bdn + ch_lane + 16 * q);
#pragma unroll
      for (int i = 0; i < NT; ++i) xr[q][i] += bd;
    }
    // The up-projection bias goes through LDS: a vector load inside the chunk loop would be YOUNGER
    // than the FM_RING weight loads in flight, and waiting for it (vmcnt) would drain the ring at
    // every chunk.  Here the ring's loads are a whole token-mixing phase old.
    *reinterpret_cast<f32x4*>(&s_bup[tid * 4]) = gload4(bp.bup + tid * 4);
    lds_barrier();   // LN2(x) (and the bias) visible to every wave
    tick(3);

    // Chunks of HC hidden units, software-pipelined over two LDS buffers:
    //   up(0); for c: { down(c-1) with the GELU of chunk c slotted between its MFMAs -> h[c & 1];
    //                   barrier; up(c+1) }; down(NC-1); barrier
    // (= the order of the weight stream: U0 U1 D0 U2 D1 ... D(NC-1)).  The GELU + pack + LDS store of
    // a chunk (4 GELUs per hidden unit and token tile, as many as the token mixing has) is VALU
    // work with no MFMA of its own to hide behind; the down-projection of the PREVIOUS chunk is
    // independent of it.  One barrier per chunk: it publishes h[c & 1] and retires the reads of
    // h[(c-1) & 1], which the GELU of chunk c+1 overwrites only after the next barrier.
    const float s_dn = qa.s_dn[b];
    f32x4 ua[RAU][NT];
    auto up = [&](int hc) {
#pragma unroll
      for (int r = 0; r < RAU; ++r) {
        const f32x4 bu = *reinterpret_cast<const f32x4*>(&s_bup[hc * HC + hid_lane + 16 * r]);
#pragma unroll
        for (int i = 0; i < NT; ++i) ua[r][i] = bu;
      }
      fused_gemm_q<RAU, NT>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (FM_RING / RAU), c, g, ua);
    };
    // GELU of value v (0 .. RAU*NT*4-1) of the up accumulators; every fourth one stores its fragment
    auto gelu_step = [&](int v, char* hbuf) {
      if (v < RAU * NT * 4) {
        const int item = v >> 2, r = item / NT, i = item % NT, k = v & 3;
        ua[r][i][k] = gelu_tanh(ua[r][i][k]) * s_dn;       // W_down's scale rides on its B operand
        if (k == 3)
          store_act4<TA>(hbuf, H_STRIDE, 16 * i + c, hid_lane + 16 * r, c, ua[r][i][0], ua[r][i][1],
                         ua[r][i][2], ua[r][i][3]);
      }
    };
    up(0);
    tick(4);
#pragma unroll
    for (int v = 0; v < RAU * NT * 4; ++v) gelu_step(v, s_h0);   // chunk 0: nothing to overlap with
    tick(5);
    lds_barrier();
    tick(7);
    for (int hc = 1; hc < NC; ++hc) {
      up(hc);
      tick(4);
      char* const hprev = s_h0 + ((hc - 1) & 1) * H_BYTES;
      char* const hcur = s_h0 + (hc & 1) * H_BYTES;
      fused_gemm_q<4, NT, DN_GROUPS>(wp, ring, hprev, H_STRIDE, DN_GROUPS, c, g, xr,
                                       [&](int step) { gelu_step(step, hcur); });
      tick(6);
      lds_barrier();
      tick(7);
    }
    fused_gemm_q<4, NT>(wp, ring, s_h0 + ((NC - 1) & 1) * H_BYTES, H_STRIDE, DN_GROUPS, c, g, xr);
    tick(6);
    lds_barrier();   // every wave is done with the activation images before the next block reuses them
    tick(7);
  }

  // ---- final LayerNorm + output Linear (tapir_model.py:154-155): 388 outputs, rows padded to 512
  {
    float mean[NT], rstd[NT];
    ln_stats(mean, rstd);
    write_xn(a.lnF, mean, rstd);
    lds_barrier();
    f32x4 oa[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o0 = ch_lane + 16 * q;
#pragma unroll
      for (int i = 0; i < NT; ++i) oa[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    fused_gemm_q<4, NT>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (FM_RING / 4), c, g, oa);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o0 = ch_lane + 16 * q;
      f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
      if (o0 < kMixOut) bo = gload4(a.bout + o0);
#pragma unroll
      for (int i = 0; i < NT; ++i) oa[q][i] = oa[q][i] * qa.s_out + bo;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o0 = ch_lane + 16 * q;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int t = NT * c + i;
        if (o0 < kMixOut && t < T) {   // (experiment: fetched on the spot, not batched as in mixer_fused.hpp)
          const long r = (long)n * T + t;
          f32x4 prev = f32x4{0.f, 0.f, 0.f, 0.f};
          EmitState st = EmitState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (a.fuse_update) {
            prev = fused_prev_feats(a, r, n, o0);
            if (o0 == 0) st = fused_prev_state(a, r);
          }
          fused_emit(a, r, o0, oa[q][i], prev, st);
        }
      }
    }
  }
  if (TRACE && a.dbg_times != nullptr && lane == 0) {
    tick(0);   // final LayerNorm + output Linear are booked with the input Linear
    long long* o = a.dbg_times + ((long)n * FM_WAVES + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (long long)tph[k];
  }
}

inline void launch_mixer_fused_fp8w(const FusedQArgs& a, hipStream_t s) {
  const int nt = (a.base.T + 15) / 16;
  const bool ragged = a.base.T % 16 != 0;
  const dim3 grid((unsigned)a.base.N), block(FM_THREADS);
  if (nt == 1) {
    if (ragged) TAPIR_LAUNCH((mixer_fused_fp8w_kernel<1, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_fp8w_kernel<1, false>), grid, block, s, a);
  } else if (nt == 2) {
    if (ragged) TAPIR_LAUNCH((mixer_fused_fp8w_kernel<2, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_fp8w_kernel<2, false>), grid, block, s, a);
  } else {
    if (ragged) TAPIR_LAUNCH((mixer_fused_fp8w_kernel<3, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_fp8w_kernel<3, false>), grid, block, s, a);
  }
}

}  // namespace tapir
