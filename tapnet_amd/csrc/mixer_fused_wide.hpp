// Wide variant of the track-resident mixer kernel (mixer_fused.hpp): SIX token tiles per workgroup --
// two tracks of up to 48 frames, or one track of up to 96 frames -- bf16 build only.
//
// Why.  With one 48-frame track per workgroup every CU streams the weights of a block (4.19 MB) for 48
// tokens and the L2 -> CU fill (64 B/clk) bounds the MLP at 64 % of the MFMA peak.  With 96 tokens per
// weight fragment (6 MFMAs instead of 3) the same stream feeds twice the arithmetic: the MLP becomes
// MFMA-bound.  It needs >= 2 tracks per CU to pay (>= 512 tracks: BASELINE configs[2], 1024 queries per
// clip), and it is what lets 96-frame clips (configs[4]) use the resident form at all.
//
// Differences from the 3-tile kernel, all forced by the register file (256 VGPRs per lane at 8 waves):
// residual 96 registers (4 x 6 fragments), so the up-projection works on chunks of 256 hidden units
// (2 x 6 accumulator fragments = 48 registers), the weight ring is 8 fragments deep (the stream has
// slack now), B fragments are single-buffered, and one hidden chunk lives in LDS (96 KiB LN2(x) +
// 48 KiB chunk): up, GELU, barrier, down, barrier per chunk.  The weight stream is packed for this
// chunking (sequential order U0 D0 U1 D1 ...; tapir_finalize_weights builds it on first use).
// Tokens of a track are interleaved over ITS tiles (column c of tile j of a track = token NTT c + j).
#pragma once
#include "mixer_fused.hpp"

namespace tapir {

constexpr int FMW_RING = 8;
constexpr int FMW_HC = 256;

inline long fused_wide_frags_per_wave(int k0_pad, int nblocks) {
  const long in = (long)(k0_pad / 32) * 4;
  const long up = (long)(FMW_HC / 8 / 16) * (kHidden / 32);
  const long dn = 4L * (FMW_HC / 32);
  return in + nblocks * (kHidden4 / FMW_HC) * (up + dn) + 4L * (kHidden / 32) + FMW_RING;
}

// NTT token tiles per track, NTRK tracks per workgroup; NTT * NTRK in 4 .. 6.
// SIM (TAPIR_EXPERIMENTS builds only, tools/kbench.py): 1 = TIMING-ONLY stand-in for a paired design in which
// two CUs share two tracks and each takes half of the hidden units of both -- one workgroup per track, token
// mixing for one track, half of the hidden chunks for both; its outputs are meaningless.
template <int NTT, int NTRK, bool RAGGED, int SIM = 0, bool TRACE = false>
__global__ __launch_bounds__(FM_THREADS) void mixer_fused_wide_kernel(FusedArgs a) {
  typedef bf16_t TA;
  constexpr int NT = NTT * NTRK;
  constexpr int KS = 32, HC = FMW_HC, RING = FMW_RING;
  constexpr int ROWS = NT * 16;
  constexpr int RAU = HC / 8 / 16;          // 2
  constexpr int NC = kHidden4 / HC;         // 8
  constexpr int XN_STRIDE = kHidden * 2, H_STRIDE = HC * 2;
  constexpr int XN_BYTES = ROWS * XN_STRIDE, H_BYTES = ROWS * H_STRIDE;
  constexpr int PAR_BYTES = kHidden * FM_MIXW * 4;
  static_assert(NT >= 4 && NT <= 6, "token tiles");
  static_assert(XN_BYTES + H_BYTES >= PAR_BYTES && 2 * ROWS * 8 * 8 <= H_BYTES, "region reuse");
  static_assert(XN_BYTES + H_BYTES + kHidden4 * 4 <= 160 * 1024, "LDS budget");
  __shared__ uint4 s_act[(XN_BYTES + H_BYTES) / 16];
  __shared__ float s_bup[kHidden4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int T = a.T;
  char* const s_xn = reinterpret_cast<char*>(s_act);
  char* const s_h = s_xn + XN_BYTES;
  // LayerNorm summaries live in the hidden-chunk region (dead outside the chunk loop)
  float2 (*const s_stat)[ROWS][8] = reinterpret_cast<float2 (*)[ROWS][8]>(s_h);
  const int ch_lane = 64 * wave + 4 * g;
  const int trk0 = SIM ? (int)(blockIdx.x & ~1u) : (int)blockIdx.x * NTRK;   // first track of this workgroup

  // TRACE (tools/kbench.py --what widetrace): shader cycles per phase, summed per wave
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

  const uint4* wp = a.stream + ((long)wave * a.frags_per_wave) * 64 + lane;
  uint4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = *wp; wp += 64; }

  // LDS row 16 i + c <-> (track trk0 + i / NTT, token NTT c + i % NTT)
  const int in_stride = a.ld_in * 2;
  {
    const int cpr = in_stride >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(a.mlp_in);
    // batches of unconditional loads from clamped coordinates, then the stores (mixer_fused.hpp: a load under a lane
    // condition is waited for on the spot)
    constexpr int STG = 8;
    const int total = ROWS * cpr;
    for (int base = tid; base < total; base += STG * FM_THREADS) {
      uint4 v[STG];
      int dst[STG];          // LDS chunk index, -1 past the end of the image
      bool live[STG];        // false: a zero row (token >= T or track >= N)
#pragma unroll
      for (int k = 0; k < STG; ++k) {
        const int id = base + k * FM_THREADS;
        const int idc = id < total ? id : total - 1;
        const int row = idc / cpr, q = idc - row * cpr;
        const int i = row >> 4;
        const int tok = NTT * (row & 15) + (i % NTT);
        const int trk = trk0 + i / NTT;
        live[k] = tok < T && trk < a.N;
        v[k] = src[((long)(trk < a.N ? trk : a.N - 1) * T + (tok < T ? tok : T - 1)) * cpr + q];
        dst[k] = id < total ? row * cpr + (q ^ (row & 15)) : -1;
      }
#pragma unroll
      for (int k = 0; k < STG; ++k) pin(v[k]);
#pragma unroll
      for (int k = 0; k < STG; ++k)
        if (dst[k] >= 0) s_act[dst[k]] = live[k] ? v[k] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  lds_barrier();

  f32x4 xr[4][NT];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = gload4(a.b0 + ch_lane + 16 * q);
#pragma unroll
    for (int i = 0; i < NT; ++i) xr[q][i] = b;
  }
  fused_gemm<TA, 4, NT, 0, NoEpilogue, RING, false>(wp, ring, s_xn, in_stride, a.ld_in / KS / (RING / 4), c, g, xr);
  lds_barrier();   // the block parameters and the LN summaries overwrite the input image
  tick(0);

  float valid[NTT];   // the same for every track of the workgroup
#pragma unroll
  for (int j = 0; j < NTT; ++j) valid[j] = (NTT * c + j < T) ? 1.0f : 0.0f;

  int ln_phase = 0;
  auto ln_stats = [&](float (&mean)[NT], float (&rstd)[NT]) {
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

  const int hid_lane = wave * (HC / 8) + 4 * g;
  constexpr int PARV = PAR_BYTES / 16 / FM_THREADS;

  for (int b = 0; b < a.nblocks; ++b) {
    const FusedBlockParams& bp = a.blocks[b];
    float mean[NT], rstd[NT];
    // the previous block's last readers of the activation region passed its final barrier.  (Through
    // registers: the LDS-DMA copy that removed 62 spilled VGPRs from the 3-tile kernel made THIS kernel
    // 9-14 % slower relative to the separate-launch path measured in the same process.)
    {
      f32x4 parv[PARV];
#pragma unroll
      for (int k = 0; k < PARV; ++k) parv[k] = gload4(bp.mixw + (tid + k * FM_THREADS) * 4);
      f32x4* dst = reinterpret_cast<f32x4*>(s_act);
#pragma unroll
      for (int k = 0; k < PARV; ++k) dst[tid + k * FM_THREADS] = parv[k];
    }
    // LN1's summaries use the chunk region, the parameters the first 64 KiB of the LN2 image
    ln_stats(mean, rstd);
    tick(1);
    // ---- token mixing, one track and one channel pair at a time (see mixer_fused.hpp)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x4* pw = reinterpret_cast<const f32x4*>(s_act) +
                          opaque((ch_lane + 16 * q + 2 * rp) >> 1) * (2 * FM_MIXW / 4);
#pragma unroll
        for (int tk = 0; tk < (SIM ? 1 : NTRK); ++tk) {
          f32x2 xc[NTT], xp[NTT], xq[NTT], s0[NTT], s1[NTT], s2[NTT];
          const f32x2 zero = f32x2{0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NTT; ++j) {
            const int i = tk * NTT + j;
            xc[j] = (f32x2{xr[q][i][2 * rp], xr[q][i][2 * rp + 1]} - mean[i]) * rstd[i];
            if (RAGGED) xc[j] = xc[j] * valid[j];
          }
#pragma unroll
          for (int j = 0; j < NTT; ++j) {
            xp[j] = j > 0 ? xc[j - 1] : f32x2{lane_up(xc[NTT - 1].x, lane), lane_up(xc[NTT - 1].y, lane)};
            xq[j] = j + 1 < NTT ? xc[j + 1] : f32x2{lane_dn(xc[0].x, lane), lane_dn(xc[0].y, lane)};
            s0[j] = zero; s1[j] = zero; s2[j] = zero;
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
            for (int j = 0; j < NTT; ++j) {
              f32x2 u = b1m;
              u = __builtin_elementwise_fma(w10, xp[j], u);
              u = __builtin_elementwise_fma(w11, xc[j], u);
              u = __builtin_elementwise_fma(w12, xq[j], u);
              f32x2 gl = gelu_tanh2(u);
              if (RAGGED) gl = gl * valid[j];
              s0[j] = __builtin_elementwise_fma(w20, gl, s0[j]);
              s1[j] = __builtin_elementwise_fma(w21, gl, s1[j]);
              s2[j] = __builtin_elementwise_fma(w22, gl, s2[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < NTT; ++j) {
            const int i = tk * NTT + j;
            const f32x2 pa = j > 0 ? s0[j - 1] : f32x2{lane_up(s0[NTT - 1].x, lane), lane_up(s0[NTT - 1].y, lane)};
            const f32x2 pb = j + 1 < NTT ? s2[j + 1] : f32x2{lane_dn(s2[0].x, lane), lane_dn(s2[0].y, lane)};
            const f32x2 y = bsum + pa + s1[j] + pb;
            xr[q][i][2 * rp] += y.x;
            xr[q][i][2 * rp + 1] += y.y;
          }
          sched_fence();
        }
      }
    }

    tick(2);
    // ---- channel MLP
    ln_stats(mean, rstd);   // (its barrier also ends every wave's reads of the parameters)
    write_xn(bp.ln2, mean, rstd);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bd = gload4(bp.bdn + ch_lane + 16 * q);
#pragma unroll
      for (int i = 0; i < NT; ++i) xr[q][i] += bd;
    }
    *reinterpret_cast<f32x4*>(&s_bup[tid * 4]) = gload4(bp.bup + tid * 4);
    lds_barrier();
    tick(3);
    for (int hc = 0; hc < (SIM ? NC / 2 : NC); ++hc) {
      f32x4 ua[RAU][NT];
#pragma unroll
      for (int r = 0; r < RAU; ++r) {
        const f32x4 bu = *reinterpret_cast<const f32x4*>(&s_bup[hc * HC + hid_lane + 16 * r]);
#pragma unroll
        for (int i = 0; i < NT; ++i) ua[r][i] = bu;
      }
      fused_gemm<TA, RAU, NT, 0, NoEpilogue, RING, false>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (RING / RAU),
                                                          c, g, ua);
      tick(4);
#pragma unroll
      for (int r = 0; r < RAU; ++r)
#pragma unroll
        for (int i = 0; i < NT; ++i)
          store_act4<TA>(s_h, H_STRIDE, 16 * i + c, hid_lane + 16 * r, c, gelu_tanh(ua[r][i][0]),
                         gelu_tanh(ua[r][i][1]), gelu_tanh(ua[r][i][2]), gelu_tanh(ua[r][i][3]));
      tick(5);
      lds_barrier();
      tick(7);
      fused_gemm<TA, 4, NT, 0, NoEpilogue, RING, false>(wp, ring, s_h, H_STRIDE, (HC / KS) / (RING / 4), c, g, xr);
      tick(6);
      lds_barrier();
      tick(7);
    }
  }

  // ---- final LayerNorm + output Linear
  {
    float mean[NT], rstd[NT];
    ln_stats(mean, rstd);
    write_xn(a.lnF, mean, rstd);
    lds_barrier();
    // two passes of two output-row tiles: a third set of 4 x NT accumulators does not fit beside xr
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 oa[2][NT];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o0 = ch_lane + 16 * (2 * half + q);
        f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
        if (o0 < kMixOut) bo = gload4(a.bout + o0);
#pragma unroll
        for (int i = 0; i < NT; ++i) oa[q][i] = bo;
      }
      // what the state update adds to: fetched in one batch (mixer_fused.hpp::fused_emit), in flight under the GEMM
      f32x4 prev[2][NT];
      EmitState st[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        st[i] = EmitState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) prev[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (a.fuse_update) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int t = NTT * c + (i % NTT), trk = trk0 + i / NTT;
          const long trc = trk < a.N ? trk : a.N - 1;
          const long r = trc * T + (t < T ? t : T - 1);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int o0 = ch_lane + 16 * (2 * half + q);
            prev[q][i] = fused_prev_feats(a, r, trc, o0 < kMixOut ? o0 : kMixOut - 4);
          }
          if (half == 0 && wave == 0) st[i] = fused_prev_state(a, r);   // channels 0..3: wave 0, first row tile
        }
      }
      fused_gemm<TA, 2, NT, 0, NoEpilogue, RING, false>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (RING / 2), c, g, oa);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        pin(st[i]);
#pragma unroll
        for (int q = 0; q < 2; ++q) pin(prev[q][i]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o0 = ch_lane + 16 * (2 * half + q);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int t = NTT * c + (i % NTT), trk = trk0 + i / NTT;
          if (o0 < kMixOut && t < T && trk < a.N) fused_emit(a, (long)trk * T + t, o0, oa[q][i], prev[q][i], st[i]);
        }
      }
    }
  }
  if (TRACE && a.dbg_times != nullptr && lane == 0) {
    tick(0);
    long long* o = a.dbg_times + ((long)blockIdx.x * FM_WAVES + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (long long)tph[k];
  }
}

// shapes the wide kernel covers: bf16, non-causal; one track of 49..96 frames, or pairs of tracks of up
// to 48 frames
inline bool fused_wide_supported(int T, int k0_pad, bool causal, bool has_ctx) {
  if (causal || has_ctx || T < 1 || T > 96) return false;
  // the staged input image (96 rows of k0_pad bf16) has to fit the activation region (144 KiB)
  if (96L * k0_pad * 2 > 96L * (kHidden + FMW_HC) * 2) return false;
  return (k0_pad * 2) % 256 == 0 && (k0_pad / 32) % (FMW_RING / 4) == 0;
}

inline void launch_mixer_fused_wide(const FusedArgs& a, hipStream_t s) {
  const int T = a.T;
  const bool ragged = T % 16 != 0;
  const int ntt = T <= 48 ? (T + 15) / 16 : 0;
  const dim3 block(FM_THREADS);
#define TAPIR_WIDE(NTT_, NTRK_)                                                                          \
  do {                                                                                                   \
    const dim3 grid((unsigned)((a.N + (NTRK_) - 1) / (NTRK_)));                                            \
    if (ragged) TAPIR_LAUNCH((mixer_fused_wide_kernel<NTT_, NTRK_, true>), grid, block, s, a);            \
    else TAPIR_LAUNCH((mixer_fused_wide_kernel<NTT_, NTRK_, false>), grid, block, s, a);                  \
  } while (0)
#ifdef TAPIR_EXPERIMENTS
  if (a.dbg_times != nullptr && ntt == 3 && !ragged && !a.pair_sim) {   // phase trace (tools/kbench.py --what widetrace)
    hipLaunchKernelGGL((mixer_fused_wide_kernel<3, 2, false, 0, true>), dim3((unsigned)((a.N + 1) / 2)), block, 0, s, a);
    return;
  }
  if (a.pair_sim && ntt == 3 && !ragged) {
    hipLaunchKernelGGL((mixer_fused_wide_kernel<3, 2, false, 1>), dim3((unsigned)a.N), block, 0, s, a);
    return;
  }
#endif
  if (ntt == 2) TAPIR_WIDE(2, 2);
  else if (ntt == 3) TAPIR_WIDE(3, 2);
  else if (T <= 64) TAPIR_WIDE(4, 1);
  else if (T <= 80) TAPIR_WIDE(5, 1);
  else TAPIR_WIDE(6, 1);
#undef TAPIR_WIDE
}

}  // namespace tapir
