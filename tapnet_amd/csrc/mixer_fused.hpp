// Track-resident PIPs MLP-mixer: the WHOLE PIPSMLPMixer (tapnet/models/tapir_model.py:127-156) of
// one track -- input Linear, num_mixer_blocks x PIPsConvBlock (:101-124: LayerNorm, temporal
// depthwise convs :39-89, GELU, LayerNorm, channel MLP 512 -> 2048 -> 512 :92-98, both skips), final
// LayerNorm and output Linear -- in ONE workgroup, with the residual stream of the track held in
// registers from the first instruction to the last.
//
// Why this shape.  With separate launches (mixer.hpp + gemm.hpp) every block writes and re-reads
// x [R,512] f32 twice, LN(x) and the 2048-wide hidden tensor (100 MB per block at 12288 token rows,
// 4.8 GB per call) and pays 37 launches per refinement iteration; the two GEMMs of a block sat at
// 27-30 % of the MFMA peak.  Tokens of different tracks never interact inside the mixer, and the
// only operator that couples the frames of a track is a depthwise convolution over time, so a
// track (T <= 48 frames x 512 channels) is a closed unit: nothing but the mixer input row and the
// 388 outputs of a token ever has to touch HBM.
//
// Layout.  512 threads = 8 waves.  Wave w owns output channels [64 w, 64 w + 64) of every
// 512-wide tensor.  The residual x[t][ch] lives in MFMA accumulator layout (v_mfma 16x16, weights
// on the A port, tokens on the B port): register xr[a][i][r] of lane l (c = l & 15, g = l >> 4)
// is channel 64 w + 16 a + 4 g + r of token t = NT c + i (NT = token tiles).  In this layout
//   * the channel MLP's second GEMM accumulates straight into the residual (+ skip for free),
//   * a lane holds 4 consecutive channels of one token: LayerNorm / GELU outputs go to LDS as
//     8-byte (bf16) stores in the [token][channel] image the next GEMM reads as its B operand,
//   * tokens are interleaved over the token tiles (column c of tile i = token NT c + i), so the
//     temporal convolutions find t-1 / t+1 in the same lane, one DPP row shift at the tile ends.
// Weights are never shared between waves (a wave multiplies ITS 64 output rows, or its slice of the
// hidden rows, by all tokens), so they do not go through LDS at all: the host packs them per wave
// into one linear stream of 1-KiB MFMA A-fragments in exactly the order the wave consumes them
// (tapir_finalize_weights), and the wave keeps FM_RING = 8 fragment loads (global_load_dwordx4
// straight to VGPRs, 32 VGPRs) in flight across phase boundaries and barriers.  LDS holds only what
// waves exchange: LN2(x) [T,512] and one 512-wide (bf16; 256 f32) chunk of the hidden tensor.
//
// Roofline.  A CU streams both weight matrices of a block (4.2 MB bf16) from L2 for ONE track:
// 64 B/clk/CU on the L2 -> CU path = ~32 us per block, against 21 us of MFMA time at the dense peak
// (2 x 2 x 48 x 512 x 2048 flop per block and track) -- the kernel is bound by the L2 -> CU fill at
// T = 48, i.e. at 64 % of the MFMA peak, plus the VALU time of the temporal convolutions (two GELUs
// per channel and frame, quarter-rate transcendentals) which nothing overlaps within one track.
#pragma once
#include "common.hpp"
#include "gemm.hpp"    // MfmaStep
#include "mixer.hpp"   // kLnEps
#include "pips.hpp"    // PatchArgs, patch_row (fuse_patch)

namespace tapir {

constexpr int FM_WAVES = 8;
constexpr int FM_THREADS = FM_WAVES * 64;
// weight fragments (1 KiB each) in flight per wave.  8, not 16: interleaved on one box the launch is
// 2.5-3 % faster (757 / 749 against 782 / 770 us at 256 tracks) -- the 32 registers it frees take the
// kernel from 38 spilled VGPRs to 5, and the fill-bound stream does not need the deeper queue.
#ifndef TAPIR_FM_RING
#define TAPIR_FM_RING 8
#endif
constexpr int FM_RING = TAPIR_FM_RING;
constexpr int FM_MIXW = 32;          // floats per channel of packed temporal-conv parameters
constexpr int FM_OUT_PAD = 512;      // output Linear rows padded to 8 waves x 64
constexpr int FM_MAX_BLOCKS = 16;    // per-block parameter pointers travel in the kernel arguments

template <typename TA> struct FusedCfg;
template <> struct FusedCfg<bf16_t> {   // one 16-byte chunk = 8 bf16: a 16x16x32 MFMA k-slice
  static constexpr int EPC = 8, KS = 32, HC = 512, MAX_NT = 3;
};
template <> struct FusedCfg<float> {    // one 16-byte chunk = 4 f32: four 16x16x4 MFMAs, k = 16
  static constexpr int EPC = 4, KS = 16, HC = 128, MAX_NT = 3;
};

struct FusedBlockParams {
  const float* mixw;   // [256 channel pairs][4 multipliers m][8][2]: per channel and m: w1[m][0..2] * ln1,
                       // b1[m], w2[m][0..2], (m == 0: sum_m b2[m]) -- the two channels of a pair interleaved
  const float* ln2;    // [512]
  const float* bup;    // [2048]
  const float* bdn;    // [512]
};

struct FusedArgs {
  const void* mlp_in;            // [N*T, ld_in] operand type (patch_corr_kernel output)
  int ld_in;                     // = k0_pad: ld_in * sizeof(TA) is a multiple of 256 bytes
  const uint4* stream;           // [8 waves][frags_per_wave][64 lanes] packed A fragments
  long frags_per_wave;
  const float* b0;               // [512] bias of the input Linear
  FusedBlockParams blocks[FM_MAX_BLOCKS];   // by value: kernel-argument (scalar) loads
  int nblocks;
  const float* lnF;              // [512] final LayerNorm scale
  const float* bout;             // [388]
  float* res;                    // [N*T, 388]
  int N, T;
  long long* dbg_times;          // TRACE build: [N][8 waves][8] shader-cycle totals per phase
  int pair_sim;                  // TAPIR_EXPERIMENTS builds: timing-only stand-in of the wide kernel (mixer_fused_wide.hpp)
  // refine_pips's state update (tapir_model.py:613-623, 1026-1039) applied by the output stage itself instead of a
  // separate launch that re-reads res [R,388]: pos += d_xy * (orig / resized), occ += d, expd += d, feats += d,
  // this iteration's output slices.  fuse_update = 0: plain res output (tapir_pips_mixer).
  int fuse_update;
  UpdateArgs upd;
  // refine_pips's front half (tapir_model.py:496-594; pips.hpp) inside the prologue: every wave builds the input rows of
  // its share of the track's frames -- header, features, 7x7 correlations -- straight into the LDS input image, instead
  // of a separate launch writing mlp_in [R, ld] to HBM for this kernel to read back.  fuse_patch = 0: rows from mlp_in.
  int fuse_patch;
  PatchArgs patch;
};

// Output of the mixer for token row r, output channels o0 .. o0 + 3 (o0 a multiple of 4, < 388): either stored to
// res, or applied to the running estimate exactly as update_kernel does (same operations in the same order: the two
// forms are bit-identical).  Channels 0..3 = [dx, dy, d_occ, d_expd] sit in ONE lane; channels 4.. are feats[o0 - 4 ..].
//
// What the update adds TO is fetched ahead of the output Linear (fused_prev_feats / fused_prev_state), not inside
// fused_emit: there every one of a lane's 4 x NT read-modify-writes was a load under a lane condition, which hipcc
// waits for on the spot -- twelve dependent memory round trips at the tail of every launch (round 4, found in the ISA
// like patch_corr_kernel's).  The fetches are unconditional and back to back from clamped coordinates (a lane that
// emits nothing reads something readable) and arrive at a common::pin.  Every (row, channel) is read and written by
// the same lane only, so reading early changes nothing.
struct EmitState { float px, py, oc, ex, oc0, ex0; };
__device__ __forceinline__ f32x4 fused_prev_feats(const FusedArgs& a, long r, long bq, int o0) {
  const UpdateArgs& u = a.upd;
  const int f = o0 >= 4 ? o0 - 4 : 0;            // feats channels f .. f + 3 (f and the 128 boundary are multiples of 4)
  const float* src = !u.first_of_level ? u.feats + r * kFeatDim + f
                     : (f < kHiresDim ? u.q_hires + bq * kHiresDim + f : u.q_lowres + bq * kLowresDim + (f - kHiresDim));
  return *reinterpret_cast<const f32x4*>(src);
}
__device__ __forceinline__ EmitState fused_prev_state(const FusedArgs& a, long r) {
  const UpdateArgs& u = a.upd;
  EmitState s;
  s.px = u.pos[r * 2 + 0]; s.py = u.pos[r * 2 + 1]; s.oc = u.occ[r]; s.ex = u.expd[r];
  s.oc0 = 0.f; s.ex0 = 0.f;
  if (u.last_of_level) { s.oc0 = u.occ0[r]; s.ex0 = u.expd0[r]; }
  return s;
}
__device__ __forceinline__ void pin(EmitState& s) {
  pin(s.px); pin(s.py); pin(s.oc); pin(s.ex); pin(s.oc0); pin(s.ex0);
}
__device__ __forceinline__ void fused_emit(const FusedArgs& a, long r, int o0, const f32x4& v, const f32x4& prev,
                                           const EmitState& st) {
  if (!a.fuse_update) {
    *reinterpret_cast<f32x4*>(a.res + r * kMixOut + o0) = v;
    return;
  }
  const UpdateArgs& u = a.upd;
  if (o0 == 0) {
    const float px = st.px + v[0] * u.sx;
    const float py = st.py + v[1] * u.sy;
    const float oc = st.oc + v[2];
    const float ex = st.ex + v[3];
    u.pos[r * 2 + 0] = px; u.pos[r * 2 + 1] = py;
    u.out_tracks[r * 2 + 0] = px * u.vx; u.out_tracks[r * 2 + 1] = py * u.vy;
    u.out_occ[r] = oc; u.out_expd[r] = ex;
    u.occ[r] = u.last_of_level ? st.oc0 : oc;
    u.expd[r] = u.last_of_level ? st.ex0 : ex;
  } else {
    *reinterpret_cast<f32x4*>(u.feats + r * kFeatDim + (o0 - 4)) = v + prev;
  }
}

// Number of A fragments in one wave's stream (host packing and kernel must agree).
template <typename TA>
inline long fused_frags_per_wave(int k0_pad, int nblocks) {
  using CF = FusedCfg<TA>;
  const long in = (long)(k0_pad / CF::KS) * 4;
  const long up = (long)(CF::HC / 8 / 16) * (kHidden / CF::KS);   // per chunk
  const long dn = 4L * (CF::HC / CF::KS);
  const long blk = (kHidden4 / CF::HC) * (up + dn);
  const long out = 4L * (kHidden / CF::KS);
  return in + nblocks * blk + out + FM_RING;   // + one ring of padding (prefetched, never used)
}

// ---- time shifts.  Tokens are INTERLEAVED over the token tiles: column c of tile i holds token
// NT * c + i, so the neighbours t-1 / t+1 of a token sit in the same lane (tiles i-1 / i+1) except at
// the first / last tile, where they are tile NT-1 of lane c-1 / tile 0 of lane c+1: ONE DPP row shift
// with zero fill (bound_ctrl), which is also exactly the SAME zero padding at both clip ends.  (With
// contiguous tiles, token = 16 i + c, every shifted value took two DPP moves plus their wait states.)
// lane_up(v): lane c <- lane c-1 (0 into lane 0);  lane_dn(v): lane c <- lane c+1 (0 into lane 15)
__device__ __forceinline__ float lane_up(float v, int lane) {
#ifdef TAPIR_HIPEMU
  const float a = __shfl(v, (lane & 48) | ((lane - 1) & 15));
  return (lane & 15) ? a : 0.f;
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
#endif
}
__device__ __forceinline__ float lane_dn(float v, int lane) {
#ifdef TAPIR_HIPEMU
  const float a = __shfl(v, (lane & 48) | ((lane + 1) & 15));
  return ((lane & 15) != 15) ? a : 0.f;
#else
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
#endif
}

// 16-byte load through an explicit global-address-space pointer: pointers taken from a struct are
// generic to the compiler, and a flat_load counts on vmcnt AND lgkmcnt (it would tie the LDS waits of
// the GEMM loops to the weight stream).
__device__ __forceinline__ f32x4 gload4(const float* p) {
#ifdef TAPIR_HIPEMU
  return *reinterpret_cast<const f32x4*>(p);
#else
  typedef const __attribute__((address_space(1))) f32x4* gptr;
  return *(gptr)(uintptr_t)p;
#endif
}

// 4 consecutive channels of one token -> the [token][channel] LDS image (operand type), 16-byte
// chunks XOR-swizzled by the token's low 4 bits so that the B-fragment ds_read_b128 of a 16-lane
// group (16 tokens, one chunk column) covers all 64 banks.
template <typename TA>
__device__ __forceinline__ void store_act4(char* base, int stride, int row, int ch0, int c,
                                           float v0, float v1, float v2, float v3) {
  constexpr int EPC = 16 / (int)sizeof(TA);
  char* p = base + row * stride + (((ch0 / EPC) ^ c) << 4) + (ch0 % EPC) * (int)sizeof(TA);
  if (sizeof(TA) == 2) {
    uint2 o;
    o.x = pack_bf16x2(v0, v1);
    o.y = pack_bf16x2(v2, v3);
    *reinterpret_cast<uint2*>(p) = o;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(v0, v1, v2, v3);
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// gelu_tanh (common.hpp) on two values at once: the same operations in the same order, the
// polynomial part as packed f32 math (v_pk_mul_f32 / v_pk_fma_f32), the two transcendentals per value
// scalar (there is no packed v_exp / v_rcp).
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
  const float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const float c3 = c1 * 0.044715f;
  const f32x2 z = x * __builtin_elementwise_fma(f32x2{c3, c3}, x * x, f32x2{c1, c1});
  const f32x2 d = f32x2{fast_exp2(z.x), fast_exp2(z.y)} + 1.0f;
  return x * f32x2{fast_rcp(d.x), fast_rcp(d.y)};
}

struct NoEpilogue {
  __device__ __forceinline__ void operator()(int) const {}
};

// One GEMM phase of a wave: acc[r][i] += W_frag(r, k) . act(token tile i, k) over `groups` x G
// k-steps, G = FM_RING / RA.  A fragments come from the wave's register ring (slot order = stream
// order; every consumed slot is refilled with the fragment FM_RING positions further down the
// stream), B fragments from the swizzled LDS image at bbase.  epi(step) runs after the MFMAs of
// every fragment (step = running fragment index of the phase): VALU work of ANOTHER stage slotted
// between the MFMAs (the matrix pipe and the VALU are separate); GROUPS > 0 unrolls the group loop
// so that `step` is a compile-time constant inside epi.
// Schedule (pinned with scheduling fences: left alone, hipcc sinks the FM_RING refill loads of a
// group to the bottom of the loop body, i.e. issues each fragment load right before its use):
//   B fragments of k-step s+1 are read from LDS before the MFMAs of k-step s (one step ahead);
//   each A fragment is refilled right after its last MFMA, so FM_RING - 1 loads stay in flight.
template <typename TA, int RA, int NT, int GROUPS = 0, typename Epi = NoEpilogue, int RING = FM_RING, bool DB = true>
__device__ __forceinline__ void fused_gemm(const uint4*& wp, uint4 (&ring)[RING],
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
        const uint4 fa = ring[kk * RA + r];
#pragma unroll
        for (int i = 0; i < NT; ++i) MfmaStep<TA>::run(fa, cur[i], acc[r][i]);
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
template <typename TA, int NT, bool RAGGED, bool TRACE = false>
__global__ __launch_bounds__(FM_THREADS) void mixer_fused_kernel(FusedArgs a) {
  using CF = FusedCfg<TA>;
  constexpr int EPC = CF::EPC, KS = CF::KS, HC = CF::HC;
  constexpr int ROWS = NT * 16;
  constexpr int RAU = HC / 8 / 16;          // hidden-row tiles of a wave per chunk
  constexpr int NC = kHidden4 / HC;         // hidden chunks per block
  constexpr int XN_STRIDE = kHidden * (int)sizeof(TA);
  constexpr int H_STRIDE = HC * (int)sizeof(TA);
  constexpr int XN_BYTES = ROWS * XN_STRIDE, H_BYTES = ROWS * H_STRIDE;
  constexpr int DN_GROUPS = (HC / KS) / (FM_RING / 4);
  static_assert(NT >= 1 && NT <= CF::MAX_NT, "token tiles");
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
  const uint4* wp = a.stream + ((long)wave * a.frags_per_wave) * 64 + lane;
  uint4 ring[FM_RING];
#pragma unroll
  for (int s = 0; s < FM_RING; ++s) { ring[s] = *wp; wp += 64; }

  // ---- stage the mixer-input rows of this track: [ROWS][ld_in], rows >= T zero
  const int in_stride = a.ld_in * (int)sizeof(TA);
  if (a.fuse_patch) {
    // built here (pips.hpp::patch_row): wave w takes frames w, w + 8, ...; LDS row 16 i + c holds token NT c + i, the
    // 16-byte chunks of a row are XOR-swizzled by the row's low 4 bits (as below)
    char* const img = reinterpret_cast<char*>(s_act);
    for (int tok = wave; tok < ROWS; tok += FM_WAVES) {
      const int row = 16 * (tok % NT) + tok / NT;
      char* const rp = img + row * in_stride;
      auto put = [&](int ccol, float v) {
        Elem<TA>::st(reinterpret_cast<TA*>(rp + (((ccol / EPC) ^ (row & 15)) << 4)) + (ccol % EPC), v);
      };
      if (tok < T) {
        patch_row<TA>(a.patch, (long)n * T + tok, lane, put);
      } else {
        for (int ccol = lane; ccol < a.ld_in; ccol += 64) put(ccol, 0.f);
      }
    }
  } else {
    const int cpr = in_stride >> 4;   // 16-byte chunks per row (a multiple of 16)
    const uint4* src = reinterpret_cast<const uint4*>(
        reinterpret_cast<const char*>(a.mlp_in) + (long)n * T * in_stride);
    // batches of STG chunks per thread, loaded unconditionally from clamped coordinates, then stored (a load under
    // `tok < T` is waited for on the spot: 8 dependent round trips per launch at 48 frames x 1280 bytes)
    constexpr int STG = 8;
    const int total = ROWS * cpr;
    for (int base = tid; base < total; base += STG * FM_THREADS) {
      uint4 v[STG];
      int dst[STG];          // LDS chunk index, -1 past the end of the image
      bool live[STG];        // false: a zero row (token >= T)
#pragma unroll
      for (int k = 0; k < STG; ++k) {
        const int id = base + k * FM_THREADS;
        const int idc = id < total ? id : total - 1;
        const int row = idc / cpr, q = idc - row * cpr;
        const int tok = NT * (row & 15) + (row >> 4);   // LDS row 16 i + c holds token NT c + i
        live[k] = tok < T;
        v[k] = src[(live[k] ? tok : T - 1) * cpr + q];
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

  // ---- residual stream <- input Linear (tapir_model.py:139): x = mlp_in . W0^T + b0
  f32x4 xr[4][NT];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = gload4(a.b0 + ch_lane + 16 * q);
#pragma unroll
    for (int i = 0; i < NT; ++i) xr[q][i] = b;
  }
  fused_gemm<TA, 4, NT>(wp, ring, s_xn, in_stride, a.ld_in / KS / (FM_RING / 4), c, g, xr);
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
    write_xn(bp.ln2, mean, rstd);
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
    f32x4 ua[RAU][NT];
    auto up = [&](int hc) {
#pragma unroll
      for (int r = 0; r < RAU; ++r) {
        const f32x4 bu = *reinterpret_cast<const f32x4*>(&s_bup[hc * HC + hid_lane + 16 * r]);
#pragma unroll
        for (int i = 0; i < NT; ++i) ua[r][i] = bu;
      }
      fused_gemm<TA, RAU, NT>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (FM_RING / RAU), c, g, ua);
    };
    // GELU of value v (0 .. RAU*NT*4-1) of the up accumulators; every fourth one stores its fragment
    auto gelu_step = [&](int v, char* hbuf) {
      if (v < RAU * NT * 4) {
        const int item = v >> 2, r = item / NT, i = item % NT, k = v & 3;
        ua[r][i][k] = gelu_tanh(ua[r][i][k]);
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
      fused_gemm<TA, 4, NT, DN_GROUPS>(wp, ring, hprev, H_STRIDE, DN_GROUPS, c, g, xr,
                                       [&](int step) { gelu_step(step, hcur); });
      tick(6);
      lds_barrier();
      tick(7);
    }
    fused_gemm<TA, 4, NT>(wp, ring, s_h0 + ((NC - 1) & 1) * H_BYTES, H_STRIDE, DN_GROUPS, c, g, xr);
    tick(6);
    lds_barrier();   // every wave is done with the activation images before the next block reuses them
    tick(7);
  }

  // ---- final LayerNorm + output Linear (tapir_model.py:154-155): 388 outputs, rows padded to 512
  {
    // what the state update will add to: in flight under the LayerNorm (see fused_emit)
    f32x4 prev[4][NT];
    EmitState st[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      st[i] = EmitState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) prev[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.fuse_update) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int t = NT * c + i;
        const long r = (long)n * T + (t < T ? t : T - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o0 = ch_lane + 16 * q;
          prev[q][i] = fused_prev_feats(a, r, n, o0 < kMixOut ? o0 : kMixOut - 4);
        }
        if (wave == 0) st[i] = fused_prev_state(a, r);   // channels 0..3 sit in wave 0 (q = 0, g = 0)
      }
    }
    float mean[NT], rstd[NT];
    ln_stats(mean, rstd);
    write_xn(a.lnF, mean, rstd);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      pin(st[i]);
#pragma unroll
      for (int q = 0; q < 4; ++q) pin(prev[q][i]);
    }
    f32x4 oa[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o0 = ch_lane + 16 * q;
      f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
      if (o0 < kMixOut) bo = gload4(a.bout + o0);
#pragma unroll
      for (int i = 0; i < NT; ++i) oa[q][i] = bo;
    }
    fused_gemm<TA, 4, NT>(wp, ring, s_xn, XN_STRIDE, (kHidden / KS) / (FM_RING / 4), c, g, oa);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o0 = ch_lane + 16 * q;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int t = NT * c + i;
        if (o0 < kMixOut && t < T) fused_emit(a, (long)n * T + t, o0, oa[q][i], prev[q][i], st[i]);
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

// true when the fused kernel covers this shape (non-causal whole clips of up to MAX_NT x 16 = 48 frames;
// longer clips, the online model and small track counts run the separate launches of mixer.hpp / gemm.hpp)
template <typename TA>
inline bool fused_mixer_supported(int T, int k0_pad, bool causal, bool has_ctx) {
  using CF = FusedCfg<TA>;
  if (causal || has_ctx || T < 1 || T > 16 * CF::MAX_NT) return false;
  if ((k0_pad * (int)sizeof(TA)) % 256 != 0) return false;
  if ((k0_pad / CF::KS) % (FM_RING / 4) != 0) return false;
  // the staged input image has to fit the activation region of the kernel instance (ACT_BYTES there)
  const long rows = 16L * ((T + 15) / 16), es = (long)sizeof(TA);
  const long images = rows * kHidden * es + 2 * rows * CF::HC * es, par = (long)kHidden * FM_MIXW * 4;
  if (rows * k0_pad * es > (images > par ? images : par)) return false;
  return true;
}

template <typename TA>
inline void launch_mixer_fused(const FusedArgs& a, hipStream_t s) {
  const int nt = (a.T + 15) / 16;
  const bool ragged = a.T % 16 != 0;
  const dim3 grid((unsigned)a.N), block(FM_THREADS);
#ifdef TAPIR_EXPERIMENTS
  if (a.dbg_times != nullptr && nt == 3 && !ragged) {   // phase trace (tools/kbench.py --what fusedtrace)
    hipLaunchKernelGGL((mixer_fused_kernel<TA, 3, false, true>), grid, block, 0, s, a);
    return;
  }
#endif
  if (nt == 1) {
    if (ragged) TAPIR_LAUNCH((mixer_fused_kernel<TA, 1, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_kernel<TA, 1, false>), grid, block, s, a);
  } else if (nt == 2) {
    if (ragged) TAPIR_LAUNCH((mixer_fused_kernel<TA, 2, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_kernel<TA, 2, false>), grid, block, s, a);
  } else {
    if (ragged) TAPIR_LAUNCH((mixer_fused_kernel<TA, 3, true>), grid, block, s, a);
    else TAPIR_LAUNCH((mixer_fused_kernel<TA, 3, false>), grid, block, s, a);
  }
}

}  // namespace tapir
