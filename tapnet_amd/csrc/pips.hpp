// Front half of TAPIR.refine_pips (tapnet/models/tapir_model.py:496-594):
// 7x7 bilinear patch correlation around the current position estimate on each
// pyramid level, and assembly of the mixer input row
//   [0, 0, occ, expd, feats(384), corr_hires(49), corr_lowres(49), (corr_pooled(49))]
//
// The 49 sample positions are the estimate plus INTEGER offsets (:509-515), so
// they share one pair of fractional weights: the kernel computes the raw dot
// products of the query vector with the 8x8 integer window of grid cells and
// blends neighbouring products (the reference's own TPU path :542-562 is the
// same linear re-association).  Cells outside the grid contribute zero
// (interp(mode='constant') :524).
//
// One workgroup per token (b, q, t); wave L handles pyramid level L (see level_corr).
#pragma once
#include "backbone.hpp"   // Vec16
#include "common.hpp"

namespace tapir {

struct PyrLevel {
  const void* grid;     // [B, T, h, w, C] operand type
  const float* query;   // [B*Q, C] f32 (used when feats == null)
  int h, w, C;
  int feat_off;         // offset of this level's slice in feats[384]: 0 (hires) or 128
};

struct PatchArgs {
  PyrLevel lvl[kMaxLevels];
  int n_levels;
  const float* pos;     // [R, 2] (x, y) in initial_resolution pixels
  const float* occ;     // [R]
  const float* expd;    // [R]
  const float* feats;   // [R, 384] or null (first iteration of a level: tiled query features)
  void* mlp_in;         // [R, ld] operand type
  int ld;               // row stride (>= 388 + 49*n_levels, zero padded)
  int B, Q, T;
  float orig_h, orig_w; // initial_resolution
};

// 8x8 window of dot products for one level; returns the 7x7 blended value for
// lane (i=lane>>3, j=lane&7) (garbage for i==7 or j==7).
//
// Lane (i = lane >> 3, cs = lane & 7) owns window row i and an eighth of the channels (16-byte
// chunks cs, 8 + cs, ...): for each of the 8 cells of its row it accumulates the partial dot
// product over its channels (the 8 lanes of a row read 128 contiguous bytes of the cell per load
// instruction), then a transposed butterfly over the 3 low lane bits (7 shuffles) leaves the
// full dot product of cell (i, j) in lane i*8 + j.  (The first version gave every lane 4
// channels of all 64 cells: 63 shuffles per level.)
template <int C, typename TG>
__device__ __forceinline__ float level_corr(const PyrLevel& L, const float* __restrict__ qsrc, long frame,
                                            float px, float py, float orig_w, float orig_h,
                                            int lane) {
  constexpr int SL = C / 8;                       // channels per lane
  constexpr int EPL = 16 / (int)sizeof(TG);       // elements per 16-byte load
  // coords = pos * grid_size / orig_size (transforms.py:75-76), then -0.5 (model_utils.py:199)
  const float gx = px * (float)L.w / orig_w - 0.5f;
  const float gy = py * (float)L.h / orig_h - 0.5f;
  const float fx0 = floorf(gx), fy0 = floorf(gy);
  const float fx = gx - fx0, fy = gy - fy0;
  const int x0 = (int)fx0 - 3, y0 = (int)fy0 - 3;   // window origin (offset -3)
  const int i = lane >> 3, cs = lane & 7;
  // channel slice of lane cs, INTERLEAVED in 16-byte chunks: chunk m of the lane = channels
  // [(m*8 + cs) * EPL, +EPL), so that the 8 lanes of a window row read 128 CONTIGUOUS bytes per load
  // instruction (one cache line per row; a contiguous slice per lane would touch 2-4 lines per row
  // and instruction and use a quarter of each)
  float qv[SL];
#pragma unroll
  for (int m = 0; m < SL / EPL; ++m)
#pragma unroll
    for (int u = 0; u < EPL; u += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qsrc + (m * 8 + cs) * EPL + u);
      qv[m * EPL + u] = t.x; qv[m * EPL + u + 1] = t.y; qv[m * EPL + u + 2] = t.z; qv[m * EPL + u + 3] = t.w;
    }
  const int y = y0 + i;
  const bool vy = (y >= 0) && (y < L.h);
  const int yc = min(max(y, 0), L.h - 1);
  const TG* rowp = reinterpret_cast<const TG*>(L.grid) + (frame * L.h + yc) * ((long)L.w * C) + cs * EPL;
  // All loads of a batch of cells are issued back to back, UNCONDITIONALLY, from clamped addresses, and masked on use:
  // under a lane condition hipcc waits for every load on the spot (the 128-channel level ran as sixteen dependent
  // L2 round trips per token: profiles/r04_patch_corr.txt).  <= 16 loads of 16 bytes in flight per lane.
  constexpr int NL = SL / EPL;                      // 16-byte loads per cell and lane
  constexpr int CB = NL <= 2 ? 8 : 16 / NL;         // cells per batch
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) u32x4_t* gptr16;
  float part[8];
#pragma unroll
  for (int j0 = 0; j0 < 8; j0 += CB) {
    uint4 raw[CB][NL];
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const int xc = min(max(x0 + j0 + jj, 0), L.w - 1);
      const TG* p = rowp + (long)xc * C;
#pragma unroll
      for (int m = 0; m < NL; ++m) {
#ifdef TAPIR_HIPEMU
        raw[jj][m] = *reinterpret_cast<const uint4*>(p + m * 8 * EPL);
#else
        raw[jj][m] = __builtin_bit_cast(uint4, *(gptr16)(uintptr_t)(p + m * 8 * EPL));   // chunk m sits (m * 8 + cs) * EPL channels in
#endif
      }
    }
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const int x = x0 + j0 + jj;
      const bool v = vy && (x >= 0) && (x < L.w);
      float d = 0.f;
#pragma unroll
      for (int m = 0; m < NL; ++m) {
        float e[EPL];
        Vec16<TG>::unpack(raw[jj][m], e);
#pragma unroll
        for (int u = 0; u < EPL; ++u) d = fmaf(e[u], qv[m * EPL + u], d);
      }
      part[j0 + jj] = v ? d : 0.f;
    }
  }
  // transposed butterfly over lane bits 2..0: after the step with offset `off`, a lane keeps
  // the half of its values whose cell index j has bit `off` equal to the lane's
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < off; ++k) {
      const float send = up ? part[k] : part[k + off];
      const float keep = up ? part[k + off] : part[k];
      part[k] = keep + __shfl_xor(send, off);
    }
  }
  const float d00 = part[0];              // lane p holds cell p = i*8 + j
  const float d01 = __shfl_down(d00, 1);
  const float d10 = __shfl_down(d00, 8);
  const float d11 = __shfl_down(d00, 9);
  const float wy0 = 1.0f - fy, wx0 = 1.0f - fx;
  return d00 * (wy0 * wx0) + d01 * (wy0 * fx) + d10 * (fy * wx0) + d11 * (fy * fx);
}

// One WAVE per token (b, q, t): it writes the token's mixer-input row (header, features, zero
// tail) and the 7x7 correlations of every pyramid level.  Waves never synchronise, so a 256-thread
// workgroup is just four independent tokens: 32 tokens in flight per CU.  (The first version gave a
// token a whole workgroup with one wave per level and two idle ones: 8 tokens in flight per CU and
// six dispatch rounds of dependent memory round trips -- position -> window addresses -> grid.)
//
// Wave -> token.  Workgroups are dealt to the 8 XCDs round robin, each XCD has its own 4 MiB L2 and
// a frame's two grids are 1.5 MiB (bf16, config 2).  When the frame count divides by 8, XCD x owns
// the frames fr = x (mod 8) and walks them one at a time, all queries of a frame back to back, so
// a frame's grids are fetched into that L2 once; otherwise tokens are taken in memory order.
constexpr int PATCH_TOKENS_PER_WG = 4;
inline unsigned patch_corr_grid(long B, long Q, long T) {
  const long frames = B * T;
  if ((frames & 7) == 0) {
    const long per_xcd = Q * (frames / 8);                                        // tokens of one XCD
    return (unsigned)(8 * ((per_xcd + PATCH_TOKENS_PER_WG - 1) / PATCH_TOKENS_PER_WG));
  }
  return (unsigned)((B * Q * T + PATCH_TOKENS_PER_WG - 1) / PATCH_TOKENS_PER_WG);
}

// The mixer-input row of token r = (b * Q + q) * T + t: header, features, zero tail and the 7x7 correlations of
// every pyramid level, handed element by element to st(c, v) (c = column of the row) by ONE wave.  Used by
// patch_corr_kernel (stores to HBM) and by the track-resident mixer's prologue (stores into its LDS input image:
// mixer_fused.hpp, fuse_patch).
template <typename TG, typename St>
__device__ __forceinline__ void patch_row(const PatchArgs& a, long r, int lane, St&& st) {
  const int t = (int)(r % a.T);
  const long bq = r / a.T;
  const int b = (int)(bq / a.Q);
  // the position first: the window addresses of both levels depend on it, and the stores of the header loop below
  // would otherwise sit in front of this load (the compiler cannot prove that they do not alias it)
  const float px = a.pos[r * 2 + 0], py = a.pos[r * 2 + 1];
  // header + features + zero padding of the K tail: every lane's sources first (one unconditional load each, from a
  // clamped address), then the stores -- element by element the loop was a chain of load / wait / store round trips
  const int ncorr0 = kMixOut;
  constexpr int HDR = 12;                       // ld <= 768 columns
  const float* const f0 = a.feats != nullptr ? a.feats + r * kFeatDim : nullptr;
  float hv[HDR];
#pragma unroll
  for (int k = 0; k < HDR; ++k) {
    const int c = lane + 64 * k;
    const int f = min(max(c - 4, 0), kFeatDim - 1);
    const float* src = f0 != nullptr ? f0 + f
                       : (f < kHiresDim ? a.lvl[0].query + bq * kHiresDim + f : a.lvl[1].query + bq * kLowresDim + (f - kHiresDim));
    if (c == 2) src = a.occ + r;
    if (c == 3) src = a.expd + r;
    hv[k] = *src;
  }
#ifndef TAPIR_HIPEMU
  // (pins the twelve loads HERE, in flight together: left alone, hipcc sinks each of them into the conditional store
  // below and waits for it on the spot)
#pragma unroll
  for (int k = 0; k < HDR; ++k) asm volatile("" : "+v"(hv[k]));
#endif
#pragma unroll
  for (int k = 0; k < HDR; ++k) {
    const int c = lane + 64 * k;
    if (c >= a.ld) break;
    if (c >= ncorr0 && c < ncorr0 + kPatch * a.n_levels) continue;   // correlation slots: written below
    const bool zero = c < 2 || c >= ncorr0;                          // position channels are always zero (:583); K tail
    st(c, zero ? 0.f : hv[k]);
  }
  for (int c = lane + 64 * HDR; c < a.ld; c += 64) st(c, 0.f);      // (rows longer than 768 columns: zero tail)
  const long frame = (long)b * a.T + t;
  const int i = lane >> 3, j = lane & 7;
  for (int l = 0; l < a.n_levels; ++l) {
    const PyrLevel& L = a.lvl[l];
    // the query vector of this level: the refined per-token feature, or the (tiled) query feature
    const float* qsrc = (a.feats != nullptr) ? a.feats + r * kFeatDim + L.feat_off : L.query + bq * L.C;
    float corr;
    if (L.C == 256) corr = level_corr<256, TG>(L, qsrc, frame, px, py, a.orig_w, a.orig_h, lane);
    else corr = level_corr<128, TG>(L, qsrc, frame, px, py, a.orig_w, a.orig_h, lane);
    if (i < 7 && j < 7) st(ncorr0 + kPatch * l + i * 7 + j, corr);
  }
}

template <typename TG, typename TO>
__global__ __launch_bounds__(256) void patch_corr_kernel(PatchArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const long frames = (long)a.B * a.T;
  long r;
  if ((frames & 7) == 0) {
    const long x = blockIdx.x & 7, s = (long)(blockIdx.x >> 3) * PATCH_TOKENS_PER_WG + wave;
    if (s >= (long)a.Q * (frames >> 3)) return;
    const long fr = (s / a.Q) * 8 + x, q = s % a.Q;
    const long fb = fr / a.T, ft = fr - fb * a.T;
    r = (fb * a.Q + q) * a.T + ft;
  } else {
    r = (long)blockIdx.x * PATCH_TOKENS_PER_WG + wave;
    if (r >= frames * a.Q) return;
  }
  TO* out = reinterpret_cast<TO*>(a.mlp_in) + r * a.ld;
  patch_row<TG>(a, r, lane, [&](int c, float v) { Elem<TO>::st(out + c, v); });
}

// ---- query features: trilinear sample with index clamping
// (get_query_features tapir_model.py:781-849; interp(mode='nearest') model_utils.py:177-206)
struct SampleArgs {
  const float* grid;     // [B, T, h, w, C]
  const float* qpts;     // [B, Q, 3] (t, y, x) in video coordinates
  float* out;            // [B, Q, C]
  int B, Q, T, h, w, C;
  float vid_h, vid_w;
};
__global__ __launch_bounds__(128) void query_feature_kernel(SampleArgs a) {
  const long bq = blockIdx.x;
  const int b = (int)(bq / a.Q);
  const float* q = a.qpts + bq * 3;
  // position_in_grid = q * [T,h,w] / [T,H,W]; t unshifted, y/x - 0.5
  const float ct = q[0] * (float)a.T / (float)a.T;
  const float cy = q[1] * (float)a.h / a.vid_h - 0.5f;
  const float cx = q[2] * (float)a.w / a.vid_w - 0.5f;
  const float ft = floorf(ct), fy = floorf(cy), fx = floorf(cx);
  const float wt = ct - ft, wy = cy - fy, wx = cx - fx;
  const int it = (int)ft, iy = (int)fy, ix = (int)fx;
  const float* g = a.grid + (long)b * a.T * a.h * a.w * a.C;
  for (int c = threadIdx.x; c < a.C; c += 128) {
    float acc = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int tt = min(max(it + dt, 0), a.T - 1);
      const float wtt = dt ? wt : 1.0f - wt;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yy = min(max(iy + dy, 0), a.h - 1);
        const float wyy = dy ? wy : 1.0f - wy;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int xx = min(max(ix + dx, 0), a.w - 1);
          const float wxx = dx ? wx : 1.0f - wx;
          acc += g[(((long)tt * a.h + yy) * a.w + xx) * a.C + c] * (wtt * wyy * wxx);
        }
      }
    }
    a.out[bq * a.C + c] = acc;
  }
}

// ---- 2x2 average pool over (h, w) of a channels-last grid (tapir_model.py:995-1000),
// optionally converting to the operand type; also the plain cast (pool = 0).
struct PoolArgs {
  const float* in; void* out; long frames; int h, w, C; int pool;
  // optional second copy (plain cast of a 256-channel grid to bf16 only) in the cost-volume kernel's A-operand order:
  // [frame][tile of 16 cells][32 chunks of 8 channels][16 cells][8 bf16] -- the 16 lanes of an MFMA row group read 256
  // contiguous bytes (costvol_rows.hpp: row-major, the same 16 lanes touch 16 different 512-byte rows and the vector
  // L1 serves one line per cycle).  Cells past the end of the last tile are never written (the buffer is zeroed once).
  void* tiled;
};
template <typename TO>
__global__ __launch_bounds__(256) void pool_cast_kernel(PoolArgs a) {
  const int oh = a.pool ? a.h / 2 : a.h, ow = a.pool ? a.w / 2 : a.w;
  const long total = a.frames * oh * ow * (a.C / 4);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c4 = (int)(idx % (a.C / 4));
    long p = idx / (a.C / 4);
    const int x = (int)(p % ow); p /= ow;
    const int y = (int)(p % oh);
    const long f = p / oh;
    float4 v;
    if (a.pool) {
      const float* s = a.in + ((f * a.h + 2 * y) * a.w + 2 * x) * a.C + c4 * 4;
      const float4 v00 = *reinterpret_cast<const float4*>(s);
      const float4 v01 = *reinterpret_cast<const float4*>(s + a.C);
      const float4 v10 = *reinterpret_cast<const float4*>(s + (long)a.w * a.C);
      const float4 v11 = *reinterpret_cast<const float4*>(s + (long)a.w * a.C + a.C);
      v.x = (v00.x + v01.x + v10.x + v11.x) * 0.25f;
      v.y = (v00.y + v01.y + v10.y + v11.y) * 0.25f;
      v.z = (v00.z + v01.z + v10.z + v11.z) * 0.25f;
      v.w = (v00.w + v01.w + v10.w + v11.w) * 0.25f;
    } else {
      v = *reinterpret_cast<const float4*>(a.in + ((f * a.h + y) * a.w + x) * a.C + c4 * 4);
    }
    TO* o = reinterpret_cast<TO*>(a.out) + ((f * oh + y) * ow + x) * a.C + c4 * 4;
    Elem<TO>::st(o, v.x); Elem<TO>::st(o + 1, v.y); Elem<TO>::st(o + 2, v.z); Elem<TO>::st(o + 3, v.w);
    if (sizeof(TO) == 2 && a.tiled != nullptr) {
      const int cell = y * ow + x, ntile = (oh * ow + 15) >> 4;
      TO* q = reinterpret_cast<TO*>(a.tiled) +
              ((((f * ntile + (cell >> 4)) * 32 + (c4 >> 1)) * 16 + (cell & 15)) * 8 + (c4 & 1) * 4);
      Elem<TO>::st(q, v.x); Elem<TO>::st(q + 1, v.y); Elem<TO>::st(q + 2, v.z); Elem<TO>::st(q + 3, v.w);
    }
  }
}

}  // namespace tapir
