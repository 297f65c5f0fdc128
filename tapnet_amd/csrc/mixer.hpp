// Token-mixing half of a PIPsConvBlock and the small row-wise kernels of the
// PIPs MLP-mixer (tapnet/models/tapir_model.py:33-156).
//
// mix_kernel fuses, per track and time chunk:
//   LN1 (scale only, eps 1e-5)  :111      -> depthwise Conv1D k=3 x4 channels :59-66
//   -> GELU(tanh) :67 -> depthwise Conv1D k=3 :75-82 -> sum of each group of 4 :87
//   -> + skip :119 -> LN2 :121
// and writes the new residual stream (f32) plus LN2(x) in the GEMM operand
// type.  The 2048-wide intermediate never leaves registers.  Temporal padding
// is SAME (zeros at both clip ends) or causal (two frames on the left taken
// from the causal context, zeros if there is none) :64,80,49-53,69-73.
#pragma once
#include "common.hpp"
#include "gemm.hpp"   // Store4

namespace tapir {

constexpr int MIX_THREADS = 256;     // 2 channels per thread x 256 = 512
constexpr int MIX_MAX_TC = 16;       // max frames per time chunk
constexpr int MIX_MAX_ROWS = MIX_MAX_TC + 4;   // + halo rows (2 each side / 4 on the left if causal)
constexpr float kLnEps = 1e-5f;      // hk.LayerNorm default

struct MixArgs {
  const float* x_in;      // [N, T, 512]
  float* x_out;           // [N, T, 512]   (must not alias x_in: halo rows are re-read)
  void* xn2;              // [N*T, 512] LN2(x_out) * scale, operand type
  const float* ln1;       // [512]
  const float* w1;        // [2048, 3]  out channel 4c+m <- in channel c  (mlp1_up)
  const float* b1;        // [2048]
  const float* w2;        // [2048, 3]  (mlp1_up_1)
  const float* b2;        // [2048]
  const float* ln2;       // [512]
  const float* ctx1_in;   // [N, 2, 512]  or null   (block_i_causal_1)
  const float* ctx2_in;   // [N, 2, 2048] or null   (block_i_causal_2)
  float* ctx1_out;        // or null
  float* ctx2_out;        // or null
  int T;                  // frames per track
  int TC;                 // frames per workgroup (<= MIX_MAX_TC)
  int causal;             // use_causal_conv
  long long* dbg_times;   // null, or [units][6] wall-clock stamps (tools/kbench.py --mix-trace)
  // x_in given as the pieces the previous block's one-launch channel MLP left (gemm.hpp mlp_small_kernel): parts != null:
  // x_in[r] = ((((p_0 + p_1) + ...) + p_{nparts-1}) + bias) + resid[r], added in that fixed order while the rows are staged
  const float* parts;     // [nparts][N * T, 512] f32
  const float* pbias;     // [512] the down-projection's bias
  const float* presid;    // [N * T, 512] the residual stream the MLP read (the previous mix_kernel's x_out)
  int nparts;
};

// one row-major value pair (columns 2 j, 2 j + 1 of row r) of the sum above; nparts == MLP_PARTS (gemm.hpp: 2048 / MLP_HS), a
// compile-time count so that all of a thread's loads are in flight at once
__device__ __forceinline__ float2 parts_sum2(const float* parts, int /*nparts*/, long rows, const float* bias, const float* resid,
                                             long r, int col) {
  float2 w[MLP_PARTS];
#pragma unroll
  for (int p = 0; p < MLP_PARTS; ++p) w[p] = *reinterpret_cast<const float2*>(parts + ((long)p * rows + r) * kHidden + col);
  const float2 b = *reinterpret_cast<const float2*>(bias + col);
  const float2 x = *reinterpret_cast<const float2*>(resid + r * kHidden + col);
  float2 v = w[0];
#pragma unroll
  for (int p = 1; p < MLP_PARTS; ++p) { v.x += w[p].x; v.y += w[p].y; }
  return make_float2((v.x + b.x) + x.x, (v.y + b.y) + x.y);
}

// the same with the residual's two values in registers (mixer_online.hpp)
__device__ __forceinline__ float2 parts_sum2v(const float* parts, long rows, const float* bias, float2 x, long r, int col) {
  float2 w[MLP_PARTS];
#pragma unroll
  for (int p = 0; p < MLP_PARTS; ++p) w[p] = *reinterpret_cast<const float2*>(parts + ((long)p * rows + r) * kHidden + col);
  const float2 b = *reinterpret_cast<const float2*>(bias + col);
  float2 v = w[0];
#pragma unroll
  for (int p = 1; p < MLP_PARTS; ++p) { v.x += w[p].x; v.y += w[p].y; }
  return make_float2((v.x + b.x) + x.x, (v.y + b.y) + x.y);
}

// Row statistics by ONE wave: the lane holds channels [4*lane, 4*lane+4) and
// [256+4*lane, 256+4*lane+4) of the row.
__device__ __forceinline__ void wave_row_stats(const float4& u, const float4& v, float& mean,
                                               float& rstd) {
  mean = wave_sum((u.x + u.y) + (u.z + u.w) + (v.x + v.y) + (v.z + v.w)) * (1.0f / kHidden);
  const float d0 = u.x - mean, d1 = u.y - mean, d2 = u.z - mean, d3 = u.w - mean;
  const float d4 = v.x - mean, d5 = v.y - mean, d6 = v.z - mean, d7 = v.w - mean;
  const float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
  rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / kHidden) + kLnEps);
}

// The same for K rows at once (independent reductions interleaved).
template <int K>
__device__ __forceinline__ void wave_row_stats_n(const float4 (&u)[K], const float4 (&v)[K],
                                                 float (&mean)[K], float (&rstd)[K]) {
  float s[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
    s[k] = (u[k].x + u[k].y) + (u[k].z + u[k].w) + (v[k].x + v[k].y) + (v[k].z + v[k].w);
  wave_sum_n<K>(s);
  float q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    mean[k] = s[k] * (1.0f / kHidden);
    const float d0 = u[k].x - mean[k], d1 = u[k].y - mean[k], d2 = u[k].z - mean[k], d3 = u[k].w - mean[k];
    const float d4 = v[k].x - mean[k], d5 = v[k].y - mean[k], d6 = v[k].z - mean[k], d7 = v[k].w - mean[k];
    q[k] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
  }
  wave_sum_n<K>(q);
#pragma unroll
  for (int k = 0; k < K; ++k) rstd[k] = 1.0f / sqrtf(q[k] * (1.0f / kHidden) + kLnEps);
}

// One workgroup = one track x one time chunk.  The chunk's input rows (with
// halo) are staged in LDS by one fully parallel, coalesced pass, so that no
// step of the temporal recurrence waits on global memory:
//   0. rows [xlo, xhi] -> LDS                      (all loads in flight at once)
//   1. LayerNorm-1 statistics, one wave per row    (shuffles only)
//   2. temporal stream, 2 channels per thread      (registers; x_new written back into LDS)
//   3. LayerNorm-2 + stores, one wave per row      (16-byte coalesced stores)
template <typename TO>
__global__ __launch_bounds__(MIX_THREADS) void mix_kernel(MixArgs a) {
  __shared__ float s_x[MIX_MAX_ROWS][kHidden];   // 40 KiB
  __shared__ float s_mean[MIX_MAX_ROWS];
  __shared__ float s_rstd[MIX_MAX_ROWS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const int T = a.T;
  const int t0 = blockIdx.x * a.TC;
  const int t1 = min(T, t0 + a.TC);
  const int off0 = a.causal ? -2 : -1;        // tap k reads frame t + off0 + k
  const int c0 = tid * 2;                      // this thread's two channels in phase 2
  const float* __restrict__ xin = a.x_in + (long)n * T * kHidden;

  // ---- phase 0: stage every input row this chunk touches: the rows are contiguous in global
  // memory (2 KiB each), so they go to LDS by DMA, 1 KiB (half a row) per wave instruction
  const int xlo = max(0, t0 + 2 * off0);
  const int xhi = min(T - 1, t1 - 1 + 2 * off0 + 4);
  const int nrows = xhi - xlo + 1;
  if (a.parts != nullptr) {   // the rows are sums of the previous block's MLP pieces (few rows: the online model)
    const long rows_all = (long)gridDim.y * T, rbase = (long)n * T + xlo;
    for (int r = 0; r < nrows; ++r) {
      const float2 v = parts_sum2(a.parts, a.nparts, rows_all, a.pbias, a.presid, rbase + r, 2 * tid);
      *reinterpret_cast<float2*>(&s_x[r][2 * tid]) = v;
    }
  } else {
    const float* src = xin + (long)xlo * kHidden;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int k = wave_u; k < nrows * 2; k += MIX_THREADS / 64)
      glds16(src + k * 256 + lane * 4, &s_x[0][0] + k * 256);
  }
  __syncthreads();   // drains the DMA (vmcnt) and makes the rows visible

  // ---- phase 1: LayerNorm-1 statistics, wave w takes rows w, w+4, ...
  for (int r = wave; r < nrows; r += 4) {
    const float4 u = *reinterpret_cast<const float4*>(&s_x[r][lane * 4]);
    const float4 v = *reinterpret_cast<const float4*>(&s_x[r][256 + lane * 4]);
    float mean, rstd;
    wave_row_stats(u, v, mean, rstd);
    if (lane == 0) { s_mean[r] = mean; s_rstd[r] = rstd; }
  }
  __syncthreads();

  // ---- per-thread weights (2 channels x 4 multipliers x 3 taps, twice)
  float w1[2][4][3], b1[2][4], w2[2][4][3], b2[2][4], sc1[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    sc1[ch] = a.ln1[c0 + ch];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int o = 4 * (c0 + ch) + m;
      b1[ch][m] = a.b1[o];
      b2[ch][m] = a.b2[o];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        w1[ch][m][k] = a.w1[o * 3 + k];
        w2[ch][m][k] = a.w2[o * 3 + k];
      }
    }
  }

  // LN1(x)[tau, c0..c0+1]; outside the clip: causal context or zeros
  auto load_xn = [&](int tau, float& o0, float& o1) {
    if (tau >= 0 && tau < T) {
      const float2 v = *reinterpret_cast<const float2*>(&s_x[tau - xlo][c0]);
      const float m = s_mean[tau - xlo], rs = s_rstd[tau - xlo];
      o0 = (v.x - m) * rs * sc1[0];
      o1 = (v.y - m) * rs * sc1[1];
    } else if (tau < 0 && tau >= -2 && a.ctx1_in != nullptr) {
      const float2 v = *reinterpret_cast<const float2*>(a.ctx1_in + ((long)n * 2 + (tau + 2)) * kHidden + c0);
      o0 = v.x; o1 = v.y;
    } else {
      o0 = 0.f; o1 = 0.f;
    }
  };

  float xw[2][3];          // LN1(x) window: frames tau+off0 .. tau+off0+2
  float gw[2][4][3];       // GELU window:   frames tau-2 .. tau
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int m = 0; m < 4; ++m) gw[ch][m][0] = gw[ch][m][1] = gw[ch][m][2] = 0.f;

  const int g_lo = t0 + off0;                 // first / last frame of g this chunk needs
  const int g_hi = t1 - 1 + off0 + 2;
  load_xn(g_lo + off0, xw[0][0], xw[1][0]);
  load_xn(g_lo + off0 + 1, xw[0][1], xw[1][1]);

  // ---- phase 2: temporal stream
  for (int tau = g_lo; tau <= g_hi; ++tau) {
    load_xn(tau + off0 + 2, xw[0][2], xw[1][2]);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        gw[ch][m][0] = gw[ch][m][1];
        gw[ch][m][1] = gw[ch][m][2];
      }
    // g(tau): GELU(conv1(LN1 x)) inside the clip, causal context / zeros outside
    if (tau >= 0 && tau < T) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float u = b1[ch][m];
          u = fmaf(w1[ch][m][0], xw[ch][0], u);
          u = fmaf(w1[ch][m][1], xw[ch][1], u);
          u = fmaf(w1[ch][m][2], xw[ch][2], u);
          gw[ch][m][2] = gelu_tanh(u);
        }
    } else if (tau < 0 && tau >= -2 && a.ctx2_in != nullptr) {
      const float* p = a.ctx2_in + ((long)n * 2 + (tau + 2)) * kHidden4 + 4 * c0;
      const float4 v0 = *reinterpret_cast<const float4*>(p);
      const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
      gw[0][0][2] = v0.x; gw[0][1][2] = v0.y; gw[0][2][2] = v0.z; gw[0][3][2] = v0.w;
      gw[1][0][2] = v1.x; gw[1][1][2] = v1.y; gw[1][2][2] = v1.z; gw[1][3][2] = v1.w;
    } else {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int m = 0; m < 4; ++m) gw[ch][m][2] = 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) { xw[ch][0] = xw[ch][1]; xw[ch][1] = xw[ch][2]; }

    // output frame whose last tap is g(tau): x_new = x + sum_m conv2(g)  (in place in LDS;
    // this thread is the only reader/writer of its two channels during the stream)
    const int t = tau - off0 - 2;
    if (t >= t0 && t < t1) {
      float y[2];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float v = b2[ch][m];
          v = fmaf(w2[ch][m][0], gw[ch][m][0], v);
          v = fmaf(w2[ch][m][1], gw[ch][m][1], v);
          v = fmaf(w2[ch][m][2], gw[ch][m][2], v);
          acc += v;
        }
        y[ch] = acc;
      }
      float2* px = reinterpret_cast<float2*>(&s_x[t - xlo][c0]);
      const float2 xv = *px;
      *px = make_float2(xv.x + y[0], xv.y + y[1]);
    }
  }

  // ---- new causal context: last two frames of [ctx ; new] (tapir_model.py:58,73)
  if (a.ctx1_out != nullptr && t1 == T) {
    // after the loop the windows hold frames T-3..T-1 in slots 0..2 of gw and
    // frames T-2, T-1 of LN1(x) in slots 0, 1 of xw
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<float2*>(a.ctx1_out + ((long)n * 2 + j) * kHidden + c0) =
          make_float2(xw[0][j], xw[1][j]);
      float* p = a.ctx2_out + ((long)n * 2 + j) * kHidden4 + 4 * c0;
      *reinterpret_cast<float4*>(p) = make_float4(gw[0][0][1 + j], gw[0][1][1 + j], gw[0][2][1 + j], gw[0][3][1 + j]);
      *reinterpret_cast<float4*>(p + 4) = make_float4(gw[1][0][1 + j], gw[1][1][1 + j], gw[1][2][1 + j], gw[1][3][1 + j]);
    }
  }
  __syncthreads();

  // ---- phase 3: LayerNorm-2 and stores, wave w takes output rows t0+w, t0+w+4, ...
  const float4 s2a = *reinterpret_cast<const float4*>(a.ln2 + lane * 4);
  const float4 s2b = *reinterpret_cast<const float4*>(a.ln2 + 256 + lane * 4);
  for (int t = t0 + wave; t < t1; t += 4) {
    const float4 u = *reinterpret_cast<const float4*>(&s_x[t - xlo][lane * 4]);
    const float4 v = *reinterpret_cast<const float4*>(&s_x[t - xlo][256 + lane * 4]);
    float mean, rs;
    wave_row_stats(u, v, mean, rs);
    const long row = (long)n * T + t;
    float* xo = a.x_out + row * kHidden;
    *reinterpret_cast<float4*>(xo + lane * 4) = u;
    *reinterpret_cast<float4*>(xo + 256 + lane * 4) = v;
    TO* o = reinterpret_cast<TO*>(a.xn2) + row * kHidden;
    Store4<TO>::run(o + lane * 4, (u.x - mean) * rs * s2a.x, (u.y - mean) * rs * s2a.y,
                    (u.z - mean) * rs * s2a.z, (u.w - mean) * rs * s2a.w);
    Store4<TO>::run(o + 256 + lane * 4, (v.x - mean) * rs * s2b.x, (v.y - mean) * rs * s2b.y,
                    (v.z - mean) * rs * s2b.z, (v.w - mean) * rs * s2b.w);
  }
}

// ---------------------------------------------------------------------------------------------
// mix_stream_kernel: the non-causal token-mixing kernel for whole clips (SAME padding).
//
// Same data flow as mix_kernel, restructured around what PMC counters showed it to be: VALU-bound
// (2200 VALU instructions per wave at 4 cycles each, a third of them register-window moves,
// branches and scalar selects).  Here the temporal stream is fully unrolled over a compile-time
// chunk length TC, so the 3-frame windows are static registers (no moves); the LayerNorm-1 scale
// is folded into the first convolution's weights and the second convolution's biases into one
// constant; and every operation acts on the thread's TWO ADJACENT channels, so the compiler emits
// packed f32 math (v_pk_fma_f32 / v_pk_mul_f32: 2 lanes-worth per issue on the SIMD-16 VALU).
// (Writing the stream with explicit 2-wide vector types packs the GELU's polynomial too -- 1941 ->
// 1781 VALU instructions -- but needs 134 VGPRs; capped at 128 for four waves per SIMD it spills and
// measured 29.3 us against 27.9, tools/ab_mix.sh.)
//   rows  r = 0 .. TC+3  <->  frames tau = t0 - 2 + r;  outputs o = t0 .. t0 + TC - 1
//   xn_r            = LN1(x)[tau]            (0 outside the clip: SAME padding of conv 1)
//   g_{r-1}[m]      = gelu(b1[m] + sum_k w1'[m][k] xn_{r-2+k})     (0 outside the clip)
//   x'[o = tau - 2] = x[o] + B2 + sum_m sum_k w2[m][k] g_{o-1+k}[m]
// One (track, time chunk) unit of mix_stream_kernel, split into "stage the rows" and "process",
// so that a persistent workgroup can copy the rows of its NEXT unit while it computes the current
// one (two row buffers) and let the stores of a unit drain under the next unit's arithmetic: with
// one unit per workgroup every workgroup of the chip is in the same phase at the same time
// (all loading, then all computing, then all storing) and HBM idles two thirds of the time.
template <int TC>
struct MixUnit {
  static constexpr int ROWS = TC + 4;
  int n, t0, rlo, rhi;
  __device__ __forceinline__ void set(int u, int nch, int T) {
    n = u / nch;
    t0 = (u - n * nch) * TC;
    const int lo = max(0, t0 - 2), hi = min(T - 1, t0 + TC + 1);   // staged frames
    rlo = lo - (t0 - 2); rhi = hi - (t0 - 2);                      // LDS row = tau - (t0 - 2)
  }
};

// DB = true: persistent workgroups with two row buffers (64 KiB: two workgroups = 8 waves per CU);
// DB = false: one unit per workgroup, one row buffer (32 KiB: four workgroups = 16 waves per CU,
// the loads of one workgroup overlap the arithmetic of the others).
template <typename TO, int TC, bool DB = true>
__global__ __launch_bounds__(MIX_THREADS) void mix_stream_kernel(MixArgs a, int units, int nch) {
  constexpr int ROWS = TC + 4;
  // two separate objects: reads of one buffer must not wait for the DMA into the other
  __shared__ float s_xa[ROWS][kHidden];
  __shared__ float s_xb[DB ? ROWS : 1][DB ? kHidden : 2];
  __shared__ float2 s_stata[ROWS];   // (mean, rstd) of LayerNorm-1
  __shared__ float2 s_statb[DB ? ROWS : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int T = a.T;
  const int c0 = tid * 2;
  int u = blockIdx.x;
  if (u >= units) return;

  // rows tau = t0-2 .. t0+TC+1 that exist, by DMA (1 KiB = half a row per wave op); rows outside
  // the clip are zero-filled with zero statistics, so that the stream needs no branches
  auto stage = [&](const MixUnit<TC>& q, float (*sx)[kHidden], float2* sstat) {
    const float* src = a.x_in + ((long)q.n * T + (q.t0 - 2 + q.rlo)) * kHidden;
    float* dst = &sx[q.rlo][0];
    for (int k = wave_u; k < (q.rhi - q.rlo + 1) * 2; k += MIX_THREADS / 64)
      glds16(src + k * 256 + lane * 4, dst + k * 256);
    for (int r = 0; r < ROWS; ++r)
      if (r < q.rlo || r > q.rhi) *reinterpret_cast<float2*>(&sx[r][c0]) = make_float2(0.f, 0.f);
    if (tid < ROWS && (tid < q.rlo || tid > q.rhi)) sstat[tid] = make_float2(0.f, 0.f);
  };

  MixUnit<TC> cur, nxt;
  cur.set(u, nch, T);
  stage(cur, s_xa, s_stata);

  // per-thread weights (once per workgroup): 2 channels x 4 multipliers x 3 taps, twice
  float2 w1[4][3], b1[4], w2[4][3];
  float2 B2 = make_float2(0.f, 0.f);
  {
    const float2 sc = *reinterpret_cast<const float2*>(a.ln1 + c0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int oa = 4 * c0 + m, ob = 4 * (c0 + 1) + m;
      b1[m] = make_float2(a.b1[oa], a.b1[ob]);
      B2.x += a.b2[oa]; B2.y += a.b2[ob];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        w1[m][k] = make_float2(a.w1[oa * 3 + k] * sc.x, a.w1[ob * 3 + k] * sc.y);
        w2[m][k] = make_float2(a.w2[oa * 3 + k], a.w2[ob * 3 + k]);
      }
    }
  }
  const float4 s2a = *reinterpret_cast<const float4*>(a.ln2 + lane * 4);
  const float4 s2b = *reinterpret_cast<const float4*>(a.ln2 + 256 + lane * 4);

  auto stamp = [&](int unit, int k) {
#ifndef TAPIR_NO_STAMPS
    if (a.dbg_times != nullptr && tid == 0) a.dbg_times[(long)unit * 6 + k] = wall_clock64();
#endif
  };
  stamp(u, 0);
  auto process = [&](const MixUnit<TC>& q, float (*s_x)[kHidden], float2* s_stat, int unit) {
    const int t0 = q.t0;
    stamp(unit, 2);
    // ---- phase 1: LayerNorm-1 statistics, wave w takes rows w, w+4, ... -- all of them at once
    // (a row past the end repeats the last one and is not written)
    {
      constexpr int K1 = (ROWS + 3) / 4;
      float4 u4[K1], v4[K1];
      float mean[K1], rstd[K1];
#pragma unroll
      for (int k = 0; k < K1; ++k) {
        const int r = min(q.rlo + wave + 4 * k, q.rhi);
        u4[k] = *reinterpret_cast<const float4*>(&s_x[r][lane * 4]);
        v4[k] = *reinterpret_cast<const float4*>(&s_x[r][256 + lane * 4]);
      }
      wave_row_stats_n<K1>(u4, v4, mean, rstd);
#pragma unroll
      for (int k = 0; k < K1; ++k) {
        const int r = q.rlo + wave + 4 * k;
        if (lane == 0 && r <= q.rhi) s_stat[r] = make_float2(mean[k], rstd[k]);
      }
    }
    lds_barrier();
    stamp(unit, 3);

    // ---- phase 2: the unrolled temporal stream (straight-line code)
    float2 xn[3];          // LN1(x) of rows r-2, r-1, r
    float2 g[3][4];        // GELU outputs of rows r-3, r-2, r-1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      xn[k] = make_float2(0.f, 0.f);
#pragma unroll
      for (int m = 0; m < 4; ++m) g[k][m] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int tau = t0 - 2 + r;
      {
        const float2 v = *reinterpret_cast<const float2*>(&s_x[r][c0]);
        const float2 st = s_stat[r];
        xn[r % 3] = make_float2((v.x - st.x) * st.y, (v.y - st.x) * st.y);
      }
      if (r >= 2) {
        // g of row r-1 (frame tau-1) from xn rows r-2, r-1, r; zero outside the clip
        const float gm = (tau - 1 >= 0 && tau - 1 < T) ? 1.0f : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float2 uu = b1[m];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float2 xv = xn[(r - 2 + k) % 3];
            uu.x = fmaf(w1[m][k].x, xv.x, uu.x);
            uu.y = fmaf(w1[m][k].y, xv.y, uu.y);
          }
          g[(r - 1) % 3][m] = make_float2(gelu_tanh(uu.x) * gm, gelu_tanh(uu.y) * gm);
        }
      }
      if (r >= 4) {
        // output frame o = tau - 2 (row r-2) from g rows r-3, r-2, r-1
        float2 y = B2;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float2 gv = g[(r - 3 + k) % 3][m];
            y.x = fmaf(w2[m][k].x, gv.x, y.x);
            y.y = fmaf(w2[m][k].y, gv.y, y.y);
          }
        float2* px = reinterpret_cast<float2*>(&s_x[r - 2][c0]);
        const float2 xv = *px;
        *px = make_float2(xv.x + y.x, xv.y + y.y);
      }
    }
    lds_barrier();
    stamp(unit, 4);

    // ---- phase 3: LayerNorm-2 and stores, wave w takes output frames t0+w, t0+w+4, ... at once
    const int t1 = min(T, t0 + TC);
    {
      constexpr int K3 = (TC + 3) / 4;
      float4 u4[K3], v4[K3];
      float mean[K3], rs[K3];
#pragma unroll
      for (int k = 0; k < K3; ++k) {
        const int t = min(t0 + wave + 4 * k, t1 - 1);
        u4[k] = *reinterpret_cast<const float4*>(&s_x[t - t0 + 2][lane * 4]);
        v4[k] = *reinterpret_cast<const float4*>(&s_x[t - t0 + 2][256 + lane * 4]);
      }
      wave_row_stats_n<K3>(u4, v4, mean, rs);
#pragma unroll
      for (int k = 0; k < K3; ++k) {
        const int t = t0 + wave + 4 * k;
        if (t < t1) {
          const long row = (long)q.n * T + t;
          float* xo = a.x_out + row * kHidden;
          *reinterpret_cast<float4*>(xo + lane * 4) = u4[k];
          *reinterpret_cast<float4*>(xo + 256 + lane * 4) = v4[k];
          TO* o = reinterpret_cast<TO*>(a.xn2) + row * kHidden;
          Store4<TO>::run(o + lane * 4, (u4[k].x - mean[k]) * rs[k] * s2a.x, (u4[k].y - mean[k]) * rs[k] * s2a.y,
                          (u4[k].z - mean[k]) * rs[k] * s2a.z, (u4[k].w - mean[k]) * rs[k] * s2a.w);
          Store4<TO>::run(o + 256 + lane * 4, (v4[k].x - mean[k]) * rs[k] * s2b.x, (v4[k].y - mean[k]) * rs[k] * s2b.y,
                          (v4[k].z - mean[k]) * rs[k] * s2b.z, (v4[k].w - mean[k]) * rs[k] * s2b.w);
        }
      }
    }
    stamp(unit, 5);
  };

  if (!DB) {
    dma_lds_wait<0>();
    block_barrier();
    stamp(u, 1);
    process(cur, s_xa, s_stata, u);
    return;
  }
  auto s_xb2 = reinterpret_cast<float (*)[kHidden]>(&s_xb[0][0]);
  for (;;) {
    // the rows of `cur` have landed in buffer a (and the stores of the previous unit are done)
    dma_lds_wait<0>();
    block_barrier();
    stamp(u, 1);
    int un = u + gridDim.x;
    if (un < units) { nxt.set(un, nch, T); stage(nxt, s_xb2, s_statb); stamp(un, 0); }
    process(cur, s_xa, s_stata, u);
    if (un >= units) break;
    u = un; cur = nxt;
    dma_lds_wait<0>();
    block_barrier();
    stamp(u, 1);
    un = u + gridDim.x;
    if (un < units) { nxt.set(un, nch, T); stage(nxt, s_xa, s_stata); stamp(un, 0); }
    process(cur, s_xb2, s_statb, u);
    if (un >= units) break;
    u = un; cur = nxt;
  }
}

// Launches the token-mixing kernel of one block: the streamed kernel for whole clips, the
// general one (time chunks chosen at run time, causal padding and state) otherwise.
template <typename TO>
inline void launch_mix(const MixArgs& m_in, int N, hipStream_t s, int force_tc = 0) {   // force_tc: grid cap (tests)
  MixArgs m = m_in;
  const bool plain = !m.causal && !m.ctx1_in && !m.ctx2_in && !m.ctx1_out && !m.ctx2_out;
  if (plain && m.T >= 12) {
    // 12-frame chunks
    const int nch = (m.T + 11) / 12;
    const int units = N * nch;
    // Default: ONE unit per workgroup with a single row buffer (32 KiB -> four workgroups = 16 waves
    // per CU).  Measured at config 2 (tools/kbench.py --what mix): 28.3 us against 31.6 us for the
    // persistent double-buffered form (force_tc = grid cap, two workgroups = 8 waves per CU): the
    // kernel is VALU-latency bound and the extra resident waves are worth more than the overlap of a
    // workgroup's own loads with its arithmetic; chunk lengths 8 / 12 / 16 measure the same, and so does
    // starting half of the workgroups 1-2 us late (longer delays cost their length).
    if (force_tc <= 0) {
      TAPIR_LAUNCH((mix_stream_kernel<TO, 12, false>), dim3(units), dim3(MIX_THREADS), s, m, units, nch);
      return;
    }
    const int grid = std::min(units, force_tc);
    TAPIR_LAUNCH((mix_stream_kernel<TO, 12>), dim3(grid), dim3(MIX_THREADS), s, m, units, nch);
    return;
  }
  const int nch = (m.T + m.TC - 1) / m.TC;
  TAPIR_LAUNCH((mix_kernel<TO>), dim3(nch, N), dim3(MIX_THREADS), s, m);
}

// Row-wise LayerNorm (scale only) -> operand type; one wave per row of 512.
struct LnArgs {
  const float* x; const float* scale; void* out; long rows;
  const float* parts; const float* pbias; const float* presid; int nparts;   // parts != null: x as MixArgs::parts describes it
};
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  float e[8];
  if (a.parts != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 t = parts_sum2(a.parts, a.nparts, a.rows, a.pbias, a.presid, row, lane * 8 + 2 * k);
      e[2 * k] = t.x; e[2 * k + 1] = t.y;
    }
  } else {
    const float* p = a.x + row * kHidden + lane * 8;
    const float4 u = *reinterpret_cast<const float4*>(p);
    const float4 v = *reinterpret_cast<const float4*>(p + 4);
    e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = v.x; e[5] = v.y; e[6] = v.z; e[7] = v.w;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += e[i];
  const float mean = wave_sum(s) * (1.0f / kHidden);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] -= mean; q += e[i] * e[i]; }
  const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / kHidden) + kLnEps);
  TO* o = reinterpret_cast<TO*>(a.out) + row * kHidden + lane * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) Elem<TO>::st(o + i, e[i] * rs * a.scale[lane * 8 + i]);
}

// (UpdateArgs: gemm.hpp -- the few-row output Linear applies the update in its epilogue)
__global__ __launch_bounds__(128) void update_kernel(UpdateArgs a) {
  const long r = blockIdx.x;
  const int tid = threadIdx.x;
  const float* res = a.res + r * kMixOut;
  const long bq = r / a.T;
  for (int c = tid; c < kFeatDim; c += 128) {
    float prev;
    if (a.first_of_level)
      prev = (c < kHiresDim) ? a.q_hires[bq * kHiresDim + c] : a.q_lowres[bq * kLowresDim + (c - kHiresDim)];
    else
      prev = a.feats[r * kFeatDim + c];
    a.feats[r * kFeatDim + c] = res[4 + c] + prev;
  }
  if (tid == 0) {
    const float px = a.pos[r * 2 + 0] + res[0] * a.sx;
    const float py = a.pos[r * 2 + 1] + res[1] * a.sy;
    const float oc = a.occ[r] + res[2];
    const float ex = a.expd[r] + res[3];
    a.pos[r * 2 + 0] = px; a.pos[r * 2 + 1] = py;
    a.out_tracks[r * 2 + 0] = px * a.vx; a.out_tracks[r * 2 + 1] = py * a.vy;
    a.out_occ[r] = oc; a.out_expd[r] = ex;
    a.occ[r] = a.last_of_level ? a.occ0[r] : oc;
    a.expd[r] = a.last_of_level ? a.expd0[r] : ex;
  }
}

}  // namespace tapir
