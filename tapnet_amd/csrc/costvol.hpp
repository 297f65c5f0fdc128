// TAPIR.tracks_from_cost_volume after the einsum (tapnet/models/tapir_model.py:438-471)
// fused per (b, query, frame) heat map; nothing wider than the 1-channel cost
// map ever leaves the chip:
//   Conv 1->16 3x3 SAME + ReLU :443-444     (kept in LDS, zero halo)
//   Conv 16->1 3x3 SAME :446 -> softmax(temp * x) over (h, w) :454
//   soft-argmax radius 5 -> (x, y) * (W/w, H/h), query-frame override :455
//     (model_utils.soft_argmax_heatmap :209-247, heatmaps_to_points :250-314)
//   Conv 16->32 3x3 stride 2 XLA-SAME + ReLU -> mean(h, w) -> Linear 32->16 + ReLU
//   -> Linear 16->2 = [occlusion, expected_dist] logits :459-470
// One 256-thread workgroup per map; cells are strided over the threads.
#pragma once
#include "common.hpp"

namespace tapir {

constexpr int CV_THREADS = 256;
// Two instantiations, chosen by the launcher from the grid size:
//   <1156, 4>: up to 32x32 cells (the 256x256 model), 80 KiB LDS -> 2 workgroups / CU
//   <1936, 7>: up to 1600 cells / (h+2)(w+2) <= 1936 (e.g. 40x40), 132 KiB LDS
constexpr int CV_SMALL_PAD = 1156, CV_SMALL_PPT = 4;
constexpr int CV_LARGE_PAD = 1936, CV_LARGE_PPT = 7;

struct CvHeadWeights {
  const float* w1;   // [16][9]
  const float* b1;   // [16]
  const float* w2;   // [16][9]
  const float* b2;   // [1]
  const float* w3;   // [(ci*9+tap)][32]   re-laid-out from [32,16,3,3]
  const float* b3;   // [32]
  const float* w4;   // [16][32]
  const float* b4;   // [16]
  const float* w5;   // [2][16]
  const float* b5;   // [2]
  const uint4* w3b;  // bf16 build: conv-3 weights as MFMA B fragments [5 k-steps][2 n-tiles][64 lanes] x 8 bf16
};

struct CvHeadArgs {
  const float* cv;        // [maps, h*w] cost volume, map index = (b*Q + q)*T + t
  long ld;                // 0, or floats between the rows of consecutive queries (>= T*h*w: rows padded to 16 bytes)
  CvHeadWeights wt;
  const float* qpts;      // [B*Q, 3] (t, y, x) in initial_resolution coordinates, or null
  float* points;          // [maps, 2] (x, y) in initial_resolution pixels
  float* occ;             // [maps]
  float* expd;            // [maps]
  int T, h, w;
  long maps;              // number of maps (cv_heads_mfma_kernel walks them persistently)
  float temperature;
  float img_h, img_w;     // initial_resolution
  long long* dbg_times;   // null, or [workgroups][8] wall-clock phase totals (tools/kbench.py --what cvtrace)
};

template <int CV_MAX_PAD, int CV_PPT>
__global__ __launch_bounds__(CV_THREADS) void cv_heads_kernel(CvHeadArgs a) {
  __shared__ float s_cm[CV_MAX_PAD];            // cost map with zero halo
  __shared__ float s_h1[16 * CV_MAX_PAD];       // relu(hid1) with zero halo
  __shared__ float s_red[8][4];
  __shared__ int s_redi[4];
  __shared__ float s_vec[32 + 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const long map = blockIdx.x;
  const int h = a.h, w = a.w, hw = h * w;
  const int pw = w + 2, ph = h + 2, pn = pw * ph;
  const float* cv = a.ld ? a.cv + (map / a.T) * a.ld + (map % a.T) * hw : a.cv + map * hw;

  // ---- cost map + zero halos
  for (int i = tid; i < pn; i += CV_THREADS) {
    const int y = i / pw - 1, x = i % pw - 1;
    const bool in = (y >= 0) && (y < h) && (x >= 0) && (x < w);
    s_cm[i] = in ? cv[y * w + x] : 0.f;
    if (!in) {
#pragma unroll
      for (int c = 0; c < 16; ++c) s_h1[c * pn + i] = 0.f;
    }
  }
  __syncthreads();

  // ---- hid1 = relu(conv3x3(cost) + b)
  for (int p = tid; p < hw; p += CV_THREADS) {
    const int y = p / w, x = p % w;
    float v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = s_cm[(y + dy) * pw + (x + dx)];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float acc = a.wt.b1[c];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc = fmaf(a.wt.w1[c * 9 + k], v[k], acc);
      s_h1[c * pn + (y + 1) * pw + (x + 1)] = fmaxf(acc, 0.f);
    }
  }
  __syncthreads();

  // ---- logits = conv3x3(hid1) + b, scaled by the temperature
  float z[CV_PPT];
  float zmax = -3.0e38f;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * CV_THREADS;
    z[s] = -3.0e38f;
    if (p < hw) {
      const int y = p / w, x = p % w;
      float acc = a.wt.b2[0];
      for (int c = 0; c < 16; ++c) {
        const float* hp = s_h1 + c * pn + y * pw + x;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc = fmaf(a.wt.w2[c * 9 + dy * 3 + dx], hp[dy * pw + dx], acc);
      }
      z[s] = acc * a.temperature;
      zmax = fmaxf(zmax, z[s]);
    }
  }
  // block max
  zmax = wave_max(zmax);
  if (lane == 0) s_red[0][wave] = zmax;
  __syncthreads();
  zmax = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
  float esum = 0.f;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * CV_THREADS;
    z[s] = (p < hw) ? fast_exp(z[s] - zmax) : 0.f;
    esum += z[s];
  }
  esum = wave_sum(esum);
  if (lane == 0) s_red[1][wave] = esum;
  __syncthreads();
  esum = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  // softmax values; argmax = FIRST maximum (jnp.argmax), model_utils.py:232
  float best = -1.f;
  int besti = 0x7fffffff;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * CV_THREADS;
    if (p < hw) {
      z[s] = z[s] / esum;
      if (z[s] > best) { best = z[s]; besti = p; }   // p increases with s: keeps the first
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(besti, off);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) { s_red[2][wave] = best; s_redi[wave] = besti; }
  __syncthreads();
  best = s_red[2][0]; besti = s_redi[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float ob = s_red[2][k]; const int oi = s_redi[k];
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  const float ax = (float)(besti % w) + 0.5f, ay = (float)(besti / w) + 0.5f;
  float sx = 0.f, sy = 0.f, sw = 0.f;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * CV_THREADS;
    if (p < hw) {
      const float cx = (float)(p % w) + 0.5f, cy = (float)(p / w) + 0.5f;
      const float d2 = (cx - ax) * (cx - ax) + (cy - ay) * (cy - ay);
      if (d2 < 25.0f) { sx += cx * z[s]; sy += cy * z[s]; sw += z[s]; }   // threshold 5, strict
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sw = wave_sum(sw);
  if (lane == 0) { s_red[3][wave] = sx; s_red[4][wave] = sy; s_red[5][wave] = sw; }

  // ---- occlusion head: conv 16->32, 3x3, stride 2, XLA SAME (pad_lo = total/2)
  const int oh = (h + 1) / 2, ow = (w + 1) / 2;
  const int ply = max((oh - 1) * 2 + 3 - h, 0) / 2, plx = max((ow - 1) * 2 + 3 - w, 0) / 2;
  float o3[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) o3[k] = 0.f;
  for (int p = tid; p < oh * ow; p += CV_THREADS) {
    const int oy = p / ow, ox = p % ow;
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = a.wt.b3[k];
    for (int c = 0; c < 16; ++c) {
      // padded coordinates: input row 2*oy + dy - ply, +1 for the halo
      const float* hp = s_h1 + c * pn + (2 * oy - ply + 1) * pw + (2 * ox - plx + 1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float v = hp[dy * pw + dx];
          const float* wp = a.wt.w3 + (c * 9 + dy * 3 + dx) * 32;
#pragma unroll
          for (int k = 0; k < 32; ++k) acc[k] = fmaf(wp[k], v, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) o3[k] += fmaxf(acc[k], 0.f);
  }
  __syncthreads();              // everyone is done reading s_h1 / s_cm
  float* s_o3 = s_h1;           // reuse: [256][33]
#pragma unroll
  for (int k = 0; k < 32; ++k) s_o3[tid * 33 + k] = o3[k];
  __syncthreads();
  if (tid < 32) {
    float m = 0.f;
    for (int i = 0; i < CV_THREADS; ++i) m += s_o3[i * 33 + tid];
    s_vec[tid] = m / (float)(oh * ow);
  }
  __syncthreads();
  if (tid < 16) {
    float acc = a.wt.b4[tid];
    for (int k = 0; k < 32; ++k) acc = fmaf(a.wt.w4[tid * 32 + k], s_vec[k], acc);
    s_vec[32 + tid] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  if (tid < 2) {
    float acc = a.wt.b5[tid];
    for (int k = 0; k < 16; ++k) acc = fmaf(a.wt.w5[tid * 16 + k], s_vec[32 + k], acc);
    if (tid == 0) a.occ[map] = acc; else a.expd[map] = acc;
  }
  if (tid == 0) {
    const float fsx = s_red[3][0] + s_red[3][1] + s_red[3][2] + s_red[3][3];
    const float fsy = s_red[4][0] + s_red[4][1] + s_red[4][2] + s_red[4][3];
    const float fsw = fmaxf(s_red[5][0] + s_red[5][1] + s_red[5][2] + s_red[5][3], 1e-12f);
    float outx = (fsx / fsw) * a.img_w / (float)w;
    float outy = (fsy / fsw) * a.img_h / (float)h;
    if (a.qpts != nullptr) {
      const long bq = map / a.T;
      const int t = (int)(map % a.T);
      const float* q = a.qpts + bq * 3;
      if ((int)rintf(q[0]) == t) { outx = q[2]; outy = q[1]; }   // round-half-even like jnp.round
    }
    a.points[map * 2 + 0] = outx;
    a.points[map * 2 + 1] = outy;
  }
}

// ---------------------------------------------------------------------------------------------
// cv_heads_mfma_kernel: the same heads for the bf16 build, with the occlusion convolution
// (16 -> 32 channels, 3x3, stride 2: 80 % of the arithmetic of the VALU kernel above, and one LDS
// read per 32 FMAs) on the matrix cores as an implicit GEMM:
//   M = oh*ow output pixels, N = 32 channels, K = 9 taps x 16 channels (padded to 160 = 5 x 32)
//   A fragment (lane: pixel i = l & 15, k-group g = l >> 4) = 8 consecutive channels of ONE tap of
//   one input pixel -> hid1 is kept in LDS pixel-major ([padded pixel][16] f32, 64 B per pixel): two
//   ds_read_b128 + 4 v_cvt_pk_bf16_f32 per fragment; B fragments (weights) live in registers.
// Everything that feeds the soft-argmax (conv 1->16, conv 16->1, softmax) stays f32 VALU: only
// the occlusion / expected-distance logits see bf16 rounding.  The pixel-major layout also cuts the
// LDS reads of the 16 -> 1 convolution from 144 x b32 to 36 x b128 per cell.
// NT threads per workgroup x CV_PPT cells per thread >= h*w.  Production: (256, 4): 256 VGPRs, 8
// waves per CU.  (512, 2) capped at 128 VGPRs for 16 waves per CU (LDS holds two workgroups either
// way) spills 48 registers and measured 405 us against 340.
// TRACE: thread 0 accumulates wall-clock phase totals into a.dbg_times (tools/kbench.py --what cvtrace);
// a separate instantiation, the run-time test alone cost the production kernel 10 %.
template <int CV_MAX_PAD, int CV_PPT, int NT, bool TRACE = false>
__global__ __launch_bounds__(NT, 2) void cv_heads_mfma_kernel(CvHeadArgs a) {
  constexpr int NW = NT / 64;   // waves
  __shared__ float s_cm[CV_MAX_PAD];             // cost map with zero halo
  // relu(hid1) with zero halo, [pixel][4 channel quads]; quad c4 of pixel p sits at slot
  // c4 ^ ((p >> 2) & 3): 16 consecutive pixels then cover all 64 banks for one quad index
  // (64-byte pixel stride would otherwise be a 4-way conflict on every ds_read_b128)
  __shared__ float4 s_h1[CV_MAX_PAD][4];
  __shared__ float s_red[8][NW];
  __shared__ int s_redi[NW];
  __shared__ float s_vec[32 + 16];
  __shared__ float s_occ[NW][32];
  // conv-1 / conv-2 weights as LDS broadcasts (as wave-uniform scalars they need 300 SGPRs and
  // were spilled): s_w1[c] = {w1[c][0..8], b1[c], -, -}, s_w2[tap] = w2[0..15][tap]
  __shared__ float4 s_w1[16][3];
  __shared__ float4 s_w2[9][4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = a.h, w = a.w, hw = h * w;
  const int pw = w + 2, ph = h + 2, pn = pw * ph;
  constexpr int CM_PT = (CV_MAX_PAD + NT - 1) / NT;   // padded cells per thread

  // B fragments of the occlusion convolution (10 KiB, L2-resident), in flight during phase A/B
  uint4 wb[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) wb[s][t] = a.wt.w3b[(s * 2 + t) * 64 + lane];

  if (tid < 16) {
    const float* wp = a.wt.w1 + tid * 9;
    s_w1[tid][0] = make_float4(wp[0], wp[1], wp[2], wp[3]);
    s_w1[tid][1] = make_float4(wp[4], wp[5], wp[6], wp[7]);
    s_w1[tid][2] = make_float4(wp[8], a.wt.b1[tid], 0.f, 0.f);
  } else if (tid >= 64 && tid < 64 + 36) {
    const int tap = (tid - 64) >> 2, c4 = (tid - 64) & 3;
    s_w2[tap][c4] = make_float4(a.wt.w2[(4 * c4 + 0) * 9 + tap], a.wt.w2[(4 * c4 + 1) * 9 + tap],
                                a.wt.w2[(4 * c4 + 2) * 9 + tap], a.wt.w2[(4 * c4 + 3) * 9 + tap]);
  }
  // Persistent walk over the maps: the weights above are fetched once per workgroup, and the
  // cost values of the NEXT map are requested (into registers) before the current one is
  // processed, so their HBM latency hides under ~10 us of arithmetic.
  float cmv[CM_PT];
  auto fetch = [&](long m) {
    const float* cvp = a.ld ? a.cv + (m / a.T) * a.ld + (m % a.T) * hw : a.cv + m * hw;
#pragma unroll
    for (int s = 0; s < CM_PT; ++s) {
      const int i = tid + s * NT;
      const int y = i / pw - 1, x = i % pw - 1;
      const bool in = (i < pn) && (y >= 0) && (y < h) && (x >= 0) && (x < w);
      cmv[s] = in ? cvp[y * w + x] : 0.f;
    }
  };
  fetch(blockIdx.x);
  // zero halos of hid1 (never written afterwards)
  for (int i = tid; i < pn; i += NT) {
    const int y = i / pw - 1, x = i % pw - 1;
    if (!((y >= 0) && (y < h) && (x >= 0) && (x < w))) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s_h1[i][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int k) {   // phase k ends here (thread 0 only; no effect unless tracing)
    if (TRACE && tid == 0) { const long long t = wall_clock64(); tph[k] += t - tlast; tlast = t; }
  };
  if (TRACE && tid == 0) tlast = wall_clock64();
  for (long map = blockIdx.x; map < a.maps; map += gridDim.x) {
  __syncthreads();   // the previous map's readers of s_cm / s_h1 / s_vec are done
  stamp(0);
#pragma unroll
  for (int s = 0; s < CM_PT; ++s) {
    const int i = tid + s * NT;
    if (i < pn) s_cm[i] = cmv[s];
  }
  if (map + gridDim.x < a.maps) fetch(map + gridDim.x);
  __syncthreads();
  stamp(1);

  // ---- hid1 = relu(conv3x3(cost) + b), pixel-major.  The thread's CV_PPT cells go through the 16
  // channels TOGETHER: one set of weight broadcasts (3 x ds_read_b128) per channel serves all of
  // them (per cell it was 48 LDS instructions for 144 FMAs; the weights cannot stay in registers:
  // 160 + 144 values, and as wave-uniform scalars they spill).
  // (Pairing the cells into v_pk_fma_f32 with splat weights measured SLOWER, 380 vs 344 us: the
  // pairs have to be assembled from single ds_read_b32 values.)
  {
    float v[CV_PPT][9];
    int pp[CV_PPT];
#pragma unroll
    for (int s = 0; s < CV_PPT; ++s) {
      const int p = min(tid + s * NT, hw - 1);
      const int y = p / w, x = p % w;
      pp[s] = (y + 1) * pw + (x + 1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) v[s][dy * 3 + dx] = s_cm[(y + dy) * pw + (x + dx)];
    }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      float o[CV_PPT][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = 4 * c4 + cc;
#ifndef TAPIR_HIPEMU
        asm volatile("" ::: "memory");   // re-read the weights per channel instead of pinning 160 VGPRs
#endif
        const float4 wa = s_w1[c][0], wb4 = s_w1[c][1], wc = s_w1[c][2];
#pragma unroll
        for (int s = 0; s < CV_PPT; ++s) {
          float acc = wc.y;
          acc = fmaf(wa.x, v[s][0], acc); acc = fmaf(wa.y, v[s][1], acc); acc = fmaf(wa.z, v[s][2], acc);
          acc = fmaf(wa.w, v[s][3], acc); acc = fmaf(wb4.x, v[s][4], acc); acc = fmaf(wb4.y, v[s][5], acc);
          acc = fmaf(wb4.z, v[s][6], acc); acc = fmaf(wb4.w, v[s][7], acc); acc = fmaf(wc.x, v[s][8], acc);
          o[s][cc] = fmaxf(acc, 0.f);
        }
      }
#pragma unroll
      for (int s = 0; s < CV_PPT; ++s)
        if (tid + s * NT < hw)
          s_h1[pp[s]][c4 ^ ((pp[s] >> 2) & 3)] = make_float4(o[s][0], o[s][1], o[s][2], o[s][3]);
    }
  }
  __syncthreads();

  stamp(2);
  // ---- logits = conv3x3(hid1) + b, scaled by the temperature (f32 VALU); again all cells of the
  // thread per weight read
  float z[CV_PPT];
  float zmax = -3.0e38f;
  {
    int base[CV_PPT];
    float acc[CV_PPT];
#pragma unroll
    for (int s = 0; s < CV_PPT; ++s) {
      const int p = min(tid + s * NT, hw - 1);
      base[s] = (p / w) * pw + (p % w);
      acc[s] = a.wt.b2[0];
    }
#pragma unroll 3
    for (int k = 0; k < 9; ++k) {
#ifndef TAPIR_HIPEMU
      asm volatile("" ::: "memory");   // as above, for the conv-2 weights
#endif
      const int koff = (k / 3) * pw + (k % 3);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 wv = s_w2[k][c];
#pragma unroll
        for (int s = 0; s < CV_PPT; ++s) {
          const int q = base[s] + koff;
          const float4 hv = s_h1[q][c ^ ((q >> 2) & 3)];
          acc[s] = fmaf(wv.x, hv.x, acc[s]);
          acc[s] = fmaf(wv.y, hv.y, acc[s]);
          acc[s] = fmaf(wv.z, hv.z, acc[s]);
          acc[s] = fmaf(wv.w, hv.w, acc[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < CV_PPT; ++s) {
      const bool in = tid + s * NT < hw;
      z[s] = in ? acc[s] * a.temperature : -3.0e38f;
      zmax = fmaxf(zmax, z[s]);
    }
  }
  stamp(3);
  // block maximum AND its first position in one reduction: argmax(softmax) = argmax(logits)
  // (monotonic), ties -> the smallest index as jnp.argmax (model_utils.py:232)
  float best = -3.0e38f;
  int besti = 0x7fffffff;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * NT;
    if (p < hw && z[s] > best) { best = z[s]; besti = p; }   // p increases with s: keeps the first
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(besti, off);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) { s_red[0][wave] = best; s_redi[wave] = besti; }
  __syncthreads();
  best = s_red[0][0]; besti = s_redi[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) {
    const float ob = s_red[0][k]; const int oi = s_redi[k];
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  zmax = best;
  // un-normalised softmax weights: total and the soft-argmax window sums (radius 5 around the
  // arg max, strict) in one pass; the normalisation cancels in sx/sw and is applied to sw at the end
  const float ax = (float)(besti % w) + 0.5f, ay = (float)(besti / w) + 0.5f;
  float esum = 0.f, sx = 0.f, sy = 0.f, sw = 0.f;
#pragma unroll
  for (int s = 0; s < CV_PPT; ++s) {
    const int p = tid + s * NT;
    if (p < hw) {
      const float e = fast_exp(z[s] - zmax);
      esum += e;
      const float cx = (float)(p % w) + 0.5f, cy = (float)(p / w) + 0.5f;
      const float d2 = (cx - ax) * (cx - ax) + (cy - ay) * (cy - ay);
      if (d2 < 25.0f) { sx += cx * e; sy += cy * e; sw += e; }
    }
  }
  esum = wave_sum(esum); sx = wave_sum(sx); sy = wave_sum(sy); sw = wave_sum(sw);
  if (lane == 0) { s_red[1][wave] = esum; s_red[3][wave] = sx; s_red[4][wave] = sy; s_red[5][wave] = sw; }

  stamp(4);
  // ---- occlusion head: conv 16->32, 3x3, stride 2, XLA SAME (pad_lo = total/2), on the MFMAs
  const int oh = (h + 1) / 2, ow = (w + 1) / 2, opix = oh * ow;
  const int ply = max((oh - 1) * 2 + 3 - h, 0) / 2, plx = max((ow - 1) * 2 + 3 - w, 0) / 2;
  const int fi = lane & 15, fgp = lane >> 4;
  float osum[2] = {0.f, 0.f};   // sum over this lane's pixels of relu(conv + b), channels n and 16 + n
  const float b3a = a.wt.b3[fi], b3b = a.wt.b3[16 + fi];
  for (int mt = wave; mt * 16 < opix; mt += NT / 64) {
    const int P = min(mt * 16 + fi, opix - 1);          // this lane's A row (clamped: masked below)
    const int oy = P / ow, ox = P - oy * ow;
    const int base = (2 * oy - ply + 1) * pw + (2 * ox - plx + 1);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = min(2 * s + (fgp >> 1), 8);        // tap 9 has zero weights
      const int pp = base + (tap / 3) * pw + (tap % 3);
      const float4* hp = s_h1[pp];
      const int sw = (pp >> 2) & 3;
      const float4 u0 = hp[(2 * (fgp & 1)) ^ sw], u1 = hp[(2 * (fgp & 1) + 1) ^ sw];
      uint4 af;
      af.x = pack_bf16x2(u0.x, u0.y); af.y = pack_bf16x2(u0.z, u0.w);
      af.z = pack_bf16x2(u1.x, u1.y); af.w = pack_bf16x2(u1.z, u1.w);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                     __builtin_bit_cast(bf16x8, wb[s][0]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                     __builtin_bit_cast(bf16x8, wb[s][1]), acc1, 0, 0, 0);
    }
    // D: lane holds channel n = fi, pixels mt*16 + 4*fgp + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (mt * 16 + 4 * fgp + r < opix) {
        osum[0] += fmaxf(acc0[r] + b3a, 0.f);
        osum[1] += fmaxf(acc1[r] + b3b, 0.f);
      }
    }
  }
  // reduce over the four pixel groups of the wave, then over the waves
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    osum[t] += __shfl_xor(osum[t], 16);
    osum[t] += __shfl_xor(osum[t], 32);
  }
  if (lane < 16) { s_occ[wave][lane] = osum[0]; s_occ[wave][16 + lane] = osum[1]; }
  __syncthreads();
  stamp(5);
  // the tail runs in ONE wave: LDS operations of a wave execute in order, no further barriers
  if (wave == 0) {
    if (lane < 32)
    {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += s_occ[k][lane & 31];
      if (lane < 32) s_vec[lane] = t / (float)opix;
    }
    wave_sync();
    if (lane < 16) {
      float acc = a.wt.b4[lane];
      for (int k = 0; k < 32; ++k) acc = fmaf(a.wt.w4[lane * 32 + k], s_vec[k], acc);
      s_vec[32 + lane] = fmaxf(acc, 0.f);
    }
    wave_sync();
    if (lane < 2) {
      float acc = a.wt.b5[lane];
      for (int k = 0; k < 16; ++k) acc = fmaf(a.wt.w5[lane * 16 + k], s_vec[32 + k], acc);
      if (lane == 0) a.occ[map] = acc; else a.expd[map] = acc;
    }
    if (lane == 0) {
      float tot = 0.f, tsx = 0.f, tsy = 0.f, tsw = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) { tot += s_red[1][k]; tsx += s_red[3][k]; tsy += s_red[4][k]; tsw += s_red[5][k]; }
      const float fsx = tsx / tot;
      const float fsy = tsy / tot;
      const float fsw = fmaxf(tsw / tot, 1e-12f);
      float outx = (fsx / fsw) * a.img_w / (float)w;
      float outy = (fsy / fsw) * a.img_h / (float)h;
      if (a.qpts != nullptr) {
        const long bq = map / a.T;
        const int t = (int)(map % a.T);
        const float* q = a.qpts + bq * 3;
        if ((int)rintf(q[0]) == t) { outx = q[2]; outy = q[1]; }   // round-half-even like jnp.round
      }
      a.points[map * 2 + 0] = outx;
      a.points[map * 2 + 1] = outy;
    }
  }
  stamp(6);
  }   // maps
  if (TRACE && tid == 0)
    for (int k = 0; k < 7; ++k) a.dbg_times[(long)blockIdx.x * 8 + k] = tph[k];
}

}  // namespace tapir
