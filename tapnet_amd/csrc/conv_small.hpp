// The block convolutions of the backbone (tapnet/models/resnet.py:185-257: conv_0 / conv_1 / proj_conv of BlockV2) for
// launches of FEW frames -- the online model's single frame (tapnet/live_demo.py:51-77, tapir_model.py:1156-1203).
//
// Why another form.  conv_fused_kernel gives a 256-channel layer of ONE 256 x 256 frame 16 workgroups, each streaming the
// layer's whole 1.18 MB of weights -- cold: every layer's weights are used once per frame -- through four dependent
// 12-fragment rings and then walking 72 k-steps alone: 33-35 us per launch (profiles/r06_kbench_convflat_v1.txt, 6 frames),
// 24 launches per frame, slower than the library's split-K kernels with all their glue around them (2.67 against 2.56 ms per
// frame, profiles/r05_online.json).  Few frames need the OPPOSITE decomposition of many frames: parallelism over everything
// that is not pixels.
//   * a workgroup = one tile of whole output rows (<= 128 pixels: NT = 4 or 8 fragments of 16) x ONE fragment row of 16
//     output channels (channels cg * 64 + 16 g + 4 r + e): C_out / 16 times as many workgroups, each fetching a sixteenth
//     (C = 256) of the weights -- 74 KB, of which each of its four waves holds its quarter in registers, ALL of it in flight
//     at once behind the staging (the first version gave a workgroup 64 channels and a 12-fragment ring per wave: six
//     dependent round trips to cold weights per launch, 23-26 us for a C = 256 layer, profiles/r06_online_timeline_v1.txt);
//   * its four waves split K (tap-major k-steps kk = w, w + 4, ...) and keep partial sums of the 16 x NT*16 tile; the
//     partial tiles meet in LDS and wave 0 finishes: + shortcut, round, store, (mean, M2) per channel;
//   * operand load (relu(a x + b) into a swizzled LDS tile), weight fragments (tapir_conv_pack's stream, unchanged),
//     epilogue contract (tile summaries, in-launch merge of the next norm's pairs by the last arriver: fin_merge) are
//     conv_fused.hpp's, so the two forms are interchangeable launch by launch -- NOT bit-identical (the K split changes the
//     summation order): which form a launch takes follows the frame count of the WHOLE clip (tapir_conv_set_small), never a
//     shard or chunk.
// One 256 x 256 frame: 512 / 512 / 256 workgroups per layer at C = 64 / 128 / 256 instead of 64 / 32 / 16.  bf16 only.
#pragma once
#include "conv_fused.hpp"

namespace tapir {

constexpr int CVS_WAVES = 4;
constexpr int CVS_LDS_BYTES = 128 * 1024;     // the largest input tile (two kernel forms: tiles <= 80 KiB -- two workgroups per CU -- and <= 128 KiB)
constexpr int CVS_LDS_SMALL = 76 * 1024;     // (+ 3 KiB of pair / reduction scratch: two workgroups per CU)
constexpr int CVS_MERGE_MAXS = 32;            // tile summaries per lane the consumer-side merge holds in registers

// output rows per tile / tiles per image / fragments per wave; false: the shape stays with conv_fused_kernel
inline bool conv_small_plan(int H, int W, int cin, int cout, int ks, int stride, int* rows, int* tiles, int* nt) {
  if (!conv3_supported(cin, cout, ks, stride) || H < 1 || W < 1) return false;
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  if (Wo > 128) return false;
  int th = Wo <= 64 ? 64 / Wo : 1;             // 64 pixels per tile, or one row of up to 128
  if (th > Ho) th = Ho;
  const long in_cols = (long)stride * (Wo - 1) + ks;
  while (th >= 1 && ((long)stride * (th - 1) + ks) * in_cols * cin * 2 > CVS_LDS_BYTES) --th;
  if (th < 1) return false;
  *rows = th;
  *tiles = (Ho + th - 1) / th;
  *nt = th * Wo <= 64 ? 4 : 8;
  return true;
}
// bytes of the input tile of that plan
inline long conv_small_tile_bytes(int W, int cin, int ks, int stride, int rows) {
  const int Wo = (W + stride - 1) / stride;
  return ((long)stride * (rows - 1) + ks) * ((long)stride * (Wo - 1) + ks) * cin * 2;
}

// (a, b) pairs of image n's input norm from the producer's tile summaries, by every consuming workgroup in its prologue:
// inorm_finalize_kernel's arithmetic (backbone.hpp: weighted mean, then M2 about it, all summaries of a pass in flight at
// once) with L = 256 / CIN lanes per channel; s_pairs[c] = (rstd * gamma, beta - mean * rstd * gamma).  Needs
// slabs_in <= L * CVS_MERGE_MAXS (the host launches inorm_finalize_kernel otherwise).
template <int CIN>
__device__ __forceinline__ void cvs_merge_pairs(const Conv3Args& a, int n, float2* s_pairs, float* s_red /* [256] */) {
  constexpr int L = 256 / CIN;
  const int tid = threadIdx.x;
  const int ch = tid % CIN, q = tid / CIN;
  const int slabs = a.slabs_in, HW = a.H * a.W;
  const int per_s = a.per_s_in > 0 ? a.per_s_in : (HW + slabs - 1) / slabs;
  auto slab_n = [&](int s) { return s < slabs ? (float)max(0, min(HW, (s + 1) * per_s) - s * per_s) : 0.f; };
  const float2* ps = reinterpret_cast<const float2*>(a.part_in) + (long)n * slabs * CIN + ch;
  const float inv_hw = 1.0f / (float)HW;
  float2 v[CVS_MERGE_MAXS];
#pragma unroll
  for (int k = 0; k < CVS_MERGE_MAXS; ++k) v[k] = ps[(long)min(q + k * L, slabs - 1) * CIN];
  float s1 = 0.f;
#pragma unroll
  for (int k = 0; k < CVS_MERGE_MAXS; ++k) s1 = fmaf(slab_n(q + k * L), v[k].x, s1);
  if (L > 1) {
    s_red[tid] = s1;
    lds_barrier();
    s1 = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) s1 += s_red[j * CIN + ch];
  }
  const float mean = s1 * inv_hw;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < CVS_MERGE_MAXS; ++k) {
    const float nk = slab_n(q + k * L);
    const float d = v[k].x - mean;
    m2 += nk > 0.f ? fmaf(nk * d, d, v[k].y) : 0.f;
  }
  if (L > 1) {
    lds_barrier();        // every lane has read the first partial sums
    s_red[tid] = m2;
    lds_barrier();
    m2 = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) m2 += s_red[j * CIN + ch];
  }
  if (q == 0) {
    const float sc = (1.0f / sqrtf(m2 * inv_hw + kInEps)) * a.gamma_in[ch];
    s_pairs[ch] = make_float2(sc, a.beta_in[ch] - mean * sc);
  }
  lds_barrier();
}

// MODE 0: a block convolution (operand = relu(a x + b), epilogue = + shortcut, summaries, next norm's pairs);
// MODE 1: the first convolution of an ExtraConvs block (tapir_model.py:183-184, extra_convs.hpp): operand = x as it is (the
//         LayerNorm kernel's output), epilogue = + bias (a.ss = the bias vector [C_out]), gelu (tanh form), no summaries;
// MODE 2: its second convolution (:185): a.cin_total input channels in chunks of CIN -- the tile is re-staged per chunk, the
//         accumulators persist, the weights are xconv's [chunk][tap][k-step][row tile] stream (tapir_xconv_pack for chunks of
//         CIN) --, epilogue = + bias + skip (a.shortcut), no summaries.
template <int CIN, int COUT, int KS, int STRIDE, int NT, bool HAS_SC, int LDS_BYTES, int MODE = 0>
__global__ __launch_bounds__(CVS_WAVES * 64) void conv_small_kernel(Conv3Args a) {
  typedef bf16_t T;
  constexpr int EPC = 8, THREADS = CVS_WAVES * 64, CG = COUT / 64;
  constexpr int CB = CIN * 2;                      // bytes per input pixel
  constexpr int CPP = CIN / EPC;                   // 16-byte chunks per input pixel
  constexpr int SWZ = (CPP < 16 ? CPP : 16) - 1;
  constexpr int TAPS = KS * KS, KPT = CIN / 32, KT = TAPS * KPT;   // k-steps in all
  constexpr int KW = (KT + 3) / 4;                 // k-steps of a wave (at most)
  __shared__ uint4 s_buf[LDS_BYTES / 16];
  __shared__ float2 s_pairs[MODE == 0 ? CIN : 1];   // the input norm's (a, b) when this workgroup merges them itself
  __shared__ float s_red[MODE == 0 ? 256 : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // Consecutive workgroup ids go round the 8 XCDs (each with its own L2): the C_out / 16 workgroups of ONE tile take the same
  // XCD, so that the tile's input rows are fetched into one L2 once
  const int slot = (int)(blockIdx.x >> 3);
  const int sub = slot % (CG * 4);
  const int bt = (slot / (CG * 4)) * 8 + (int)(blockIdx.x & 7u);
  if (bt >= a.N * a.tiles) return;
  const int rsel = sub & 3;                         // fragment row of this workgroup: channels cg * 64 + 16 g + 4 rsel + e
  const int cg = sub >> 2;
  const int n = bt / a.tiles, t = bt - n * a.tiles;
  const int H = a.H, W = a.W, Wo = a.Wo;
  const int PW = STRIDE * (Wo - 1) + KS;           // columns of the input tile
  const int r0 = t * a.TH;
  const int rows = min(a.TH, a.Ho - r0);
  const int HP = (STRIDE * (rows - 1) + KS) * PW;  // pixels of the input tile
  const int TP = rows * Wo;                        // output pixels of this tile
  const int y0 = STRIDE * r0 - a.pad_y, x0 = -a.pad_x;
  char* const tile = reinterpret_cast<char*>(s_buf);

  const int mine = (KT - wave + 3) / 4;            // k-steps of this wave per chunk (0 for the high waves of a 1x1 with few channels)
  const uint4* const wbase = a.wstream + ((long)cg * a.frags_per_cg) * 64 + lane;
  const int nch = MODE == 2 ? a.cin_total / CIN : 1;   // input-channel chunks
  const int CT = MODE == 2 ? a.cin_total : CIN;        // channels per input pixel in memory

  int Pc[NT], qpix[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int q = i * 16 + c;
    qpix[i] = q;
    const int qq = q < TP ? q : 0;
    const int yy = qq / Wo, xx = qq - yy * Wo;
    Pc[i] = STRIDE * (yy * PW + xx);
  }
  const long img = ((long)n * a.Ho + r0) * Wo;     // first output pixel of the tile
  // the shortcut values the finishing wave will add (channels cg * 64 + 16 g + 4 rsel + e): in flight from here
  uint2 scv[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    scv[i] = make_uint2(0u, 0u);
    if (HAS_SC) {
      const int qq = qpix[i] < TP ? qpix[i] : 0;
      scv[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(a.shortcut) + (img + qq) * COUT + cg * 64 + 16 * g + 4 * rsel);
    }
  }

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < nch; ++ch) {
  // ---- this wave's share of the chunk's weights (fragment row rsel of the k-steps wave, wave + 4, ...): ALL in flight during
  // the staging
  uint4 wf[KW];
#pragma unroll
  for (int j = 0; j < KW; ++j) wf[j] = wbase[(long)((ch * KT + min(wave + 4 * j, KT - 1)) * 4 + rsel) * 64];   // (past the end: valid, not used)
  if (MODE == 0 && a.part_in != nullptr) cvs_merge_pairs<CIN>(a, n, s_pairs, s_red);   // (under the weight loads just issued)
  // ---- stage relu(a x + b) of the input tile (conv_fused_kernel's walk: a thread keeps one 8-channel chunk)
  {
    constexpr int PPS = THREADS / CPP;
    constexpr int U = 18;                          // loads in flight per thread: one trip covers the 4 x 34-pixel tile of the 32 x 32 maps
    const int chunk = tid % CPP, pl = tid / CPP;
    f32x4 ssv[4];
    if (MODE == 0) {
      if (a.part_in != nullptr) {
        float2 pr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = s_pairs[EPC * chunk + e];
        ssv[0] = f32x4{pr[0].x, pr[1].x, pr[2].x, pr[3].x}; ssv[1] = f32x4{pr[4].x, pr[5].x, pr[6].x, pr[7].x};
        ssv[2] = f32x4{pr[0].y, pr[1].y, pr[2].y, pr[3].y}; ssv[3] = f32x4{pr[4].y, pr[5].y, pr[6].y, pr[7].y};
      } else {
        const f32x4* sp = reinterpret_cast<const f32x4*>(a.ss + ((long)n * CIN + EPC * chunk) * 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) ssv[k] = sp[k];
      }
    }
    const T* xin = reinterpret_cast<const T*>(a.x) + (long)n * H * W * CT + ch * CIN + EPC * chunk;
    const int dq = PPS / PW, dr = PPS - dq * PW;
    int hy = pl / PW, hx = pl - hy * PW;
    for (int P0 = pl; P0 < HP; P0 += U * PPS) {
      uint4 v[U];
      int off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int P = P0 + u * PPS;
        const int y = y0 + hy, x = x0 + hx;
        const bool in = P < HP && y >= 0 && y < H && x >= 0 && x < W;
        const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        v[u] = *reinterpret_cast<const uint4*>(xin + (long)(yc * W + xc) * CT);
        off[u] = P < HP ? ((P * CB + ((chunk ^ (P & SWZ)) << 4)) | (in ? 0 : (1 << 30))) : -1;
        hx += dr; hy += dq;
        if (hx >= PW) { hx -= PW; ++hy; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const unsigned m = (off[u] >> 30) ? 0u : 0xffffffffu;
        unsigned r4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (MODE != 0) { r4[k] = w4[k] & m; continue; }
          const f32x2 xv = f32x2{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xffff0000u)};
          const f32x2 sa = f32x2{ssv[k >> 1][2 * (k & 1)], ssv[k >> 1][2 * (k & 1) + 1]};
          const f32x2 sb = f32x2{ssv[2 + (k >> 1)][2 * (k & 1)], ssv[2 + (k >> 1)][2 * (k & 1) + 1]};
          const f32x2 yv = __builtin_elementwise_fma(xv, sa, sb);
          r4[k] = relu_bf16x2(pack_bf16x2(yv.x, yv.y)) & m;
        }
        if (off[u] >= 0) *reinterpret_cast<uint4*>(tile + (off[u] & 0x3fffffff)) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
      }
    }
  }
  lds_barrier();

  // ---- this wave's k-steps over the whole tile
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    const int kk = min(wave + 4 * j, KT - 1);
    const int tap = kk / KPT, ks = kk - tap * KPT;
    const int dy = (tap * 11) >> 5;                // tap / 3 for tap < 9
    const int toff = KS == 3 ? dy * PW + (tap - 3 * dy) : 0;
    uint4 fb[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int P = Pc[i] + toff;
      fb[i] = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
    }
    if (j < mine) {
#pragma unroll
      for (int i = 0; i < NT; ++i) MfmaStep<T>::run(wf[j], fb[i], acc[i]);
    }
  }
  lds_barrier();   // every wave is done with the tile: the region now takes the next chunk / the partial tiles
  }

  // ---- the four partial tiles meet: s_part[wave][i][lane]; wave 0 finishes (16 channels x TP pixels)
  f32x4* const s_part = reinterpret_cast<f32x4*>(s_buf);
#pragma unroll
  for (int i = 0; i < NT; ++i) s_part[(wave * NT + i) * 64 + lane] = acc[i];
  lds_barrier();
  const bool fin_wave = wave == 0;
  f32x4 v[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    v[i] = s_part[(0 * NT + i) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v[i] += s_part[(w * NT + i) * 64 + lane];
  }
  uint2 pk[NT];
  T* const ytile = reinterpret_cast<T*>(a.y) + img * COUT;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (MODE != 0) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.ss + cg * 64 + 16 * g + 4 * rsel);
      v[i] += bv;
      if (MODE == 1) { v[i][0] = gelu_tanh(v[i][0]); v[i][1] = gelu_tanh(v[i][1]); v[i][2] = gelu_tanh(v[i][2]); v[i][3] = gelu_tanh(v[i][3]); }
    }
    if (HAS_SC) {
      v[i][0] += __uint_as_float(scv[i].x << 16); v[i][1] += __uint_as_float(scv[i].x & 0xffff0000u);
      v[i][2] += __uint_as_float(scv[i].y << 16); v[i][3] += __uint_as_float(scv[i].y & 0xffff0000u);
    }
    pk[i].x = pack_bf16x2(v[i][0], v[i][1]);
    pk[i].y = pack_bf16x2(v[i][2], v[i][3]);
    if (fin_wave && qpix[i] < TP) *reinterpret_cast<uint2*>(ytile + (long)qpix[i] * COUT + cg * 64 + 16 * g + 4 * rsel) = pk[i];
  }
  if (MODE != 0 || a.part == nullptr) return;

  // ---- (mean, M2) of the STORED values of the tile, per channel: wave 0 holds all TP pixels of the 4 x 4 channels
  if (fin_wave) {
    const float inv_cnt = 1.0f / (float)TP;
    float vv[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool ok = qpix[i] < TP;
      vv[i][0] = ok ? __uint_as_float(pk[i].x << 16) : 0.f;
      vv[i][1] = ok ? __uint_as_float(pk[i].x & 0xffff0000u) : 0.f;
      vv[i][2] = ok ? __uint_as_float(pk[i].y << 16) : 0.f;
      vv[i][3] = ok ? __uint_as_float(pk[i].y & 0xffff0000u) : 0.f;
    }
    float mean[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i) s += vv[i][e];
      mean[e] = s;
    }
    row_sum_n<4>(mean);
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] *= inv_cnt;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const float d = qpix[i] < TP ? vv[i][e] - mean[e] : 0.f;
        s = fmaf(d, d, s);
      }
      m2[e] = s;
    }
    row_sum_n<4>(m2);
    float* const part = a.part + (((long)n * a.tiles + t) * COUT + cg * 64 + 16 * g + 4 * rsel) * 2;
    if (c == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.fin.ss != nullptr) agent_store_f2(part + 2 * e, mean[e], m2[e]);   // write-through: read by another workgroup
        else *reinterpret_cast<float2*>(part + 2 * e) = make_float2(mean[e], m2[e]);
      }
    }
  }
  if (a.fin.ss == nullptr) return;
  // ---- the next norm's (a, b) pairs by the image's last arriver (conv_fused.hpp: cv3_epilogue / fin_merge; here an image
  // has tiles x C_out / 16 arrivals)
  dma_wait<0>();
  lds_barrier();          // (also: the partial tiles are dead, the region is reused by fin_merge)
  int* const s_flag = reinterpret_cast<int*>(s_buf);
  if (tid == 0) {
    const int last = agent_fetch_add(a.fin.arrive + n, 1) == a.tiles * CG * 4 - 1;
    if (last) agent_store_int(a.fin.arrive + n, 0);
    *s_flag = last;
  }
  lds_barrier();
  if (*s_flag == 0) return;
  fin_merge<COUT, THREADS>(a.fin, a.part + (long)n * a.tiles * COUT * 2, n, a.tiles, a.TH * a.Wo, a.Ho * a.Wo,
                           reinterpret_cast<float*>(s_buf) + 16);
}

inline void launch_conv_small(const Conv3Args& a, int cin, int cout, int ks, int stride, int nt, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((a.N * a.tiles + 7) / 8) * (cout / 16))), block(CVS_WAVES * 64);
  const bool lds_small = conv_small_tile_bytes(a.W, cin, ks, stride, a.TH) <= CVS_LDS_SMALL;   // (partials: <= 32 KiB, in the dead tile)
#define TAPIR_CVS_L(CI_, CO_, K_, S_, NT_, SC_)                                                                      \
  do {                                                                                                              \
    if (lds_small) TAPIR_LAUNCH((conv_small_kernel<CI_, CO_, K_, S_, NT_, SC_, CVS_LDS_SMALL>), grid, block, s, a); \
    else TAPIR_LAUNCH((conv_small_kernel<CI_, CO_, K_, S_, NT_, SC_, CVS_LDS_BYTES>), grid, block, s, a);           \
  } while (0)
#define TAPIR_CVS(CI_, CO_, K_, S_)                                                        \
  do {                                                                                     \
    if ((K_) == 3 && (S_) == 1 && a.shortcut) {                                            \
      if (nt == 4) TAPIR_CVS_L(CI_, CO_, K_, S_, 4, (K_ == 3 && S_ == 1));                 \
      else TAPIR_CVS_L(CI_, CO_, K_, S_, 8, (K_ == 3 && S_ == 1));                         \
    } else {                                                                               \
      if (nt == 4) TAPIR_CVS_L(CI_, CO_, K_, S_, 4, false);                                \
      else TAPIR_CVS_L(CI_, CO_, K_, S_, 8, false);                                        \
    }                                                                                      \
  } while (0)
  if (stride == 1) {
    if (ks == 3) {
      if (cin == 64) TAPIR_CVS(64, 64, 3, 1);
      else if (cin == 128) TAPIR_CVS(128, 128, 3, 1);
      else TAPIR_CVS(256, 256, 3, 1);
    } else {
      if (cin == 64) TAPIR_CVS(64, 64, 1, 1);
      else TAPIR_CVS(256, 256, 1, 1);
    }
  } else {
    if (ks == 3) {
      if (cin == 64) TAPIR_CVS(64, 128, 3, 2);
      else TAPIR_CVS(128, 256, 3, 2);
    } else {
      if (cin == 64) TAPIR_CVS(64, 128, 1, 2);
      else TAPIR_CVS(128, 256, 1, 2);
    }
  }
#undef TAPIR_CVS
#undef TAPIR_CVS_L
  (void)cout;
}

// the 256 -> C_out (a multiple of 64) ExtraConvs convolution with bias + gelu in the few-frame form; false: not covered
inline bool launch_xconv_small(const Conv3Args& a, int cin, int cout, int nt, hipStream_t s) {
  if (cin == 1024 && cout == 256 && a.shortcut != nullptr) {   // the block's second convolution: four chunks of 256, + bias + skip
    const dim3 grid2((unsigned)(8 * ((a.N * a.tiles + 7) / 8) * (cout / 16))), block2(CVS_WAVES * 64);
    const bool small2 = conv_small_tile_bytes(a.W, 256, 3, 1, a.TH) <= CVS_LDS_SMALL;
    if (nt == 4) {
      if (small2) TAPIR_LAUNCH((conv_small_kernel<256, 256, 3, 1, 4, true, CVS_LDS_SMALL, 2>), grid2, block2, s, a);
      else TAPIR_LAUNCH((conv_small_kernel<256, 256, 3, 1, 4, true, CVS_LDS_BYTES, 2>), grid2, block2, s, a);
    } else {
      if (small2) TAPIR_LAUNCH((conv_small_kernel<256, 256, 3, 1, 8, true, CVS_LDS_SMALL, 2>), grid2, block2, s, a);
      else TAPIR_LAUNCH((conv_small_kernel<256, 256, 3, 1, 8, true, CVS_LDS_BYTES, 2>), grid2, block2, s, a);
    }
    return true;
  }
  if (cin != 256 || cout != 1024) return false;
  const dim3 grid((unsigned)(8 * ((a.N * a.tiles + 7) / 8) * (cout / 16))), block(CVS_WAVES * 64);
  const bool lds_small = conv_small_tile_bytes(a.W, cin, 3, 1, a.TH) <= CVS_LDS_SMALL;
  if (nt == 4) {
    if (lds_small) TAPIR_LAUNCH((conv_small_kernel<256, 1024, 3, 1, 4, false, CVS_LDS_SMALL, 1>), grid, block, s, a);
    else TAPIR_LAUNCH((conv_small_kernel<256, 1024, 3, 1, 4, false, CVS_LDS_BYTES, 1>), grid, block, s, a);
  } else {
    if (lds_small) TAPIR_LAUNCH((conv_small_kernel<256, 1024, 3, 1, 8, false, CVS_LDS_SMALL, 1>), grid, block, s, a);
    else TAPIR_LAUNCH((conv_small_kernel<256, 1024, 3, 1, 8, false, CVS_LDS_BYTES, 1>), grid, block, s, a);
  }
  return true;
}

}  // namespace tapir
