// The 3x3 / stride-1 / 256 -> 256 block convolutions of ResNet groups 2 and 3 (tapnet/models/resnet.py:185-257:
// conv_0 / conv_1 of BlockV2 on the 32 x 32 maps, seven of the eight 3x3 layers of those groups; with the 1x1
// projection of a group's first block fused in, conv_fused.hpp DUAL) in a FLAT tiling of the whole launch.
//
// Why.  conv_fused_kernel cuts every image into tiles of 64 output pixels (2 rows of a 32-wide map): a 48-frame layer is
// 768 workgroups for 512 resident slots -- 1.5 rounds, i.e. two, the second half empty -- and every workgroup streams the
// layer's whole 1.18 MB of weights through its CU's 64 B/clk L2 path for 64 pixels, which takes as long as its 75 MFLOP
// take on the matrix pipe (profiles/r05_conv_phase_trace.txt: 62 k cycles per tile against an 18.6 k MFMA floor).
// Here the (image, row) space of the WHOLE launch is one flat list of slabs (a slab = one tile of the kernel above:
// the same 64 pixel slots, the same InstanceNorm summary slot), and a workgroup owns CVL_SLABS = 3 CONSECUTIVE slabs --
// 192 pixels, wherever the image boundaries fall: 48 frames x 16 slabs / 3 = 256 workgroups = ONE round of the chip with
// one workgroup per CU, a third of the weight bytes per pixel, and 8 input rows staged per 6 rows of output instead of
// 4 per 2.  A workgroup whose slabs straddle two images stages two segments (each with its own image's (a, b) pairs)
// around ONE shared zero row: the bottom padding of the first image is the top padding of the second.
//
// Bit-identical to conv_fused_kernel, by construction and by test (tests/test_conv_flat_emulated.py, tests/test_gpu_conv.py):
//   * every output pixel sees the same k order (tap-major, then the 8 k-steps of a tap) through the same MFMAs;
//   * the (mean, M2) summary of a slab is computed by ONE wave from the same four 16-pixel fragments in the same order
//     as the 4-wave kernel's wave does it -- the slab that straddles the two pixel groups of the workgroup gets the
//     other group's two fragments through LDS (the ROUNDED values, 8 bytes per lane and fragment);
//   * the in-launch merge of the next norm's (a, b) pairs takes one ticket per SLAB, so an image's last arriver is
//     found exactly as before (conv_fused.hpp fin_merge, unchanged).
// So the choice between the two forms is free per launch (it follows the number of slabs of the launch; no "whole clip
// selects the form" rule as for the ExtraConvs forms), and frame shards / chunks stay bit-identical to the whole clip.
#pragma once
#include "conv_fused.hpp"

namespace tapir {

constexpr int CVL_SLABS = 3;                 // slabs (tiles of conv_fused_kernel's 4-wave form) per workgroup
constexpr int CVL_NT = 6;                    // 16-pixel fragments per wave: 2 pixel groups x 6 x 16 = 3 slabs x 64 pixel slots
constexpr int CVL_WAVES = 8;                 // 4 output-channel groups x 2 pixel groups
constexpr int CVL_RING = 12;                 // A fragments in flight per wave (4 k-steps)
constexpr int CVL_LDS_BYTES = 157 * 1024;    // (3 TH + 3) rows x (W + 2) pixels x 512 bytes <= 156 672 for TH W <= 64, W <= 32

// shapes the flat form takes: what conv3_plan gives the 4-wave 256 -> 256 kernel, whole slabs per image, at least one
// workgroup's worth of slabs per image (at most ONE image boundary inside a workgroup)
inline bool conv_flat_supported(int H, int W, int cin, int cout, int ks, int stride, int esize) {
  if (esize != 2 || cin != 256 || cout != 256 || ks != 3 || stride != 1) return false;
  int rows = 0, tiles = 0, waves = 0;
  if (!conv3_plan(H, W, cin, cout, ks, stride, esize, &rows, &tiles, &waves) || waves != 4) return false;
  if (H % rows != 0 || tiles < CVL_SLABS || rows * W > 64) return false;
  return (long)(CVL_SLABS * rows + 3) * (W + 2) * cin * esize <= CVL_LDS_BYTES;
}

// (mean, M2) of one slab = four 16-pixel fragments of ROUNDED values, exactly as cv3_epilogue's wave computes its tile
// (pg = 0, NT = 4): lane (c, g) holds channels 16 g + 4 r + e of pixel slot 16 j + c of fragment j.
__device__ __forceinline__ void cvf_slab_stats(const uint2 (&p)[4][4] /* [r][fragment] */, int TPs, bool slab_ok,
                                               float2* out /* [64 channels of this wave] */) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int cnt = slab_ok ? min(64, TPs) : 0;
  const float inv_cnt = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = slab_ok && 16 * j + c < TPs;
      const uint2 q = p[r][j];
      v[j][0] = ok ? __uint_as_float(q.x << 16) : 0.f;
      v[j][1] = ok ? __uint_as_float(q.x & 0xffff0000u) : 0.f;
      v[j][2] = ok ? __uint_as_float(q.y << 16) : 0.f;
      v[j][3] = ok ? __uint_as_float(q.y & 0xffff0000u) : 0.f;
    }
    float mean[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[j][e];
      mean[e] = s;
    }
    row_sum_n<4>(mean);
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] *= inv_cnt;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = (slab_ok && 16 * j + c < TPs) ? v[j][e] - mean[e] : 0.f;
        s = fmaf(d, d, s);
      }
      m2[e] = s;
    }
    row_sum_n<4>(m2);
    if (c == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[16 * g + 4 * r + e] = make_float2(mean[e], m2[e]);
    }
  }
}

// Tile geometry shared by the prologue and the epilogue of a workgroup.
struct CvfTile {
  int fs0;        // first flat slab (slab index = image * tiles + tile)
  int nslabs;     // slabs of the launch
  int TH, W, PW;  // rows per slab, row length, row length of the LDS image
  int TPs;        // pixels per slab
  int jb;         // output rows [0, jb) belong to the first image of the tile, [jb, 3 TH) to the next one
  long pix0;      // first output pixel of the tile in the flat [N * H * W] pixel space
};

// Stores the wave's 4 x NT accumulator fragments (rounded to bf16) and, with part != null, emits the slab summaries and
// the in-launch merge.  `scratch` = the dead input tile.
__device__ __forceinline__ void cvf_epilogue(const f32x4 (&acc)[4][CVL_NT], const int (&gq)[CVL_NT], unsigned okm,
                                             const CvfTile& tl, bf16_t* y, float* part, char* scratch, const FinArgs& fin,
                                             int tiles, int HW) {
  constexpr int NT = CVL_NT, COUT = 256, CG = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int cg = wave % CG, pg = wave / CG;
  uint2 pk[4][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pk[r][i].x = pack_bf16x2(acc[r][i][0], acc[r][i][1]);
      pk[r][i].y = pack_bf16x2(acc[r][i][2], acc[r][i][3]);
    }
    if ((okm >> i) & 1u) {
      uint4* yp = reinterpret_cast<uint4*>(y + (tl.pix0 + gq[i]) * COUT + cg * 64 + 16 * g);
      yp[0] = make_uint4(pk[0][i].x, pk[0][i].y, pk[1][i].x, pk[1][i].y);
      yp[1] = make_uint4(pk[2][i].x, pk[2][i].y, pk[3][i].x, pk[3][i].y);
    }
  }
  if (part == nullptr) return;
  // scratch: [0, 6 KiB) slab summaries [3][CG][64] float2; [8, 24 KiB) the straddling slab's fragments of pixel group 1
  // [CG][4 r][2 fragments][64 lanes] uint2; [32 KiB, ..) flags and fin_merge's partial sums
  float2 (*const s_stat)[CG][64] = reinterpret_cast<float2 (*)[CG][64]>(scratch);
  uint2 (*const s_xch)[4][2][64] = reinterpret_cast<uint2 (*)[4][2][64]>(scratch + 8 * 1024);
  if (pg == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_xch[cg][r][0][lane] = pk[r][0]; s_xch[cg][r][1][lane] = pk[r][1]; }
  }
  lds_barrier();
  const bool ok0 = tl.fs0 + 0 < tl.nslabs, ok1 = tl.fs0 + 1 < tl.nslabs, ok2 = tl.fs0 + 2 < tl.nslabs;
  if (pg == 0) {
    uint2 q[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { q[r][0] = pk[r][0]; q[r][1] = pk[r][1]; q[r][2] = pk[r][2]; q[r][3] = pk[r][3]; }
    cvf_slab_stats(q, tl.TPs, ok0, &s_stat[0][cg][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) { q[r][0] = pk[r][4]; q[r][1] = pk[r][5]; q[r][2] = s_xch[cg][r][0][lane]; q[r][3] = s_xch[cg][r][1][lane]; }
    cvf_slab_stats(q, tl.TPs, ok1, &s_stat[1][cg][0]);
  } else {
    uint2 q[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { q[r][0] = pk[r][2]; q[r][1] = pk[r][3]; q[r][2] = pk[r][4]; q[r][3] = pk[r][5]; }
    cvf_slab_stats(q, tl.TPs, ok2, &s_stat[2][cg][0]);
  }
  lds_barrier();
  for (int idx = tid; idx < CVL_SLABS * COUT; idx += CVL_WAVES * 64) {
    const int s = idx / COUT, ch = idx - s * COUT;
    if (tl.fs0 + s >= tl.nslabs) continue;
    float cn = 0.f, mean = 0.f, m2 = 0.f;
    const float2 sv = s_stat[s][ch >> 6][ch & 63];
    merge_stats(cn, mean, m2, (float)min(64, tl.TPs), sv.x, sv.y);
    float* po = part + ((long)(tl.fs0 + s) * COUT + ch) * 2;
    if (fin.ss != nullptr) agent_store_f2(po, mean, m2);      // write-through: read by another workgroup
    else *reinterpret_cast<float2*>(po) = make_float2(mean, m2);
  }
  if (fin.ss == nullptr) return;
  dma_wait<0>();          // every wave: its summary stores have been written through
  lds_barrier();
  int* const s_flag = reinterpret_cast<int*>(scratch + 32 * 1024);
  if (tid == 0) {
    // one RELAXED ticket per slab (conv_fused.hpp cv3_epilogue: why relaxed, and what orders the hand-off instead)
#pragma unroll
    for (int s = 0; s < CVL_SLABS; ++s) {
      int last = 0;
      if (tl.fs0 + s < tl.nslabs) {
        const int n = (tl.fs0 + s) / tiles;
        last = agent_fetch_add(fin.arrive + n, 1) == tiles - 1;
        if (last) agent_store_int(fin.arrive + n, 0);
      }
      s_flag[s] = last;
    }
  }
  lds_barrier();
#pragma unroll 1
  for (int s = 0; s < CVL_SLABS; ++s) {
    if (s_flag[s] == 0) continue;
    const int n = (tl.fs0 + s) / tiles;
    fin_merge<COUT, CVL_WAVES * 64>(fin, part + (long)n * tiles * COUT * 2, n, tiles, tl.TPs, HW,
                                    reinterpret_cast<float*>(scratch + 32 * 1024) + 16);
  }
}

// TRACE (-DTAPIR_EXPERIMENTS builds, tools/kbench.py --what convflattrace): shader cycles per phase and wave into a.dbg_times
template <bool HAS_SC, bool DUAL, bool TRACE = false>
__global__ __launch_bounds__(CVL_WAVES * 64, 2) void conv_flat_kernel(Conv3Args a) {
  typedef bf16_t T;
  constexpr int CIN = 256, COUT = 256, NT = CVL_NT, WAVES = CVL_WAVES, THREADS = WAVES * 64;
  constexpr int CG = COUT / 64;
  constexpr int CB = CIN * 2, CPP = CIN / 8, SWZ = 15;
  constexpr int TAPS = 9, KPT = CIN / 32;
  // A fragments in flight per wave: CVL_RING (4 k-steps); B fragments in ONE buffer, refilled in place: the MFMAs of a
  // k-step run pixel-fragment-major (the four channel rows of fragment i, then fragment i + 1), so fragment i of the NEXT
  // k-step is read from LDS right behind its last MFMA and has the other five fragments' twenty MFMAs to arrive.  (r05's
  // order -- two B buffers, 12 A fragments -- needs 96 + 48 + 48 registers here and spilled inside the k loop; a ring of 8
  // left a lone workgroup waiting on L2 for half of its k loop: 61 us per workgroup against 35 for a third of the pixels
  // in conv_fused_kernel, profiles/r06_kbench_convflat_v1.txt.)  The STREAM is the one conv_fused_kernel reads
  // (tapir_conv_pack / tapir_conv_pack_dual: fragments in the order they are multiplied; the projection's k-steps padded to
  // whole turns of THAT kernel's ring = 9), so the 3x3 loop of a DUAL launch starts at ring phase KP % G.
  constexpr int RING = CVL_RING, G = RING / 4, UNR = 2 * G;
  static_assert((TAPS * KPT) % UNR == 0, "whole loop trips");
  static_assert(!DUAL || !HAS_SC, "the projection is fused into conv_0 (no shortcut)");
  constexpr int KP = DUAL ? conv3_proj_ksteps_c(CIN, 32) : 0;
  constexpr int PH = KP % G;                       // ring phase of the 3x3 loop's first k-step
  __shared__ uint4 s_tile[CVL_LDS_BYTES / 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int cg = wave % CG, pg = wave / CG;
  CvfTile tl;
  tl.nslabs = a.N * a.tiles;
  // consecutive workgroup ids go round the 8 XCDs: XCD x takes the x-th contiguous eighth of the flat tile list
  const int total = (tl.nslabs + CVL_SLABS - 1) / CVL_SLABS;
  const int per_xcd = (total + 7) >> 3;
  const int bid = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (bid >= total) return;
  const int H = a.H, W = a.W;
  tl.fs0 = bid * CVL_SLABS;
  tl.TH = a.TH; tl.W = W; tl.PW = W + 2; tl.TPs = a.TH * W;
  const int PW = tl.PW;
  const int FR = CVL_SLABS * a.TH;                 // output rows of this workgroup
  const int R0 = tl.fs0 * a.TH;                    // first flat row (image n, row y  <->  n * H + y)
  const int nA = R0 / H, yA0 = R0 - nA * H;
  tl.jb = min(H - yA0, FR);
  tl.pix0 = (long)R0 * W;
  const bool segB = tl.jb < FR && nA + 1 < a.N;    // a second image inside the tile
  char* const tile = reinterpret_cast<char*>(s_tile);
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int k) {
#ifndef TAPIR_HIPEMU
    if (TRACE) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
      if (k >= 0) tph[k] += t_ - tlast;
      tlast = t_;
    }
#endif
  };
  tick(-1);

  // ---- the wave's weight stream: the first ring of fragments is in flight during the staging
  const uint4* wp = a.wstream + ((long)cg * a.frags_per_cg) * 64 + lane;
  uint4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = *wp; wp += 64; }

  // ---- this lane's output pixel of each of the wave's fragments: LDS pixel of its tap (0, 0), flat output pixel
  int Pc[NT], gq[NT];
  unsigned okm = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int q = (pg * NT + i) * 16 + c;          // pixel slot of the tile: slab q / 64, slot q % 64 of it
    const int s = q >> 6, qq = q & 63;
    const bool ok = qq < tl.TPs && tl.fs0 + s < tl.nslabs;
    const int qc = ok ? qq : 0;
    const int rr = qc / W, xx = qc - rr * W;
    const int j = (ok ? s : 0) * a.TH + rr;        // output row of the tile
    Pc[i] = (j + (j >= tl.jb ? 1 : 0)) * PW + xx;  // rows of the second image sit one LDS row lower (the shared zero row)
    gq[i] = j * W + xx;
    okm |= ok ? (1u << i) : 0u;
  }
  f32x4 acc[4][NT];
  auto bf4 = [](unsigned p, unsigned q) {
    return f32x4{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u),
                 __uint_as_float(q << 16), __uint_as_float(q & 0xffff0000u)};
  };
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_SC) {   // the raw 32 bytes are parked in acc[0] / acc[1] and converted after the staging (conv_fused.hpp)
      const f32x4* sp = reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.shortcut) +
                                                       (tl.pix0 + gq[i]) * COUT + cg * 64 + 16 * g);
      acc[0][i] = sp[0];
      acc[1][i] = sp[1];
    }
  }

  // ---- stage relu(a x + b): one or two segments (image n, input rows y0 .. y0 + nrows - 1 -> LDS rows L0 ..);
  // rows outside the image are literal zeros (the convolution's padding, and the row two images share)
  {
    constexpr int PPS = THREADS / CPP;             // pixels per sweep (16)
    constexpr int U = 8;                           // loads in flight per thread
    const int chunk = tid % CPP, pl = tid / CPP;
    auto stage = [&](int n, int y0, int nrows, int L0) {
      f32x4 ssv[4];
      {
        const f32x4* sp = reinterpret_cast<const f32x4*>(a.ss + ((long)n * CIN + 8 * chunk) * 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) ssv[k] = sp[k];
      }
      const T* xin = reinterpret_cast<const T*>(a.x) + (long)n * H * W * CIN + 8 * chunk;
      const int HP = nrows * PW;
      const int PL0 = L0 * PW;
      const int dq = PPS / PW, dr = PPS - dq * PW;
      int hy = pl / PW, hx = pl - hy * PW;
      for (int P0 = pl; P0 < HP; P0 += U * PPS) {
        uint4 v[U];
        int off[U];                                // LDS byte offset, -1: past the segment; bit 30: outside the image
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int P = P0 + u * PPS;
          const int y = y0 + hy, x = hx - 1;
          const bool in = P < HP && y >= 0 && y < H && x >= 0 && x < W;
          const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
          v[u] = *reinterpret_cast<const uint4*>(xin + (yc * W + xc) * CIN);
          const int PL = PL0 + P;
          off[u] = P < HP ? ((PL * CB + ((chunk ^ (PL & SWZ)) << 4)) | (in ? 0 : (1 << 30))) : -1;
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
            const f32x2 xv = f32x2{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xffff0000u)};
            const f32x2 sa = f32x2{ssv[k >> 1][2 * (k & 1)], ssv[k >> 1][2 * (k & 1) + 1]};
            const f32x2 sb = f32x2{ssv[2 + (k >> 1)][2 * (k & 1)], ssv[2 + (k >> 1)][2 * (k & 1) + 1]};
            const f32x2 yv = __builtin_elementwise_fma(xv, sa, sb);
            r4[k] = relu_bf16x2(pack_bf16x2(yv.x, yv.y)) & m;
          }
          if (off[u] >= 0) *reinterpret_cast<uint4*>(tile + (off[u] & 0x3fffffff)) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
        }
      }
    };
    // first image: input rows yA0 - 1 .. yA0 + jb -> LDS rows 0 .. jb + 1 (its last row is the zero row when the image
    // ends inside the tile); second image: input rows 0 .. FR - jb -> LDS rows jb + 2 .. FR + 2
    stage(nA, yA0 - 1, tl.jb + 2, 0);
    if (segB) stage(nA + 1, 0, FR - tl.jb + 1, tl.jb + 2);
  }

  if (HAS_SC) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const uint4 s0 = __builtin_bit_cast(uint4, acc[0][i]), s1 = __builtin_bit_cast(uint4, acc[1][i]);
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool ok = (okm >> i) & 1u;
      acc[0][i] = ok ? bf4(s0.x, s0.y) : zero; acc[1][i] = ok ? bf4(s0.z, s0.w) : zero;
      acc[2][i] = ok ? bf4(s1.x, s1.y) : zero; acc[3][i] = ok ? bf4(s1.z, s1.w) : zero;
    }
  }
  tick(0);
  lds_barrier();
  tick(1);

  if constexpr (DUAL) {
    // ---- proj_conv: the tap that reads input pixel (y, x) of output pixel (y, x) -- tap (1, 1) (conv_fused.hpp DUAL)
    const int toffp = a.pad_y * PW + a.pad_x;
    auto read_p = [&](int ks, int i, uint4& f) {
      const int P = Pc[i] + toffp;
      f = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
    };
    uint4 fb[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) read_p(0, i, fb[i]);
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int ksn = kk + 1 < KPT ? kk + 1 : KPT - 1;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) MfmaStep<T>::run(ring[(kk % G) * 4 + r], fb[i], acc[r][i]);
        read_p(ksn, i, fb[i]);
        sched_fence();
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { ring[(kk % G) * 4 + r] = *wp; wp += 64; }
      sched_fence();
    }
    cvf_epilogue(acc, gq, okm, tl, reinterpret_cast<T*>(a.y_proj), nullptr, tile, FinArgs{}, a.tiles, H * W);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- 9 taps x 8 k-steps; fragment i of the next k-step is read behind fragment i's MFMAs, the A fragments of a k-step are
  // refilled behind its last MFMA
  auto read_b = [&](int tap, int ks, int i, uint4& f) {
    const int dy = (tap * 11) >> 5;                // tap / 3 for tap < 9
    const int toff = dy * PW + (tap - 3 * dy);
    const int P = Pc[i] + toff;
    f = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
  };
  {
    uint4 fb[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) read_b(0, 0, i, fb[i]);
    int tap = 0, ks = 0;
    for (int grp = 0; grp < TAPS * KPT / UNR; ++grp) {
#pragma unroll
      for (int kk = 0; kk < UNR; ++kk) {
        int ks1 = ks + 1, tap1 = tap;
        if (ks1 == KPT) { ks1 = 0; tap1 = tap + 1; }
        if (tap1 == TAPS) tap1 = 0;                // past the end: any valid address (not used)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) MfmaStep<T>::run(ring[((kk + PH) % G) * 4 + r], fb[i], acc[r][i]);
          read_b(tap1, ks1, i, fb[i]);
          sched_fence();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ring[((kk + PH) % G) * 4 + r] = *wp; wp += 64; }
        sched_fence();
        tap = tap1; ks = ks1;
      }
    }
  }
  tick(2);
  lds_barrier();   // every wave is done with the tile: the region is reused for the summaries
  tick(3);

  cvf_epilogue(acc, gq, okm, tl, reinterpret_cast<T*>(a.y), a.part, tile, a.fin, a.tiles, H * W);
  tick(4);
  if (TRACE && a.dbg_times != nullptr && lane == 0) {
    long long* o = a.dbg_times + ((long)bid * WAVES + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (long long)tph[k];
  }
}

// a: what conv_fused_impl builds for conv_fused_kernel's 4-wave form (TH, tiles of conv3_plan)
inline void launch_conv_flat(const Conv3Args& a, hipStream_t s) {
  const int total = (a.N * a.tiles + CVL_SLABS - 1) / CVL_SLABS;
  const dim3 grid((unsigned)(8 * ((total + 7) / 8))), block((unsigned)(CVL_WAVES * 64));
#ifdef TAPIR_EXPERIMENTS
  if (a.dbg_times != nullptr && a.y_proj == nullptr) {
    if (a.shortcut != nullptr) hipLaunchKernelGGL((conv_flat_kernel<true, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_flat_kernel<false, false, true>), grid, block, 0, s, a);
    return;
  }
#endif
  if (a.y_proj != nullptr) TAPIR_LAUNCH((conv_flat_kernel<false, true>), grid, block, s, a);
  else if (a.shortcut != nullptr) TAPIR_LAUNCH((conv_flat_kernel<true, false>), grid, block, s, a);
  else TAPIR_LAUNCH((conv_flat_kernel<false, false>), grid, block, s, a);
}

}  // namespace tapir
