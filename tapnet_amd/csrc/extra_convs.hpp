// BootsTAPIR's ExtraConvs (tapnet/models/tapir_model.py:159-186; torch twin tapnet/torch/nets.py:25-89):
// five blocks on the low-resolution map [N, h, w, 256]
//     y = LayerNorm(x) * scale + offset                 (per pixel, over the 256 channels, eps 1e-5)
//     r = gelu(Conv3x3(256 -> 1024)(y) + b)             (jax.nn.gelu = tanh form)
//     x = y + Conv3x3(1024 -> 256)(r) + b'              (the skip adds to the NORMALISED tensor: `x` is
//                                                        rebound to layernorm(x) at :176 before `x +=` :185)
// 2.3 TFLOP per 48-frame clip -- 2.6x the whole ResNet -- which round 2 still ran as MIOpen convolutions
// with torch LayerNorm / GELU / permutes around them.
//
// Two kernels:
//   ln_affine_kernel : the per-pixel LayerNorm (one 16-byte piece of a pixel per lane, two-pass
//                      statistics across the lanes of the pixel), writes y in the element type;
//   xconv_kernel     : 3x3 / stride 1 / SAME convolution as an implicit GEMM on the matrix cores in the
//                      layout of conv_fused.hpp -- A = weights, a per-wave packed stream of 1-KiB fragments
//                      through a register ring, never in LDS; B = the haloed input tile in LDS, 16-byte
//                      chunks XOR-swizzled by the pixel index, the nine taps nine constant offsets -- with
//                      what an ExtraConvs convolution needs on top:
//                        * many output-channel groups per tile (1024 = 16 groups of 64): a workgroup
//                          (4 waves, 64 pixels) takes ONE pass of 256 output channels, the 4 passes of a
//                          tile are neighbours in the launch (same XCD: the tile comes out of one L2);
//                        * input channels in chunks (1024 x 136 pixels does not fit LDS): the accumulators
//                          persist across the chunks, the tile is re-staged per chunk (two workgroups per
//                          CU: the staging of one runs under the matrix phase of the other);
//                        * epilogue: + bias, then gelu (first convolution) or + skip (second), rounded to
//                          the element type, 16-byte stores.
//   The weight stream of a wave is packed in exactly its consumption order [chunk][tap][k-step][row tile]
//   (tapir_xconv_pack), so the ring runs through chunk boundaries and barriers.
// Element type bf16 (16x16x32 MFMA) or f32 (parity build: exact-f32 16x16x4 MFMA, chunks half as wide).
#pragma once
#include "conv_fused.hpp"

namespace tapir {

constexpr float kLnAffineEps = 1e-5f;   // hk.LayerNorm / nn.LayerNorm default

struct LnAffineArgs {
  const void* x;        // [pixels, C] (T)
  const float* gamma;   // [C]
  const float* beta;    // [C]
  void* y;              // [pixels, C] (T)
  long pixels;
  int C;
};

// grid-stride over pixels; G = C / EPT lanes per pixel (a power of two <= 64)
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void ln_affine_kernel(LnAffineArgs a) {
  constexpr int EPT = Vec16<T>::EPT;
  const int G = a.C / EPT;
  const int PP = NORM_THREADS / G;
  const int cg = threadIdx.x % G, pl = threadIdx.x / G;
  float ga[EPT], be[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { ga[e] = a.gamma[cg * EPT + e]; be[e] = a.beta[cg * EPT + e]; }
  const float inv_c = 1.0f / (float)a.C;
  for (long p = (long)blockIdx.x * PP + pl; p < a.pixels; p += (long)gridDim.x * PP) {
    float v[EPT];
    Vec16<T>::load(reinterpret_cast<const T*>(a.x) + p * a.C + cg * EPT, v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) s += v[e];
    for (int m = G >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
    for (int m = G >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    const float rstd = 1.0f / sqrtf(q * inv_c + kLnAffineEps);
#pragma unroll
    for (int e = 0; e < EPT; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
    Vec16<T>::store(reinterpret_cast<T*>(a.y) + p * a.C + cg * EPT, v);
  }
}

constexpr int XC_WAVES = 4;
constexpr int XC_NT = 4;                       // pixel tiles (16 pixels) per wave: 64 pixels per workgroup (the small form)
constexpr int XC_NT_WIDE = 8;                  // the wide form: 128 pixels per workgroup -- every weight fragment a wave loads
                                               // feeds twice the pixels (see xconv_kernel)
constexpr int XC_RING = 12;                    // A fragments in flight per wave (3 k-steps)
constexpr int XC_LDS_BYTES = 72 * 1024;        // two workgroups per CU

struct XConvArgs {
  const void* x;          // [N, H, W, CIN] (T): the convolution's input as it is (already normalised / activated)
  const uint4* wstream;   // [COUT / 64][frags_per_cg][64 lanes] packed A fragments (tapir_xconv_pack)
  long frags_per_cg;
  const float* bias;      // [COUT]
  const void* skip;       // null, or [N, H, W, COUT] (T) added before rounding
  void* y;                // [N, H, W, COUT] (T)
  int N, H, W, cin, cout;
  int TH, tiles;          // output rows per tile, tiles per image
  int passes;             // COUT / 256
  int nt;                 // pixel tiles per wave: XC_NT or XC_NT_WIDE (xconv_plan)
};

// fragments of one 64-channel group's stream: chunks x 9 taps x k-steps per chunk x 4 row tiles (+ one ring
// of padding: prefetched, never used)
inline long xconv_frags_per_cg(int cin, int kstep) { return 9L * (cin / kstep) * 4 + XC_RING; }

// rows per tile / tiles per image / input channels per chunk / pixel tiles per wave for an [H, W, cin] map; false: not
// covered.  The wide form (128 pixels per workgroup) is taken whenever it gives a tile more rows than the small one and
// the tile fits LDS with one of the instantiated chunk widths: a workgroup then streams the SAME weight fragments for
// twice the pixels -- with 64 pixels the fragments of two co-resident workgroups (2 x 1.18 MB per 256-channel chunk) need
// as long on the L2 -> CU path as their products on the matrix pipe and the kernel sat at 0.46 of the MFMA peak -- and
// stages 6 rows per 4 rows of output instead of 4 per 2 (64-wide maps: 4 per 2 instead of 3 per 1).
// It needs enough work to pay: a wide workgroup runs ~1.6 x as long as a small one, so while ALL small workgroups of a
// clip's launches are resident at once (frames x small tiles x output-channel passes < 2 workgroups x 256 CUs) the small
// form finishes first --
// measured on BootsTAPIR clips of 4 .. 48 frames (profiles/r05_ab_xconv_wide.txt): small wins up to 24 frames, equal at
// 32, wide from there.  n_frames = frames of the WHOLE clip (0: unknown, taken as many), never of one launch or shard:
// the two forms add the input channels in different orders, and a clip must not change form with how it is cut up.
// force_nt (tests, A/B): 0 = choose, XC_NT / XC_NT_WIDE = that form only.
inline bool xconv_plan(int H, int W, int cin, int cout, int esize, int* rows, int* tiles, int* cch, int* nt = nullptr,
                       int force_nt = 0, int n_frames = 0) {
  if (H < 1 || W < 1 || cin % 256 || cout % 256 || cin < 256 || cout < 256) return false;
  auto fit = [&](int ntile, int* th_out, int* cc_out) {
    if (W > ntile * 16) return false;
    int th = (ntile * 16) / W;
    if (th > H) th = H;
    for (int cc = esize == 2 ? 256 : 128; cc >= (esize == 2 ? 128 : 64); cc >>= 1) {   // (the instantiated chunk widths)
      if ((long)(th + 2) * (W + 2) * cc * esize <= XC_LDS_BYTES) { *th_out = th; *cc_out = cc; return true; }
    }
    return false;
  };
  int th4 = 0, cc4 = 0, th8 = 0, cc8 = 0;
  const bool ok4 = force_nt != XC_NT_WIDE && fit(XC_NT, &th4, &cc4);
  const bool ok8 = force_nt != XC_NT && fit(XC_NT_WIDE, &th8, &cc8);
  bool wide = ok8 && (!ok4 || th8 > th4);
  if (wide && ok4 && force_nt == 0 && n_frames > 0 && (long)n_frames * ((H + th4 - 1) / th4) * (cout / 256) < 512) wide = false;
  if (!wide && !ok4) return false;
  const int th = wide ? th8 : th4;
  *rows = th; *tiles = (H + th - 1) / th; *cch = wide ? cc8 : cc4;
  if (nt) *nt = wide ? XC_NT_WIDE : XC_NT;
  return true;
}

template <typename T, int CCH, bool GELU, bool SKIP, int NT = XC_NT>
__global__ __launch_bounds__(XC_WAVES * 64, 2) void xconv_kernel(XConvArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EPC = CvT<T>::EPC, KSTEP = CvT<T>::KSTEP;
  constexpr int WAVES = XC_WAVES, THREADS = WAVES * 64;
  constexpr int RING = NT == 8 ? 8 : XC_RING;      // (a k-step of the wide form is 32 MFMAs: two k-steps of fragments in flight cover it)
  constexpr int NH = NT / 4;                       // groups of 4 pixel tiles: the B fragments are double-buffered per group
  static_assert(NT == 4 || NT == 8, "4 or 8 pixel tiles per wave");
  constexpr int CB = CCH * (int)sizeof(T);         // bytes per tile pixel (one chunk of input channels)
  constexpr int CPP = CCH / EPC;                   // 16-byte pieces per tile pixel
  constexpr int SWZ = (CPP < 16 ? CPP : 16) - 1;
  constexpr int KPT = CCH / KSTEP;                 // k-steps per tap and chunk
  constexpr int G = RING / 4, UNR = 2 * G;         // k-steps per ring turn / per loop trip
  static_assert((9 * KPT) % UNR == 0, "whole loop trips per chunk");
  __shared__ uint4 s_tile[XC_LDS_BYTES / 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // XCD x takes the x-th contiguous eighth of the (image, tile, pass) units: the passes of a tile and its
  // neighbour tiles (shared halo rows) read their input through one L2
  const int total = a.N * a.tiles * a.passes;
  const int per_xcd = (total + 7) >> 3;
  const int bid = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (bid >= total) return;
  const int pass = bid % a.passes;
  const int nt = bid / a.passes;
  const int n = nt / a.tiles, t = nt - n * a.tiles;
  const int H = a.H, W = a.W, CIN = a.cin, COUT = a.cout;
  const int PW = W + 2;
  const int r0 = t * a.TH;
  const int rows = min(a.TH, H - r0);
  const int HP = (rows + 2) * PW;                  // pixels of the input tile
  const int TP = rows * W;                         // output pixels of this tile
  const int y0 = r0 - 1, x0 = -1;                  // input coordinates of tile pixel (0, 0)
  char* const tile = reinterpret_cast<char*>(s_tile);
  const int cg = pass * WAVES + wave;              // this wave's group of 64 output channels

  const uint4* wp = a.wstream + ((long)cg * a.frags_per_cg) * 64 + lane;
  uint4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = *wp; wp += 64; }

  int Pc[NT], qpix[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int q = i * 16 + c;
    qpix[i] = q;
    const int qq = q < TP ? q : 0;
    const int yy = qq / W, xx = qq - yy * W;
    Pc[i] = yy * PW + xx;                          // tile pixel of tap (0, 0)
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto read_b = [&](int tap, int ks, int h, uint4 (&fb)[4]) {   // pixel tiles 4 h .. 4 h + 3
    const int dy = (tap * 11) >> 5;                // tap / 3 for tap < 9
    const int toff = dy * PW + (tap - 3 * dy);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int P = Pc[4 * h + i] + toff;
      fb[i] = *reinterpret_cast<const uint4*>(tile + P * CB + (((4 * ks + g) ^ (P & SWZ)) << 4));
    }
  };

  const int nchunks = CIN / CCH;
  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- stage chunk ch of the input tile (zero outside the image): every load unconditional with a
    // clamped address, the mask applied on use (conv_fused.hpp)
    {
      constexpr int PPS = THREADS / CPP;           // pixels per sweep
      constexpr int U = NT == 8 ? 9 : 18;          // loads in flight per thread (the wide form's accumulators leave room for 9)
      const int piece = tid % CPP, pl = tid / CPP;
      const T* xin = reinterpret_cast<const T*>(a.x) + (long)n * H * W * CIN + ch * CCH + EPC * piece;
      const int dq = PPS / PW, dr = PPS - dq * PW;
      int hy = pl / PW, hx = pl - hy * PW;
      for (int P0 = pl; P0 < HP; P0 += U * PPS) {
        uint4 v[U];
        int off[U];                                // LDS byte offset, -1: past the tile; bit 30: outside the image
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int P = P0 + u * PPS;
          const int y = y0 + hy, x = x0 + hx;
          const bool in = P < HP && y >= 0 && y < H && x >= 0 && x < W;
          const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
          v[u] = *reinterpret_cast<const uint4*>(xin + ((long)yc * W + xc) * CIN);
          off[u] = P < HP ? ((P * CB + ((piece ^ (P & SWZ)) << 4)) | (in ? 0 : (1 << 30))) : -1;
          hx += dr; hy += dq;
          if (hx >= PW) { hx -= PW; ++hy; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned m = (off[u] >> 30) ? 0u : 0xffffffffu;
          if (off[u] >= 0)
            *reinterpret_cast<uint4*>(tile + (off[u] & 0x3fffffff)) = make_uint4(v[u].x & m, v[u].y & m, v[u].z & m, v[u].w & m);
        }
      }
    }
    lds_barrier();

    // ---- 9 taps x KPT k-steps on this chunk; B fragments one group of 4 pixel tiles ahead (the next k-step's first
    // group behind the last group of this one), A fragments refilled right after their last MFMA (the ring keeps
    // running into the next chunk's fragments).  NT = 8: every A fragment multiplies two groups of B fragments.
    {
      uint4 fb0[4], fb1[4];
      read_b(0, 0, 0, fb0);
      int tap = 0, ks = 0;
      for (int grp = 0; grp < 9 * KPT / UNR; ++grp) {
#pragma unroll
        for (int kk = 0; kk < UNR; ++kk) {
          int ks1 = ks + 1, tap1 = tap;
          if (ks1 == KPT) { ks1 = 0; tap1 = tap + 1; }
          if (tap1 == 9) tap1 = 0;                 // past the end: any valid address (not used)
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const int step = kk * NH + h;          // (static under the unrolls: buffer parity)
            uint4 (&nxt)[4] = (step & 1) ? fb0 : fb1;
            uint4 (&cur)[4] = (step & 1) ? fb1 : fb0;
            if (h + 1 < NH) read_b(tap, ks, h + 1, nxt);
            else read_b(tap1, ks1, 0, nxt);
            sched_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const uint4 fa = ring[(kk % G) * 4 + r];
#pragma unroll
              for (int i = 0; i < 4; ++i) MfmaStep<T>::run(fa, cur[i], acc[r][4 * h + i]);
              if (h == NH - 1) {
                ring[(kk % G) * 4 + r] = *wp;
                wp += 64;
              }
              sched_fence();
            }
          }
          tap = tap1; ks = ks1;
        }
      }
    }
    lds_barrier();   // every wave is done with this chunk's tile before the next one overwrites it
  }

  // ---- epilogue: lane group g of pixel column c holds channels cg * 64 + 16 g + 4 r + e of pixel qpix[i]
  // (the host packing permutes the fragment rows, tapir_xconv_pack): + bias, gelu or + skip, round, store
  f32x4 b4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) b4[r] = *reinterpret_cast<const f32x4*>(a.bias + cg * 64 + 16 * g + 4 * r);
  const long img = ((long)n * H + r0) * W;         // first output pixel of the tile
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (qpix[i] >= TP) continue;
    const long o = (img + qpix[i]) * COUT + cg * 64 + 16 * g;
    f32x4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r][i] + b4[r];
    if (SKIP) {
      const T* sp = reinterpret_cast<const T*>(a.skip) + o;
      if (BF) {
        const uint4 s0 = reinterpret_cast<const uint4*>(sp)[0], s1 = reinterpret_cast<const uint4*>(sp)[1];
        const unsigned w8[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r][0] += __uint_as_float(w8[2 * r] << 16); v[r][1] += __uint_as_float(w8[2 * r] & 0xffff0000u);
          v[r][2] += __uint_as_float(w8[2 * r + 1] << 16); v[r][3] += __uint_as_float(w8[2 * r + 1] & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += reinterpret_cast<const f32x4*>(sp)[r];
      }
    }
    if (GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[r][e] = gelu_tanh(v[r][e]);
    }
    T* yq = reinterpret_cast<T*>(a.y) + o;
    if (BF) {
      uint4* yp = reinterpret_cast<uint4*>(yq);
      yp[0] = make_uint4(pack_bf16x2(v[0][0], v[0][1]), pack_bf16x2(v[0][2], v[0][3]),
                         pack_bf16x2(v[1][0], v[1][1]), pack_bf16x2(v[1][2], v[1][3]));
      yp[1] = make_uint4(pack_bf16x2(v[2][0], v[2][1]), pack_bf16x2(v[2][2], v[2][3]),
                         pack_bf16x2(v[3][0], v[3][1]), pack_bf16x2(v[3][2], v[3][3]));
    } else {
      f32x4* yp = reinterpret_cast<f32x4*>(yq);
#pragma unroll
      for (int r = 0; r < 4; ++r) yp[r] = v[r];
    }
  }
}

template <typename T>
inline bool launch_xconv(const XConvArgs& a, int cch, bool gelu, hipStream_t s) {
  const dim3 grid((unsigned)(8 * ((a.N * a.tiles * a.passes + 7) / 8))), block(XC_WAVES * 64);
  const bool skip = a.skip != nullptr;
  if (gelu && skip) return false;
#define TAPIR_XC_NT(CCH_, NT_)                                                                         \
  do {                                                                                                 \
    if (gelu) TAPIR_LAUNCH((xconv_kernel<T, CCH_, true, false, NT_>), grid, block, s, a);              \
    else if (skip) TAPIR_LAUNCH((xconv_kernel<T, CCH_, false, true, NT_>), grid, block, s, a);         \
    else TAPIR_LAUNCH((xconv_kernel<T, CCH_, false, false, NT_>), grid, block, s, a);                  \
  } while (0)
#define TAPIR_XC(CCH_)                                                                                 \
  do {                                                                                                 \
    if (a.nt == XC_NT_WIDE) TAPIR_XC_NT(CCH_, XC_NT_WIDE); else TAPIR_XC_NT(CCH_, XC_NT);              \
  } while (0)
  if (a.nt != XC_NT && a.nt != XC_NT_WIDE) return false;
  if (cch == 256) {
    if constexpr (sizeof(T) == 2) TAPIR_XC(256);
    else return false;                             // (f32: 256 channels x 136 pixels do not fit the tile)
  } else if (cch == 128) {
    TAPIR_XC(128);
  } else if (cch == 64) {
    if constexpr (sizeof(T) == 4) TAPIR_XC(64);    // (f32 on 64-wide maps; bf16 chunks are never this narrow)
    else return false;
  } else {
    return false;
  }
#undef TAPIR_XC
#undef TAPIR_XC_NT
  return true;
}

}  // namespace tapir
