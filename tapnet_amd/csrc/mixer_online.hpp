// mixer_online.hpp -- the PIPs mixer of the ONLINE model (one frame per call: tapir_model.py:33-156 with use_causal_conv and
// the causal context of :1156-1203) as ONE persistent launch over all its blocks.
//
// With one frame the mixer is M = tracked points rows x 12 blocks of (token mixing over [context ; frame], LayerNorm, channel
// MLP 512 -> 2048 -> 512).  As separate launches (mixer.hpp mix_kernel + gemm.hpp mlp_small_kernel) a block is two dependent
// launches of 11.5 + 15.5 us that do ~1 us of arithmetic each: 1.3 ms of the 2.1 ms frame are 96 launch boundaries
// (profiles/r06_online_timeline_mlp.txt).  A grid-wide barrier is no cheaper than a launch boundary on this chip (5.5-13.6 us,
// profiles/r05_grid_barrier_bench.txt) -- but nothing in the mixer couples rows: a tile of 32 rows is a closed unit through
// every block, and what has to be spread over many CUs is only the 4 MB of weights per block.  So:
//
//   cluster  = 32 workgroups that own 32 rows for the whole launch (8 clusters = 256 rows = 256 workgroups, one per CU;
//              cluster = blockIdx % 8, member = blockIdx / 8: the dispatcher deals workgroups round-robin over the 8 XCDs
//              (observed: XCC_ID == blockIdx % 8), so a cluster's exchange and its counter stay on ONE XCD -- an expectation
//              about speed only (1.60 against 1.71 ms per frame with the members that read the same weight slice on one XCD
//              instead); every exchange below is agent-scope correct wherever the members run);
//   member k = row k of the tile in the row phases (its residual stream lives in registers from the first block to the last),
//              unit k = (hidden group k % 16 of 128 units, half k / 16 of the output columns) in the MLP phases;
//   block    = row phase (x = 16 partials + bias + residual; LayerNorm, causal temporal convolutions over [context ; frame],
//              new context, LayerNorm -> one operand row) | cluster barrier | MLP phase (gemm.hpp MlpSmallTile: the weights of
//              the unit were requested during the previous block) | cluster barrier.
//
// Hand-offs follow the MI355X guide's recipe for data that crosses workgroups inside a launch (Guideline 16, R1): the payload
// (an operand row; a 32 KB slab of partial sums) is stored WRITE-THROUGH (sc1), every storing wave drains its vector-memory
// queue, one lane adds 1 to the cluster's counter (relaxed, agent scope; one 64-byte line per cluster), the readers poll that
// word relaxed and then load the payload past L1 (sc1 loads; ACQ = true: one agent-scope acquire and plain loads instead).
// No release fence: it writes back the XCD's whole L2 (a first form with release / acquire fences around every barrier took
// 36 us per block, profiles/r06_ab_mixer_online.txt).  EVERY spin is bounded: a member that waits too long writes the error word, every poll reads it, all members
// leave, the rows are written as NaN (loud in the data) and tapir_online_sync_error() reports it.  The launch needs its 256
// workgroups resident at once (LDS 45 KB, <= 256 VGPRs: any idle MI355X); the host checks the CU count.
//
// Arithmetic: the row phase is mix_kernel's for T = 1 (same operations in the same order), the MLP phase IS mlp_small_kernel's
// tile code, the last row phase is layernorm_kernel's -- results are bit-identical to the two-launch form
// (tests/test_gpu_bf16_stages.py).  Not built for the host emulator (its workgroups run eight at a time).
#pragma once
#include "common.hpp"
#include "gemm.hpp"
#include "mixer.hpp"

namespace tapir {

struct OnlineBlockW {   // one block's parameters (device pointers)
  const float *ln1, *w1, *b1, *w2, *b2, *ln2;
  const void* Wup; const float* bup; const void* Wdn; const float* bdn;
};
constexpr int ONL_CLUSTERS = 8, ONL_MEMBERS = MLP_UNITS;       // 32 members: one (hidden group, column half) unit each
constexpr int ONL_SYNC_WORDS = ONL_CLUSTERS * 16 + 32;         // one 64-byte line per cluster + the error word's line + the exit counter's
constexpr unsigned ONL_SPIN_LIMIT = 1u << 21;                  // polls (~1 us each) before a member gives up
struct MixerOnlineArgs {
  const float* x_in;           // [M, 512] the input Linear's output
  void* xn;                    // [M, 512] operand-type exchange rows (LayerNorm-2 of every block; at the end: the final LayerNorm)
  float* part;                 // [MLP_PARTS][M, 512]
  const OnlineBlockW* blocks;  // [nb] device table
  const float* lnF;
  const float *ctx1_in, *ctx2_in;   // [nb][M, 2, 512] / [nb][M, 2, 2048] or null (zeros)
  float *ctx1_out, *ctx2_out;       // same shapes or null
  unsigned* sync;              // [ONL_SYNC_WORDS], zero at launch: zeroed when allocated, and by the LAST workgroup of every launch to leave
  int M, nb;
  int by_xcd;                  // 1 (default): cluster = blockIdx % 8 (a cluster on ONE XCD under round-robin dispatch); 0 (tests): cluster = blockIdx / 32 -- the same bits, slower
  long long* dbg_times;        // null, or [256][nb][8] wall-clock stamps of lane 0 (tools/probe_online_mixer.py)
  unsigned spin_limit;         // polls before a member gives up (ONL_SPIN_LIMIT)
  int drop_member;             // (tests) 1: member 0 of cluster 0 leaves before its first arrival: the others must time out, not hang
};

#ifndef TAPIR_HIPEMU
typedef unsigned onl_u32x2 __attribute__((ext_vector_type(2)));
// 32 arrivals per barrier on the cluster's counter; false = gave up (the error word is set).  Every wave has stored its payload
// write-through; it drains its queue, the workgroup meets, one lane arrives and polls.
template <bool ACQ>
__device__ __forceinline__ bool onl_cluster_barrier(unsigned* ctr, unsigned* err, unsigned target, int* s_flag, unsigned limit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(err, 0x80000000u | target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// (Pointers read from the device table are generic to the compiler; every load through them says "global" -- common.hpp ldg*:
// as FLAT loads they count on the LDS counter too, and each LDS wait of the row phase waited for the 128 KB of weights: 8 us per block.)
// mixer.hpp parts_sum2v on slabs other workgroups of this launch stored: 8-byte loads past L1 (sc1) -- requested, then (behind
// whatever else the caller requests meanwhile) added in the same order: ((((p_0 + p_1) + ...) + p_15) + bias) + residual
__device__ __forceinline__ void onl_parts_request(onl_u32x2 (&w)[MLP_PARTS], __amdgpu_buffer_rsrc_t parts, int rows, int r, int col) {
#pragma unroll
  for (int p = 0; p < MLP_PARTS; ++p) w[p] = __builtin_amdgcn_raw_buffer_load_b64(parts, ((p * rows + r) * kHidden + col) * 4, 0, 16);
}
__device__ __forceinline__ float2 onl_parts_finish(const onl_u32x2 (&w)[MLP_PARTS], const float* bias, float2 x, int col) {
  const float2 b = ldg_f2(bias + col);
  float2 v = make_float2(__uint_as_float(w[0].x), __uint_as_float(w[0].y));
#pragma unroll
  for (int p = 1; p < MLP_PARTS; ++p) { v.x += __uint_as_float(w[p].x); v.y += __uint_as_float(w[p].y); }
  return make_float2((v.x + b.x) + x.x, (v.y + b.y) + x.y);
}

template <typename TA, bool ACQ>
__global__ __launch_bounds__(256) void mixer_online_kernel(MixerOnlineArgs a) {
  using Tile = MlpSmallTile<TA>;
  __shared__ f32x4 s_part[4][Tile::NF1][64];
  __shared__ __attribute__((aligned(16))) TA s_hid[32 * Tile::LDH];
  __shared__ __attribute__((aligned(16))) float s_x[2][kHidden];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = a.by_xcd ? blockIdx.x % ONL_CLUSTERS : blockIdx.x / ONL_MEMBERS;
  const int k = a.by_xcd ? blockIdx.x / ONL_CLUSTERS : blockIdx.x % ONL_MEMBERS;
  const int m0 = 32 * cl;
  // The counters are zero at launch because the last workgroup to leave the PREVIOUS launch zeroed them (and the allocation was
  // zeroed once).  Not a memset node in front of the launch: replayed from a captured hipGraph, hipMemsetAsync of these 576 bytes
  // filled them with a 16-byte pattern of stale host data every other replay (profiles/r06_graph_memset_hazard.txt), and it
  // cost a dependent dispatch (~5 us) per launch.  The error word is NOT cleared here: a launch that finds it set leaves at once
  // (NaN rows) until the host has read it (tapir_online_sync_error reads and clears everything).
  auto leave = [&]() {
    if (tid == 0) {
      unsigned* done = a.sync + 16 * ONL_CLUSTERS + 16;
      if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        for (int q = 0; q < ONL_CLUSTERS; ++q) __hip_atomic_store(a.sync + 16 * q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (m0 >= a.M) { leave(); return; }           // the whole cluster has no rows (uniform over its members)
  if (a.drop_member && cl == 0 && k == 0) return;   // (test: never arrives, never leaves)
  const int r = m0 + k;
  const bool has_row = r < a.M;
  unsigned* ctr = a.sync + 16 * cl;
  unsigned* err = a.sync + 16 * ONL_CLUSTERS;
  unsigned arrivals = 0;
  const int c0 = 2 * tid;
  float2 xres = make_float2(0.f, 0.f);          // this thread's two channels of the row's residual stream
  Tile t;
  t.init();
  bool ok = true;
  // the two exchange buffers as buffer resources (wave-uniform: kernel arguments only)
  const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, MLP_PARTS * a.M * kHidden * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_xn = __builtin_amdgcn_make_buffer_rsrc(a.xn, 0, a.M * kHidden * (int)sizeof(TA), 0x00020000);

  auto stamp = [&](int i, int j) {
    if (a.dbg_times != nullptr && tid == 0) a.dbg_times[((long)blockIdx.x * a.nb + i) * 8 + j] = wall_clock64();
  };
  // A block's row-phase operands -- parameters (2 channels x 4 multipliers x 3 taps, twice), causal context (LN1(x) and the GELU
  // outputs of the two previous frames; zeros without one) -- and the MLP phase's weights: none of them depends on anything this
  // launch computes.  They are requested during the PREVIOUS block's MLP phase, each into registers that phase has just finished
  // with (W_up behind its first product, W_dn and the row operands behind its second), in front of the slab stores: the drain
  // those stores need anyway covers them, and behind the barrier the 16 partial sums are alone in the wave's load queue (loads
  // return in order: requested in front of them, 128 KB of weights cost 7 us per block).
  float w1[2][4][3], b1[2][4], w2[2][4][3], b2[2][4], sc1[2];
  float c1[2][2], g0[2][2][4];      // [frame][channel]([multiplier])
  float2 s2 = make_float2(0.f, 0.f);
  const bool has1 = a.ctx1_in != nullptr, has2 = a.ctx2_in != nullptr;
  const int rr = min(r, a.M - 1);    // (a member without a row requests the tile's last row and drops it: no branch in the run)
  auto request = [&](int ib, const OnlineBlockW& b) {
    {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        sc1[ch] = ldg_f(b.ln1 + c0 + ch);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int o = 4 * (c0 + ch) + m;
          b1[ch][m] = ldg_f(b.b1 + o);
          b2[ch][m] = ldg_f(b.b2 + o);
#pragma unroll
          for (int q = 0; q < 3; ++q) { w1[ch][m][q] = ldg_f(b.w1 + o * 3 + q); w2[ch][m][q] = ldg_f(b.w2 + o * 3 + q); }
        }
      }
      s2 = ldg_f2(b.ln2 + c0);
      // (no branch: without a context the loads read the input row and the values are dropped)
      const float* p1 = has1 ? a.ctx1_in + (((long)ib * a.M + rr) * 2) * kHidden + c0 : a.x_in + (long)rr * kHidden + c0;
      const float* p2 = has2 ? a.ctx2_in + (((long)ib * a.M + rr) * 2) * kHidden4 + 4 * c0 : a.x_in + (long)rr * kHidden;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float2 v = ldg_f2(p1 + (has1 ? j * kHidden : 0));
        c1[j][0] = has1 ? v.x : 0.f; c1[j][1] = has1 ? v.y : 0.f;
        const tapir_f32x4 v0 = ldg_f4(p2 + (has2 ? j * kHidden4 : 0));
        const tapir_f32x4 v1 = ldg_f4(p2 + (has2 ? j * kHidden4 : 0) + 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) { g0[j][0][m] = has2 ? v0[m] : 0.f; g0[j][1][m] = has2 ? v1[m] : 0.f; }
      }
    }
  };
  const int hg = k % MLP_PARTS, h0 = hg * MLP_HS, col0 = (k / MLP_PARTS) * (512 / MLP_CG);   // (both halves of a hidden group on one XCD)
  if (a.dbg_times != nullptr && tid == 0) {      // where this workgroup runs: XCC_ID (hwreg 20), HW_ID (hwreg 4)
    a.dbg_times[((long)blockIdx.x * a.nb) * 8 + 6] = __builtin_amdgcn_s_getreg(20 | (3 << 11));
    a.dbg_times[((long)blockIdx.x * a.nb) * 8 + 7] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  }
  OnlineBlockW bw = a.blocks[0];
  float2 x = ldg_f2(a.x_in + (long)rr * kHidden + c0);
  request(0, bw);
  t.load_up(bw.Wup, h0);
  t.load_dn(bw.Wdn, h0, col0);
  for (int i = 0; i < a.nb && ok; ++i) {
    stamp(i, 0);
    // the next block's table entry (scalar loads): under this block, not in front of the next
    const OnlineBlockW bw_next = a.blocks[min(i + 1, a.nb - 1)];
    if (has_row) {
      // ---- row phase (mix_kernel for T = 1, causal)
      *reinterpret_cast<float2*>(&s_x[0][c0]) = x;
      lds_barrier();
      stamp(i, 1);
      float mean, rstd;
      {
        const float4 u = *reinterpret_cast<const float4*>(&s_x[0][lane * 4]);
        const float4 v = *reinterpret_cast<const float4*>(&s_x[0][256 + lane * 4]);
        wave_row_stats(u, v, mean, rstd);      // (every wave computes the row's statistics: no second meeting)
      }
      float xn1[2], g[2][4], y[2];
      xn1[0] = (x.x - mean) * rstd * sc1[0];
      xn1[1] = (x.y - mean) * rstd * sc1[1];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float u = b1[ch][m];
          u = fmaf(w1[ch][m][0], c1[0][ch], u);
          u = fmaf(w1[ch][m][1], c1[1][ch], u);
          u = fmaf(w1[ch][m][2], xn1[ch], u);
          g[ch][m] = gelu_tanh(u);
        }
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float v = b2[ch][m];
          v = fmaf(w2[ch][m][0], g0[0][ch][m], v);
          v = fmaf(w2[ch][m][1], g0[1][ch][m], v);
          v = fmaf(w2[ch][m][2], g[ch][m], v);
          acc += v;
        }
        y[ch] = acc;
      }
      xres = make_float2(x.x + y[0], x.y + y[1]);
      *reinterpret_cast<float2*>(&s_x[1][c0]) = xres;
      // new causal context: the last two frames of [context ; frame] (tapir_model.py:58,73)
      if (a.ctx1_out != nullptr) {
        float* p1 = a.ctx1_out + (((long)i * a.M + r) * 2) * kHidden + c0;
        *reinterpret_cast<float2*>(p1) = make_float2(c1[1][0], c1[1][1]);
        *reinterpret_cast<float2*>(p1 + kHidden) = make_float2(xn1[0], xn1[1]);
        float* p2 = a.ctx2_out + (((long)i * a.M + r) * 2) * kHidden4 + 4 * c0;
        *reinterpret_cast<float4*>(p2) = make_float4(g0[1][0][0], g0[1][0][1], g0[1][0][2], g0[1][0][3]);
        *reinterpret_cast<float4*>(p2 + 4) = make_float4(g0[1][1][0], g0[1][1][1], g0[1][1][2], g0[1][1][3]);
        *reinterpret_cast<float4*>(p2 + kHidden4) = make_float4(g[0][0], g[0][1], g[0][2], g[0][3]);
        *reinterpret_cast<float4*>(p2 + kHidden4 + 4) = make_float4(g[1][0], g[1][1], g[1][2], g[1][3]);
      }
      lds_barrier();
      {
        const float4 u = *reinterpret_cast<const float4*>(&s_x[1][lane * 4]);
        const float4 v = *reinterpret_cast<const float4*>(&s_x[1][256 + lane * 4]);
        wave_row_stats(u, v, mean, rstd);
      }
      // the operand row, write-through: through LDS (the MLP phase's hidden tile is free now) so that one wave stores it 16 bytes per lane
      TA* s_row = s_hid;
      Elem<TA>::st2(s_row + c0, (xres.x - mean) * rstd * s2.x, (xres.y - mean) * rstd * s2.y);
      lds_barrier();
      if (tid < (int)(kHidden * sizeof(TA) / 16)) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(s_row) + 16 * tid);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_xn, r * kHidden * (int)sizeof(TA) + 16 * tid, 0, 16);
      }
    }
    stamp(i, 2);
    ok = onl_cluster_barrier<ACQ>(ctr, err, (arrivals += ONL_MEMBERS), &s_flag, a.spin_limit);
    if (!ok) break;
    stamp(i, 3);
    // ---- MLP phase: this member's unit (128 hidden units, half of the output columns) over the cluster's 32 rows
    if (ACQ) t.load_rows(a.xn, m0, a.M); else t.load_rows_shared(rs_xn, m0, a.M);
    t.load_bias(bw.bup, h0);
    request(min(i + 1, a.nb - 1), bw_next);         // the next block's row operands (its registers are free since barrier 1); with the
                                                    // weights below a wave would have more than the 63 requests in flight vmcnt can count
    t.phase1(s_part);
    t.load_up(bw_next.Wup, h0);                     // the next block's operands (after the last block: re-read and dropped)
    t.mid(s_part, s_hid);
    f32x4 acc[Tile::NJ2][2];
    t.phase2(s_hid, acc);
    t.load_dn(bw_next.Wdn, h0, col0);
    t.template store<true>(acc, a.part, m0, hg, col0, a.M, rs_part);
    stamp(i, 4);
    ok = onl_cluster_barrier<ACQ>(ctr, err, (arrivals += ONL_MEMBERS), &s_flag, a.spin_limit);
    stamp(i, 5);
    if (!ok) break;
    // ---- the next row phase's input (after the last block: the final LayerNorm's): the 16 partial sums
    onl_u32x2 pw[MLP_PARTS];
    onl_parts_request(pw, rs_part, a.M, rr, c0);
    x = onl_parts_finish(pw, bw.bdn, xres, c0);
    bw = bw_next;
  }
  leave();       // (past this workgroup's last poll)
  if (!has_row) return;
  TA* o = reinterpret_cast<TA*>(a.xn) + (long)r * kHidden;
  if (!ok) {     // a member gave up: loud in the data
    Elem<TA>::st2(o + c0, __builtin_nanf(""), __builtin_nanf(""));
    return;
  }
  // ---- the final LayerNorm (layernorm_kernel's arithmetic: a lane holds 8 consecutive channels)
  {
    lds_barrier();
    *reinterpret_cast<float2*>(&s_x[0][c0]) = x;
    lds_barrier();
    if (wave == 0) {
      float e[8];
      const float4 u = *reinterpret_cast<const float4*>(&s_x[0][lane * 8]);
      const float4 v = *reinterpret_cast<const float4*>(&s_x[0][lane * 8 + 4]);
      e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = v.x; e[5] = v.y; e[6] = v.z; e[7] = v.w;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += e[q];
      const float mean = wave_sum(s) * (1.0f / kHidden);
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { e[q] -= mean; qq += e[q] * e[q]; }
      const float rs = 1.0f / sqrtf(wave_sum(qq) * (1.0f / kHidden) + kLnEps);
#pragma unroll
      for (int q = 0; q < 8; ++q) Elem<TA>::st(o + lane * 8 + q, e[q] * rs * a.lnF[lane * 8 + q]);
    }
  }
}

template <typename TA>
inline void launch_mixer_online(const MixerOnlineArgs& a, hipStream_t stream, bool acq = false) {
  if (acq) TAPIR_LAUNCH((mixer_online_kernel<TA, true>), dim3(ONL_CLUSTERS * ONL_MEMBERS), dim3(256), stream, a);
  else TAPIR_LAUNCH((mixer_online_kernel<TA, false>), dim3(ONL_CLUSTERS * ONL_MEMBERS), dim3(256), stream, a);
}
#endif  // TAPIR_HIPEMU

inline bool mixer_online_supported(int N, int T, bool causal) {
#ifdef TAPIR_HIPEMU
  return false;
#else
  return causal && T == 1 && N >= 1 && N <= 32 * ONL_CLUSTERS;
#endif
}

}  // namespace tapir
