// Shared device/host helpers for the TAPIR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tapir {

typedef unsigned short bf16_t;  // storage type: raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;  // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 C/D fragment

constexpr int kHiresDim = 128;   // tapir_model.py:320
constexpr int kLowresDim = 256;  // tapir_model.py:321
constexpr int kFeatDim = kHiresDim + kLowresDim;  // 384
constexpr int kMixOut = 4 + kFeatDim;             // 388 = [dx,dy,docc,dexpd,dfeat]
constexpr int kHidden = 512;                      // mixer_hidden_dim
constexpr int kHidden4 = 2048;
constexpr int kPatch = 49;                        // 7x7 (patch_size=7)
constexpr int kMaxLevels = 3;                     // hires, lowres, pooled (pyramid_level<=1)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even, like v_cvt_pk_bf16_f32 (NaN not special-cased: inputs are finite)
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair (lo in bits 0..15), round-to-nearest-even: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#ifdef TAPIR_HIPEMU
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
#else
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
#endif
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void st2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ void st2(bf16_t* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = pack_bf16x2(a, b); }
};

// Loads through pointers the compiler cannot prove global (read from a device table: mixer_online.hpp) are FLAT loads, which
// count on the LDS counter as well: every LDS wait then waits for them.  These say "global" at the load.
typedef unsigned tapir_u32x4 __attribute__((ext_vector_type(4)));
typedef float tapir_f32x4 __attribute__((ext_vector_type(4)));
typedef float tapir_f32x2 __attribute__((ext_vector_type(2)));
#ifdef TAPIR_HIPEMU
__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ tapir_f32x4 ldg_f4(const float* p) { return *reinterpret_cast<const tapir_f32x4*>(p); }
__device__ __forceinline__ float2 ldg_f2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float ldg_f(const float* p) { return *p; }
#else
__device__ __forceinline__ uint4 ldg16(const void* p) {
  const tapir_u32x4 v = *(const __attribute__((address_space(1))) tapir_u32x4*)p;
  return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ tapir_f32x4 ldg_f4(const float* p) { return *(const __attribute__((address_space(1))) tapir_f32x4*)p; }
__device__ __forceinline__ float2 ldg_f2(const float* p) {
  const tapir_f32x2 v = *(const __attribute__((address_space(1))) tapir_f32x2*)p;
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float ldg_f(const float* p) { return *(const __attribute__((address_space(1))) float*)p; }
#endif

// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4).  The hardware writes lane
// l's 16 bytes to M0 + l * 16, M0 = the FIRST lane's `lds` argument: pass the wave's base address
// (wave-uniform, so that it stays in an SGPR); the global address is free per lane.  Completion is
// tracked by vmcnt.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// constant address space: a wave-uniform load through it is a scalar (s_load) load
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
__device__ __forceinline__ void glds16(const void* gptr, void* lds) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)gptr, (lds_ptr_t)(uintptr_t)lds, 16, 0, 0);
}

// Waits until at most N of this wave's vector-memory operations (LDS-DMA copies included) are
// outstanding.  Loads complete in order, so vmcnt(N) guarantees that every copy older than the
// youngest N has landed; outstanding stores only make the wait longer, never shorter.
template <int N> __device__ __forceinline__ void dma_wait() {
#ifndef TAPIR_HIPEMU
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// dma_wait<N> plus lgkmcnt(0): additionally, every LDS read this wave has issued has returned
// (used before a barrier after which another wave may overwrite what was read).
template <int N> __device__ __forceinline__ void dma_lds_wait() {
#ifndef TAPIR_HIPEMU
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#endif
}
// Value the optimiser cannot see through (blocks hoisting of what is derived from it).
__device__ __forceinline__ int opaque(int v) {
#ifndef TAPIR_HIPEMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}
// Marks a use of v at this point (the compiler places the wait for a pending load of v here).
__device__ __forceinline__ void consume(const f32x4& v) {
#ifndef TAPIR_HIPEMU
  asm volatile("" ::"v"(v));
#endif
}
// Arrival point of a batch of global loads: issued back to back, then passed through here one after the other, the
// compiler can neither sink a load into the (conditional) code that uses it nor wait for each before issuing the next.
__device__ __forceinline__ void pin(f32x4& v) {
#ifndef TAPIR_HIPEMU
  asm volatile("" : "+v"(v));
#endif
}
__device__ __forceinline__ void pin(float& v) {
#ifndef TAPIR_HIPEMU
  asm volatile("" : "+v"(v));
#endif
}
__device__ __forceinline__ void pin(uint4& v) {
#ifndef TAPIR_HIPEMU
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t t = __builtin_bit_cast(u32x4_t, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(uint4, t);
#endif
}
// "Arrival point" of an LDS read: the value is passed through an empty asm, so the compiler waits
// for it HERE (with LDS-DMA in flight hipcc only ever emits lgkmcnt(0)) and treats every later
// use as a plain register -- reads issued after this point stay in flight under those uses.
__device__ __forceinline__ void arrive(uint4& v) {
#ifndef TAPIR_HIPEMU
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t t = __builtin_bit_cast(u32x4_t, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(uint4, t);
#endif
}
// Workgroup barrier that orders LDS traffic only (lgkmcnt), leaving vector-memory operations --
// LDS-DMA copies, stores -- in flight.
__device__ __forceinline__ void lds_barrier() {
#ifdef TAPIR_HIPEMU
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#endif
}
// Orders the LDS traffic of ONE wave across its lanes: the hardware executes a wave in lockstep, so
// only the compiler needs to be told; the host emulator runs lanes as fibers and needs a rendezvous.
__device__ __forceinline__ void wave_sync() {
#ifdef TAPIR_HIPEMU
  (void)__shfl_xor(0, 1);
#else
  __builtin_amdgcn_wave_barrier();
#endif
}
// Compiler scheduling fence: no instruction is moved across it (bounds live ranges).
__device__ __forceinline__ void sched_fence() {
#ifndef TAPIR_HIPEMU
  __builtin_amdgcn_sched_barrier(0);
#endif
}
// Workgroup barrier WITHOUT the memory fence of __syncthreads(): the fence would drain vmcnt and
// with it every LDS-DMA copy in flight.  The caller orders what it needs with dma_wait<>; LDS
// reads are complete before the MFMAs that consume them, i.e. before the barrier.
__device__ __forceinline__ void block_barrier() {
#ifdef TAPIR_HIPEMU
  __syncthreads();
#else
  __builtin_amdgcn_s_barrier();
#endif
}

// Wave-wide (64 lanes) reductions on the DPP data path: four row-local steps (quad swaps, half-row
// and row mirrors), two row broadcasts, then a v_readlane of lane 63.  Seven VALU instructions
// with no LDS round trip -- __shfl_xor is a ds_bpermute per step (~60 cycles of dependent latency
// each; the LayerNorm statistics spent 3 us per 16 rows in them).  The result is wave-uniform.
#ifndef TAPIR_HIPEMU
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                    ROW_MASK, 0xf, false));
}
#endif
__device__ __forceinline__ float wave_sum(float v) {
#ifdef TAPIR_HIPEMU
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
#else
  v += dpp_f32<0xB1, 0xf>(0.f, v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xf>(0.f, v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xf>(0.f, v);   // row_half_mirror
  v += dpp_f32<0x140, 0xf>(0.f, v);   // row_mirror: every lane holds its row's sum
  v += dpp_f32<0x142, 0xa>(0.f, v);   // row_bcast:15 -> rows 1, 3
  v += dpp_f32<0x143, 0xc>(0.f, v);   // row_bcast:31 -> rows 2, 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}
// K independent wave sums, step by step side by side: a single DPP butterfly is a chain of seven
// dependent instructions (with their DPP / readlane wait states); K rows interleaved fill them.
template <int K>
__device__ __forceinline__ void wave_sum_n(float (&v)[K]) {
#ifdef TAPIR_HIPEMU
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += __shfl_xor(v[k], m);
#else
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0xB1, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x4E, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x141, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x140, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x142, 0xa>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x143, 0xc>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
#endif
}
// K independent sums over the 16 lanes of each DPP row (lanes 16 g .. 16 g + 15), every lane of
// the row receives its row's total: the four row-local steps of wave_sum_n.
template <int K>
__device__ __forceinline__ void row_sum_n(float (&v)[K]) {
#ifdef TAPIR_HIPEMU
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += __shfl_xor(v[k], m);
#else
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0xB1, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x4E, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x141, 0xf>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f32<0x140, 0xf>(0.f, v[k]);
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef TAPIR_HIPEMU
  v = fmaxf(v, __shfl_xor(v, 32));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 8));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 1));
  return v;
#else
  v = fmaxf(v, dpp_f32<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xa>(v, v));   // lanes not written keep v (old = v)
  v = fmaxf(v, dpp_f32<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}

// Raw hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp): exp2f() and operator/ expand to
// ~10-instruction sequences (denormal scaling, Newton steps, div_fixup) that made the GELU
// epilogues VALU-bound.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// e^x; relative error ~|x|*1e-7.
__device__ __forceinline__ float fast_exp(float x) { return fast_exp2(x * 1.4426950408889634f); }

// jax.nn.gelu(approximate=True) == F.gelu(approximate='tanh') (tapir_model.py:67,96):
// 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3); 0.5(1+tanh u) = 1/(1+e^{-2u})
//   = 1 / (1 + 2^(x (c1 + c3 x^2))),  c1 = -2 sqrt(2/pi) log2(e),  c3 = 0.044715 c1.
// 7 VALU instructions, two of them transcendental.  x -> -inf gives x * rcp(inf) = -0.
__device__ __forceinline__ float gelu_tanh(float x) {
  const float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const float c3 = c1 * 0.044715f;
  const float e = fast_exp2(x * fmaf(c3, x * x, c1));
  return x * fast_rcp(1.0f + e);
}

// Timing of ONE kernel launch by the dispatch itself: hipExtLaunchKernelGGL writes the kernel's own
// start / stop timestamps into two events (what rocprofv3 reports as the kernel's duration).
// hipEventRecord markers around a launch instead add ~3 us of marker latency to a 30 us kernel and
// sit in the stream as extra packets.  The profiling scope arms the timer, the next TAPIR_LAUNCH
// consumes it.
struct LaunchTimer { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
inline LaunchTimer& launch_timer() { static thread_local LaunchTimer t; return t; }

}  // namespace tapir

// Device-scope ("agent") relaxed accesses for hand-offs between workgroups inside one launch: the per-XCD L2s are
// not coherent with each other and a CU's L1 is never refreshed by another CU's stores, so the producer stores
// write-through (8-byte sc1 store), drains its vector-memory queue (dma_wait<0>) and only then takes a ticket; the
// consumer loads past L1 / the non-coherent L2 (sc1 load).  (MI355X guide, in-launch reduction recipe.)
namespace tapir {
__device__ __forceinline__ void agent_store_f2(float* p, float a, float b) {
  const unsigned long long bits = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
#ifdef TAPIR_HIPEMU
  __atomic_store_n(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_SEQ_CST);
#else
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float2 agent_load_f2(const float* p) {
#ifdef TAPIR_HIPEMU
  const unsigned long long bits = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_SEQ_CST);
#else
  const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
#endif
  return make_float2(__uint_as_float((unsigned)bits), __uint_as_float((unsigned)(bits >> 32)));
}
__device__ __forceinline__ int agent_fetch_add(int* p, int v) {
#ifdef TAPIR_HIPEMU
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
#else
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void agent_store_int(int* p, int v) {
#ifdef TAPIR_HIPEMU
  __atomic_store_n(p, v, __ATOMIC_SEQ_CST);
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
}  // namespace tapir

// occupancy the register allocator has to respect (waves per SIMD); nothing for the host emulator
#ifdef TAPIR_HIPEMU
#define TAPIR_WAVES_PER_EU(lo, hi)
#else
#define TAPIR_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#ifdef TAPIR_HIPEMU
#define TAPIR_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#else
#include <hip/hip_ext.h>
#define TAPIR_LAUNCH(kernel, grid, block, stream, ...)                                              \
  do {                                                                                               \
    tapir::LaunchTimer& lt_ = tapir::launch_timer();                                                 \
    if (lt_.start != nullptr && !lt_.used) {                                                         \
      lt_.used = true;                                                                               \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, lt_.start, lt_.stop, 0, __VA_ARGS__);    \
    } else {                                                                                         \
      hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                               \
    }                                                                                                \
  } while (0)
#endif
