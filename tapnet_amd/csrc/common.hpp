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

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4).  The hardware writes to
// M0 (the FIRST lane's LDS address) + lane * 16: `lds` must be wave-uniform base + lane*16;
// the global address is free per lane.  Completion is tracked by vmcnt; __syncthreads() drains it.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const void* gptr, void* lds) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)gptr, (lds_ptr_t)(uintptr_t)lds, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 32));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 8));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 1));
  return v;
}

// e^x through the hardware exp2 (v_exp_f32); relative error ~|x|*1e-7.
__device__ __forceinline__ float fast_exp(float x) { return exp2f(x * 1.4426950408889634f); }

// jax.nn.gelu(approximate=True) == F.gelu(approximate='tanh') (tapir_model.py:67,96):
// 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3); 0.5(1+tanh u) = 1/(1+e^{-2u}).
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k = 0.7978845608028654f;
  float u = k * (x + 0.044715f * x * x * x);
  return x / (1.0f + fast_exp(-2.0f * u));
}

}  // namespace tapir
