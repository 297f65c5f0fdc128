// Stand-alone VICTIM of the co-tenant fault (profiles/r06_cotenant_fault.txt): packed-f32 VALU instructions of one wave
// next to dense MFMA traffic of ANOTHER wave on the same SIMD (tools/micro/cotenant_aggressors.hip, kind 3 / 4).  In the
// engine the fault shows in mix_kernel (mixer.hpp): the low half of `v_pk_fma_f32 d, w, x, d op_sel:[0,1,0]` is lost in lanes
// 48-63.  This kernel has no dependency on the engine: every thread runs each packed form below on its own operands, next
// to the same arithmetic in scalar v_fma_f32 / v_mul_f32 / v_add_f32 (exactly rounded: on a correct machine every count is 0
// whatever runs beside it), and counts mismatches per FORM and result half, plus a lane histogram.
//   form 0  v_pk_fma_f32 d, a, b, c                          (no modifiers)
//   form 1  v_pk_fma_f32 d, a, b, c op_sel_hi:[1,0,1]        (src1: low half for both results)
//   form 2  v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]           (src1: high half for both results)
//   form 3  two dependent: d = pk_fma(a, b, c) op_sel_hi:[1,0,1]; d = pk_fma(e, b, d) op_sel:[0,1,0]   (mix_kernel's pair)
//   form 4  v_pk_mul_f32 d, a, b
//   form 5  v_pk_add_f32 d, a, b
//   form 6  v_pk_fma_f32 d, a, b, d op_sel:[0,1,0] with d preloaded by v_mov (accumulate in place, no producer in flight)
//   form 7  v_fma_f32 x2 (scalar control through the same checking code)
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/pk_forms_victim.hip -o tools/micro/libpkforms.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int FORMS = 8;
struct FormsReport { unsigned long long bad[FORMS][2]; unsigned long long lanes[4]; };   // [form][half]; faults per quarter wave

__global__ __launch_bounds__(256) void pk_forms_kernel(const float* __restrict__ seed, int iters, FormsReport* rep) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  auto S = [&](int k) { return seed[(tid * 29 + k * 131) & 4095]; };
  f32x2 a = {S(0), S(1)}, b = {S(2), S(3)}, c = {S(4), S(5)}, e = {S(6), S(7)};
  unsigned bad[FORMS][2] = {};
  for (int it = 0; it < iters; ++it) {
    f32x2 d[FORMS], w[FORMS];
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d[0]) : "v"(a), "v"(b), "v"(c));
    w[0] = f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d[1]) : "v"(a), "v"(b), "v"(c));
    w[1] = f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[0], c[1])};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d[2]) : "v"(a), "v"(b), "v"(c));
    w[2] = f32x2{fmaf(a[0], b[1], c[0]), fmaf(a[1], b[1], c[1])};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %4, %2, %0 op_sel:[0,1,0]"
                 : "=&v"(d[3]) : "v"(a), "v"(b), "v"(c), "v"(e));
    w[3] = f32x2{fmaf(e[0], b[1], fmaf(a[0], b[0], c[0])), fmaf(e[1], b[1], fmaf(a[1], b[0], c[1]))};
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[4]) : "v"(a), "v"(b));
    w[4] = f32x2{a[0] * b[0], a[1] * b[1]};
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d[5]) : "v"(a), "v"(b));
    w[5] = f32x2{a[0] + b[0], a[1] + b[1]};
    d[6] = c;
    asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(d[6]) : "v"(a), "v"(b));
    w[6] = f32x2{fmaf(a[0], b[1], c[0]), fmaf(a[1], b[1], c[1])};
    asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(d[7][0]), "=&v"(d[7][1])
                 : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]));
    w[7] = f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#pragma unroll
    for (int f = 0; f < FORMS; ++f)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (__float_as_uint(d[f][h]) != __float_as_uint(w[f][h])) ++bad[f][h];
    // next operands: bounded recurrences
    const float n0 = b[1] * -0.8f + a[0] * 0.3f + 0.05f, n1 = a[1] * 0.7f - b[0] * 0.2f + 0.01f;
    a = f32x2{b[0], n1}; b = f32x2{n0, a[0]}; c = f32x2{c[1] * 0.5f + 0.1f, c[0] * -0.5f + 0.2f}; e = f32x2{e[1], e[0] * 0.9f + 0.03f};
  }
  unsigned tot = 0;
#pragma unroll
  for (int f = 0; f < FORMS; ++f)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (bad[f][h]) { atomicAdd(&rep->bad[f][h], (unsigned long long)bad[f][h]); tot += bad[f][h]; }
  if (tot) atomicAdd(&rep->lanes[(threadIdx.x & 63) >> 4], (unsigned long long)tot);
}

extern "C" int pk_forms_check(const float* seed_dev, int blocks, int iters, void* report_dev, void* stream) {
  hipLaunchKernelGGL(pk_forms_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed_dev, iters, (FormsReport*)report_dev);
  return (int)hipGetLastError();
}
