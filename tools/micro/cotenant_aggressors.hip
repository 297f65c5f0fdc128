// Aggressor kernels for the co-tenant fault hunt (tools/probe_mix_fault.py --setting pair --aggr micro:<kind>).
//
// What is known (profiles/r06_cotenant_fault.txt): mix_kernel (mixer.hpp) running in process A loses the low half of one
// v_pk_fma_f32 result in lanes 48-63 -- but ONLY while process B runs gemm_small_kernel (gemm.hpp) on the same GPU; B
// running mix_kernel, the split-K GEMM pair, or rocBLAS matmuls never does it.  gemm_small_kernel is the only kernel of the
// library that hipcc gave ACCUMULATION registers (AGPRs: MFMA C/D in a[..], v_accvgpr_mov_b32 zeroing, ds_write_b128 straight
// from a[..]).  The kernels below isolate one ingredient each, with no dependency on the engine, so that the fault can be
// pinned on an instruction class (or on nothing but the register allocation) of the NEIGHBOURING wave:
//   0  valu      control: v_pk_add_f32 / v_fma_f32 on arch VGPRs only, no AGPR allocated
//   1  alloc     allocates 16 AGPRs (clobber list) and never touches them
//   2  accmov    v_accvgpr_write_b32 / v_accvgpr_mov_b32 / v_accvgpr_read_b32 traffic
//   3  mfma_a    v_mfma_f32_16x16x32_bf16 with C/D in AGPRs
//   4  mfma_v    the same MFMAs with C/D in arch VGPRs
//   5  ds_a      ds_write_b128 from AGPRs + ds_read_b128 into AGPRs
//   6  mimic     the shape of gemm_small_kernel: 16-byte global loads -> MFMA into AGPRs -> ds_write_b128 from AGPRs ->
//                barrier -> ds_read_b128 -> v_pk_add_f32 -> store
// Every kernel: 256 threads, `iters` trips; launch with a small grid again and again, as the engine does.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/cotenant_aggressors.hip -o tools/micro/libcotenant.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void aggr_valu(float* sink, int iters) {
  f32x2 p = {threadIdx.x * 0.001f, 0.5f}, q = {0.25f, 0.125f};
  float c = 0.75f;
  for (int i = 0; i < iters; ++i)
    asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(p), "+v"(c) : "v"(q));
  if (p[0] + c == 12345.f) *sink = p[1];
}

__global__ __launch_bounds__(256) void aggr_alloc(float* sink, int iters) {
  float a = threadIdx.x * 0.001f;
  for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\ts_nop 0" : "+v"(a) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  if (a == 12345.f) *sink = a;
}

__global__ __launch_bounds__(256) void aggr_accmov(float* sink, int iters) {
  float a = threadIdx.x * 0.001f;
  asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(a) : "a0");
  for (int i = 0; i < iters; ++i)
    asm volatile("v_accvgpr_mov_b32 a1, a0\n\tv_accvgpr_mov_b32 a2, a0\n\tv_accvgpr_mov_b32 a3, a0\n\tv_accvgpr_mov_b32 a7, a3\n\t"
                 "v_accvgpr_mov_b32 a11, a3\n\tv_accvgpr_mov_b32 a15, a3\n\tv_accvgpr_mov_b32 a6, a2\n\tv_accvgpr_mov_b32 a5, a1\n\t"
                 "v_accvgpr_mov_b32 a4, a0\n\tv_accvgpr_mov_b32 a10, a2\n\tv_accvgpr_mov_b32 a9, a1\n\tv_accvgpr_mov_b32 a8, a0\n\t"
                 "v_accvgpr_mov_b32 a14, a2\n\tv_accvgpr_mov_b32 a13, a1\n\tv_accvgpr_mov_b32 a12, a0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a15"
                 : "+v"(a) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  if (a == 12345.f) *sink = a;
}

__global__ __launch_bounds__(256) void aggr_mfma_a(float* sink, int iters) {
  f32x4 x = {threadIdx.x * 0.001f, 0.5f, 0.25f, 0.125f}, y = {0.1f, 0.2f, 0.3f, 0.4f};
  float r = 0.f;
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_mov_b32 a1, a0\n\tv_accvgpr_mov_b32 a2, a0\n\tv_accvgpr_mov_b32 a3, a0\n\t"
               "v_accvgpr_mov_b32 a4, a0\n\tv_accvgpr_mov_b32 a5, a0\n\tv_accvgpr_mov_b32 a6, a0\n\tv_accvgpr_mov_b32 a7, a0\n\t"
               "v_accvgpr_mov_b32 a8, a0\n\tv_accvgpr_mov_b32 a9, a0\n\tv_accvgpr_mov_b32 a10, a0\n\tv_accvgpr_mov_b32 a11, a0\n\t"
               "v_accvgpr_mov_b32 a12, a0\n\tv_accvgpr_mov_b32 a13, a0\n\tv_accvgpr_mov_b32 a14, a0\n\tv_accvgpr_mov_b32 a15, a0"
               ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  for (int i = 0; i < iters; ++i)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[12:15], %0, %1, a[12:15]\n\tv_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, a[8:11]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[4:7], %0, %1, a[4:7]\n\tv_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]"
                 :: "v"(x), "v"(y) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  asm volatile("s_nop 15\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, a0" : "=v"(r) :: "a0");
  if (r == 12345.f) *sink = r;
}

__global__ __launch_bounds__(256) void aggr_mfma_v(float* sink, int iters) {
  f32x4 x = {threadIdx.x * 0.001f, 0.5f, 0.25f, 0.125f}, y = {0.1f, 0.2f, 0.3f, 0.4f};
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y));
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) *sink = c0[0];
}

__global__ __launch_bounds__(256) void aggr_ds_a(float* sink, int iters) {
  __shared__ f32x4 s_part[4][4][64];
  const unsigned addr = (unsigned)(uintptr_t)&s_part[threadIdx.x >> 6][0][threadIdx.x & 63];
  float r = 0.f;
  asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_mov_b32 a1, a0\n\tv_accvgpr_mov_b32 a2, a0\n\tv_accvgpr_mov_b32 a3, a0"
               :: "v"(threadIdx.x * 0.5f) : "a0", "a1", "a2", "a3");
  for (int i = 0; i < iters; ++i)
    asm volatile("ds_write_b128 %0, a[0:3]\n\tds_write_b128 %0, a[0:3] offset:1024\n\tds_write_b128 %0, a[0:3] offset:2048\n\t"
                 "ds_write_b128 %0, a[0:3] offset:3072\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b128 a[4:7], %0 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 :: "v"(addr) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "memory");
  asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(r) :: "a4");
  if (r == 12345.f) *sink = r;
}

// gemm_small_kernel's shape: per wave a K slice, 2 x 2 fragments, partial tiles meet in LDS (written from the AGPRs)
__global__ __launch_bounds__(256) void aggr_mimic(const u32x4* __restrict__ A, const u32x4* __restrict__ W, float* C, int ksteps) {
  __shared__ f32x4 s_part[4][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* pa = A + ((blockIdx.x & 1) * 32 + (lane & 15)) * 80 + (lane >> 4);
  const u32x4* pw = W + ((blockIdx.x >> 1) * 32 + (lane & 15)) * 80 + (lane >> 4);
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_mov_b32 a1, a0\n\tv_accvgpr_mov_b32 a2, a0\n\tv_accvgpr_mov_b32 a3, a0\n\t"
               "v_accvgpr_mov_b32 a4, a0\n\tv_accvgpr_mov_b32 a5, a0\n\tv_accvgpr_mov_b32 a6, a0\n\tv_accvgpr_mov_b32 a7, a0\n\t"
               "v_accvgpr_mov_b32 a8, a0\n\tv_accvgpr_mov_b32 a9, a0\n\tv_accvgpr_mov_b32 a10, a0\n\tv_accvgpr_mov_b32 a11, a0\n\t"
               "v_accvgpr_mov_b32 a12, a0\n\tv_accvgpr_mov_b32 a13, a0\n\tv_accvgpr_mov_b32 a14, a0\n\tv_accvgpr_mov_b32 a15, a0"
               ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  for (int k = wave; k < ksteps; k += 4) {
    const u32x4 a0 = pa[4 * k], a1 = pa[4 * k + 16 * 80], w0 = pw[4 * k], w1 = pw[4 * k + 16 * 80];
    asm volatile("v_mfma_f32_16x16x32_bf16 a[12:15], %2, %0, a[12:15]\n\tv_mfma_f32_16x16x32_bf16 a[8:11], %2, %1, a[8:11]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[4:7], %3, %0, a[4:7]\n\tv_mfma_f32_16x16x32_bf16 a[0:3], %3, %1, a[0:3]"
                 :: "v"(a0), "v"(a1), "v"(w0), "v"(w1)
                 : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  }
  const unsigned addr = (unsigned)(uintptr_t)&s_part[wave][0][lane];
  asm volatile("s_nop 15\n\ts_nop 7\n\tds_write_b128 %0, a[12:15]\n\tds_write_b128 %0, a[8:11] offset:1024\n\t"
               "ds_write_b128 %0, a[4:7] offset:2048\n\tds_write_b128 %0, a[0:3] offset:3072\n\ts_waitcnt lgkmcnt(0)"
               :: "v"(addr) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "memory");
  __syncthreads();
  f32x4 v = s_part[0][wave][lane];
  for (int w = 1; w < 4; ++w) v += s_part[w][wave][lane];
  *reinterpret_cast<f32x4*>(C + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = v;
}

extern "C" int aggr_launch(int kind, int grid, int iters, void* buf, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  float* f = (float*)buf;                       // >= 4 MiB: A at +0, W at +1 MiB, C at +2 MiB (kind 6)
  switch (kind) {
    case 0: hipLaunchKernelGGL(aggr_valu, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 1: hipLaunchKernelGGL(aggr_alloc, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 2: hipLaunchKernelGGL(aggr_accmov, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 3: hipLaunchKernelGGL(aggr_mfma_a, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 4: hipLaunchKernelGGL(aggr_mfma_v, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 5: hipLaunchKernelGGL(aggr_ds_a, dim3(grid), dim3(256), 0, s, f, iters); break;
    case 6: hipLaunchKernelGGL(aggr_mimic, dim3(grid), dim3(256), 0, s, (const u32x4*)buf, (const u32x4*)((char*)buf + (1 << 20)),
                               (float*)((char*)buf + (2 << 20)), 20); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
