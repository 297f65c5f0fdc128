// Structural probe for the GEMM main loop on gfx950: what each ingredient of a k-step costs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe && ./mfma_probe
// Every kernel runs `steps` k-steps of 24 MFMA 16x16x32 bf16 per wave (the 192x128 / 8-wave tile
// of gemm.hpp: 3x4 fragments, two half-steps), 256 workgroups x WAVES waves.
//   V0: MFMAs only, operands in registers
//   V1: + 14 ds_read_b128 per step (fragment reads, conflict-free pattern)
//   V2: + one s_barrier per step
//   V3: + 5 global_load_lds_dwordx4 per thread per step (40 KiB stage per workgroup, 4-stage ring)
//   V4: the LDS-DMA + wait + barrier alone (no MFMA, no ds_read): the L2 -> LDS fill rate
//   V5: V3 with the copies issued by waves 0-3 only (10 per thread): does issue cost block MFMAs?
//   V6: V4 with plain global_load_dwordx4 into registers + ds_write_b128 instead of the DMA
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const uint4* __restrict__ src, float* out, int steps) {
  constexpr int FM = 3, FN = 4;
  constexpr int STAGE = 320 * 8;   // uint4 chunks: 320 rows x 128 B
  __shared__ uint4 lds0[STAGE];
  __shared__ uint4 lds1[STAGE];
  __shared__ uint4 lds2[STAGE];
  __shared__ uint4 lds3[STAGE];
  uint4* const bufs[4] = {lds0, lds1, lds2, lds3};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int a_row = (wave >> 1) * 48 % 192 + fr, w_row = (wave & 1) * 64 + (fr >> 2) * 16 + (fr & 3);
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 fa[2][FM], fw[2][FN];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[k][i] = src[(tid * 7 + i + k * 3) & 1023];
#pragma unroll
    for (int j = 0; j < FN; ++j) fw[k][j] = src[(tid * 5 + j + k * 4) & 1023];
  }
  if (V >= 1) {   // defined LDS contents
    for (int s = 0; s < 4; ++s)
      for (int i = tid; i < STAGE; i += WAVES * 64) bufs[s][i] = src[i & 1023];
    __syncthreads();
  }
  const uint4* gsrc = src + (size_t)blockIdx.x * 4096;
  auto one = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    asm volatile("" ::: "memory");   // LDS contents count as changed: no hoisting of the fragment reads
    if (V == 3 || V == 4) {
#pragma unroll
      for (int s = 0; s < STAGE / (WAVES * 64); ++s)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc + ((tid + s * WAVES * 64) & 4095)),
                                         (lds_ptr_t)(uintptr_t)(bufs[(S + 3) % 4] + tid + s * WAVES * 64), 16, 0, 0);
    }
    if (V == 5 && wave < WAVES / 2) {
#pragma unroll
      for (int s = 0; s < 2 * STAGE / (WAVES * 64); ++s)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc + ((tid + s * WAVES * 32) & 4095)),
                                         (lds_ptr_t)(uintptr_t)(bufs[(S + 3) % 4] + tid + s * WAVES * 32), 16, 0, 0);
    }
    if (V == 6) {
      uint4 tmp[STAGE / (WAVES * 64)];
#pragma unroll
      for (int s = 0; s < STAGE / (WAVES * 64); ++s) tmp[s] = gsrc[(tid + s * WAVES * 64 + S * 64) & 4095];
#pragma unroll
      for (int s = 0; s < STAGE / (WAVES * 64); ++s) bufs[(S + 3) % 4][tid + s * WAVES * 64] = tmp[s];
    }
    if (V == 4 || V == 6) {
      if (V == 4) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (V >= 1) {
        const int c = (kk * 4 + fg) ^ (fr & 7);
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[kk][i] = bufs[S][(a_row + i * 16) * 8 + c];
#pragma unroll
        for (int j = 0; j < FN; ++j) fw[kk][j] = bufs[S][(192 + w_row + j * 4) * 8 + c];
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[kk][j]),
                                                             __builtin_bit_cast(bf16x8, fa[kk][i]), acc[i][j], 0, 0, 0);
    }
    if (V == 3) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    if (V == 5) { if (wave < WAVES / 2) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (V >= 2) __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < steps; t += 4) {
    one(std::integral_constant<int, 0>{});
    one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{});
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[tid] = s;
}

template <int V, int WAVES>
void run(const uint4* src, float* out, int steps, const char* name) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int grid : {256, 512}) {
    if (grid == 512 && WAVES == 8) continue;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((probe<V, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, out, steps);
    CHECK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<V, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, out, steps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    const double flops = (double)grid * WAVES * steps * 24 * 16384.0;
    printf("%-28s waves/WG=%d grid=%d steps=%d  %.1f us  %.0f TFLOP/s  (%.0f cycles/step/SIMD @2.4GHz)\n", name,
           WAVES, grid, steps, us, flops / (us * 1e-6) / 1e12, us * 1e-6 * 2.4e9 / steps);
  }
}

int main() {
  uint4* src; float* out;
  CHECK(hipMalloc(&src, (256 * 2 * 4096 + 4096) * sizeof(uint4)));
  CHECK(hipMalloc(&out, 4096 * sizeof(float)));
  std::vector<unsigned short> h((256 * 2 * 4096 + 4096) * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
  CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int steps = 512;
  run<0, 8>(src, out, steps, "V0 mfma only");
  run<1, 8>(src, out, steps, "V1 +ds_read");
  run<2, 8>(src, out, steps, "V2 +barrier");
  run<3, 8>(src, out, steps, "V3 +lds-dma");
  run<4, 8>(src, out, steps, "V4 lds-dma alone");
  run<5, 8>(src, out, steps, "V5 dma by 4 of 8 waves");
  run<6, 8>(src, out, steps, "V6 gload+ds_write alone");
  return 0;
}
