// What does a kernel boundary cost in front of / behind a launch shaped like mixer_fused_kernel (256 workgroups x 512 threads,
// 158 KB of LDS each, one per CU)?  The headline step shows 14-19 us of idle device on both sides of each of its four mixer
// launches (profiles/r06_step_timeline.txt) against 0-6 us between its small kernels.  Every kernel stamps wall_clock64 at its
// first and last instruction (min / max over workgroups); gap = first stamp of a kernel - last stamp of its predecessor.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/launch_gap_bench.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void stamp_begin(unsigned long long* t) { if (threadIdx.x == 0) atomicMin(t, wall_clock64()); }
__device__ __forceinline__ void stamp_end(unsigned long long* t) { __syncthreads(); if (threadIdx.x == 0) atomicMax(t + 1, wall_clock64()); }

__global__ __launch_bounds__(256) void small_kernel(unsigned long long* t, float* sink) {
  stamp_begin(t);
  sink[blockIdx.x * 256 + threadIdx.x] = threadIdx.x;
  stamp_end(t);
}
template <int LDS_BYTES, int SCRATCH, int SPIN>
__global__ __launch_bounds__(512) void big_kernel(unsigned long long* t, float* sink, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  stamp_begin(t);
  float* s = reinterpret_cast<float*>(smem);
  s[threadIdx.x] = threadIdx.x;
  s[LDS_BYTES / 4 - 1 - threadIdx.x] = 1.f;
  float acc = 0.f;
  if (SCRATCH) {                      // a small private array indexed at run time: scratch
    constexpr int NP = SCRATCH > 0 ? SCRATCH : 1;
    float priv[NP];
    for (int i = 0; i < NP; ++i) priv[i] = i * 0.5f + threadIdx.x;
    for (int i = 0; i < 4; ++i) acc += priv[(n * (i + 1) + threadIdx.x + (int)sink[i]) % NP];
  }
  __syncthreads();
  for (int i = 0; i < SPIN; ++i) acc += s[(threadIdx.x + i) & 1023] * 1.0001f;   // ~SPIN LDS round trips
  sink[blockIdx.x * 512 + threadIdx.x] = acc + s[LDS_BYTES / 4 - 1 - threadIdx.x];
  stamp_end(t);
}

template <int LDS_BYTES, int SCRATCH, int SPIN>
int run(const char* name, hipStream_t st, unsigned long long* d_t, float* sink, bool graph) {
  const int K = 7;   // small, small, BIG, small, small, BIG, small
  auto body = [&]() {
    for (int k = 0; k < K; ++k) {
      unsigned long long* t = d_t + 2 * k;
      if (k == 2 || k == 5) hipLaunchKernelGGL((big_kernel<LDS_BYTES, SCRATCH, SPIN>), dim3(256), dim3(512), LDS_BYTES, st, t, sink, k);
      else hipLaunchKernelGGL(small_kernel, dim3(256), dim3(256), 0, st, t, sink);
    }
  };
  CK(hipFuncSetAttribute((const void*)big_kernel<LDS_BYTES, SCRATCH, SPIN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  std::vector<unsigned long long> h(2 * K), init(2 * K);
  for (int k = 0; k < K; ++k) { init[2 * k] = ~0ull; init[2 * k + 1] = 0; }
  std::vector<double> gaps[K], durs[K];
  for (int rep = 0; rep < 30; ++rep) {
    CK(hipMemcpy(d_t, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    if (graph) CK(hipGraphLaunch(ge, st)); else body();
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost));
    if (rep < 5) continue;
    for (int k = 0; k < K; ++k) {
      durs[k].push_back((h[2 * k + 1] - h[2 * k]) / 100.0);
      if (k) gaps[k].push_back(((double)h[2 * k] - (double)h[2 * k - 1]) / 100.0);
    }
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-44s %s  ", name, graph ? "graph" : "eager");
  for (int k = 1; k < K; ++k) printf(" gap%d %5.2f", k, med(gaps[k]));
  printf("   | dur big %.2f small %.2f us\n", med(durs[2]), med(durs[3]));
  return 0;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned long long* d_t; float* sink;
  CK(hipMalloc(&d_t, 64 * 8));
  CK(hipMalloc(&sink, 256 * 512 * 4));
  printf("sequence: small small BIG small small BIG small; gap k = in front of kernel k (k = 2, 5: in front of BIG; 3, 6: behind BIG)\n");
  for (int graph = 0; graph < 2; ++graph) {
    if (run<16384, 0, 64>("BIG = 16 KB LDS", st, d_t, sink, graph)) return 1;
    if (run<65536, 0, 64>("BIG = 64 KB LDS", st, d_t, sink, graph)) return 1;
    if (run<81920, 0, 64>("BIG = 80 KB LDS", st, d_t, sink, graph)) return 1;
    if (run<131072, 0, 64>("BIG = 128 KB LDS", st, d_t, sink, graph)) return 1;
    if (run<161792, 0, 64>("BIG = 158 KB LDS (the mixer's)", st, d_t, sink, graph)) return 1;
    if (run<161792, 300, 64>("BIG = 158 KB LDS + scratch", st, d_t, sink, graph)) return 1;
    if (run<16384, 300, 64>("BIG = 16 KB LDS + scratch", st, d_t, sink, graph)) return 1;
    if (run<161792, 0, 20000>("BIG = 158 KB LDS, long (~0.5 ms)", st, d_t, sink, graph)) return 1;
    if (run<16384, 0, 20000>("BIG = 16 KB LDS, long (~0.5 ms)", st, d_t, sink, graph)) return 1;
  }
  return 0;
}
