// Cost of a grid-wide barrier inside ONE launch on MI355X (8 XCDs, non-coherent L2s): what a persistent
// form of the online mixer (mix -> barrier -> up -> barrier -> down -> barrier per block) would pay
// 36 times per refinement iteration instead of 36 kernel boundaries.  Stand-alone:
//     hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
// Every spin is BOUNDED (a workgroup that never sees the generation gives up and flags it): a barrier that cannot
// complete -- fewer resident workgroups than the grid -- ends as an error count, not as a hung GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned kSpinLimit = 1u << 20;

struct Bar { unsigned* cnt; unsigned* gen; unsigned* xcnt; unsigned* fail; };

// flat: one agent-scope counter, one generation word
__device__ __forceinline__ bool barrier_flat(const Bar& b, unsigned n, unsigned& g) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    ++g;
    const unsigned prev = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == n * g - 1) {
      __hip_atomic_store(b.gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned it = 0;
      while (__hip_atomic_load(b.gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < g) {
        if (++it > kSpinLimit) { atomicAdd(b.fail, 1u); s_ok = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
  return s_ok != 0;   // false: gave up -- the caller leaves the kernel
}

// two levels: the workgroups of an XCD (blockIdx & 7) meet on their own counter (its own 128-byte line), the last
// of each XCD goes to the global one
__device__ __forceinline__ bool barrier_xcd(const Bar& b, unsigned n, unsigned& g) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    ++g;
    const unsigned x = blockIdx.x & 7u;
    const unsigned nx = (n + 7 - x) / 8;     // workgroups of this XCD
    bool last = false;
    const unsigned p = __hip_atomic_fetch_add(b.xcnt + 32 * x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (p == nx * g - 1) {
      const unsigned q = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned groups = n < 8 ? n : 8;
      if (q == groups * g - 1) { __hip_atomic_store(b.gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); last = true; }
    }
    if (!last) {
      unsigned it = 0;
      while (__hip_atomic_load(b.gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < g) {
        if (++it > kSpinLimit) { atomicAdd(b.fail, 1u); s_ok = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
  return s_ok != 0;   // false: gave up -- the caller leaves the kernel
}


// no read-modify-write at all: every workgroup STORES its generation into its own word, workgroup 0 polls the words
// (one per thread), then publishes the generation everyone else polls
__device__ __forceinline__ bool barrier_flags(const Bar& b, unsigned n, unsigned& g) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  ++g;
  if (blockIdx.x == 0) {
    for (unsigned w = threadIdx.x + 1; w < n; w += blockDim.x) {
      unsigned it = 0;
      while (__hip_atomic_load(b.xcnt + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < g) {
        if (++it > kSpinLimit) { atomicAdd(b.fail, 1u); s_ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(b.gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else if (threadIdx.x == 0) {
    __hip_atomic_store(b.xcnt + blockIdx.x, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned it = 0;
    while (__hip_atomic_load(b.gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < g) {
      if (++it > kSpinLimit) { atomicAdd(b.fail, 1u); s_ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return s_ok != 0;
}

// tuned: release fence + RELAXED arrival atomic; the last arriver publishes EIGHT copies of the generation (one per
// blockIdx & 7, 256 bytes apart: 32 pollers per word instead of 256); pollers use relaxed loads and fence once
__device__ __forceinline__ bool barrier_tuned(const Bar& b, unsigned n, unsigned& g) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    ++g;
    __atomic_thread_fence(__ATOMIC_RELEASE);   // (agent scope by default for device code)
    const unsigned prev = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == n * g - 1) {
      for (int x = 0; x < 8; ++x) __hip_atomic_store(b.xcnt + 64 * x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned it = 0;
      unsigned* mine = b.xcnt + 64 * (blockIdx.x & 7u);
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g) {
        if (++it > kSpinLimit) { atomicAdd(b.fail, 1u); s_ok = 0; break; }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
  return s_ok != 0;
}

template <int MODE>
__device__ __forceinline__ bool barrier_any(const Bar& b, unsigned n, unsigned& g) {
  if (MODE == 0) return barrier_flat(b, n, g);
  if (MODE == 1) return barrier_xcd(b, n, g);
  if (MODE == 2) return barrier_flags(b, n, g);
  return barrier_tuned(b, n, g);
}

template <int MODE>
__global__ __launch_bounds__(256) void bench_kernel(Bar b, int iters, float* data, int work) {
  unsigned g = 0;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (work) {   // a phase's worth of traffic between barriers: every workgroup writes a slab, reads its neighbour's
      const int n = gridDim.x, me = blockIdx.x, nb = (me + 1 + (i % (n - 1))) % n;
      for (int k = threadIdx.x; k < work; k += 256)
        __builtin_nontemporal_store((float)(i + k), data + (long)me * work + k);
      __threadfence();
      if (!barrier_any<MODE>(b, gridDim.x, g)) return;
      for (int k = threadIdx.x; k < work; k += 256)
        acc += __hip_atomic_load(data + (long)nb * work + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (float)(i + k);
    }
    if (!barrier_any<MODE>(b, gridDim.x, g)) return;
  }
  if (acc != 0.f) atomicAdd(b.fail, 1u << 16);   // a stale read shows up here
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, cus);
  unsigned* mem;
  CK(hipMalloc(&mem, 4096 * 4));
  float* data;
  const int work_max = 4096;
  CK(hipMalloc(&data, (size_t)1024 * work_max * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode)
    for (int work : {0, 1024, 4096})
      for (int grid : {cus / 2, cus, 2 * cus}) {
        CK(hipMemset(mem, 0, 4096 * 4));
        Bar b{mem, mem + 64, mem + 128, mem + 2048};
        for (int rep = 0; rep < 2; ++rep) {   // (first pass warms up)
          CK(hipMemset(mem, 0, 4096 * 4));
          CK(hipEventRecord(e0));
          if (mode == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(grid), dim3(256), 0, 0, b, iters, data, work);
          else if (mode == 1) hipLaunchKernelGGL(bench_kernel<1>, dim3(grid), dim3(256), 0, 0, b, iters, data, work);
          else if (mode == 2) hipLaunchKernelGGL(bench_kernel<2>, dim3(grid), dim3(256), 0, 0, b, iters, data, work);
          else hipLaunchKernelGGL(bench_kernel<3>, dim3(grid), dim3(256), 0, 0, b, iters, data, work);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned fail = 0;
        CK(hipMemcpy(&fail, mem + 2048, 4, hipMemcpyDeviceToHost));
        const int nb = work ? 2 : 1;
        printf("%s grid %4d  slab %5d floats: %.3f us per barrier%s (%d barriers, fail word 0x%x)\n",
               mode == 0 ? "flat   " : mode == 1 ? "per-XCD" : mode == 2 ? "flags  " : "tuned  ", grid, work, ms * 1e3 / (iters * nb), work ? " + half a phase" : "", iters * nb, fail);
      }
  return 0;
}
