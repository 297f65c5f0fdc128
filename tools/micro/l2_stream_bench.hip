// What the L2 -> CU path delivers when EVERY workgroup streams the SAME weight fragments, the access pattern of the
// track-resident mixer (mixer_fused.hpp: 8 waves per workgroup, each wave walks its own region of one packed stream
// through an 8-deep ring of 1-KiB loads; all 256 workgroups walk the same stream).  Stand-alone:
//     hipcc --offload-arch=gfx950 -O3 tools/micro/l2_stream_bench.hip -o /tmp/l2s && /tmp/l2s
// Prints bytes per clock and CU for: the stream size (fits one L2 / the 50-MB mixer stream), workgroups in lock step or
// each starting at its own offset, ring depth, waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { const uint4* p; long frags_per_wave; int passes; int rotate; unsigned* sink; };

template <int RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long fpw = a.frags_per_wave;
  const uint4* base = a.p + (long)wave * fpw * 64 + lane;
  // rotate: workgroup b starts b / gridDim of the way into the stream (and wraps): the workgroups are NOT in lock step
  long start = a.rotate ? (fpw * (long)blockIdx.x / (long)gridDim.x) : 0;
  if (a.rotate == 2) start = (fpw * (long)(blockIdx.x >> 3) / (long)(gridDim.x >> 3));   // the same offset within an XCD's neighbours
  unsigned acc = 0;
  uint4 ring[RING];
  long i = start % fpw;
  const long total = fpw * a.passes;
#pragma unroll
  for (int s = 0; s < RING; ++s) { ring[s] = base[i * 64]; if (++i == fpw) i = 0; }
  for (long k = 0; k < total; k += RING) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      const uint4 v = ring[s];
      acc += (v.x ^ v.y) + (v.z ^ v.w);
      ring[s] = base[i * 64];
      if (++i == fpw) i = 0;
    }
  }
  if (acc == 0x12345678u) *a.sink = acc;
}

template <int RING, int WAVES>
static float run(const Args& a, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<RING, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best;
}


// ---- load flavours (8 waves, ring 8, lock step): plain / non-temporal / LDS-DMA (global_load_lds 16 bytes per lane, the
// fragment then read back from LDS) / plain with only a quarter of the CUs active
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int FLAVOUR>
__global__ __launch_bounds__(512) void flavour_kernel(Args a) {
  __shared__ uint4 s_ring[8][8][64];   // [wave][slot][lane] 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long fpw = a.frags_per_wave;
  const uint4* base = a.p + (long)wave * fpw * 64 + lane;
  unsigned acc = 0;
  const long total = fpw * a.passes;
  if (FLAVOUR == 2) {
    long i = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + i * 64), (lds_ptr_t)(uintptr_t)&s_ring[wave][s][0], 16, 0, 0); if (++i == fpw) i = 0; }
    for (long k = 0; k < total; k += 8) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        const uint4 v = s_ring[wave][s][lane];
        acc += (v.x ^ v.y) + (v.z ^ v.w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + i * 64), (lds_ptr_t)(uintptr_t)&s_ring[wave][s][0], 16, 0, 0);
        if (++i == fpw) i = 0;
      }
    }
  } else {
    uint4 ring[8];
    long i = 0;
    auto ld = [&](long idx) {
      const uint4* q = base + idx * 64;
      if (FLAVOUR == 1) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q));
        return make_uint4(t.x, t.y, t.z, t.w);
      }
      return *q;
    };
#pragma unroll
    for (int s = 0; s < 8; ++s) { ring[s] = ld(i); if (++i == fpw) i = 0; }
    for (long k = 0; k < total; k += 8) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 v = ring[s];
        acc += (v.x ^ v.y) + (v.z ^ v.w);
        ring[s] = ld(i);
        if (++i == fpw) i = 0;
      }
    }
  }
  if (acc == 0x12345678u) *a.sink = acc;
}

template <int FLAVOUR>
static float run_flavour(const Args& a, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((flavour_kernel<FLAVOUR>), dim3(grid), dim3(512), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best;
}

// ---- what the counters count: s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz) over one streaming
// workgroup, and the same stream out of L1 (2 KiB per wave re-read: 16 KiB per CU)
__global__ __launch_bounds__(512) void clock_kernel(Args a, long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long fpw = a.frags_per_wave;
  const uint4* base = a.p + (long)wave * fpw * 64 + lane;
  unsigned acc = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  uint4 ring[8];
  long i = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) { ring[s] = base[i * 64]; if (++i == fpw) i = 0; }
  const long total = fpw * a.passes;
  for (long k = 0; k < total; k += 8) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const uint4 v = ring[s];
      acc += (v.x ^ v.y) + (v.z ^ v.w);
      ring[s] = base[i * 64];
      if (++i == fpw) i = 0;
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (acc == 0x12345678u) *a.sink = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("%d CUs, %.2f GHz (peak clock; bytes per clock below use it)\n", cus, ghz);
  const size_t bytes_max = 64u << 20;
  uint4* buf;
  CK(hipMalloc(&buf, bytes_max));
  CK(hipMemset(buf, 1, bytes_max));
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  printf("%-10s %-6s %-5s %-5s %-7s %10s %12s %14s\n", "stream", "waves", "ring", "grid", "rotate", "us", "TB/s (all)", "B/clk/CU");
  for (size_t mb : {2, 50}) {
    for (int waves : {8, 4}) {
      for (int ring : {8, 16}) {
        for (int grid : {cus, 2 * cus}) {
          if (grid == 2 * cus && waves == 8) continue;
          for (int rotate : {0, 1, 2}) {
            const long fpw = (long)(mb << 20) / 1024 / waves;
            const int passes = mb == 2 ? 24 : 1;
            Args a{buf, fpw, passes, rotate, sink};
            float ms;
            if (waves == 8) ms = ring == 8 ? run<8, 8>(a, grid) : run<16, 8>(a, grid);
            else ms = ring == 8 ? run<8, 4>(a, grid) : run<16, 4>(a, grid);
            const double total = (double)grid * waves * fpw * passes * 1024.0;
            printf("%-10s %-6d %-5d %-5d %-7d %10.1f %12.2f %14.1f\n", mb == 2 ? "2 MB x24" : "50 MB", waves, ring, grid, rotate,
                   ms * 1e3, total / (ms * 1e-3) * 1e-12, total / (ms * 1e-3) / (ghz * 1e9) / cus);
          }
        }
      }
    }
  }
  printf("\nload flavours (8 waves, ring 8, lock step):\n");
  for (size_t mb : {2, 50}) {
    const long fpw = (long)(mb << 20) / 1024 / 8;
    const int passes = mb == 2 ? 24 : 1;
    Args a{buf, fpw, passes, 0, sink};
    struct { const char* name; int f; int grid; } rows[] = {{"plain", 0, cus}, {"non-temporal", 1, cus}, {"LDS-DMA + ds_read", 2, cus},
                                                          {"plain, 64 workgroups", 0, 64}, {"plain, 32 workgroups", 0, 32}, {"LDS-DMA, 64 workgroups", 2, 64}};
    for (auto& r : rows) {
      const float ms = r.f == 0 ? run_flavour<0>(a, r.grid) : r.f == 1 ? run_flavour<1>(a, r.grid) : run_flavour<2>(a, r.grid);
      const double total = (double)r.grid * 8 * fpw * passes * 1024.0;
      printf("%-10s %-26s %10.1f us %8.2f TB/s %8.1f B/clk per ACTIVE workgroup\n", mb == 2 ? "2 MB x24" : "50 MB", r.name, ms * 1e3,
             total / (ms * 1e-3) * 1e-12, total / (ms * 1e-3) / (ghz * 1e9) / r.grid);
    }
  }
  {
    long long* out;
    CK(hipMalloc(&out, 16));
    for (int l1 : {0, 1}) {
      const long fpw = l1 ? 2 : (long)(2 << 20) / 1024 / 8;
      const int passes = l1 ? 24 * 128 : 24;
      Args a{buf, fpw, passes, 0, sink};
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(clock_kernel, dim3(cus), dim3(512), 0, 0, a, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      long long h[2];
      CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      const double bytes = 8.0 * fpw * passes * 1024.0;
      printf("%s: %.1f us by events; workgroup 0: %lld s_memtime ticks, %lld s_memrealtime ticks (100 MHz -> %.1f us): s_memtime runs at %.3f GHz; "
             "%.1f B per s_memtime tick and CU, %.1f GB/s per CU\n", l1 ? "L1-resident (2 KiB per wave)" : "L2-resident (2 MB)", ms * 1e3, h[0], h[1],
             h[1] / 100.0, h[0] / (h[1] * 10.0), bytes / h[0], bytes / (h[1] * 10.0));
    }
  }
  return 0;
}
