// TEST INFRASTRUCTURE ONLY -- a lane-accurate host emulator for the HIP subset
// used by tapnet_amd/csrc/*.hip.
//
// The GPU box is reachable for a few minutes per round only, so the kernels'
// index math (LDS tiling, MFMA fragment layouts, wave shuffles, halo handling)
// is exercised on the CPU first: the *unmodified* .hip sources are compiled
// with the host clang++ against this header (it shadows <hip/hip_runtime.h>)
// into tests/hipemu/libtapir_emu.so, which exports the same C ABI as the real
// gfx950 library.  Only tests/ loads it; the product (tapnet_amd/) loads the
// gfx950 library exclusively and has no CPU path.
//
// Execution model: every workgroup runs as N cooperative fibers (one per
// work-item) on one OS thread; __syncthreads and the wave collectives
// (shuffles, MFMA) are rendezvous points.  Fibers are scheduled in a
// pseudo-random order between rendezvous points so that a missing barrier
// shows up as a wrong result instead of passing by luck.  Two schedules:
//   fair   -- every fiber advances to its next rendezvous each round: waves
//             stay within one collective of each other;
//   skewed -- one wave at a time runs alone until all its lanes wait at a
//             workgroup barrier (or finish), waves in a random order: a wave
//             gets a whole barrier interval ahead of the others, which is
//             what a missing barrier between two phases needs to show up.
// HIPEMU_SKEW=0/1 forces one schedule for every workgroup; unset, odd
// workgroups run skewed and even ones fair.  Workgroups run in parallel
// over OS threads (OpenMP).
//
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   16x16x32 bf16 : A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15], j<8
//                   D[row=4*(l>>4)+r][col=l&15], r<4
//   16x16x4  f32  : A[l&15][k=l>>4], B[k=l>>4][l&15], D as above
//   32x32x16 bf16 : A[i=l&31][k=8*(l>>5)+j], B[k][n=l&31]
//                   D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31], r<16
//   32x32x2  f32  : A[l&31][k=l>>5], B[k=l>>5][l&31], D as above
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#define TAPIR_HIPEMU 1

// ---------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

// ---------------------------------------------------------------- basic types
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned uint2 __attribute__((ext_vector_type(2)));
typedef unsigned uint4 __attribute__((ext_vector_type(4)));
typedef unsigned short ushort2 __attribute__((ext_vector_type(2)));
typedef unsigned short ushort4 __attribute__((ext_vector_type(4)));
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4
static inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) {
  *p = (T*)aligned_alloc(256, (n + 255) / 256 * 256);
  return *p ? hipSuccess : 2;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }

typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (the emulator runs every launch to completion)
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }

// ---------------------------------------------------------------- fibers
namespace hipemu {

extern "C" void hipemu_switch(void** from_sp, void* to_sp);

// AddressSanitizer builds (build_emu.sh --asan): the fibers run on heap-allocated stacks, which the sanitizer has to be
// told about at every switch (it keeps the current stack's bounds per thread); no-ops otherwise.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#endif
#endif
#ifdef HIPEMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif

struct Block;
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  bool in_barrier = false;
  long wait_gen = 0;
  unsigned tid = 0;
  Block* block = nullptr;
  void* fake = nullptr;                  // (ASAN fake-stack handle of this fiber)
};

struct WaveBuf {           // one rendezvous buffer (two per wave, alternating)
  int arrived = 0;
  uint32_t v[64][16];      // per-lane deposit (shuffle value, or MFMA a/b)
  float acc[64][16];       // per-lane MFMA accumulator in/out
};
struct Wave {
  long seq = 0;            // completed collectives
  WaveBuf buf[2];
};

struct Block {
  unsigned nthreads = 0;
  dim3 bidx, bdim, gdim;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int bar_arrived = 0;
  long bar_gen = 0;
  unsigned live = 0;
  void* sched_sp = nullptr;
  void* sched_fake = nullptr;            // (ASAN: fake-stack handle / bounds of the scheduler's own stack)
  const void* sched_bottom = nullptr;
  size_t sched_size = 0;
  Fiber* cur = nullptr;
  const std::function<void()>* body = nullptr;
};

inline thread_local Block* g_block = nullptr;

struct Idx3 { unsigned x, y, z; };
inline thread_local Idx3 threadIdx_, blockIdx_, blockDim_, gridDim_;

inline void yield() {
  Block* b = g_block;
#ifdef HIPEMU_ASAN
  Fiber* me = b->cur;
  __sanitizer_start_switch_fiber(me->done ? nullptr : &me->fake, b->sched_bottom, b->sched_size);
#endif
  hipemu_switch(&b->cur->sp, b->sched_sp);
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(me->fake, nullptr, nullptr);
#endif
}

// scheduler -> fiber f and back
inline void enter_fiber(Block& b, Fiber& f, char* stack_lo, size_t stack_size) {
#ifdef HIPEMU_ASAN
  __sanitizer_start_switch_fiber(&b.sched_fake, stack_lo, stack_size);
#else
  (void)stack_lo; (void)stack_size;
#endif
  hipemu_switch(&b.sched_sp, f.sp);
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(b.sched_fake, nullptr, nullptr);
#endif
}

inline void fiber_entry() {
  Block* b = g_block;
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &b->sched_bottom, &b->sched_size);
#endif
  (*b->body)();
  b->cur->done = true;
  b->live--;
  // a finished fiber counts as arrived at every later barrier
  yield();
  abort();
}

static constexpr size_t kStack = 256 * 1024;

inline void run_block(Block& b, const std::function<void()>& body) {
  g_block = &b;
  b.body = &body;
  b.live = b.nthreads;
  b.fibers.resize(b.nthreads);
  b.waves.assign((b.nthreads + 63) / 64, Wave());
  char* stacks = (char*)aligned_alloc(64, kStack * b.nthreads);
  for (unsigned t = 0; t < b.nthreads; ++t) {
    Fiber& f = b.fibers[t];
    f.tid = t; f.block = &b; f.done = false;
    f.stack = stacks + kStack * t;
    // initial frame for hipemu_switch: 6 callee-saved regs + return address
    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);       // so that rsp%16==8 at fiber_entry
    *--sp = (void*)&fiber_entry;         // return address
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
  }
  // scheduling rounds in a shuffled order
  std::vector<unsigned> order(b.nthreads);
  for (unsigned i = 0; i < b.nthreads; ++i) order[i] = i;
  uint32_t rng = 0x9e3779b9u ^ (b.bidx.x * 2654435761u);
  auto run_fiber = [&](Fiber& f) {
    b.cur = &f;
    unsigned t = f.tid;
    threadIdx_.x = t % b.bdim.x;
    threadIdx_.y = (t / b.bdim.x) % b.bdim.y;
    threadIdx_.z = t / (b.bdim.x * b.bdim.y);
    enter_fiber(b, f, f.stack, kStack);
  };
  bool skew = (b.bidx.x & 1u) != 0;
  if (const char* e = getenv("HIPEMU_SKEW")) skew = e[0] == '1';
  const unsigned nwaves = (b.nthreads + 63) / 64;
  if (skew && nwaves > 1) {
    std::vector<unsigned> worder(nwaves);
    for (unsigned i = 0; i < nwaves; ++i) worder[i] = i;
    while (b.live > 0) {
      for (unsigned i = nwaves - 1; i > 0; --i) {
        rng = rng * 1664525u + 1013904223u;
        unsigned j = (rng >> 8) % (i + 1);
        unsigned tmp = worder[i]; worder[i] = worder[j]; worder[j] = tmp;
      }
      bool progress = false;
      for (unsigned wi = 0; wi < nwaves; ++wi) {
        const unsigned lo = worder[wi] * 64, hi = lo + 64 < b.nthreads ? lo + 64 : b.nthreads;
        for (;;) {   // this wave alone, until every lane waits at a barrier or is done
          bool runnable = false;
          for (unsigned k = lo; k < hi; ++k) {
            Fiber& f = b.fibers[lo + (k - lo + (rng >> 10)) % (hi - lo)];
            if (f.done || (f.in_barrier && f.wait_gen == b.bar_gen)) continue;
            runnable = true;
            progress = true;
            run_fiber(f);
          }
          rng = rng * 1664525u + 1013904223u;
          if (!runnable) break;
        }
      }
      // lanes that returned early can complete a barrier the waiters must re-evaluate
      if (!progress)
        for (unsigned k = 0; k < b.nthreads; ++k)
          if (!b.fibers[k].done) run_fiber(b.fibers[k]);
    }
    free(stacks);
    g_block = nullptr;
    return;
  }
  while (b.live > 0) {
    for (unsigned i = b.nthreads - 1; i > 0; --i) {
      rng = rng * 1664525u + 1013904223u;
      unsigned j = (rng >> 8) % (i + 1);
      unsigned tmp = order[i]; order[i] = order[j]; order[j] = tmp;
    }
    for (unsigned k = 0; k < b.nthreads; ++k) {
      Fiber& f = b.fibers[order[k]];
      if (f.done) continue;
      b.cur = &f;
      unsigned t = f.tid;
      threadIdx_.x = t % b.bdim.x;
      threadIdx_.y = (t / b.bdim.x) % b.bdim.y;
      threadIdx_.z = t / (b.bdim.x * b.bdim.y);
      enter_fiber(b, f, f.stack, kStack);
    }
  }
  free(stacks);
  g_block = nullptr;
}

inline void restore_tid() {
  Block* b = g_block;
  unsigned t = b->cur->tid;
  threadIdx_.x = t % b->bdim.x;
  threadIdx_.y = (t / b->bdim.x) % b->bdim.y;
  threadIdx_.z = t / (b->bdim.x * b->bdim.y);
}

inline void syncthreads() {
  Block* b = g_block;
  long gen = b->bar_gen;
  Fiber* me = b->cur;
  b->bar_arrived++;
  me->in_barrier = true;
  me->wait_gen = gen;
  for (;;) {
    if (b->bar_gen != gen) break;
    if (b->bar_arrived >= (int)b->live) {  // everyone alive has arrived
      b->bar_arrived = 0;
      b->bar_gen++;
      break;
    }
    yield();
  }
  me->in_barrier = false;
}

// Wave collective: every live lane of the wave deposits, the last arriver runs
// `compute` over the whole buffer, then every lane reads its result.
template <class Dep, class Comp, class Rd>
inline void wave_collective(Dep dep, Comp comp, Rd rd) {
  Block* b = g_block;
  unsigned t = b->cur->tid;
  Wave& w = b->waves[t / 64];
  unsigned lane = t % 64;
  long my = w.seq;
  WaveBuf& wb = w.buf[my & 1];
  dep(wb, lane);
  wb.arrived++;
  unsigned wave_size = b->nthreads - (t / 64) * 64;
  if (wave_size > 64) wave_size = 64;
  // lanes of this wave that already returned never arrive: count live lanes.  A waiting lane
  // re-counts every time it is scheduled: lanes that left a loop early finish (and stop being
  // live) only after the others have started to wait -- the hardware analogue is the exec mask.
  for (;;) {
    if (w.seq != my) break;   // somebody completed the collective
    unsigned live_lanes = 0;
    for (unsigned l = 0; l < wave_size; ++l) live_lanes += !b->fibers[(t / 64) * 64 + l].done;
    if (wb.arrived >= (int)live_lanes) {
      comp(wb);
      wb.arrived = 0;
      w.seq = my + 1;
      break;
    }
    yield();
  }
  rd(wb, lane);
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& f) {
  std::function<void()> body = f;
  long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic)
  for (long bi = 0; bi < nblocks; ++bi) {
    Block b;
    b.bdim = block; b.gdim = grid;
    b.nthreads = block.x * block.y * block.z;
    b.bidx = dim3(bi % grid.x, (bi / grid.x) % grid.y, bi / ((long)grid.x * grid.y));
    blockIdx_ = {b.bidx.x, b.bidx.y, b.bidx.z};
    blockDim_ = {block.x, block.y, block.z};
    gridDim_ = {grid.x, grid.y, grid.z};
    run_block(b, body);
  }
}

}  // namespace hipemu

#define threadIdx (hipemu::threadIdx_)
#define blockIdx (hipemu::blockIdx_)
#define blockDim (hipemu::blockDim_)
#define gridDim (hipemu::gridDim_)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::syncthreads(); }

// ---------------------------------------------------------------- shuffles
template <class T> static inline T hipemu_shfl_idx(T v, int src) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  T out;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) { memcpy(&wb.v[lane][0], &v, 4); },
      [&](hipemu::WaveBuf&) {},
      [&](hipemu::WaveBuf& wb, unsigned lane) { (void)lane; memcpy(&out, &wb.v[src & 63][0], 4); });
  return out;
}
static inline unsigned hipemu_lane() { return hipemu::g_block->cur->tid % 64; }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu_lane();
  return hipemu_shfl_idx(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return hipemu_shfl_idx(v, hipemu_lane() ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu_lane();
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu_shfl_idx(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = hipemu_lane();
  int src = lane - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu_shfl_idx(v, src);
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu_shfl_idx(v, 0); }

// ---------------------------------------------------------------- MFMA
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

static inline float hipemu_bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f;
}

static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(
    hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
  hipemu_f32x4 out;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) {
        memcpy(&wb.v[lane][0], &a, 16); memcpy(&wb.v[lane][4], &b, 16);
        for (int r = 0; r < 4; ++r) wb.acc[lane][r] = c[r];
      },
      [&](hipemu::WaveBuf& wb) {
        float d[16][16];
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
          // accumulator of D[i][n] lives in lane (n + 16*(i/4)), reg i%4
          float s = wb.acc[n + 16 * (i / 4)][i % 4];
          for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) {
            const uint16_t* pa = (const uint16_t*)&wb.v[i + 16 * g][0];
            const uint16_t* pb = (const uint16_t*)&wb.v[n + 16 * g][4];
            s += hipemu_bf16_to_f32(pa[j]) * hipemu_bf16_to_f32(pb[j]);
          }
          d[i][n] = s;
        }
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
          wb.acc[l][r] = d[4 * (l >> 4) + r][l & 15];
      },
      [&](hipemu::WaveBuf& wb, unsigned lane) { for (int r = 0; r < 4; ++r) out[r] = wb.acc[lane][r]; });
  return out;
}

static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(
    float a, float b, hipemu_f32x4 c, int, int, int) {
  hipemu_f32x4 out;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) {
        memcpy(&wb.v[lane][0], &a, 4); memcpy(&wb.v[lane][1], &b, 4);
        for (int r = 0; r < 4; ++r) wb.acc[lane][r] = c[r];
      },
      [&](hipemu::WaveBuf& wb) {
        float d[16][16];
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
          float s = wb.acc[n + 16 * (i / 4)][i % 4];
          for (int k = 0; k < 4; ++k) {
            float fa, fb;
            memcpy(&fa, &wb.v[i + 16 * k][0], 4); memcpy(&fb, &wb.v[n + 16 * k][1], 4);
            s = fmaf(fa, fb, s);
          }
          d[i][n] = s;
        }
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
          wb.acc[l][r] = d[4 * (l >> 4) + r][l & 15];
      },
      [&](hipemu::WaveBuf& wb, unsigned lane) { for (int r = 0; r < 4; ++r) out[r] = wb.acc[lane][r]; });
  return out;
}

static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(
    hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
  hipemu_f32x16 out;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) {
        memcpy(&wb.v[lane][0], &a, 16); memcpy(&wb.v[lane][4], &b, 16);
        for (int r = 0; r < 16; ++r) wb.acc[lane][r] = c[r];
      },
      [&](hipemu::WaveBuf& wb) {
        static thread_local float d[32][32];
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
          d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = wb.acc[l][r];
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
          float s = d[i][n];
          for (int g = 0; g < 2; ++g) for (int j = 0; j < 8; ++j) {
            const uint16_t* pa = (const uint16_t*)&wb.v[i + 32 * g][0];
            const uint16_t* pb = (const uint16_t*)&wb.v[n + 32 * g][4];
            s += hipemu_bf16_to_f32(pa[j]) * hipemu_bf16_to_f32(pb[j]);
          }
          d[i][n] = s;
        }
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
          wb.acc[l][r] = d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31];
      },
      [&](hipemu::WaveBuf& wb, unsigned lane) { for (int r = 0; r < 16; ++r) out[r] = wb.acc[lane][r]; });
  return out;
}

static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(
    float a, float b, hipemu_f32x16 c, int, int, int) {
  hipemu_f32x16 out;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) {
        memcpy(&wb.v[lane][0], &a, 4); memcpy(&wb.v[lane][1], &b, 4);
        for (int r = 0; r < 16; ++r) wb.acc[lane][r] = c[r];
      },
      [&](hipemu::WaveBuf& wb) {
        static thread_local float d[32][32];
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
          d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = wb.acc[l][r];
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
          float s = d[i][n];
          for (int k = 0; k < 2; ++k) {
            float fa, fb;
            memcpy(&fa, &wb.v[i + 32 * k][0], 4); memcpy(&fb, &wb.v[n + 32 * k][1], 4);
            s = fmaf(fa, fb, s);
          }
          d[i][n] = s;
        }
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
          wb.acc[l][r] = d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31];
      },
      [&](hipemu::WaveBuf& wb, unsigned lane) { for (int r = 0; r < 16; ++r) out[r] = wb.acc[lane][r]; });
  return out;
}

// ---------------------------------------------------------------- LDS DMA
// global_load_lds: every lane copies `size` bytes to (first lane's LDS address + lane*size).
// The emulator copies synchronously and CHECKS the addressing rule the hardware imposes.
static inline void __builtin_amdgcn_global_load_lds(
    const __attribute__((address_space(1))) void* gptr, __attribute__((address_space(3))) void* lptr,
    unsigned size, int offset, unsigned aux) {
  (void)aux;
  uintptr_t l = (uintptr_t)lptr + (uintptr_t)offset;
  uintptr_t g = (uintptr_t)gptr + (uintptr_t)offset;
  uintptr_t base = 0;
  hipemu::wave_collective(
      [&](hipemu::WaveBuf& wb, unsigned lane) { memcpy(&wb.v[lane][0], &l, sizeof(l)); },
      [&](hipemu::WaveBuf&) {},
      [&](hipemu::WaveBuf& wb, unsigned lane) { (void)lane; memcpy(&base, &wb.v[0][0], sizeof(base)); });
  // the hardware uses M0 = the first lane's LDS address and adds lane * size itself: the argument
  // must be wave-uniform (the base) or already base + lane * size
  if (l != base && l != base + (uintptr_t)hipemu_lane() * size) {
    fprintf(stderr, "hipemu: global_load_lds destination is neither wave-uniform nor base + lane*size\n");
    abort();
  }
  memcpy((void*)(base + (uintptr_t)hipemu_lane() * size), (const void*)g, size);
}

// ---------------------------------------------------------------- misc device math
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline long long wall_clock64() { return 0; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
// HIP exposes integer min/max in device code
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
