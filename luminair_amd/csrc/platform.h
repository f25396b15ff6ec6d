// Platform layer: the product build is HIP for gfx950 (hipcc).  Defining LMN_EMU instead builds the
// same kernel sources for a *test-only* host emulation (one OS thread per GPU thread, real
// barriers) used by tests/emu to debug host orchestration and kernel indexing on a machine without
// a GPU.  The emulation library is never loaded by the luminair_amd package; the shipped
// libluminair_hip.so is built without LMN_EMU and fails loudly when no HIP device is present.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#ifndef LMN_EMU
#include <hip/hip_runtime.h>
#include <time.h>
#include <chrono>
#define LMN_HD __host__ __device__ __forceinline__
#define LMN_D __device__ __forceinline__
#ifdef LMN_BATCH
#define LMN_KERNEL __device__ void   /* kernel bodies; the one __global__ is lmn_batch_tramp (batch.h) */
#else
#define LMN_KERNEL __global__ void
#endif
#define LMN_DYN_SMEM(T, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; T* name = reinterpret_cast<T*>(name##_raw)
#define LMN_SHARED __shared__
typedef hipStream_t lmn_stream_t;
#define lmn_shfl_xor(v, mask) __shfl_xor((v), (mask), 64)
#define lmn_shfl_up(v, delta) __shfl_up((v), (delta), 64)   // lanes below `delta` of a wave keep their own value
// value of lane (quad base + ((CTRL >> 2*(lane&3)) & 3)): DPP quad_perm, no LDS round trip
#define lmn_quad_perm(v, CTRL) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (CTRL), 0xf, 0xf, true))
#define LMN_ASSUME(x) __builtin_assume(x)

struct LmnError : std::runtime_error {
  int code;
  LmnError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define LMN_HIP_CHECK(expr)                                                                        \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      throw LmnError(-100, std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                               std::to_string(__LINE__));                                          \
  } while (0)

#ifdef LMN_BATCH
#include "batch.h"
#define LMN_LAUNCH(kernel, grid, block, smem, stream, ...) ::lmn::batch_launch<&kernel>(grid, block, smem, stream, __VA_ARGS__)
#else
#define LMN_LAUNCH(kernel, grid, block, smem, stream, ...)                          \
  do {                                                                              \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);             \
    LMN_HIP_CHECK(hipGetLastError());                                               \
  } while (0)
#endif

inline void* lmn_dev_malloc(size_t bytes) {
  void* p = nullptr;
  LMN_HIP_CHECK(hipMalloc(&p, bytes));
  return p;
}
inline void lmn_dev_free(void* p) { (void)hipFree(p); }
#ifdef LMN_BATCH
#define LMN_BATCH_COPY(dst, src, n, dir) if (::lmn::batch_copy((dst), (src), (n), (dir))) return
#else
#define LMN_BATCH_COPY(dst, src, n, dir) do { } while (0)
#endif
inline void lmn_h2d(void* dst, const void* src, size_t n, lmn_stream_t s) {
  LMN_BATCH_COPY(dst, src, n, 0);
  LMN_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
}
inline void lmn_d2h(void* dst, const void* src, size_t n, lmn_stream_t s) {
  LMN_BATCH_COPY(dst, src, n, 1);
  LMN_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s));
}
inline void lmn_d2d(void* dst, const void* src, size_t n, lmn_stream_t s) {
  LMN_BATCH_COPY(dst, src, n, 2);
  LMN_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
}
inline void lmn_memset(void* dst, int v, size_t n, lmn_stream_t s) {
  LMN_BATCH_COPY(dst, (const void*)(uintptr_t)(unsigned char)v, n, 3);
  LMN_HIP_CHECK(hipMemsetAsync(dst, v, n, s));
}
namespace lmn { extern std::atomic<int> g_proofs_in_flight; }   // context.cpp: proofs this process is proving right now
// Spin on hipStreamQuery instead of blocking in hipStreamSynchronize: the prover synchronises 6
// times per proof and the blocking wake-up latency (tens of microseconds) would sit on the critical path.
inline void lmn_sync(lmn_stream_t s) {
#ifdef LMN_BATCH
  ::lmn::batch_sync(s);   // a rendezvous of the lock-step group: one stream wait for all members
  return;
#endif
  static const int mode = getenv("LMN_SYNC_MODE") ? atoi(getenv("LMN_SYNC_MODE")) : 0;  // 0 spin, 1 hipStreamSynchronize, 2 hybrid, 3 blocking event
  if (mode == 1) {
    LMN_HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  if (mode == 3) {   // sleep on an interrupt: a blocking-sync event recorded behind the stream's work (no polling at all)
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int ev_dev = -1;
    int dev = 0;
    LMN_HIP_CHECK(hipGetDevice(&dev));
    if (!ev || ev_dev != dev) {
      LMN_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
      ev_dev = dev;
    }
    LMN_HIP_CHECK(hipEventRecord(ev, s));
    LMN_HIP_CHECK(hipEventSynchronize(ev));
    return;
  }
  // Mode 0 polls.  A proof that has the GPU to itself never waits longer than about a millisecond, and polling is what keeps
  // its latency low; under concurrent load every wait lasts several milliseconds (the GPU is shared), eight polling contexts
  // kept eight CPUs busy, and N ranks in one CPU-limited container starved each other's launch threads
  // (tools/host_cpu_per_proof.py).  So a wait that outlasts LMN_SPIN_US goes on in 50 us sleeps, and a thread whose recent
  // waits did so starts sleeping after 100 us already; a few short waits bring it back to polling.  Default 3000 us since
  // round 6 (1200 before): with the quotient step on the device a solo 2^20-row proof has ONE wait of 2 ms in front of its
  // decommitment instead of three below a millisecond - at 1200 us it went to sleep in it and woke up 50 - 100 us late.
  static const long spin_us = getenv("LMN_SPIN_US") ? atol(getenv("LMN_SPIN_US")) : 3000;
  static thread_local int long_waits = 0;   // 0 .. 8: how many of the recent waits on this thread outlasted spin_us
  // (end of round 6) the direct signal: while this process proves several proofs at once the GPU is shared whatever this
  // thread's own history says - with the limit at 3000 us a wait of 1 - 3 ms under load (the gather launch queued behind other
  // proofs' work) kept a thread polling and reset its history: 3.2 instead of 1.9 ms of host CPU per proof at 24 in flight
#ifdef LMN_NO_SHARED_SIGNAL   // (experiment build: the policy before this signal)
  const bool shared = false;
#else
  const bool shared = ::lmn::g_proofs_in_flight.load(std::memory_order_relaxed) > 1;
#endif
  const long limit_us = (shared || long_waits >= 2) ? 100 : spin_us;
  const auto t_start = std::chrono::steady_clock::now();
  auto waited_us = [&] {
    return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start).count();
  };
  bool sleeping = false;
  for (int it = 0;; ++it) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) LMN_HIP_CHECK(e);
    if (mode == 2 && it > 64) {
      LMN_HIP_CHECK(hipStreamSynchronize(s));
      return;
    }
    if (sleeping) {
      struct timespec ts = {0, 50000};
      nanosleep(&ts, nullptr);
      continue;
    }
    if (spin_us >= 0 && (it & 15) == 15 && waited_us() > limit_us) sleeping = true;
#if defined(__x86_64__)
    for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
#endif
  }
  if (spin_us >= 0) {
    if (sleeping && waited_us() > spin_us) {
      if (long_waits < 8) ++long_waits;
    } else if (long_waits > 0) {
      --long_waits;
    }
  }
}
inline void* lmn_host_alloc_pinned(size_t bytes) {
  void* p = nullptr;
  LMN_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
#ifdef LMN_BATCH
  ::lmn::batch_register_pinned(p, bytes);
#endif
  return p;
}
inline void lmn_host_free_pinned(void* p) {
#ifdef LMN_BATCH
  ::lmn::batch_unregister_pinned(p);
#endif
  (void)hipHostFree(p);
}
inline void lmn_host_register_range(void* p, size_t bytes) {
  LMN_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
#ifdef LMN_BATCH
  ::lmn::batch_register_pinned(p, bytes);
#endif
}
inline void lmn_host_unregister_range(void* p) {
#ifdef LMN_BATCH
  ::lmn::batch_unregister_pinned(p);
#endif
  LMN_HIP_CHECK(hipHostUnregister(p));
}
typedef hipEvent_t lmn_event_t;
inline lmn_event_t lmn_event_create() {
  hipEvent_t e;
  LMN_HIP_CHECK(hipEventCreate(&e));
  return e;
}
inline void lmn_event_destroy(lmn_event_t e) { (void)hipEventDestroy(e); }
inline void lmn_event_record(lmn_event_t e, lmn_stream_t s) { LMN_HIP_CHECK(hipEventRecord(e, s)); }
inline float lmn_event_elapsed_ms(lmn_event_t a, lmn_event_t b) {
  float ms = 0.f;
  LMN_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}
inline lmn_event_t lmn_event_create_sync() {   // ordering only (no timestamps)
  hipEvent_t e;
  LMN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}
inline void lmn_stream_wait_event(lmn_stream_t s, lmn_event_t e) { LMN_HIP_CHECK(hipStreamWaitEvent(s, e, 0)); }

#else  // ------------------------------------------------------------------ LMN_EMU (tests only)
#include <barrier>
#include <functional>
#include <thread>
#include <vector>

#define LMN_HD inline
#define LMN_D inline
#define LMN_KERNEL static void
#define LMN_SHARED static
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3e {
  unsigned x, y, z;
};
extern thread_local uint3e threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
extern thread_local unsigned char* lmn_emu_dyn_smem;
void lmn_emu_syncthreads();
#define __syncthreads() lmn_emu_syncthreads()
#define LMN_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(lmn_emu_dyn_smem)
typedef int lmn_stream_t;
#define LMN_ASSUME(x) ((void)0)
extern unsigned lmn_emu_shfl_scratch[1024];
inline unsigned lmn_shfl_xor(unsigned v, int mask) {  // all lanes of the block must call it together
  lmn_emu_shfl_scratch[threadIdx.x] = v;
  lmn_emu_syncthreads();
  unsigned r = lmn_emu_shfl_scratch[threadIdx.x ^ (unsigned)mask];
  lmn_emu_syncthreads();
  return r;
}
inline unsigned lmn_shfl_up(unsigned v, unsigned delta) {  // all lanes of the block must call it together
  lmn_emu_shfl_scratch[threadIdx.x] = v;
  lmn_emu_syncthreads();
  unsigned r = (threadIdx.x & 63u) >= delta ? lmn_emu_shfl_scratch[threadIdx.x - delta] : v;
  lmn_emu_syncthreads();
  return r;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {  // fibers are cooperative: no real concurrency
  unsigned o = *p;
  *p += v;
  return o;
}
inline unsigned lmn_quad_perm(unsigned v, int ctrl) {  // all lanes of the block must call it together
  lmn_emu_shfl_scratch[threadIdx.x] = v;
  lmn_emu_syncthreads();
  unsigned r = lmn_emu_shfl_scratch[(threadIdx.x & ~3u) | (((unsigned)ctrl >> (2 * (threadIdx.x & 3u))) & 3u)];
  lmn_emu_syncthreads();
  return r;
}
struct uint4 {
  unsigned x, y, z, w;
};
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline unsigned __brev(unsigned x) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
  return r;
}

struct LmnError : std::runtime_error {
  int code;
  LmnError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void lmn_emu_run(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem);
#ifdef LMN_BATCH
// ---- emulation of the lock-step batch build (tests/emu/build_emu.sh batch; round 6): batch.h / batch.cpp - rendezvous, member
// fibers, argument tables, the batched copy list - compiled as they are, the dozen HIP calls they make served from host
// memory, the trampoline run through lmn_emu_run with blockIdx.z = the member.  Lets the CPU suite and the sanitizers reach
// the one part of the host code that otherwise exists only in the gfx950 build.
typedef int hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorNotReady = 600;
constexpr unsigned hipHostMallocDefault = 0u, hipStreamNonBlocking = 1u;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s_, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated HIP call failed"; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s_, unsigned) { *s_ = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
#define LMN_HIP_CHECK(expr)                                                          \
  do {                                                                               \
    if ((expr) != hipSuccess) throw LmnError(-100, "emulated HIP call failed");      \
  } while (0)
#define __global__
#define __device__
#define __forceinline__ inline
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) lmn_emu_run([&]() { kernel(__VA_ARGS__); }, grid, block, smem)
#include "batch.h"
#define LMN_LAUNCH(kernel, grid, block, smem, stream, ...) ::lmn::batch_launch<&kernel>(grid, block, smem, stream, __VA_ARGS__)
#define LMN_BATCH_COPY(dst, src, n, dir) if (::lmn::batch_copy((dst), (src), (n), (dir))) return
#else
#define LMN_LAUNCH(kernel, grid, block, smem, stream, ...) \
  lmn_emu_run([&]() { kernel(__VA_ARGS__); }, grid, block, smem)
#define LMN_BATCH_COPY(dst, src, n, dir) do { } while (0)
#endif

inline void* lmn_dev_malloc(size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p) throw LmnError(-100, "emu malloc failed");
  return p;
}
inline void lmn_dev_free(void* p) { free(p); }
// (emulated batch build: the members' transfers go through batch_copy's list and their waits are rendezvous, as on the GPU)
inline void lmn_h2d(void* d, const void* s, size_t n, lmn_stream_t) {
  LMN_BATCH_COPY(d, s, n, 0);
  memcpy(d, s, n);
}
inline void lmn_d2h(void* d, const void* s, size_t n, lmn_stream_t) {
  LMN_BATCH_COPY(d, s, n, 1);
  memcpy(d, s, n);
}
inline void lmn_d2d(void* d, const void* s, size_t n, lmn_stream_t) {
  LMN_BATCH_COPY(d, s, n, 2);
  memmove(d, s, n);
}
inline void lmn_memset(void* d, int v, size_t n, lmn_stream_t) {
  LMN_BATCH_COPY(d, (const void*)(uintptr_t)(unsigned char)v, n, 3);
  memset(d, v, n);
}
#ifdef LMN_BATCH
inline void lmn_sync(lmn_stream_t s) { ::lmn::batch_sync(s); }
inline void* lmn_host_alloc_pinned(size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  ::lmn::batch_register_pinned(p, bytes);
  return p;
}
inline void lmn_host_free_pinned(void* p) {
  ::lmn::batch_unregister_pinned(p);
  free(p);
}
inline void lmn_host_register_range(void* p, size_t bytes) { ::lmn::batch_register_pinned(p, bytes); }
inline void lmn_host_unregister_range(void* p) { ::lmn::batch_unregister_pinned(p); }
#else
inline void lmn_sync(lmn_stream_t) {}
inline void* lmn_host_alloc_pinned(size_t bytes) { return malloc(bytes ? bytes : 1); }
inline void lmn_host_free_pinned(void* p) { free(p); }
inline void lmn_host_register_range(void*, size_t) {}
inline void lmn_host_unregister_range(void*) {}
#endif
#include <chrono>
typedef double* lmn_event_t;
inline lmn_event_t lmn_event_create() { return new double(0.0); }
inline void lmn_event_destroy(lmn_event_t e) { delete e; }
inline void lmn_event_record(lmn_event_t e, lmn_stream_t) {
  *e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline float lmn_event_elapsed_ms(lmn_event_t a, lmn_event_t b) { return (float)(*b - *a); }
inline lmn_event_t lmn_event_create_sync() { return new double(0.0); }
inline void lmn_stream_wait_event(lmn_stream_t, lmn_event_t) {}
#endif
