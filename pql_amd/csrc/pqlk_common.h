// Internal helpers shared by the libpqlk.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "../../include/pqlk.h"

#define PQLK_WAVE 64

#define PQLK_REQUIRE(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

// Kernel launches are asynchronous; hipGetLastError catches configuration errors only.
#define PQLK_LAUNCH_CHECK()                    \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return -(int)e__;   \
  } while (0)

static inline int64_t pqlk_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline hipStream_t pqlk_s(pqlk_stream_t s) { return (hipStream_t)s; }
static inline bool pqlk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Record layout shared by the replay ring and the n-step window: every field starts on a 16-B
// boundary so a record can be moved with dwordx4 accesses whatever O and A are.
//   [ obs (O) pad4 | next_obs (O) pad4 | action (A) pad4 | reward, done, 0, 0 | zero pad to 32 floats ]
struct RecLayout {
  int O, A;        // A < 0: obs-only record
  int o4, a4;      // field widths rounded up to 4 floats
  int off_nobs, off_act, off_rd;
  int used;        // floats in use (multiple of 4)
  int ld;          // record stride (multiple of 32)
};

__host__ __device__ static inline RecLayout rec_layout(int O, int A) {
  RecLayout L;
  L.O = O; L.A = A;
  L.o4 = (O + 3) & ~3;
  if (A < 0) {
    L.a4 = 0; L.off_nobs = L.off_act = L.off_rd = L.o4; L.used = L.o4;
  } else {
    L.a4 = (A + 3) & ~3;
    L.off_nobs = L.o4; L.off_act = 2 * L.o4; L.off_rd = 2 * L.o4 + L.a4; L.used = L.off_rd + 4;
  }
  L.ld = (L.used + 31) & ~31;
  return L;
}

// "have I raised this kernel's dynamic-LDS limit on the CURRENT device yet?"  hipFuncSetAttribute acts on the current
// device only, and one process may drive several (sim on GPU 0, learners on GPU 1): the flag is per device.
// ctypes drops the GIL, so two host threads (the V and the P learner of `algo.async_learners`) may make a kernel's first call
// at once: the device is marked done only AFTER the attribute call returned success, and a second caller waits on the mutex
// until then (never launching with > 64 KB of dynamic LDS before the limit is raised).  The fast path is one acquire load.
struct PqlkPerDeviceOnce {
  std::atomic<bool> done[64] = {};
  std::mutex mu;
  // f() -> 0 on success, else the (negative) error code to hand back; it is run again by the next call after a failure
  template <class F>
  int run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return f();
    if (done[dev].load(std::memory_order_acquire)) return 0;
    std::lock_guard<std::mutex> g(mu);
    if (done[dev].load(std::memory_order_relaxed)) return 0;
    const int rc = f();
    if (rc == 0) done[dev].store(true, std::memory_order_release);
    return rc;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
