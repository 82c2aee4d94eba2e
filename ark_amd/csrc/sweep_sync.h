// Hand-off protocol of the persistent GRU kernels (gru_sweep.hip: small batches of long sequences; until round 4 also gru_fat.hip:
// the forward recurrence of a full batch with register-resident weights): write-through payload stores, one padded
// MONOTONE counter per hand-off point, an epoch word per `sync` workspace, bounded spins and a STICKY error word.
// See the header comment of gru_sweep.hip for the protocol and MI355X_MICROARCH.md (visibility section) for why it is valid.
#pragma once
#include "gemm_core.h"

namespace ark {

typedef _Float16 shalf4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// s_memrealtime runs at 100 MHz: 2 s.  (0.25 s until round 4: with TWO processes on one card -- the data-parallel tests -- a
// sweep was seen to give up although nothing was wrong: its queue is time-sliced against the other process's, and a wait that
// straddles a slice is as long as the slice.  The bound only has to turn a real deadlock into an error instead of a hang.)
constexpr unsigned long long kSweepTimeoutTicks = 200000000ull;
constexpr int kSweepLds = 96 * 1024;                             // > half of a CU's LDS: one workgroup per CU
constexpr int kSweepSyncHdr = 32;                                // words in front of the counters: [0] sticky error, [1] who / where,
                                                                 // [2] epoch = launches completed on this workspace, [3] workgroups
                                                                 // of the running launch that have left; written with the error:
                                                                 // [4] the counter value waited for, [5] / [6] the values last seen
                                                                 // of the first / second counter polled (0xFFFFFFFF: not polled),
                                                                 // [7] 100-MHz ticks spent in the wait that gave up; [8] signals per
                                                                 // counter and launch (a counter's value = launches x [8] when idle)
constexpr int kSweepCntStride = 32;                              // words per counter: each on a 128-byte line of its own (atomics and
                                                                 // polls of different (layer, step, row block) never queue on one line)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sweep_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16);   // aux 16 = sc1
}
__device__ __forceinline__ void st_sc1(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 16);
}

// whole wave polls ONE word (one request); false = gave up (timeout here or elsewhere)
__device__ __forceinline__ bool sweep_wait(unsigned* cnt, unsigned need, unsigned* sync, unsigned code) {
  unsigned long long t0 = 0;
  for (unsigned spins = 1;; ++spins) {
    if ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0) return true;   // (wrap-safe: monotone counters)
    if ((spins & 31u) == 0u) {
      if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > kSweepTimeoutTicks) {
        // the record first, the error word last: need / seen tell a producer that never started this launch (its counter
        // still holds the previous launch's final value) from one that was merely late
        __hip_atomic_store(sync + 4, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 5, __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 6, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 7, (unsigned)(now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// two counters at once: both polls are in flight together (one round trip through the memory system instead of two in a row
// -- the second dependency of a step is normally satisfied long before the first)
__device__ __forceinline__ bool sweep_wait2(unsigned* c0, unsigned* c1, unsigned need, unsigned* sync, unsigned code) {
  unsigned long long t0 = 0;
  for (unsigned spins = 1;; ++spins) {
    const unsigned a = __hip_atomic_load(c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned b = __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(a - need) >= 0 && (int)(b - need) >= 0) return true;
    if ((spins & 31u) == 0u) {
      if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > kSweepTimeoutTicks) {
        __hip_atomic_store(sync + 4, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 5, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 6, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 7, (unsigned)(now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 1, code | ((int)(a - need) >= 0 ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}


// first thing a workgroup does: the launch's epoch (-> the counter value that means "this launch's producer is done") and
// whether the workspace is poisoned by an earlier failure
__device__ __forceinline__ unsigned sweep_enter(unsigned* sync, int* lflag, unsigned NS) {
  const unsigned e = __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sync + 8, NS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the host's reading of [4]-[6])
  if (threadIdx.x == 0) *lflag = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1 : 0;
  return (e + 1u) * NS;
}
// last thing: the workgroup that leaves last closes the epoch (every workgroup read it when it entered, and none can have
// left before all had entered their first step: they wait for each other)
__device__ __forceinline__ void sweep_leave(unsigned* sync) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gone = __hip_atomic_fetch_add(sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == gridDim.x - 1u) {
      __hip_atomic_store(sync + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


}  // namespace ark
