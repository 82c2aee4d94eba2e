// Micro-benchmark for DESIGN.md section 10 ("a layer per XCD"): what does ONE hand-off between two workgroups cost
//   (a) as the persistent sweeps do it today -- 16-byte write-through (sc1) payload store, vmcnt(0), agent-scope atomic add;
//       the consumer polls the counter (agent-scope load) and reads the payload with an sc1 load --
//   (b) with both partners on ONE XCD and everything kept in that XCD's L2 -- plain store, vmcnt(0), workgroup-scope atomic
//       add (executes in L2); the consumer polls with sc0 loads (L1 bypassed) and reads the payload with an sc0 load?
// Workgroup i lands on XCD i % 8 (observed dispatch order): partners {i, i + 8} share an XCD, partners {i, i + 1} do not;
// every workgroup reports its HW_REG_XCC_ID so the pairing can be checked.  A ping-pong of `iters` round trips per pair; every
// wait is bounded (a pair that stalls reports -1 instead of hanging).
// Build + run on the GPU box:   hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o /tmp/hb && /tmp/hb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// MODE 0: today's protocol (sc1 / agent scope).  MODE 1: L2-local (sc0 loads, plain stores, workgroup-scope atomics).
// pair p = workgroups (a, b); slot p owns two 128-byte lines of counters (a->b, b->a) and two of payload.
template <int MODE>
__global__ __launch_bounds__(64) void pingpong(unsigned* buf, int stride_partner, int iters, long long* ticks, int* xcc) {
  const int wg = blockIdx.x;
  const int span = 2 * stride_partner;                       // ids [k*span, k*span + stride) are "a", the next `stride` are "b"
  const int grp = wg / span, in = wg % span;
  const bool is_a = in < stride_partner;
  const int pair = grp * stride_partner + (is_a ? in : in - stride_partner);
  unsigned* base = buf + (long)pair * 128;                   // 512 bytes per pair
  unsigned* c_ab = base, *c_ba = base + 32;
  unsigned* pay_ab = base + 64, *pay_ba = base + 96;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[wg] = (int)(id & 0xf);
  }
  const __amdgpu_buffer_rsrc_t r_mine = rsrc(is_a ? pay_ab : pay_ba, 128), r_theirs = rsrc(is_a ? pay_ba : pay_ab, 128);
  unsigned* c_mine = is_a ? c_ab : c_ba, *c_theirs = is_a ? c_ba : c_ab;
  const int lane = threadIdx.x;
  long long t0 = 0;
  unsigned acc = 0;
  bool ok = true;
  auto signal = [&](unsigned v) {
    // payload: one 16-byte store per lane 0..7 (a whole 128-byte line), then the counter
    if (lane < 8) {
      const u32x4 d = {v, v + 1, v + 2, v + 3};
      __builtin_amdgcn_raw_buffer_store_b128(d, r_mine, lane * 16, 0, MODE == 0 ? 16 : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (MODE == 0) __hip_atomic_fetch_add(c_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(c_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto wait_for = [&](unsigned need) -> bool {
    const __amdgpu_buffer_rsrc_t rc = rsrc(c_theirs, 4);
    for (int spin = 0; spin < 2000000; ++spin) {
      unsigned c;
      if (MODE == 0) c = __hip_atomic_load(c_theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else {
        asm volatile("" ::: "memory");   // (a plain buffer load is loop-invariant to the compiler: re-issue it every spin)
        c = __builtin_amdgcn_raw_buffer_load_b32(rc, 0, 0, 1);   // aux 1 = sc0: L1 bypassed, served by the XCD's L2
      }
      if ((int)(c - need) >= 0) return true;
      __builtin_amdgcn_s_sleep(1);
    }
    return false;
  };
  for (int it = 0; it < iters + 16 && ok; ++it) {
    if (it == 16 && lane == 0) t0 = (long long)__builtin_amdgcn_s_memrealtime();
    if (is_a) {
      signal((unsigned)it);
      ok = wait_for((unsigned)it + 1);
    } else {
      ok = wait_for((unsigned)it + 1);
      if (ok) signal((unsigned)it);
    }
    if (ok && lane < 8) {   // the payload the partner sent with this signal
      const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r_theirs, lane * 16, 0, MODE == 0 ? 16 : 1);
      acc += d[0];
    }
  }
  if (lane == 0) ticks[wg] = ok ? (long long)__builtin_amdgcn_s_memrealtime() - t0 : -1;
  if (acc == 0xdeadbeefu && lane == 0) ticks[wg] = -2;   // (keeps the payload loads alive)
}

template <int MODE>
static int run(const char* what, int stride, int n_pairs_groups, int iters) {
  const int n_wg = 2 * stride * n_pairs_groups, n_pairs = stride * n_pairs_groups;
  unsigned* buf; long long* ticks; int* xcc;
  CK(hipMalloc(&buf, (size_t)n_pairs * 512));
  CK(hipMemset(buf, 0, (size_t)n_pairs * 512));
  CK(hipMalloc(&ticks, n_wg * sizeof(long long)));
  CK(hipMalloc(&xcc, n_wg * sizeof(int)));
  hipLaunchKernelGGL(pingpong<MODE>, dim3(n_wg), dim3(64), 0, 0, buf, stride, iters, ticks, xcc);
  CK(hipDeviceSynchronize());
  std::vector<long long> t(n_wg); std::vector<int> x(n_wg);
  CK(hipMemcpy(t.data(), ticks, n_wg * sizeof(long long), hipMemcpyDeviceToHost));
  CK(hipMemcpy(x.data(), xcc, n_wg * sizeof(int), hipMemcpyDeviceToHost));
  int same = 0, stalled = 0; std::vector<double> us;
  for (int g = 0; g < n_pairs_groups; ++g)
    for (int i = 0; i < stride; ++i) {
      const int a = g * 2 * stride + i, b = a + stride;
      same += x[a] == x[b];
      if (t[a] < 0 || t[b] < 0) { ++stalled; continue; }
      us.push_back((double)t[a] / 100.0 / iters);   // 100-MHz ticks -> us per round trip (two hand-offs)
    }
  std::sort(us.begin(), us.end());
  if (us.empty()) { printf("%-58s all %d pairs stalled\n", what, n_pairs); return 0; }
  printf("%-58s %3d pairs (%3d on one XCD, %d stalled): round trip median %.2f us (min %.2f, max %.2f) = %.2f us per hand-off\n", what,
         n_pairs, same, stalled, us[us.size() / 2], us.front(), us.back(), us[us.size() / 2] / 2);
  (void)hipFree(buf); (void)hipFree(ticks); (void)hipFree(xcc);
  return 0;
}

int main() {
  const int iters = 2000;
  // one pair at a time first (no contention), then 48 pairs at once (the sweep's 96 workgroups)
  for (int groups : {1, 6}) {
    printf("-- %d pair group(s)\n", groups);
    if (run<0>("sc1 + agent scope, partners i / i+1 (different XCDs)", 1, groups * 8, iters)) return 1;
    if (run<0>("sc1 + agent scope, partners i / i+8 (one XCD)", 8, groups, iters)) return 1;
    if (run<1>("sc0 + L2 atomics, partners i / i+8 (one XCD)", 8, groups, iters)) return 1;
  }
  return 0;
}
