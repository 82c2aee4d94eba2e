// Micro-benchmark behind DESIGN.md section 6: how fast can ONE CU pull L2-resident operand panels
//   (a) through LDS-DMA (global_load_lds_dwordx4, what the ring engines use), and
//   (b) through plain global_load_dwordx4 into VGPRs,
// as a function of workgroups per CU and bytes kept in flight?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/l2_stream_bench.hip -o /tmp/l2b && /tmp/l2b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every workgroup streams `iters` stages of STAGE bytes from a `span`-byte window (L2-resident when small)
template <int STAGE, int NBUF>
__global__ __launch_bounds__(256) void dma_kernel(const char* src, long span, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int PIECES = STAGE / 1024, PPW = PIECES / 4;
  long base = ((long)blockIdx.x * 7919 * 1024) % span;
  auto issue = [&](int s) {
    char* dst = lds + (s % NBUF) * STAGE;
    const long off = (base + (long)s * STAGE) % span;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + piece * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < NBUF - 1 && s < iters; ++s) issue(s);
  for (int s = 0; s < iters; ++s) {
    if (s + NBUF - 1 < iters) issue(s + NBUF - 1);
    // wait for stage s: allow the younger NBUF-1 stages to stay in flight
    if (s + NBUF - 1 < iters) { if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                                else if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
                                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0 && sink) sink[blockIdx.x] = lds[0];
}

// operand-panel pattern of the ring engines: a 1-KB piece = 8 rows x 128 B, rows `rs` bytes apart; a stage is
// ROWS rows x 128 B; consecutive stages advance 128 B along the rows (the k direction)
template <int ROWS, int NBUF>
__global__ __launch_bounds__(256) void dma_rows_kernel(const char* src, long rs, int n_rows_total, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int STAGE = ROWS * 128, PIECES = ROWS / 8, PPW = PIECES / 4;
  const long row0 = ((long)blockIdx.x * ROWS) % n_rows_total;
  auto issue = [&](int s) {
    char* dst = lds + (s % NBUF) * STAGE;
    const long koff = ((long)s * 128) % rs;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + 4 * i;
      const long r = (row0 + piece * 8 + (lane >> 3)) % n_rows_total;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * rs + koff + (lane & 7) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < NBUF - 1 && s < iters; ++s) issue(s);
  for (int s = 0; s < iters; ++s) {
    if (s + NBUF - 1 < iters) issue(s + NBUF - 1);
    if (s + NBUF - 1 < iters) { if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0 && sink) sink[blockIdx.x] = lds[0];
}

// same traffic through VGPRs: each thread keeps DEPTH 16-byte loads in flight
template <int STAGE, int DEPTH>
__global__ __launch_bounds__(256) void reg_kernel(const char* src, long span, int iters, int* sink) {
  const int tid = threadIdx.x;
  long base = ((long)blockIdx.x * 7919 * 1024) % span;
  constexpr int PER_T = STAGE / (256 * 16);   // 16-byte loads per thread per stage
  int4 acc = {0, 0, 0, 0};
  for (int s = 0; s < iters; s += DEPTH) {
    int4 v[DEPTH][PER_T];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long off = (base + (long)(s + d) * STAGE) % span;
#pragma unroll
      for (int i = 0; i < PER_T; ++i) v[d][i] = *reinterpret_cast<const int4*>(src + off + (long)(i * 256 + tid) * 16);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < PER_T; ++i) { acc.x ^= v[d][i].x; acc.y ^= v[d][i].y; acc.z ^= v[d][i].z; acc.w ^= v[d][i].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345 && sink) sink[blockIdx.x] = 1;
}

template <class K>
static float time_kernel(K launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  const long span_small = 3L << 20, span_big = 512L << 20;   // 3 MB (L2-resident per XCD) / 512 MB (HBM)
  char* buf; int* sink;
  CK(hipMalloc(&buf, span_big + (1 << 20)));
  CK(hipMemset(buf, 1, span_big + (1 << 20)));
  CK(hipMalloc(&sink, 4096 * sizeof(int)));
  const int iters = 512;
  printf("%-44s %8s %10s %12s\n", "case", "us", "TB/s chip", "GB/s per CU");
  auto report = [&](const char* name, float us, long wgs, long stage) {
    const double bytes = (double)wgs * iters * stage;
    printf("%-44s %8.1f %10.2f %12.1f\n", name, us, bytes / us / 1e6, bytes / us / 1e3 / 256.0);
  };
  for (long span : {span_small, span_big}) {
    printf("--- window %ld MB\n", span >> 20);
    for (int wgs : {256, 512, 768}) {
      char nm[128];
      hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<32768, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      snprintf(nm, sizeof nm, "LDS-DMA 32KB x2 ring, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((dma_kernel<32768, 2>), dim3(wgs), dim3(256), 65536, 0, buf, span, iters, sink); }, 5), wgs, 32768);
      hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<16384, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
      snprintf(nm, sizeof nm, "LDS-DMA 16KB x3 ring, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((dma_kernel<16384, 3>), dim3(wgs), dim3(256), 49152, 0, buf, span, iters, sink); }, 5), wgs, 16384);
      hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<16384, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      snprintf(nm, sizeof nm, "LDS-DMA 16KB x4 ring, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((dma_kernel<16384, 4>), dim3(wgs), dim3(256), 65536, 0, buf, span, iters, sink); }, 5), wgs, 16384);
      snprintf(nm, sizeof nm, "VGPR loads 16KB stage depth 2, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((reg_kernel<16384, 2>), dim3(wgs), dim3(256), 0, 0, buf, span, iters, sink); }, 5), wgs, 16384);
      snprintf(nm, sizeof nm, "VGPR loads 16KB stage depth 4, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((reg_kernel<16384, 4>), dim3(wgs), dim3(256), 0, 0, buf, span, iters, sink); }, 5), wgs, 16384);
      snprintf(nm, sizeof nm, "VGPR loads 32KB stage depth 4, %d WGs", wgs);
      report(nm, time_kernel([&] { hipLaunchKernelGGL((reg_kernel<32768, 4>), dim3(wgs), dim3(256), 0, 0, buf, span, iters, sink); }, 5), wgs, 32768);
    }
  }
  printf("--- operand-panel pattern: 160 rows x 128 B per stage (64 + 96), x2 ring; rows rs bytes apart\n");
  for (long rs : {1024L, 3072L, 1024L + 128L}) {
    for (int wgs : {256, 512, 768}) {
      char nm[128];
      const int n_rows = (int)((3L << 20) / rs);   // ~3 MB of rows: L2-resident
      hipFuncSetAttribute(reinterpret_cast<const void*>(dma_rows_kernel<160, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
      snprintf(nm, sizeof nm, "rows rs=%ld, %d WGs", rs, wgs);
      const int it = (int)(rs / 128) * 8;
      float us = time_kernel([&] { hipLaunchKernelGGL((dma_rows_kernel<160, 2>), dim3(wgs), dim3(256), 40960, 0, buf, rs, n_rows, it, sink); }, 5);
      const double bytes = (double)wgs * it * 160 * 128;
      printf("%-44s %8.1f %10.2f %12.1f\n", nm, us, bytes / us / 1e6, bytes / us / 1e3 / 256.0);
    }
  }
  return 0;
}
