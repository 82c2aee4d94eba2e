// Micro-benchmark: does the L2 -> LDS stream of ONE workgroup per CU depend on how many
// WAVES issue the LDS-DMA instructions?  One workgroup per CU (grid 256), NW waves, ring of NBUF slots of 128 rows x 128 B
// (16 one-KB pieces per stage, dealt over the waves), NBUF - 1 stages in flight, counted vmcnt + one barrier per stage,
// no consumer.  Rows are 1 KB apart (a [rows, 512] 16-bit operand), 3 MB window (L2 / Infinity-Cache resident).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_waves_bench.hip -o /tmp/dwb && /tmp/dwb
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NW, int NBUF, int ROWS>
__global__ __launch_bounds__(64 * NW) void dma_waves_kernel(const char* src, long rs, int n_rows_total, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE = ROWS * 128, PIECES = ROWS / 8, PPW = PIECES / NW;
  static_assert(PIECES % NW == 0, "even dealing");
  const long row0 = ((long)blockIdx.x * ROWS) % n_rows_total;
  auto issue = [&](int s) {
    char* dst = lds + (s % NBUF) * STAGE;
    const long koff = ((long)s * 128) % rs;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + NW * i;
      const long r = (row0 + piece * 8 + (lane >> 3)) % n_rows_total;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * rs + koff + (lane & 7) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < NBUF - 1 && s < iters; ++s) issue(s);
  for (int s = 0; s < iters; ++s) {
    if (s + NBUF - 1 < iters) {
      issue(s + NBUF - 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * PPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0 && sink) sink[blockIdx.x] = lds[0];
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

template <int NW, int NBUF, int ROWS>
static void run(const char* buf, int* sink, int wgs) {
  const long rs = 1024;
  const int n_rows = (int)((3L << 20) / rs);
  const int iters = 256;
  const int ldsb = NBUF * ROWS * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_waves_kernel<NW, NBUF, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  float us = time_kernel([&] { hipLaunchKernelGGL((dma_waves_kernel<NW, NBUF, ROWS>), dim3(wgs), dim3(64 * NW), ldsb, 0, buf, rs, n_rows, iters, sink); }, 5);
  const double bytes = (double)wgs * iters * ROWS * 128;
  printf("waves %2d  slots %d x %3d KB (in flight %3d KB)  WGs %4d : %8.1f us  %6.2f TB/s chip  %6.1f GB/s per CU\n", NW, NBUF,
         ROWS * 128 / 1024, (NBUF - 1) * ROWS * 128 / 1024, wgs, us, bytes / us / 1e6, bytes / us / 1e3 / 256.0);
}

int main() {
  char* buf; int* sink;
  CK(hipMalloc(&buf, (4L << 20)));
  CK(hipMemset(buf, 1, (4L << 20)));
  CK(hipMalloc(&sink, 4096 * sizeof(int)));
  printf("--- one workgroup per CU, 7 x 16 KB in flight, waves varied\n");
  run<2, 8, 128>(buf, sink, 256);
  run<4, 8, 128>(buf, sink, 256);
  run<8, 8, 128>(buf, sink, 256);
  run<16, 8, 128>(buf, sink, 256);
  printf("--- one workgroup per CU, 4 waves, depth varied\n");
  run<4, 2, 128>(buf, sink, 256);
  run<4, 3, 128>(buf, sink, 256);
  run<4, 5, 128>(buf, sink, 256);
  printf("--- one workgroup per CU, 8 waves, depth varied\n");
  run<8, 2, 128>(buf, sink, 256);
  run<8, 3, 128>(buf, sink, 256);
  run<8, 5, 128>(buf, sink, 256);
  printf("--- three workgroups per CU (4 waves each), 2 x 16 KB ring each\n");
  run<4, 2, 128>(buf, sink, 768);
  run<4, 3, 128>(buf, sink, 768);
  printf("--- 96 workgroups (one per CU on 96 CUs), as the wd-articles launches\n");
  run<4, 8, 128>(buf, sink, 96);
  run<8, 8, 128>(buf, sink, 96);
  run<16, 8, 128>(buf, sink, 96);
  return 0;
}
