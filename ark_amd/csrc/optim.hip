// Fused Adam over the flat parameter buffer, bias-gradient column sums, dropout masks.
// Replaces torch.optim.Adam.step (reference kgvae/experiments/ablation_study.py:571,76) and the
// bias / broadcast reductions autograd performs for nn.Linear / nn.GRU biases.
#include "common.h"
#include "../../include/ark_amd.h"

namespace ark {

// advance the optimiser step counter and refresh the bias corrections (1 thread; keeps the step
// state on the device so a captured graph replays correctly)
__global__ void adam_tick_kernel(float* hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = hyper[ARK_HP_ADAM_STEP] + 1.0f;
    hyper[ARK_HP_ADAM_STEP] = t;
    hyper[ARK_HP_ADAM_BC1] = 1.0f - powf(hyper[ARK_HP_ADAM_B1], t);
    hyper[ARK_HP_ADAM_BC2] = 1.0f - powf(hyper[ARK_HP_ADAM_B2], t);
  }
}

// torch.optim.Adam (no weight decay / amsgrad), same operation order as torch's single-tensor path:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// 16 B/lane streams: 28 B of HBM traffic per parameter (read p,g,m,v; write p,m,v).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ hyper) {
  const float lr = hyper[ARK_HP_LR], b1 = hyper[ARK_HP_ADAM_B1], b2 = hyper[ARK_HP_ADAM_B2];
  const float eps = hyper[ARK_HP_ADAM_EPS], gs = hyper[ARK_HP_GRAD_SCALE];
  const float step_size = lr / hyper[ARK_HP_ADAM_BC1];
  const float inv_sqrt_bc2 = 1.0f / sqrtf(hyper[ARK_HP_ADAM_BC2]);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gs;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
      vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
      pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps));
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gg = g[i] * gs;
    const float mm = b1 * m[i] + (1.0f - b1) * gg;
    const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

// out[batch, n] (+)= sum_m X[batch, m, n]   64 columns x row chunks per workgroup, LDS cross-wave
// reduction, one float atomic per (workgroup, column).  `out` is zeroed by the host wrapper.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, long batch_stride_in,
                                                     float* __restrict__ out, long batch_stride_out, int M, int N,
                                                     int rows_per_wg) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
  const float* x = X + (long)blockIdx.z * batch_stride_in;
  float s = 0.f;
  if (col < N)
    for (int r = m0 + wave; r < m1; r += 4) s += x[(long)r * ld + col];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) {
    s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    atomicAdd(&out[(long)blockIdx.z * batch_stride_out + col], s);
  }
}

// counter-based dropout mask: mask[i] = keep ? 1/(1-p) : 0, keep ~ Bernoulli(1-p) from a
// splitmix64 hash of (seed, offset + i).  Statistically equivalent to torch's inter-layer GRU
// dropout, not bit-identical (the reference draws from the CPU/cuRAND generator).
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long n, float p, uint64_t seed,
                                                           const float* __restrict__ hyper) {
  // the optimiser step counter is folded into the stream so a replayed graph draws fresh masks
  const uint64_t step = (uint64_t)hyper[ARK_HP_ADAM_STEP];
  const float keep_scale = 1.0f / (1.0f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    mask[i] = dropout_keep_scale(seed, step, (uint64_t)i, p, keep_scale);
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

}  // namespace ark

extern "C" int ark_version(void) { return 100; }

extern "C" int ark_adam_tick(float* hyper, void* stream) {
  if (!hyper) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream) {
  if (!p || !g || !m || !v || !hyper || n <= 0) return ARK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15) return ARK_ERR_ALIGN;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ark::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_colsum(const float* x, int64_t ld, int64_t batch_stride_in, float* out, int64_t batch_stride_out,
                          int M, int N, int n_batch, int accumulate, void* stream) {
  if (!x || !out || M <= 0 || N <= 0 || n_batch <= 0) return ARK_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (accumulate) {
    // out already holds the value to add to (e.g. the zeroed gradient buffer)
  } else if (n_batch == 1 || batch_stride_out == N) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N * n_batch, st);
    if (e != hipSuccess) return (int)e;
  } else {
    for (int b = 0; b < n_batch; ++b) {
      hipError_t e = hipMemsetAsync(out + b * batch_stride_out, 0, sizeof(float) * (size_t)N, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  int rows_per_wg = 128;
  const int col_tiles = (N + 63) / 64;
  while (rows_per_wg > 16 && (long)col_tiles * ((M + rows_per_wg - 1) / rows_per_wg) * n_batch < 512) rows_per_wg >>= 1;
  hipLaunchKernelGGL(ark::colsum_kernel, dim3(col_tiles, (M + rows_per_wg - 1) / rows_per_wg, n_batch), dim3(256), 0, st, x,
                     (long)ld, (long)batch_stride_in, out, (long)batch_stride_out, M, N, rows_per_wg);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const float* hyper, void* stream) {
  if (!mask || !hyper || n <= 0 || p < 0.f || p >= 1.f) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ark::dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, (long)n, p,
                     seed, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ark::mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n);
  ARK_LAUNCH_CHECK();
  return 0;
}
