// Shared device/host helpers for the ark_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ark {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

enum : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };
// Operand layouts.  KMAJ: element (row, k) lives at base[row * ld + k]  (reduction index contiguous).
//                   MMAJ: element (row, k) lives at base[k * ld + row]  (row index contiguous).
enum : int { LAY_KMAJ = 0, LAY_MMAJ = 1 };

constexpr int kWave = 64;  // CDNA wavefront width

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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// exact-erf GELU, as torch.nn.GELU() default (reference kgvae/model/models.py:37)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// counter-based dropout: keep-scale of element i of a layer's [B*L, D] output at optimiser step `step`.
// One definition shared by the mask kernel (register-staged path) and the in-kernel use of the
// LDS-DMA path (forward cell epilogue and the input-gradient product), so both directions agree.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float dropout_keep_scale(uint64_t seed, uint64_t step, uint64_t i, float p, float keep_scale) {
  const uint64_t h = splitmix64(seed ^ splitmix64(step * 0x100000001B3ull + i));
  const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
  return (u >= p) ? keep_scale : 0.f;
}

// XCD-aware workgroup remap (MI355X: 8 XCDs, private 4 MB L2 each; workgroups are dealt round-robin
// over XCDs, so ids b and b+8 share an L2).  Gives each XCD a CONTIGUOUS range of logical tile ids
// so neighbouring tiles -- which share an operand panel -- hit the same L2.  Bijective for any grid
// size; placement is a speed matter only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

}  // namespace ark

// Error plumbing for the C-ABI: 0 = ok, >0 = hipError_t, <0 = argument error.
#define ARK_ERR_ARG (-1)
#define ARK_ERR_SHAPE (-2)
#define ARK_ERR_ALIGN (-3)
#define ARK_LAUNCH_CHECK()                       \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)
