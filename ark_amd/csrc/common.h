// Shared device/host helpers for the ark_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernels of the step's DEPENDENT CHAIN raise their waves' issue priority (s_setprio): in the tail they share CUs with the
// chip-filling weight-gradient / Adam launches of the side queues, whose older waves otherwise win the SIMD arbitration.
#ifndef ARK_NO_CHAIN_PRIO
#define ARK_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define ARK_CHAIN_PRIO() ((void)0)
#endif

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

// GELU / GELU' for the 16-bit-operand product epilogues (gemm16.hip): erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7,
// i.e. fp32 rounding noise) on the hardware exp2 / rcp -- 14 instructions where erff + expf + a division took ~90, which one
// workgroup per CU cannot hide; GELU' shares the exponential between the density and the erf.  The exact-fp32 mode
// (gemm.hip) keeps erff.
__device__ __forceinline__ void erf_parts(float x, float& erf_abs, float& e) {   // erf(|x| / sqrt 2), exp(-x^2 / 2)
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  erf_abs = fmaf(-poly, e, 1.0f);
}
__device__ __forceinline__ float gelu_fast(float x) {
  float ea, e;
  erf_parts(x, ea, e);
  return 0.5f * x * (1.0f + copysignf(ea, x));
}
__device__ __forceinline__ float dgelu_fast(float x) {
  float ea, e;
  erf_parts(x, ea, e);
  return fmaf(0.39894228040143267794f * x, e, 0.5f * (1.0f + copysignf(ea, x)));
}

// hardware transcendentals for the 16-bit-operand (mixed precision) GRU epilogues: v_exp_f32 / v_rcp_f32 are ~1 ulp,
// far inside the fp16 operand rounding of that path; 4-5 instructions instead of ~25 (expf + division) / ~40 (tanhf).
// The exact-fp32 path (gru.hip) keeps expf / tanhf.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2/(1 + e^{2x}); exp2 -> inf / 0 gives the +-1 limits
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// counter-based dropout: keep-scale of element i of a layer's [B*L, D] output for dropout draw `step`
// (hyper[ARK_HP_DROP_STEP], a uint32 the token gather bumps once per training forward).  One 64-bit hash
// serves FOUR consecutive elements (a quad = one lane's 4 accumulator rows in the tile-native layout),
// 16 bits each, so p is quantised to 1/65536.  One definition shared by the mask kernel, the forward cell
// epilogue and the backward cell, so both directions agree.  (Round 1 hashed every element with two
// splitmix64 rounds -- 64-bit multiplies -- ~80 VALU instructions per element.)
constexpr int kHpDropStep = 12;   // == ARK_HP_DROP_STEP (include/ark_amd.h; static_assert in optim.hip)
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// The draw counter is hashed on its own (hs) before it meets the element index: a LINEAR mix i*A + step*B would give two
// (index, step) pairs with di*A + ds*B = 0 (mod 2^32) the same stream, i.e. step s + 433 983 would replay step s's values
// shifted by 1 511 elements (found on the noise generator, which shared the construction).
struct DropCtx { uint32_t s0, s1, hs, p16; float ks; };
__device__ __forceinline__ uint32_t step_hash(uint32_t step, uint32_t s0, uint32_t s1) {
  return fmix32(fmix32(step ^ s1) + 0x85EBCA77u) + s0;
}
__device__ __forceinline__ DropCtx drop_ctx(uint64_t seed, const float* hyper, float p) {
  DropCtx c;
  c.s0 = (uint32_t)seed; c.s1 = (uint32_t)(seed >> 32);
  c.hs = step_hash(reinterpret_cast<const uint32_t*>(hyper)[kHpDropStep], c.s0, c.s1);
  c.p16 = (uint32_t)(p * 65536.0f + 0.5f);
  c.ks = 1.0f / (1.0f - p);
  return c;
}
__device__ __forceinline__ f32x4 dropout_quad(const DropCtx& c, uint64_t quad) {
  const uint32_t q0 = (uint32_t)quad, q1 = (uint32_t)(quad >> 32);
  uint32_t a = fmix32(q0 * 0x9E3779B1u + c.hs);
  a = fmix32(a ^ (q1 * 0xC2B2AE3Du + c.s1));
  const uint32_t b = fmix32(a + 0x6C8E9CF5u);
  f32x4 m;
  m[0] = ((a & 0xFFFFu) >= c.p16) ? c.ks : 0.f;
  m[1] = ((a >> 16) >= c.p16) ? c.ks : 0.f;
  m[2] = ((b & 0xFFFFu) >= c.p16) ? c.ks : 0.f;
  m[3] = ((b >> 16) >= c.p16) ? c.ks : 0.f;
  return m;
}
__device__ __forceinline__ float dropout_one(const DropCtx& c, uint64_t i) {
  const f32x4 m = dropout_quad(c, i >> 2);
  const int e = (int)(i & 3);
  return e == 0 ? m[0] : e == 1 ? m[1] : e == 2 ? m[2] : m[3];
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
