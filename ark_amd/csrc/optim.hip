// Fused Adam over the flat parameter buffer, bias-gradient column sums, dropout masks.
// Replaces torch.optim.Adam.step (reference kgvae/experiments/ablation_study.py:571,76) and the
// bias / broadcast reductions autograd performs for nn.Linear / nn.GRU biases.
#include "gemm_core.h"
#include "../../include/ark_amd.h"

namespace ark {

static_assert(kHpDropStep == ARK_HP_DROP_STEP, "common.h and ark_amd.h disagree on the dropout-draw slot");

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

// counter-based dropout mask: mask[i] = keep ? 1/(1-p) : 0, keep ~ Bernoulli(1-p) from the counter hash of
// common.h (seed, draw counter hyper[ARK_HP_DROP_STEP], element index).  Statistically equivalent to torch's
// inter-layer GRU dropout, not bit-identical (the reference draws from the CPU/cuRAND generator).
// n % 4 == 0: one hash serves a quad of consecutive elements.
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long n, float p, uint64_t seed,
                                                           const float* __restrict__ hyper) {
  const DropCtx dc = drop_ctx(seed, hyper, p);
  const long n4 = n >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(mask)[q] = dropout_quad(dc, (uint64_t)q);
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

// ---- 16-bit shadows (plain and transposed) of a list of fp32 matrices, one launch ------------
struct ShadowJob { const float* src; void* dst; void* dstT; int R, C, prec, precT, ldT; };   // dstT[c*ldT + r]
struct ShadowJobs { ShadowJob j[12]; int n; };

__global__ __launch_bounds__(256) void weight_shadow_kernel(ShadowJobs jobs) {
  __shared__ float tile[32][33];
  const ShadowJob jb = jobs.j[blockIdx.z];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (r0 >= jb.R || c0 >= jb.C) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < jb.R && c < jb.C) {
      v = jb.src[(long)r * jb.C + c];
      if (jb.dst) put16(jb.dst, (long)r * jb.C + c, v, jb.prec);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (jb.dstT)
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (r < jb.R && c < jb.C) put16(jb.dstT, (long)c * jb.ldT + r, tile[tx][i], jb.precT);
    }
}

// ---- Adam over the flat buffer WITH the 16-bit weight shadows written from the same registers ----------
// A job is either a LINEAR range of the flat buffer (embeddings, biases: 1024 floats per workgroup) or a
// MATRIX [R, C] whose plain / transposed 16-bit shadows the next forward / backward stream by LDS-DMA: a
// workgroup updates one 32x32 tile (128-byte row segments of p, g, m, v), stores the plain 16-bit copy from
// registers and the transposed one through LDS.  Round 1 ran a separate shadow launch that re-read every
// parameter Adam had just written (114 MB, 24 us per step).
struct AdamJob { long off; void* dst; void* dstT; int R, C, prec, precT, ldT, tile0; };   // R == 0: linear, C = length
struct AdamJobs { AdamJob j[ARK_ADAM_MAX_JOBS]; int n; };

struct AdamScalars { float b1, b2, eps, gs, step_size, inv_sqrt_bc2; };
__device__ __forceinline__ AdamScalars adam_scalars(const float* hyper) {
  AdamScalars a;
  a.b1 = hyper[ARK_HP_ADAM_B1]; a.b2 = hyper[ARK_HP_ADAM_B2]; a.eps = hyper[ARK_HP_ADAM_EPS]; a.gs = hyper[ARK_HP_GRAD_SCALE];
  a.step_size = hyper[ARK_HP_LR] / hyper[ARK_HP_ADAM_BC1];
  a.inv_sqrt_bc2 = 1.0f / sqrtf(hyper[ARK_HP_ADAM_BC2]);
  return a;
}
// G16: the gradient is read from a bf16 buffer (the all-reduced transport copy of a data-parallel bucket) instead of the fp32
// one -- what ark_uncast16 + this kernel did in two passes (72 MB less HBM traffic per step at syn-paths)
template <bool G16>
__device__ __forceinline__ f32x4 adam_quad(const AdamScalars& a, float* p, const void* g, float* m, float* v, long i) {
  f32x4 pp = *reinterpret_cast<f32x4*>(p + i);
  f32x4 gg;
  if constexpr (G16) {
    const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(g) + i);
    gg = f32x4{(float)g4[0], (float)g4[1], (float)g4[2], (float)g4[3]} * a.gs;
  } else {
    gg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + i) * a.gs;
  }
  f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mm[e] = a.b1 * mm[e] + (1.0f - a.b1) * gg[e];
    vv[e] = a.b2 * vv[e] + (1.0f - a.b2) * gg[e] * gg[e];
    pp[e] -= a.step_size * (mm[e] / (sqrtf(vv[e]) * a.inv_sqrt_bc2 + a.eps));
  }
  *reinterpret_cast<f32x4*>(p + i) = pp;
  *reinterpret_cast<f32x4*>(m + i) = mm;
  *reinterpret_cast<f32x4*>(v + i) = vv;
  return pp;
}

// store 4 consecutive 16-bit elements of run-time type
__device__ __forceinline__ void put16_quad(void* base, long e, f32x4 x, int prec) {
  if (prec == PREC_F16) {
    using PT = PrecTraits<PREC_F16>;
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(base) + e) = f16x4{PT::cvt(x[0]), PT::cvt(x[1]), PT::cvt(x[2]), PT::cvt(x[3])};
  } else {
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + e) = bf16x4{(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
  }
}

// Round 5: a workgroup takes a 64 x 64 tile (a linear job: 4096 floats), FOUR quads per thread with all sixteen 16-byte
// loads (p, g, m, v of the four quads) in flight before the first update -- round 4's 32 x 32 tiles kept four loads in
// flight per thread and touched memory in 128-byte (fp32) / 64-byte (16-bit) row segments: 3.85 TB/s.  Now every access
// of a wave-instruction is 256 contiguous bytes of a row (fp32), 128 (plain 16-bit copy) or 128 (transposed copy: 64
// source rows of one column).
constexpr int kAdamTile = 64;
template <bool G16>
__device__ __forceinline__ void adam_load(const float* p, const void* g, const float* m, const float* v, long i, f32x4& pp,
                                          f32x4& gg, f32x4& mm, f32x4& vv) {
  pp = *reinterpret_cast<const f32x4*>(p + i);
  if constexpr (G16) {
    const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(g) + i);
    gg = f32x4{(float)g4[0], (float)g4[1], (float)g4[2], (float)g4[3]};
  } else {
    gg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + i);
  }
  mm = *reinterpret_cast<const f32x4*>(m + i);
  vv = *reinterpret_cast<const f32x4*>(v + i);
}
__device__ __forceinline__ void adam_update(const AdamScalars& a, f32x4& pp, f32x4 gg, f32x4& mm, f32x4& vv) {
  gg *= a.gs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mm[e] = a.b1 * mm[e] + (1.0f - a.b1) * gg[e];
    vv[e] = a.b2 * vv[e] + (1.0f - a.b2) * gg[e] * gg[e];
    pp[e] -= a.step_size * (mm[e] / (sqrtf(vv[e]) * a.inv_sqrt_bc2 + a.eps));
  }
}

template <bool G16>
__global__ __launch_bounds__(256, 4) void adam_shadow_kernel(float* __restrict__ p, const void* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, AdamJobs jobs,
                                                          const float* __restrict__ hyper) {
  __shared__ float tile[kAdamTile][kAdamTile + 1];
  int ji = 0;
  for (int k = 1; k < jobs.n; ++k)
    if ((int)blockIdx.x >= jobs.j[k].tile0) ji = k;
  const AdamJob jb = jobs.j[ji];
  const int t = blockIdx.x - jb.tile0;
  const AdamScalars a = adam_scalars(hyper);
  f32x4 pp[4], gg[4], mm[4], vv[4];
  long idx[4];
  bool ok[4];
  if (jb.R == 0) {   // linear range, length jb.C (% 4 == 0): 4096 floats per workgroup
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long e = (long)t * 4096 + 1024 * i + 4 * threadIdx.x;
      ok[i] = e < jb.C;
      idx[i] = jb.off + e;
    }
  } else {
    const int tiles_c = (jb.C + kAdamTile - 1) / kAdamTile;
    const int r0 = (t / tiles_c) * kAdamTile, c0 = (t % tiles_c) * kAdamTile;
    const int ty = threadIdx.x >> 4, c = c0 + (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 16 * i;
      ok[i] = r < jb.R && c < jb.C;   // C % 4 == 0 (host-checked): the quad is whole
      idx[i] = jb.off + (long)r * jb.C + c;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pp[i] = gg[i] = mm[i] = vv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok[i]) adam_load<G16>(p, g, m, v, idx[i], pp[i], gg[i], mm[i], vv[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!ok[i]) continue;
    adam_update(a, pp[i], gg[i], mm[i], vv[i]);
    *reinterpret_cast<f32x4*>(p + idx[i]) = pp[i];
    *reinterpret_cast<f32x4*>(m + idx[i]) = mm[i];
    *reinterpret_cast<f32x4*>(v + idx[i]) = vv[i];
    if (jb.R != 0 && jb.dst) put16_quad(jb.dst, idx[i] - jb.off, pp[i], jb.prec);
  }
  if (jb.R == 0 || !jb.dstT) return;   // block-uniform
  {
    const int ty = threadIdx.x >> 4, tx4 = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[ty + 16 * i][tx4 + e] = pp[i][e];
  }
  __syncthreads();
  // transposed copy: thread -> column c0 + ty + 16 i of the source tile, 4 consecutive source rows r0 + tx4 .. + 3
  const int tiles_c = (jb.C + kAdamTile - 1) / kAdamTile;
  const int r0 = (t / tiles_c) * kAdamTile, c0 = (t % tiles_c) * kAdamTile;
  const int ty = threadIdx.x >> 4, tx4 = (threadIdx.x & 15) * 4;
  const int rT = r0 + tx4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = ty + 16 * i, cT = c0 + cl;
    if (cT >= jb.C || rT >= jb.R) continue;
    const long eT = (long)cT * jb.ldT + rT;   // ldT % 4 == 0 (host-checked): 8-byte aligned
    const f32x4 x = {tile[tx4][cl], tile[tx4 + 1][cl], tile[tx4 + 2][cl], tile[tx4 + 3][cl]};
    if (rT + 3 < jb.R) {
      put16_quad(jb.dstT, eT, x, jb.precT);
    } else {   // last rows of a matrix whose row count is not a multiple of 4 (e.g. a 55-token vocabulary)
      for (int e = 0; e < 4 && rT + e < jb.R; ++e) put16(jb.dstT, eT + e, x[e], jb.precT);
    }
  }
}

}  // namespace ark

extern "C" int ark_version(void) { return 210; }

namespace ark {
__global__ void stamp_kernel(unsigned long long* buf, int slot) {
  if (threadIdx.x == 0) buf[slot] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole device
}
}  // namespace ark
// Diagnostics (tools/step_stamps.py): buf[slot] = the device's 100-MHz real-time counter when this one-thread launch
// runs, i.e. right after everything queued before it on `stream`.  Captured with the step, it gives the end time of every
// launch of a replay without a profiler in the process (rocprofv3 stretches the two-queue parts of a captured step up to 2x).
extern "C" int ark_stamp(unsigned long long* buf, int slot, void* stream) {
  if (!buf || slot < 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, slot);
  ARK_LAUNCH_CHECK();
  return 0;
}

// Adam (as ark_adam_step) over the jobs' ranges of the flat buffers, writing the 16-bit weight shadows of
// every matrix job from the updated values.  Job i: R[i] == 0 -> linear range [off[i], off[i] + C[i]);
// else matrix [R[i], C[i]] at off[i] with optional plain (dst, prec) / transposed (dstT, precT, ldT) shadows.
static int adam_step_shadows_impl(float* p, const void* g, bool g16, float* m, float* v, int n_jobs, const int64_t* off,
                                  const int* R, const int* C, void* const* dst, void* const* dstT, const int* prec,
                                  const int* precT, const int* ldT, const float* hyper, void* stream) {
  using namespace ark;
  if (!p || !g || !m || !v || !hyper || n_jobs <= 0 || n_jobs > ARK_ADAM_MAX_JOBS || !off || !R || !C) return ARK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15) return ARK_ERR_ALIGN;
  AdamJobs jobs;
  jobs.n = n_jobs;
  long tiles = 0;
  for (int i = 0; i < n_jobs; ++i) {
    if (off[i] < 0 || (off[i] & 3) || C[i] <= 0 || R[i] < 0 || (C[i] & 3)) return ARK_ERR_ALIGN;
    void* d = dst ? dst[i] : nullptr;
    void* dT = dstT ? dstT[i] : nullptr;
    int ld = 0;
    if (R[i] > 0) {
      if ((d || dT) && (!prec || !precT)) return ARK_ERR_ARG;
      if (d && prec[i] != PREC_F16 && prec[i] != PREC_BF16) return ARK_ERR_ARG;
      if (dT) {
        if (precT[i] != PREC_F16 && precT[i] != PREC_BF16) return ARK_ERR_ARG;
        ld = (ldT && ldT[i] > 0) ? ldT[i] : R[i];
        if (ld < R[i] || (ld & 3)) return ARK_ERR_ALIGN;
      }
    }
    jobs.j[i] = AdamJob{(long)off[i], R[i] > 0 ? d : nullptr, R[i] > 0 ? dT : nullptr, R[i], C[i], prec ? prec[i] : 0,
                        precT ? precT[i] : 0, ld, (int)tiles};
    tiles += R[i] > 0 ? (long)((R[i] + kAdamTile - 1) / kAdamTile) * ((C[i] + kAdamTile - 1) / kAdamTile) : ((long)C[i] + 4095) / 4096;
    if (tiles > 0x7fffffffL) return ARK_ERR_SHAPE;
  }
  if (g16) hipLaunchKernelGGL(adam_shadow_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p, g, m, v, jobs, hyper);
  else hipLaunchKernelGGL(adam_shadow_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p, g, m, v, jobs, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_adam_step_shadows(float* p, const float* g, float* m, float* v, int n_jobs, const int64_t* off,
                                     const int* R, const int* C, void* const* dst, void* const* dstT, const int* prec,
                                     const int* precT, const int* ldT, const float* hyper, void* stream) {
  return adam_step_shadows_impl(p, g, false, m, v, n_jobs, off, R, C, dst, dstT, prec, precT, ldT, hyper, stream);
}

// the same with the gradient read from a bf16 buffer g16 (same element offsets as the flat fp32 buffers): the all-reduced
// transport copy of a data-parallel gradient bucket goes straight into the update, no widening pass in between
extern "C" int ark_adam_step_shadows_g16(float* p, const void* g16, float* m, float* v, int n_jobs, const int64_t* off,
                                         const int* R, const int* C, void* const* dst, void* const* dstT, const int* prec,
                                         const int* precT, const int* ldT, const float* hyper, void* stream) {
  if (reinterpret_cast<uintptr_t>(g16) & 7) return ARK_ERR_ALIGN;
  return adam_step_shadows_impl(p, g16, true, m, v, n_jobs, off, R, C, dst, dstT, prec, precT, ldT, hyper, stream);
}


// up to 12 jobs: dst = cast(src [R,C]) in `prec`, dstT = cast(src^T [C,R]) in `precT` (either may be NULL)
extern "C" int ark_weight_shadows(int n_jobs, const float* const* src, void* const* dst, void* const* dstT, const int* R,
                                  const int* C, const int* prec, const int* precT, const int* ldT, void* stream) {
  using namespace ark;
  if (n_jobs <= 0 || n_jobs > 12 || !src || !dst || !dstT || !R || !C || !prec || !precT) return ARK_ERR_ARG;
  ShadowJobs jobs;
  jobs.n = n_jobs;
  int maxR = 0, maxC = 0;
  for (int i = 0; i < n_jobs; ++i) {
    if (!src[i] || R[i] <= 0 || C[i] <= 0) return ARK_ERR_ARG;
    const int ld = (ldT && ldT[i] > 0) ? ldT[i] : R[i];
    if (ld < R[i]) return ARK_ERR_ARG;
    jobs.j[i] = ShadowJob{src[i], dst[i], dstT[i], R[i], C[i], prec[i], precT[i], ld};
    if (R[i] > maxR) maxR = R[i];
    if (C[i] > maxC) maxC = C[i];
  }
  hipLaunchKernelGGL(weight_shadow_kernel, dim3((maxC + 31) / 32, (maxR + 31) / 32, n_jobs), dim3(256), 0,
                     (hipStream_t)stream, jobs);
  ARK_LAUNCH_CHECK();
  return 0;
}

namespace ark {
// N(0,1) by Box-Muller on two counter hashes per PAIR of elements (common.h fmix32); one workgroup, so the draw counter
// can be read by everybody before thread 0 bumps it
__global__ __launch_bounds__(1024) void normal_fill_kernel(float* __restrict__ out, long n, uint32_t s0, uint32_t s1,
                                                           float* __restrict__ hyper) {
  uint32_t* ctr = reinterpret_cast<uint32_t*>(hyper) + ARK_HP_NOISE_STEP;
  const uint32_t step = *ctr;
  __syncthreads();
  const long pairs = (n + 1) >> 1;
  const uint32_t hs = step_hash(step, s0, s1);   // (the draw counter hashed on its own: no linear (index, draw) aliasing)
  for (long i = threadIdx.x; i < pairs; i += blockDim.x) {
    uint32_t a = fmix32((uint32_t)i * 0x9E3779B1u + hs);
    a = fmix32(a ^ ((uint32_t)(i >> 32) * 0xC2B2AE3Du + s1));
    const uint32_t b = fmix32(a + 0x6C8E9CF5u);
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]: 24 bits
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;          // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    out[2 * i] = r * cs;
    if (2 * i + 1 < n) out[2 * i + 1] = r * sn;
  }
  if (threadIdx.x == 0) *ctr = step + 1u;
}
}  // namespace ark

extern "C" int ark_normal_fill(float* out, int64_t n, uint64_t seed, float* hyper, void* stream) {
  if (!out || !hyper || n <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::normal_fill_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, (long)n, (uint32_t)seed,
                     (uint32_t)(seed >> 32), hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

namespace ark {
// plain kernel (an ordinary node of a captured graph, ordered like every other launch): 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, long n_words) {
  const long n4 = n_words >> 2;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<u32x4*>(p)[i] = u32x4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[4 * n4 + threadIdx.x] = 0u;
}
}  // namespace ark

namespace ark {
__global__ __launch_bounds__(256) void copy_kernel(uint32_t* __restrict__ d, const uint32_t* __restrict__ s, long n_words) {
  const long n4 = n_words >> 2;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) d[4 * n4 + threadIdx.x] = s[4 * n4 + threadIdx.x];
}
}  // namespace ark

extern "C" int ark_copy(void* dst, const void* src, int64_t nbytes, void* stream) {
  if (!dst || !src || nbytes < 0 || (nbytes & 3) != 0) return ARK_ERR_ARG;
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) != 0) return ARK_ERR_ALIGN;
  if (nbytes == 0) return 0;
  const long words = nbytes >> 2;
  long blocks = (words / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ark::copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t*>(dst),
                     reinterpret_cast<const uint32_t*>(src), words);
  ARK_LAUNCH_CHECK();
  return 0;
}

namespace ark {
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
}  // namespace ark

extern "C" int ark_axpy(float* y, const float* x, int64_t n, float a, void* stream) {
  if (!y || !x || n <= 0) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ark::axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, x, (long)n, a);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_zero(void* ptr, int64_t nbytes, void* stream) {
  if (!ptr || nbytes < 0 || (nbytes & 3) != 0) return ARK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(ptr) & 15) != 0) return ARK_ERR_ALIGN;
  if (nbytes == 0) return 0;
  const long words = nbytes >> 2;
  long blocks = (words / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ark::zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t*>(ptr), words);
  ARK_LAUNCH_CHECK();
  return 0;
}

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
  if (n & 3) return ARK_ERR_SHAPE;
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
