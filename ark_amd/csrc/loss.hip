// Latent head (clamp / reparameterise / KL), z-projection, token cross-entropy and the ELBO scalar.
// Replaces: models.py:61-63 (clamp, randn_like reparam), models.py:199-200 (kl_mean),
// models.py:139 (tanh(z_proj(z))), F.cross_entropy(ignore_index=PAD) + `ce + b*kl`
// (kgvae/experiments/ablation_study.py:65-71) and their autograd backward.
//
// Scalars that change from step to step live in a small DEVICE array `hyper` (see ark_amd.h
// ARK_HP_*), so a captured hipGraph of the whole step can be replayed unchanged.
#include "common.h"
#include "../../include/ark_amd.h"

namespace ark {

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (wave == 0) {
    t = (lane < (int)(blockDim.x >> 6)) ? red[lane] : 0.f;
    t = wave_sum(t);
  }
  __syncthreads();
  return t;  // valid in wave 0
}

// head[b, 0:Z] = mu, head[b, Z:2Z] = raw logv.  Single workgroup -> deterministic KL sum.
__global__ __launch_bounds__(1024) void latent_fwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                                          float* __restrict__ mu, float* __restrict__ logv,
                                                          float* __restrict__ z, float* __restrict__ kl_out, int B, int Z,
                                                          int clamp) {
  __shared__ float red[16];
  float acc = 0.f;
  const int n = B * Z;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = i / Z, j = i % Z;
    const float m = head[(long)b * 2 * Z + j];
    float lv = head[(long)b * 2 * Z + Z + j];
    if (clamp) lv = fminf(fmaxf(lv, -10.0f), 10.0f);   // (the MLP encoder clamps, models.py:62; the Transformer encoder does not, :93)
    mu[i] = m;
    logv[i] = lv;
    z[i] = m + (eps ? eps[i] : 0.f) * expf(0.5f * lv);
    acc += 1.0f + lv - m * m - expf(lv);
  }
  const float tot = block_sum_1024(acc, red);
  if (threadIdx.x == 0) kl_out[0] = -0.5f * tot / (float)n;
}

// dhead = d(beta*kl)/d(mu,logv) + dz routed through the reparameterisation; clamp passes gradient
// only where the raw logv lies inside [-10, 10] (torch.clamp semantics, boundary inclusive).
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ head,
                                                         const float* __restrict__ eps, const float* __restrict__ hyper,
                                                         float* __restrict__ dhead, int B, int Z, int clamp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Z) return;
  const int b = i / Z, j = i % Z;
  const float ks = hyper[ARK_HP_BETA] * hyper[ARK_HP_KL_NORM];  // beta / (B_global * Z)
  const float m = head[(long)b * 2 * Z + j];
  const float raw = head[(long)b * 2 * Z + Z + j];
  const float lv = clamp ? fminf(fmaxf(raw, -10.0f), 10.0f) : raw;
  const float g = dz[i];
  const float dmu = g + ks * m;
  float dlv = g * (eps ? eps[i] : 0.f) * 0.5f * expf(0.5f * lv) + ks * 0.5f * (expf(lv) - 1.0f);
  if (clamp && (raw < -10.0f || raw > 10.0f)) dlv = 0.f;
  dhead[(long)b * 2 * Z + j] = dmu;
  dhead[(long)b * 2 * Z + Z + j] = dlv;
}

// h0[b, d] = tanh(bz[d] + sum_j z[b,j] Wz[d,j]), written to n_copies layer slots
__global__ __launch_bounds__(256) void zproj_fwd_kernel(const float* __restrict__ z, const float* __restrict__ Wz,
                                                        const float* __restrict__ bz, float* __restrict__ h0,
                                                        long copy_stride, int n_copies, int B, int Z, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * D) return;
  const int b = (int)(i / D), d = (int)(i % D);
  float a = bz[d];
  for (int j = 0; j < Z; ++j) a += z[(long)b * Z + j] * Wz[(long)d * Z + j];
  const float h = tanhf(a);
  for (int c = 0; c < n_copies; ++c) h0[c * copy_stride + i] = h;
}

// dzp = dh0 * (1 - h0^2) (written back in place for the dWz kernel) and dz[b,j] = sum_d dzp[b,d] Wz[d,j].
// One wave per row b, lanes over d; all ZT >= Z latent accumulators stay in registers, so every
// dzp element and every Wz row (Z contiguous floats) is read exactly once.
template <int ZT>
__global__ __launch_bounds__(256) void zproj_bwd_dz_kernel(float* __restrict__ dh0, const float* __restrict__ h0,
                                                           const float* __restrict__ Wz, float* __restrict__ dz, int B, int Z,
                                                           int D) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc[ZT];
#pragma unroll
  for (int j = 0; j < ZT; ++j) acc[j] = 0.f;
  for (int d = lane; d < D; d += 64) {
    const long i = (long)b * D + d;
    const float h = h0[i];
    const float g = dh0[i] * (1.0f - h * h);
    dh0[i] = g;
    const float* wr = Wz + (long)d * Z;
#pragma unroll
    for (int j = 0; j < ZT; ++j)
      if (j < Z) acc[j] += g * wr[j];
  }
#pragma unroll
  for (int j = 0; j < ZT; ++j)
    if (j < Z) {
      const float a = wave_sum(acc[j]);
      if (lane == 0) dz[(long)b * Z + j] = a;
    }
}

// The whole per-row part of the latent backward in ONE launch (it sits on the dependent chain between
// the GRU's initial-state gradient and the encoder MLP backward):
//   dzp = dh0 * (1 - h0^2)              (in place, for the batch reductions that follow off-chain)
//   dz  = dzp Wz                        (z-projection, models.py:139)
//   dhead = d(beta*kl)/d(mu,logv) + dz through the reparameterisation (+ external dmu/dlogv)
//   dA  = (dhead W_head) * gelu'(pre)   (heads models.py:43-44, last MLP activation models.py:32-41)
// The reductions over the batch (dWz, dbz, db_head, dW_head) only need dzp / dhead and run elsewhere.
//
// RW = 4 rows per 256-thread workgroup, one wave per row for the [D] -> [Z] reduction (Wz^T staged once per
// workgroup in LDS, row stride D+1: conflict-free fill and reads), then all 256 threads form the [RW, H] block
// of dA with every W_head element loaded once per workgroup.  Round 1 used 2 rows per workgroup with half of
// the waves idle in the first phase and scalar, strided Wz reads: 39 us alone (104 us beside the weight
// gradients) for 31 MFLOP; this one is bound by its ~15 MB of HBM traffic.
template <int ZT, int RW, bool BATCHED>
__global__ __launch_bounds__(256) void latent_chain_bwd_kernel(float* __restrict__ dh0, const float* __restrict__ h0,
                                                               const float* __restrict__ Wz, const float* __restrict__ head,
                                                               const float* __restrict__ eps, const float* __restrict__ hyper,
                                                               const float* __restrict__ ext_dhead,
                                                               const float* __restrict__ Whead, const float* __restrict__ pre,
                                                               float* __restrict__ dhead, float* __restrict__ dA, void* dA16,
                                                               int prec16, float* __restrict__ dA_colsum, int B, int Z, int D,
                                                               int H) {
  ARK_CHAIN_PRIO();
  static_assert((ZT <= 64 && RW == 4) || (ZT > 64 && RW == 1), "one wave per row; the wide-latent form: one workgroup per row");
  extern __shared__ __attribute__((aligned(16))) char smem_lc[];
  float* zs = reinterpret_cast<float*>(smem_lc);   // Wz^T: [Z][D + 1]
  __shared__ float sh[RW][2 * ZT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int DS = D + 1;
  // BATCHED: every loop keeps a batch of loads in flight before it touches the results (with 4 waves per CU nobody
  // else hides a load's latency).  It costs registers (255 VGPRs against ~100), which matters where the kernel has to
  // squeeze in beside the weight-gradient launch: measured in situ, same box, alternating rounds -- wd-movies
  // (Z = 64, 64 workgroups) 2.50 -> 2.44 ms/step, syn-types (Z = 24) equal, syn-paths (Z = 10, 256 workgroups beside
  // 480 weight-gradient workgroups) 1.28 -> 1.30 ms/step.  So: batched for Z > 32 only.
  constexpr bool WIDE = ZT > 64;   // 64 < Z <= 128 (wd-articles): Wz^T does not fit LDS -- lanes own latent columns instead
  if constexpr (WIDE) {
  } else if constexpr (!BATCHED) {
    for (int i = threadIdx.x; i < D * Z; i += 256) zs[(i % Z) * DS + i / Z] = Wz[i];   // coalesced read, stride-(D+1) write
  } else {
    const int N4 = (D * Z) >> 2;   // D % 4 == 0 (host check) -> whole float4s; parameter blocks are 16-byte aligned
    const f32x4* W4 = reinterpret_cast<const f32x4*>(Wz);
    for (int base = threadIdx.x; base < N4; base += 256 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = base + 256 * u;
        v[u] = q < N4 ? W4[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = base + 256 * u;
        if (q < N4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            zs[(i % Z) * DS + i / Z] = v[u][e];   // row stride D+1: conflict-free fill and reads
          }
        }
      }
    }
  }
  const int b = WIDE ? (int)blockIdx.x : (int)blockIdx.x * RW + wave;
  const bool valid = b < B;
  const int bb = valid ? b : B - 1;
  __syncthreads();
  float mydz2[1] = {0.f};   // dz[b, lane]; WIDE: dz[b, thread] for threads 0..127
  if constexpr (WIDE) {
    // ONE row per workgroup.  dzp = dh0 (1 - h0^2) goes to LDS (and back to dh0); thread (j = t % 128, half = t / 128) sums
    // dzp[d] Wz[d, j] over its half of d with coalesced reads of Wz rows (L2-resident: 256 KB) -- no transposed copy of Wz,
    // sixteen loads in flight per thread; the two halves meet in LDS
    float* gs = zs;            // [D]
    float* ph = zs + D;        // [2][128] partial sums
    for (int d = threadIdx.x; d < D; d += 256) {
      const float h = h0[(long)bb * D + d];
      const float g = dh0[(long)bb * D + d] * (1.0f - h * h);
      if (valid) dh0[(long)bb * D + d] = g;
      gs[d] = g;
    }
    __syncthreads();
    const int zj = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int dlo = half * (D / 2), dhi = dlo + D / 2;
    float a0 = 0.f;
    if (zj < Z) {
#pragma unroll 16
      for (int d = dlo; d < dhi; ++d) a0 += gs[d] * Wz[(long)d * Z + zj];
    }
    ph[half * 128 + zj] = a0;
    __syncthreads();
    if (threadIdx.x < 128) mydz2[0] = ph[threadIdx.x] + ph[128 + threadIdx.x];
  }
  float acc[WIDE ? 1 : ZT];
#pragma unroll
  for (int j = 0; j < (WIDE ? 1 : ZT); ++j) acc[j] = 0.f;
  constexpr int DU = BATCHED ? 8 : 1;
  for (int d0 = lane; d0 < (WIDE ? 0 : D); d0 += 64 * DU) {
    float hv[DU], gv[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int d = d0 + 64 * u;
      hv[u] = d < D ? h0[(long)bb * D + d] : 0.f;
      gv[u] = d < D ? dh0[(long)bb * D + d] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int d = d0 + 64 * u;
      if (d < D) {
        const float g = gv[u] * (1.0f - hv[u] * hv[u]);
        if (valid) dh0[(long)bb * D + d] = g;
#pragma unroll
        for (int j = 0; j < (WIDE ? 1 : ZT); ++j)
          if (j < Z) acc[j] += g * zs[j * DS + d];
      }
    }
  }
  if constexpr (!WIDE) {
#pragma unroll
    for (int j = 0; j < ZT; ++j)
      if (j < Z) {
        const float a = wave_sum(acc[j]);
        if (lane == j) mydz2[0] = a;
      }
  }
  {
    const int zj = WIDE ? (int)threadIdx.x : lane;           // WIDE: threads 0..127 own the row's latent columns
    const int wv = WIDE ? 0 : wave;
    const bool mine = WIDE ? threadIdx.x < 128 : true;
    if (mine && zj < Z) {
    const float mydz = mydz2[0];
    const float ks = hyper[ARK_HP_BETA] * hyper[ARK_HP_KL_NORM];  // beta / (B_global * Z)
    const float m = head[(long)bb * 2 * Z + zj];
    const float raw = head[(long)bb * 2 * Z + Z + zj];
    const float lv = fminf(fmaxf(raw, -10.0f), 10.0f);
    float dmu = mydz + ks * m;
    float dlv = mydz * (eps ? eps[(long)bb * Z + zj] : 0.f) * 0.5f * expf(0.5f * lv) + ks * 0.5f * (expf(lv) - 1.0f);
    if (raw < -10.0f || raw > 10.0f) dlv = 0.f;
    if (ext_dhead) {
      dmu += ext_dhead[(long)bb * 2 * Z + zj];
      dlv += ext_dhead[(long)bb * 2 * Z + Z + zj];
    }
    sh[wv][zj] = dmu;
    sh[wv][Z + zj] = dlv;
    if (valid) {
      dhead[(long)bb * 2 * Z + zj] = dmu;
      dhead[(long)bb * 2 * Z + Z + zj] = dlv;
    }
    }
  }
  __syncthreads();
  if constexpr (!BATCHED) {
    // dA for the workgroup's RW rows: thread -> KC columns at a time (every W_head element is loaded once per
    // workgroup and used for all rows; each dhead value is one LDS broadcast per KC columns)
    const int row0 = blockIdx.x * RW;
    constexpr int KC = 3;
    for (int cb = 0; cb < H; cb += 256 * KC) {
      float acc4[KC][RW];
  #pragma unroll
      for (int k = 0; k < KC; ++k)
  #pragma unroll
        for (int r = 0; r < RW; ++r) acc4[k][r] = 0.f;
  #pragma unroll
      for (int j = 0; j < 2 * ZT; ++j)
        if (j < 2 * Z) {
          float dj[RW];
  #pragma unroll
          for (int r = 0; r < RW; ++r) dj[r] = sh[r][j];
  #pragma unroll
          for (int k = 0; k < KC; ++k) {
            const int c = cb + 256 * k + threadIdx.x;
            const float wv = (c < H) ? Whead[(long)j * H + c] : 0.f;
  #pragma unroll
            for (int r = 0; r < RW; ++r) acc4[k][r] += dj[r] * wv;
          }
        }
  #pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = cb + 256 * k + threadIdx.x;
        if (c >= H) continue;
        float cs = 0.f;
  #pragma unroll
        for (int r = 0; r < RW; ++r) {
          if (row0 + r >= B) break;
          const long o = (long)(row0 + r) * H + c;
          const float v = acc4[k][r] * dgelu_erf(pre[o]);
          dA[o] = v;
          cs += v;
          if (prec16 == 2) reinterpret_cast<_Float16*>(dA16)[o] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
          else reinterpret_cast<__bf16*>(dA16)[o] = (__bf16)v;
        }
        if (dA_colsum) atomicAdd(&dA_colsum[c], cs);
      }
    }
  } else {
    // dA for the workgroup's RW rows: thread -> KC columns at a time (every W_head element is loaded once per
    // workgroup and used for all rows; each dhead value is one LDS broadcast per KC columns); W_head in batches of
    // JC x KC loads, the pre-activations of the epilogue fetched before the products
    const int row0 = blockIdx.x * RW;
    constexpr int KC = 3, JC = 16;
    static_assert((2 * ZT) % JC == 0, "batches of head rows");
    for (int cb = 0; cb < H; cb += 256 * KC) {
      float acc4[KC][RW], pv[KC][RW];
  #pragma unroll
      for (int k = 0; k < KC; ++k)
  #pragma unroll
        for (int r = 0; r < RW; ++r) {
          acc4[k][r] = 0.f;
          const int c = cb + 256 * k + threadIdx.x;
          pv[k][r] = (c < H && row0 + r < B) ? pre[(long)(row0 + r) * H + c] : 0.f;
        }
  #pragma unroll 1
      for (int j0 = 0; j0 < 2 * Z; j0 += JC) {
        float wv[JC][KC];
  #pragma unroll
        for (int jj = 0; jj < JC; ++jj)
  #pragma unroll
          for (int k = 0; k < KC; ++k) {
            const int c = cb + 256 * k + threadIdx.x;
            wv[jj][k] = (j0 + jj < 2 * Z && c < H) ? Whead[(long)(j0 + jj) * H + c] : 0.f;
          }
  #pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
          float dj[RW];
  #pragma unroll
          for (int r = 0; r < RW; ++r) dj[r] = (j0 + jj < 2 * Z) ? sh[r][j0 + jj] : 0.f;
  #pragma unroll
          for (int k = 0; k < KC; ++k)
  #pragma unroll
            for (int r = 0; r < RW; ++r) acc4[k][r] += dj[r] * wv[jj][k];
        }
      }
  #pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = cb + 256 * k + threadIdx.x;
        if (c >= H) continue;
        float cs = 0.f;
  #pragma unroll
        for (int r = 0; r < RW; ++r) {
          if (row0 + r >= B) break;
          const long o = (long)(row0 + r) * H + c;
          const float v = acc4[k][r] * dgelu_erf(pv[k][r]);
          dA[o] = v;
          cs += v;
          if (prec16 == 2) reinterpret_cast<_Float16*>(dA16)[o] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
          else reinterpret_cast<__bf16*>(dA16)[o] = (__bf16)v;
        }
        if (dA_colsum) atomicAdd(&dA_colsum[c], cs);
      }
    }
  }
}

// Phases 1-2 of latent_chain_bwd_kernel as a launch of their own, built for LATENCY (round 4: the fused kernel was 72-80 us
// of mostly waiting -- one wave per SIMD walking Wz^T into LDS 20 dependent loads deep, then D / 64 dependent row loads, then
// 2 x 20 x 3 dependent W_head loads -- for 31 MFLOP):
//   dzp = dh0 * (1 - h0^2) (in place), dz = dzp Wz, dhead = d(beta * kl)/d(mu, logv) + dz through the reparameterisation
// RW = 4 rows per workgroup, one wave per row; every load of a phase is in flight before the first use (Wz: D*Z/4/256
// float4s per thread; the row: D / 64 values of dh0 and h0 per lane); Wz^T sits in LDS with row stride D + 1 (conflict-free).
// The [B, 3D] phase (dA) follows as latent_dA_kernel on a 2-D grid.
template <int ZT, int DU>
__global__ __launch_bounds__(256) void latent_dz_head_kernel(float* __restrict__ dh0, const float* __restrict__ h0,
                                                             const float* __restrict__ Wz, const float* __restrict__ head,
                                                             const float* __restrict__ eps, const float* __restrict__ hyper,
                                                             const float* __restrict__ ext_dhead, float* __restrict__ dhead,
                                                             int B, int Z, int D) {
  ARK_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem_dz[];
  float* zs = reinterpret_cast<float*>(smem_dz);   // Wz^T: [Z][D + 1]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int DS = D + 1;
  const int b = (int)blockIdx.x * 4 + wave;
  const bool valid = b < B;
  const int bb = valid ? b : B - 1;
  // this row's operands first (they are what the wave's dot products wait for), then the shared Wz tile
  float hv[DU], gv[DU];
#pragma unroll
  for (int u = 0; u < DU; ++u) {
    const int d = lane + 64 * u;
    hv[u] = d < D ? h0[(long)bb * D + d] : 0.f;
    gv[u] = d < D ? dh0[(long)bb * D + d] : 0.f;
  }
  {
    const int N4 = (D * Z) >> 2;   // (D % 4 == 0: host check)
    const f32x4* W4 = reinterpret_cast<const f32x4*>(Wz);
    for (int base = threadIdx.x; base < N4; base += 256 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = base + 256 * u;
        v[u] = q < N4 ? W4[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = base + 256 * u;
        if (q < N4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            zs[(i % Z) * DS + i / Z] = v[u][e];
          }
        }
      }
    }
  }
  __syncthreads();
  float acc[ZT];
#pragma unroll
  for (int j = 0; j < ZT; ++j) acc[j] = 0.f;
#pragma unroll
  for (int u = 0; u < DU; ++u) {
    const int d = lane + 64 * u;
    if (d < D) {
      const float g = gv[u] * (1.0f - hv[u] * hv[u]);
      if (valid) dh0[(long)bb * D + d] = g;
#pragma unroll
      for (int j = 0; j < ZT; ++j)
        if (j < Z) acc[j] += g * zs[j * DS + d];
    }
  }
  float mydz = 0.f;
#pragma unroll
  for (int j = 0; j < ZT; ++j)
    if (j < Z) {
      const float a = wave_sum(acc[j]);
      if (lane == j) mydz = a;
    }
  if (lane < Z && valid) {
    const float ks = hyper[ARK_HP_BETA] * hyper[ARK_HP_KL_NORM];  // beta / (B_global * Z)
    const float m = head[(long)b * 2 * Z + lane];
    const float raw = head[(long)b * 2 * Z + Z + lane];
    const float lv = fminf(fmaxf(raw, -10.0f), 10.0f);
    float dmu = mydz + ks * m;
    float dlv = mydz * (eps ? eps[(long)b * Z + lane] : 0.f) * 0.5f * expf(0.5f * lv) + ks * 0.5f * (expf(lv) - 1.0f);
    if (raw < -10.0f || raw > 10.0f) dlv = 0.f;
    if (ext_dhead) {
      dmu += ext_dhead[(long)b * 2 * Z + lane];
      dlv += ext_dhead[(long)b * 2 * Z + Z + lane];
    }
    dhead[(long)b * 2 * Z + lane] = dmu;
    dhead[(long)b * 2 * Z + Z + lane] = dlv;
  }
}

// The last phase of latent_chain_bwd_kernel as a launch of its own, for batches whose B/4 row workgroups would leave most
// of the chip idle while each of them walks all H columns (syn-types B = 256: 64 workgroups, 224 us for 38 MFLOP):
//   dA[b, c] = (sum_j dhead[b, j] W_head[j, c]) * gelu'(pre[b, c]), its 16-bit copy and column sums
// 8 rows x 256 columns per workgroup, W_head read coalesced, one row of it per j.
__global__ __launch_bounds__(256) void latent_dA_kernel(const float* __restrict__ dhead, const float* __restrict__ Whead,
                                                        const float* __restrict__ pre, float* __restrict__ dA, void* dA16, int prec16,
                                                        float* __restrict__ dA_colsum, int B, int Z2, int H) {
  ARK_CHAIN_PRIO();
  constexpr int RW = 8;
  __shared__ float sh[RW][256];   // (2Z <= 256: host check)
  const int row0 = blockIdx.x * RW;
  const int Z2p = (Z2 + 31) & ~31;   // (columns up to the next multiple of 32 are zero: the products below run in blocks of 32)
  for (int i = threadIdx.x; i < RW * Z2p; i += 256) {
    const int r = i / Z2p, j = i - r * Z2p;
    sh[r][j] = (row0 + r < B && j < Z2) ? dhead[(long)(row0 + r) * Z2 + j] : 0.f;
  }
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= H) return;
  float acc[RW], pv[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    acc[r] = 0.f;
    pv[r] = row0 + r < B ? pre[(long)(row0 + r) * H + c] : 0.f;
  }
  // 32 rows of W_head in flight at once (round 5): on the step's critical path beside the chip-filling weight-gradient group
  // a memory round trip costs several microseconds, and eight loads per trip made this kernel 94 us in the step (16 alone)
  for (int j0 = 0; j0 < Z2; j0 += 32) {
    float wv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wv[u] = Whead[(long)min(j0 + u, Z2 - 1) * H + c];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[r] += sh[r][j0 + u] * wv[u];
    }
  }
  float cs = 0.f;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if (row0 + r >= B) break;
    const long o = (long)(row0 + r) * H + c;
    const float v = acc[r] * dgelu_erf(pv[r]);
    dA[o] = v;
    cs += v;
    if (prec16 == 2) reinterpret_cast<_Float16*>(dA16)[o] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    else reinterpret_cast<__bf16*>(dA16)[o] = (__bf16)v;
  }
  if (dA_colsum) atomicAdd(&dA_colsum[c], cs);
}

// dWz[d,j] += sum_b dzp[b,d] z[b,j] ; dbz[d] += sum_b dzp[b,d]
// 64 d-columns x 4 row groups per workgroup over a chunk of the batch; z rows are staged in LDS
// padded to ZT latent columns so the ZT accumulators stay in registers (compile-time indices).
template <int ZT>
__global__ __launch_bounds__(256) void zproj_bwd_dw_kernel(const float* __restrict__ dzp, const float* __restrict__ z,
                                                           float* __restrict__ dWz, float* __restrict__ dbz, int B, int Z,
                                                           int D, int b_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem_z[];
  float* zs = reinterpret_cast<float*>(smem_z);            // [b_chunk][ZT]
  float* red = zs + (size_t)b_chunk * ZT;                  // [4][64] reused per j-block
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * b_chunk, nb = min(B, b0 + b_chunk) - b0;
  for (int i = threadIdx.x; i < b_chunk * ZT; i += 256) {
    const int bb = i / ZT, j = i % ZT;
    zs[i] = (bb < nb && j < Z) ? z[(long)(b0 + bb) * Z + j] : 0.f;
  }
  __syncthreads();
  float acc[ZT];
#pragma unroll
  for (int j = 0; j < ZT; ++j) acc[j] = 0.f;
  float sb = 0.f;
  if (d < D)
    for (int bb = rg; bb < nb; bb += 4) {
      const float g = dzp[(long)(b0 + bb) * D + d];
      sb += g;
#pragma unroll
      for (int j = 0; j < ZT; ++j) acc[j] += g * zs[bb * ZT + j];
    }
  // cross row-group reduction through LDS, one latent column at a time
#pragma unroll
  for (int j = 0; j <= ZT; ++j) {
    const float v = (j < ZT) ? acc[j < ZT ? j : 0] : sb;
    red[rg * 64 + lane] = v;
    __syncthreads();
    if (rg == 0 && d < D) {
      const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
      if (j < ZT) { if (j < Z) atomicAdd(&dWz[(long)d * Z + j], t); }
      else atomicAdd(&dbz[d], t);
    }
    __syncthreads();
  }
}

// Both batch reductions that follow the per-row latent backward, in ONE launch (they used to be three launches --
// z-projection weights, a column sum, a register-staged skinny product -- ~45 us on the side queue):
//   job 0:  dWz[d, j] += sum_b dzp[b, d] z[b, j]        dbz[d] += sum_b dzp[b, d]              (models.py:139)
//   job 1:  dWh[j, c] += sum_b dhead[b, j] act[b, c]    dbh[j] += sum_b dhead[b, j]            (models.py:43-44)
// i.e. OUT[wide column][skinny column] = sum over the batch of WIDE[b][.] x SKINNY[b][.] with a skinny side of
// Z / 2Z columns.  64 wide columns x 4 row groups per workgroup over a chunk of the batch; the skinny rows are staged
// in LDS padded to ZT columns so the ZT accumulators stay in registers.
struct LatentReduceArgs {
  const float* dzp; const float* z; float* dWz; float* dbz;              // job 0
  const float* dhead; const void* act16; int prec16; float* dWh; float* dbh;   // job 1
  int B, Z, D, H, b_chunk;
};

template <int ZT>
__global__ __launch_bounds__(256) void latent_reduce_bwd_kernel(LatentReduceArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_lr[];
  float* zs = reinterpret_cast<float*>(smem_lr);            // [b_chunk][ZT]
  float* red = zs + (size_t)p.b_chunk * ZT;                 // [4][64] reused per column
  const int job = blockIdx.z;   // 0: z-projection; 1, 2, ..: blocks of ZT head columns
  const int jbase = job == 0 ? 0 : (job - 1) * ZT;
  const int NTOT = job == 0 ? p.Z : 2 * p.Z;
  const int NW = job == 0 ? p.D : p.H, NS = min(NTOT - jbase, ZT);
  if ((int)blockIdx.x * 64 >= NW) return;   // block-uniform
  const float* skinny = (job == 0 ? p.z : p.dhead) + jbase;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * p.b_chunk, nb = min(p.B, b0 + p.b_chunk) - b0;
  for (int i = threadIdx.x; i < p.b_chunk * ZT; i += 256) {
    const int bb = i / ZT, j = i % ZT;
    zs[i] = (bb < nb && j < NS) ? skinny[(long)(b0 + bb) * NTOT + j] : 0.f;
  }
  __syncthreads();
  float acc[ZT];
#pragma unroll
  for (int j = 0; j < ZT; ++j) acc[j] = 0.f;
  float sb = 0.f;
  if (c < NW)
    for (int bb = rg; bb < nb; bb += 4) {
      float g;
      if (job == 0) g = p.dzp[(long)(b0 + bb) * p.D + c];
      else if (p.prec16 == PREC_F16) g = (float)reinterpret_cast<const _Float16*>(p.act16)[(long)(b0 + bb) * p.H + c];
      else g = (float)reinterpret_cast<const __bf16*>(p.act16)[(long)(b0 + bb) * p.H + c];
      sb += g;
#pragma unroll
      for (int j = 0; j < ZT; ++j) acc[j] += g * zs[bb * ZT + j];
    }
  // cross row-group reduction through LDS, one skinny column at a time (+ the wide column sum for job 0)
#pragma unroll
  for (int j = 0; j <= ZT; ++j) {
    if (j < ZT && j >= NS) continue;          // (uniform: NS is a kernel argument)
    if (j == ZT && job != 0) continue;
    const float v = (j < ZT) ? acc[j < ZT ? j : 0] : sb;
    red[rg * 64 + lane] = v;
    __syncthreads();
    if (rg == 0 && c < NW) {
      const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
      if (j == ZT) atomicAdd(&p.dbz[c], t);
      else if (job == 0) atomicAdd(&p.dWz[(long)c * p.Z + j], t);
      else atomicAdd(&p.dWh[(long)(jbase + j) * p.H + c], t);
    }
    __syncthreads();
  }
  // dbh[j] = column sums of the skinny operand itself: once per batch chunk, by the first column block
  if (job >= 1 && blockIdx.x == 0 && (int)threadIdx.x < NS) {
    float t = 0.f;
    for (int bb = 0; bb < nb; ++bb) t += zs[bb * ZT + threadIdx.x];
    atomicAdd(&p.dbh[jbase + threadIdx.x], t);
  }
}

// ---- token cross-entropy over time-major logits rows (t,b); target = seq[b, t+1] ---------------
// One wave per row: running max / sum-exp over V in 64-wide strides, wavefront reductions.
// Writes the per-row loss and (optionally, in place) dlogits = (softmax - onehot) * inv_count.
__global__ __launch_bounds__(256) void ce_kernel(float* logits, long ld, const int64_t* __restrict__ seq,
                                                 long ld_seq, const float* __restrict__ hyper, float* __restrict__ row_loss,
                                                 float* dlogits, __bf16* d16bf, _Float16* d16h, long ld16, int B, int L, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const int t = row / B, b = row % B;
  const long tgt = seq[(long)b * ld_seq + t + 1];
  const float* x = logits + (long)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < V; c += 64) se += expf(x[c] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  const bool live = (tgt != ARK_TOK_PAD);
  if (lane == 0) row_loss[row] = live ? (lse - x[tgt]) : 0.f;
  if (dlogits || d16bf || d16h) {
    const float s = live ? hyper[ARK_HP_CE_INV_COUNT] : 0.f;
    float* d = dlogits ? dlogits + (long)row * ld : nullptr;   // (16-bit copy alone: the fp32 round trip is skipped)
    const long nc = (d16bf || d16h) ? ((ld16 > ld || !d) ? ld16 : ld) : ld;
    for (int c = lane; c < nc; c += 64) {
      float g = 0.f;
      if (c < V) g = (expf(x[c] - lse) - ((long)c == tgt ? 1.0f : 0.f)) * s;
      if (d && c < ld) d[c] = g;
      // K-padded 16-bit copy (zeros beyond V) for the input-gradient product on the LDS-DMA engine
      if (c < ld16) {
        if (d16bf) d16bf[(long)row * ld16 + c] = (__bf16)g;
        if (d16h) d16h[(long)row * ld16 + c] = (_Float16)fminf(fmaxf(g, -65504.f), 65504.f);
      }
    }
  }
}


// Large-vocabulary edition: ONE workgroup per row, the row cached in LDS -- the [B*L,V] fp32 logits are read from
// HBM exactly once (the wave-per-row kernel above streams every row three times: max, sum-exp, gradient; at
// V = 24 101 that was 1.8 ms of a 5.2 ms step).  Used when the row fits LDS (V * 4 B <= 144 KB).
template <int NR>   // NR 16-byte loads per thread cover a row: NR * 1024 * 4 >= ld
__global__ __launch_bounds__(1024) void ce_row_lds_kernel(float* logits, long ld, const int64_t* __restrict__ seq, long ld_seq,
                                                          const float* __restrict__ hyper, float* __restrict__ row_loss,
                                                          float* dlogits, __bf16* d16bf, _Float16* d16h, long ld16, int B, int L,
                                                          int V) {
  extern __shared__ __attribute__((aligned(16))) char smem_ce[];
  float* xs = reinterpret_cast<float*>(smem_ce);
  __shared__ float red_m[16], red_s[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = B * L, LD4 = (int)(ld >> 2);   // ld % 4 == 0 (host-checked): rows are 16-byte aligned
  // Persistent over rows; the NEXT row travels HBM -> registers while this row's gradient goes LDS -> HBM, so the
  // read stream and the write stream overlap although only one workgroup fits a CU.
  f32x4 nx[NR];
  auto fetch = [&](int row) {
    const float* x = logits + (long)row * ld;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c4 = tid + 1024 * i;
      nx[i] = (c4 < LD4) ? *reinterpret_cast<const f32x4*>(x + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int row = blockIdx.x;
  if (row < R) fetch(row);
  for (; row < R; row += gridDim.x) {
    const int t = row / B, b = row % B;
    const long tgt = seq[(long)b * ld_seq + t + 1];
    // registers -> LDS with an online (max, sum-exp) per thread; columns >= V (row padding) do not count
    float m = -INFINITY, se = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c4 = tid + 1024 * i;
      if (c4 < LD4) {
        f32x4 v = nx[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * c4 + e >= V) v[e] = -INFINITY;
        *reinterpret_cast<f32x4*>(xs + 4 * c4) = v;
        const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (vm > m) { se *= expf(m - vm); m = vm; }
        if (m > -INFINITY) se += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
      }
    }
    const float wm = wave_max(m);
    se = wave_sum(m == -INFINITY ? 0.f : se * expf(m - wm));
    if (lane == 0) { red_m[wave] = wm; red_s[wave] = se; }
    __syncthreads();
    const int nrow = row + gridDim.x;
    if (nrow < R) fetch(nrow);   // in flight underneath the gradient phase below
    float M = red_m[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) M = fmaxf(M, red_m[i]);
    float S = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) S += (red_m[i] == -INFINITY) ? 0.f : red_s[i] * expf(red_m[i] - M);
    const float lse = M + logf(S);
    const bool live = (tgt != ARK_TOK_PAD);
    if (tid == 0) row_loss[row] = live ? (lse - xs[tgt]) : 0.f;
    if (dlogits || d16bf || d16h) {
      const float sc = live ? hyper[ARK_HP_CE_INV_COUNT] : 0.f;
      float* d = dlogits ? dlogits + (long)row * ld : nullptr;
      const long nc = (d16bf || d16h) ? ((ld16 > ld || !d) ? ld16 : ld) : ld;
      for (int c = tid; c < nc; c += 1024) {
        float g = 0.f;
        if (c < V) g = (expf(xs[c] - lse) - ((long)c == tgt ? 1.0f : 0.f)) * sc;
        if (d && c < ld) d[c] = g;
        if (c < ld16) {
          if (d16bf) d16bf[(long)row * ld16 + c] = (__bf16)g;
          if (d16h) d16h[(long)row * ld16 + c] = (_Float16)fminf(fmaxf(g, -65504.f), 65504.f);
        }
      }
    }
    __syncthreads();   // the row in LDS is done before the next one overwrites it
  }
}

// count of non-PAD targets -> hyper[CE_COUNT], hyper[CE_INV_COUNT] (single-rank case; data-parallel
// runs overwrite both with the all-reduced global count before the CE kernel runs)
__global__ __launch_bounds__(1024) void count_targets_kernel(const int64_t* __restrict__ seq, long ld_seq, int B, int L,
                                                             float* __restrict__ hyper) {
  __shared__ float red[16];
  float c = 0.f;
  for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
    const int b = i / L, t = i % L;
    c += (seq[(long)b * ld_seq + t + 1] != ARK_TOK_PAD) ? 1.f : 0.f;
  }
  const float tot = block_sum_1024(c, red);
  if (threadIdx.x == 0) { hyper[ARK_HP_CE_COUNT] = tot; hyper[ARK_HP_CE_INV_COUNT] = tot > 0.f ? 1.0f / tot : 0.f; }
}

// out[0] = loss = ce + beta*kl, out[1] = ce, out[2] = kl.  Deterministic single-workgroup sum.
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const float* __restrict__ row_loss, int n_rows,
                                                             const float* __restrict__ kl, int n_kl, float kl_scale,
                                                             const float* __restrict__ hyper, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_rows; i += blockDim.x) s += row_loss[i];
  const float tot = block_sum_1024(s, red);
  float ks = 0.f;
  if (kl && n_kl > 1) {   // per-row KL sums (ark_latent_zproj_fwd): fixed-order reduction, deterministic
    __syncthreads();      // (red[] is reused)
    float t = 0.f;
    for (int i = threadIdx.x; i < n_kl; i += blockDim.x) t += kl[i];
    ks = block_sum_1024(t, red) * kl_scale;
  }
  if (threadIdx.x == 0) {
    const float ce = tot * hyper[ARK_HP_CE_INV_COUNT];
    const float k = !kl ? 0.f : (n_kl > 1 ? ks : kl[0] * kl_scale);
    out[0] = ce + hyper[ARK_HP_BETA] * k;
    out[1] = ce;
    out[2] = k;
    out[3] = tot;  // un-normalised token loss sum (for data-parallel reporting)
  }
}

// greedy decode helper: argmax over the vocabulary of selected logits rows (first index on ties,
// as torch.topk / argmax on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld, int64_t* __restrict__ out,
                                                          int rows, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = x[(long)row * ld + c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = bi;
}

}  // namespace ark

extern "C" int ark_latent_fwd(const float* head, const float* eps, float* mu, float* logv, float* z, float* kl_out,
                              int B, int Z, void* stream) {
  using namespace ark;
  if (!head || !mu || !logv || !z || !kl_out || B <= 0 || Z <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, head, eps, mu, logv, z, kl_out, B, Z, 1);
  ARK_LAUNCH_CHECK();
  return 0;
}

// the same without the [-10, 10] clamp of logv when clamp == 0 (AutoRegEncoder of t-SAIL, reference models.py:93)
extern "C" int ark_latent_fwd_ex(const float* head, const float* eps, float* mu, float* logv, float* z, float* kl_out,
                                 int B, int Z, int clamp, void* stream) {
  using namespace ark;
  if (!head || !mu || !logv || !z || !kl_out || B <= 0 || Z <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, head, eps, mu, logv, z, kl_out, B, Z, clamp);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_latent_bwd_ex(const float* dz, const float* head, const float* eps, const float* hyper, float* dhead,
                                 int B, int Z, int clamp, void* stream) {
  using namespace ark;
  if (!dz || !head || !hyper || !dhead || B <= 0 || Z <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, (hipStream_t)stream, dz, head, eps, hyper,
                     dhead, B, Z, clamp);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_latent_bwd(const float* dz, const float* head, const float* eps, const float* hyper, float* dhead,
                              int B, int Z, void* stream) {
  using namespace ark;
  if (!dz || !head || !hyper || !dhead || B <= 0 || Z <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((B * Z + 255) / 256), dim3(256), 0, (hipStream_t)stream, dz, head, eps, hyper,
                     dhead, B, Z, 1);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_zproj_fwd(const float* z, const float* w_z, const float* b_z, float* h0, int64_t copy_stride,
                             int n_copies, int B, int Z, int D, void* stream) {
  using namespace ark;
  if (!z || !w_z || !b_z || !h0 || B <= 0 || Z <= 0 || D <= 0 || n_copies <= 0) return ARK_ERR_ARG;
  const long n = (long)B * D;
  hipLaunchKernelGGL(zproj_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, w_z, b_z, h0,
                     (long)copy_stride, n_copies, B, Z, D);
  ARK_LAUNCH_CHECK();
  return 0;
}

// dh0 (in: dL/dh0 summed over layers; out: overwritten with dL/d(pre-tanh)), dz / dWz / dbz outputs
// (dWz, dbz are overwritten).
extern "C" int ark_zproj_bwd(float* dh0, const float* h0, const float* z, const float* w_z, float* dz, float* d_w_z,
                             float* d_b_z, int B, int Z, int D, int accumulate, void* stream) {
  using namespace ark;
  if (!dh0 || !h0 || !z || !w_z || !dz || !d_w_z || !d_b_z || B <= 0 || Z <= 0 || D <= 0) return ARK_ERR_ARG;
  if (Z > 128) return ARK_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * D;
  (void)n;
#define ARK_ZDZ(ZT) hipLaunchKernelGGL(zproj_bwd_dz_kernel<ZT>, dim3((B + 3) / 4), dim3(256), 0, st, dh0, h0, w_z, dz, B, Z, D)
  if (Z <= 16) ARK_ZDZ(16);
  else if (Z <= 32) ARK_ZDZ(32);
  else if (Z <= 64) ARK_ZDZ(64);
  else ARK_ZDZ(128);
#undef ARK_ZDZ
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(d_w_z, 0, sizeof(float) * (size_t)D * Z, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d_b_z, 0, sizeof(float) * (size_t)D, st);
    if (e != hipSuccess) return (int)e;
  }
  const int b_chunk = 32;   // 256 workgroups at B=1024, D=512: the reduction over the batch is atomic anyway
  dim3 grid((D + 63) / 64, (B + b_chunk - 1) / b_chunk);
#define ARK_ZDW(ZT) hipLaunchKernelGGL(zproj_bwd_dw_kernel<ZT>, grid, dim3(256), (size_t)(b_chunk * ZT + 256) * sizeof(float), st, dh0, z, d_w_z, d_b_z, B, Z, D, b_chunk)
  if (Z <= 16) ARK_ZDW(16);
  else if (Z <= 32) ARK_ZDW(32);
  else if (Z <= 64) ARK_ZDW(64);
  else ARK_ZDW(128);
#undef ARK_ZDW
  ARK_LAUNCH_CHECK();
  return 0;
}

// batch reductions of the z-projection backward alone: dWz (+)= dzp^T z, dbz (+)= colsum(dzp)
extern "C" int ark_zproj_bwd_dw(const float* dzp, const float* z, float* d_w_z, float* d_b_z, int B, int Z, int D,
                                int accumulate, void* stream) {
  using namespace ark;
  if (!dzp || !z || !d_w_z || !d_b_z || B <= 0 || Z <= 0 || D <= 0) return ARK_ERR_ARG;
  if (Z > 128) return ARK_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(d_w_z, 0, sizeof(float) * (size_t)D * Z, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d_b_z, 0, sizeof(float) * (size_t)D, st);
    if (e != hipSuccess) return (int)e;
  }
  const int b_chunk = 32;
  dim3 grid((D + 63) / 64, (B + b_chunk - 1) / b_chunk);
#define ARK_ZDW(ZT) hipLaunchKernelGGL(zproj_bwd_dw_kernel<ZT>, grid, dim3(256), (size_t)(b_chunk * ZT + 256) * sizeof(float), st, dzp, z, d_w_z, d_b_z, B, Z, D, b_chunk)
  if (Z <= 16) ARK_ZDW(16);
  else if (Z <= 32) ARK_ZDW(32);
  else if (Z <= 64) ARK_ZDW(64);
  else ARK_ZDW(128);
#undef ARK_ZDW
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_latent_chain_bwd(float* dh0, const float* h0, const float* w_z, const float* head, const float* eps,
                                    const float* hyper, const float* ext_dhead, const float* w_head, const float* pre,
                                    float* dhead, float* dA, void* dA16, int prec16, float* dA_colsum, int B, int Z, int D, int H,
                                    void* stream) {
  using namespace ark;
  if (!dh0 || !h0 || !w_z || !head || !hyper || !w_head || !pre || !dhead || !dA || !dA16 || B <= 0 || Z <= 0 || D <= 0 || H <= 0)
    return ARK_ERR_ARG;
  if (Z > 128) return ARK_ERR_SHAPE;
  if (prec16 != 1 && prec16 != 2) return ARK_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = Z > 64 ? (size_t)(D + 256) * sizeof(float)      // the row of dzp + partial sums
                            : (size_t)Z * (D + 1) * sizeof(float);   // Wz^T
  if (lds > 150 * 1024) return ARK_ERR_SHAPE;
  // few row workgroups (B < 512) x many columns: the dA phase goes to a launch of its own with a 2-D grid (the chain kernel
  // runs with H = 0: its column loops are empty)
  // small latent x wide decoder (syn-paths Z = 10, syn-types Z = 24): two launches built for latency -- per-row dz / dhead
  // with every load of a phase in flight at once, then dA on a 2-D grid (72-80 us fused -> see profiles/r04_kernel_stats.csv)
#ifndef ARK_NO_LATENT_DZ   // (A/B builds: tools/build_variant.sh nodz loss.hip -DARK_NO_LATENT_DZ)
  if (Z <= 32 && D % 4 == 0 && D <= 1024 && B >= 64 && H >= 768) {
    const size_t l2 = (size_t)Z * (D + 1) * sizeof(float);
#define ARK_DZ(ZT, DU)                                                                                                       \
    {                                                                                                                        \
      static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_dz_head_kernel<ZT, DU>),            \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), true);         \
      (void)once;                                                                                                            \
      hipLaunchKernelGGL((latent_dz_head_kernel<ZT, DU>), dim3((B + 3) / 4), dim3(256), l2, st, dh0, h0, w_z, head, eps, hyper, \
                         ext_dhead, dhead, B, Z, D);                                                                         \
    }
    if (l2 <= 150 * 1024) {
      if (Z <= 16 && D <= 512) ARK_DZ(16, 8)
      else if (Z <= 16) ARK_DZ(16, 16)
      else if (D <= 512) ARK_DZ(32, 8)
      else ARK_DZ(32, 16)
#undef ARK_DZ
      ARK_LAUNCH_CHECK();
      hipLaunchKernelGGL(latent_dA_kernel, dim3((B + 7) / 8, (H + 255) / 256), dim3(256), 0, st, dhead, w_head, pre, dA, dA16, prec16,
                         dA_colsum, B, 2 * Z, H);
      ARK_LAUNCH_CHECK();
      return 0;
    }
  }
#endif
  const bool split = Z <= 64 && (B + 3) / 4 < 128 && H >= 768;
  const int Hc = split ? 0 : H;
#define ARK_LC(ZT)                                                                                                          \
  {                                                                                                                         \
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_bwd_kernel<ZT, 4, (ZT > 32)>),            \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), true);           \
    (void)once;                                                                                                             \
    hipLaunchKernelGGL((latent_chain_bwd_kernel<ZT, 4, (ZT > 32)>), dim3((B + 3) / 4), dim3(256), lds, st, dh0, h0, w_z, head, eps,     \
                       hyper, ext_dhead, w_head, pre, dhead, dA, dA16, prec16, dA_colsum, B, Z, D, Hc);                                 \
  }
  if (Z <= 16) ARK_LC(16)
  else if (Z <= 32) ARK_LC(32)
  else if (Z <= 64) ARK_LC(64)
  else {   // wide latent (wd-articles Z = 128): one workgroup per row
    if (D % 2 != 0) return ARK_ERR_SHAPE;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_bwd_kernel<128, 1, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), true);
    (void)once;
    hipLaunchKernelGGL((latent_chain_bwd_kernel<128, 1, true>), dim3(B), dim3(256), lds, st, dh0, h0, w_z, head, eps, hyper,
                       ext_dhead, w_head, pre, dhead, dA, dA16, prec16, dA_colsum, B, Z, D, H);
  }
#undef ARK_LC
  ARK_LAUNCH_CHECK();
  if (split) {
    hipLaunchKernelGGL(latent_dA_kernel, dim3((B + 7) / 8, (H + 255) / 256), dim3(256), 0, st, dhead, w_head, pre, dA, dA16, prec16,
                       dA_colsum, B, 2 * Z, H);
    ARK_LAUNCH_CHECK();
  }
  return 0;
}

// the two batch reductions behind ark_latent_chain_bwd in one launch; every output ACCUMULATES (+=):
// dWz [D,Z], dbz [D] from dzp [B,D] (the in-place output of ark_latent_chain_bwd) and z [B,Z];
// dWh [2Z,H], dbh [2Z] from dhead [B,2Z] and the 16-bit copy act16 [B,H] of the last encoder activation
extern "C" int ark_latent_reduce_bwd(const float* dzp, const float* z, float* d_w_z, float* d_b_z, const float* dhead,
                                     const void* act16, int prec16, float* d_w_head, float* d_b_head, int B, int Z, int D, int H,
                                     void* stream) {
  using namespace ark;
  if (!dzp || !z || !d_w_z || !d_b_z || !dhead || !act16 || !d_w_head || !d_b_head || B <= 0 || Z <= 0 || D <= 0 || H <= 0)
    return ARK_ERR_ARG;
  if (prec16 != PREC_F16 && prec16 != PREC_BF16) return ARK_ERR_ARG;
  if (Z > 128) return ARK_ERR_SHAPE;
  LatentReduceArgs p{dzp, z, d_w_z, d_b_z, dhead, act16, prec16, d_w_head, d_b_head, B, Z, D, H, 32};
  const int wide = D > H ? D : H;
  const int zt = 2 * Z <= 32 ? 32 : 2 * Z <= 64 ? 64 : 128;
  dim3 grid((wide + 63) / 64, (B + p.b_chunk - 1) / p.b_chunk, 1 + (2 * Z + zt - 1) / zt);
  hipStream_t st = (hipStream_t)stream;
#define ARK_LR(ZT) hipLaunchKernelGGL(latent_reduce_bwd_kernel<ZT>, grid, dim3(256), (size_t)(p.b_chunk * ZT + 256) * sizeof(float), st, p)
  if (2 * Z <= 32) ARK_LR(32);
  else if (2 * Z <= 64) ARK_LR(64);
  else ARK_LR(128);
#undef ARK_LR
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_count_targets(const int64_t* seq, int64_t ld_seq, int B, int L, float* hyper, void* stream) {
  using namespace ark;
  if (!seq || !hyper || B <= 0 || L <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(count_targets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seq, (long)ld_seq, B, L, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_ce_fwd_bwd(float* logits, int64_t ld, const int64_t* seq, int64_t ld_seq, const float* hyper,
                              float* row_loss, float* dlogits, void* dlogits16, int prec16, int64_t ld16, int B, int L, int V,
                              void* stream) {
  using namespace ark;
  if (!logits || !seq || !hyper || !row_loss || B <= 0 || L <= 0 || V <= 0 || ld < V) return ARK_ERR_ARG;
  if (dlogits16 && (ld16 < V || (prec16 != 1 && prec16 != 2))) return ARK_ERR_ARG;
  __bf16* dbf = prec16 == 1 ? (__bf16*)dlogits16 : nullptr;
  _Float16* dh = prec16 == 2 ? (_Float16*)dlogits16 : nullptr;
  const size_t ld_bytes = (size_t)ld * sizeof(float);
  if (V >= 4096 && ld_bytes <= 144 * 1024 && ld % 4 == 0) {   // large vocabulary: row cached in LDS, read once
    const int nr = (int)((ld / 4 + 1023) / 1024);   // 16-byte loads per thread per row (<= 9 at 144 KB)
    int grid = 256 * 1;                              // one workgroup per CU (LDS), persistent over the rows
    if (grid > B * L) grid = B * L;
#define ARK_CE_LDS(NRV)                                                                                                   \
    {                                                                                                                     \
      static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(ce_row_lds_kernel<NRV>),               \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024), true);      \
      (void)once;                                                                                                         \
      hipLaunchKernelGGL(ce_row_lds_kernel<NRV>, dim3((unsigned)grid), dim3(1024), ld_bytes, (hipStream_t)stream, logits,  \
                         (long)ld, seq, (long)ld_seq, hyper, row_loss, dlogits, dbf, dh, (long)ld16, B, L, V);            \
    }
    if (nr <= 2) ARK_CE_LDS(2) else if (nr <= 4) ARK_CE_LDS(4) else if (nr <= 6) ARK_CE_LDS(6) else ARK_CE_LDS(9)
#undef ARK_CE_LDS
  } else {
    hipLaunchKernelGGL(ce_kernel, dim3((B * L + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, seq, (long)ld_seq,
                       hyper, row_loss, dlogits, dbf, dh, (long)ld16, B, L, V);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_loss_finalize(const float* row_loss, int n_rows, const float* kl, const float* hyper, float* out4,
                                 void* stream) {
  using namespace ark;
  if (!row_loss || !hyper || !out4 || n_rows <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, n_rows, kl, 1, 1.0f, hyper,
                     out4);
  ARK_LAUNCH_CHECK();
  return 0;
}

// the same with kl = kl_scale * sum(kl_rows[0:n_kl]) (per-row KL terms of ark_latent_zproj_fwd; kl_scale = -0.5 / (rows * Z))
extern "C" int ark_loss_finalize_rows(const float* row_loss, int n_rows, const float* kl_rows, int n_kl, float kl_scale,
                                      const float* hyper, float* out4, void* stream) {
  using namespace ark;
  if (!row_loss || !hyper || !out4 || !kl_rows || n_rows <= 0 || n_kl <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, n_rows, kl_rows, n_kl, kl_scale, hyper,
                     out4);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_argmax_rows(const float* x, int64_t ld, int64_t* out, int rows, int V, void* stream) {
  using namespace ark;
  if (!x || !out || rows <= 0 || V <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long)ld, out, rows, V);
  ARK_LAUNCH_CHECK();
  return 0;
}
