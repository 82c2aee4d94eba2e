// Transformer-variant kernels (t-ARK: reference kgvae/model/models.py:349-366, the stock nn.TransformerEncoderLayer with
// its defaults -- post-norm, ReLU feed-forward, causal boolean mask): residual + LayerNorm, causal multi-head attention and
// the counter-hash dropout of its four dropout sites, forward and backward.  The dense products of the layers run on the
// generic tile engine (gemm.hip: exact-fp32 or 16-bit MFMA operands, bias / ReLU epilogues).
//
// Layout: rows are TIME-MAJOR like everything else in this library -- row (t, b) = t * B + b -- so the token gather, the
// vocabulary projection and the cross-entropy kernels are shared with the GRU models.  Head h of token (t, b) is the
// dh-wide slice [h * dh, (h + 1) * dh) of its row; qkv rows are [q | k | v] (nn.MultiheadAttention's packed in-projection).
//
// Attention is exact fp32 on the vector units: one wave per query row (scores: one key per lane; softmax by wave
// reductions; context: one head column per lane), probabilities kept for the backward pass.  L is at most 640 here
// (wd-articles: 637), head widths 8 ... 256.
#include "gemm_core.h"
#include "../../include/ark_amd.h"

namespace ark {

constexpr int kAttnMaxChunks = 10;   // keys per lane: L <= 640
constexpr int kAttnMaxDh = 768;      // head width: up to 12 columns per lane (t-SAIL encoder: 3 * 1024 / 4 heads)

// ---------------------------------------------------------------------------------------------------------------------
// y = LayerNorm(x + res) * gamma + beta over the last dimension (biased variance, eps inside the root: nn.LayerNorm);
// s_out (nullable) keeps x + res, stats keeps (mean, 1/sqrt(var + eps)) per row for the backward pass.  One wave per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ s_out, float* __restrict__ y,
                                                            float* __restrict__ stats, int rows, int D, float eps, float drop_p,
                                                            uint64_t seed, const float* __restrict__ hyper) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  const float* rr = res ? res + (long)row * D : nullptr;
  // drop_p > 0: `res` is the sublayer output BEFORE its dropout -- the keep-scale of element row * D + c of the current draw
  // (what ark_dropout_apply would have multiplied in a pass of its own) is applied on the way in
  const bool drop = drop_p > 0.f && rr;
  DropCtx dc{};
  if (drop) dc = drop_ctx(seed, hyper, drop_p);
  auto rv = [&](int c) -> float {
    if (!rr) return 0.f;
    const float v = rr[c];
    return drop ? v * dropout_one(dc, (uint64_t)((long)row * D + c)) : v;
  };
  float sum = 0.f;
  for (int c = lane; c < D; c += 64) sum += xr[c] + rv(c);
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] + rv(c) - mean;
    sq += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  for (int c = lane; c < D; c += 64) {
    const float v = xr[c] + rv(c);
    if (s_out) s_out[(long)row * D + c] = v;
    y[(long)row * D + c] = (v - mean) * rstd * gamma[c] + beta[c];
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// The same for rows that are whole multiples of 256 columns (every layer width of the Transformer variants: 512, 1024,
// 1536, 3072, 384 excepted): ONE pass -- a lane keeps its NV float4s of x + res in registers (16-byte loads, columns
// 4 (64 k + lane) ..), the dropout keep-scales of a quad come from one hash instead of one per element and pass (the
// three-pass kernel above hashed every element three times: 30 us for 10 240 x 512 where the traffic is worth 12).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ s_out, float* __restrict__ y,
                                                                float* __restrict__ stats, int rows, int D, float eps, float drop_p,
                                                                uint64_t seed, const float* __restrict__ hyper) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long ro = (long)row * D;
  const bool drop = drop_p > 0.f && res;
  DropCtx dc{};
  if (drop) dc = drop_ctx(seed, hyper, drop_p);
  f32x4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * (64 * k + lane);
    v[k] = *reinterpret_cast<const f32x4*>(x + ro + c);
    if (res) {
      f32x4 r4 = *reinterpret_cast<const f32x4*>(res + ro + c);
      if (drop) r4 *= dropout_quad(dc, (uint64_t)(ro + c) >> 2);
      v[k] += r4;
    }
    sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = v[k][i] - mean;
      sq += d * d;
    }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * (64 * k + lane);
    if (s_out) *reinterpret_cast<f32x4*>(s_out + ro + c) = v[k];
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c), b4 = *reinterpret_cast<const f32x4*>(beta + c);
    *reinterpret_cast<f32x4*>(y + ro + c) = (v[k] - mean) * rstd * g4 + b4;
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// ds = gradient w.r.t. the normalised sum s; dgamma / dbeta accumulate (+=).  With xhat = (s - mean) * rstd and
// g = dy * gamma:  ds = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)).  A workgroup takes RPW rows per wave and keeps the
// per-column sums of its rows in registers (column = lane + 64 k), combines its four waves in LDS: one atomic per column.
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            float* __restrict__ ds, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int rows, int D, int rows_per_wg) {
  __shared__ float red[2][4][64 * CPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[CPL], ab[CPL], gm[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    ag[k] = ab[k] = 0.f;
    gm[k] = (lane + 64 * k < D) ? gamma[lane + 64 * k] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  // TWO rows per wave and iteration (CPL <= 8: registers allow it): their loads are in flight together and their wave
  // reductions interleave -- a wave walked its rows one memory round trip at a time before (42 us for 10 240 x 512)
  constexpr int RW = CPL <= 8 ? 2 : 1;
  for (int row = r0 + wave * RW; row < r1; row += 4 * RW) {
    float g[RW][CPL], xh[RW][CPL], s1[RW], s2[RW], rstd[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int rw = min(row + j, r1 - 1);   // (a missing second row repeats the first: computed, not stored, not summed)
      const bool live = row + j < r1;
      const float mean = stats[2 * rw];
      rstd[j] = stats[2 * rw + 1];
      s1[j] = s2[j] = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        const bool ok = c < D;
        const float dyv = ok ? dy[(long)rw * D + c] : 0.f;
        xh[j][k] = ok ? (s[(long)rw * D + c] - mean) * rstd[j] : 0.f;
        g[j][k] = dyv * gm[k];
        s1[j] += g[j][k];
        s2[j] += g[j][k] * xh[j][k];
        if (live) {
          ag[k] += dyv * xh[j][k];
          ab[k] += dyv;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      s1[j] = wave_sum(s1[j]) / (float)D;
      s2[j] = wave_sum(s2[j]) / (float)D;
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      if (row + j >= r1) break;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        if (c < D) ds[(long)(row + j) * D + c] = rstd[j] * (g[j][k] - s1[j] - xh[j][k] * s2[j]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) { red[0][wave][lane + 64 * k] = ag[k]; red[1][wave][lane + 64 * k] = ab[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(&dgamma[c], a);
    atomicAdd(&dbeta[c], b);
  }
}

// (Measured and not kept: a 16-byte-access version of this kernel with the next row's loads issued ahead of the current
// row's reductions -- 58.9 us against 42.0 for 10 240 x 512.)

// x[i] *= keep-scale(i) of the current dropout draw (common.h dropout_quad: hyper[ARK_HP_DROP_STEP], `seed`): forward on an
// activation, backward on its gradient -- the same (seed, draw, index) gives the same mask.
__global__ __launch_bounds__(256) void dropout_apply_kernel(float* __restrict__ x, long n4, uint64_t seed,
                                                            const float* __restrict__ hyper, float p) {
  const DropCtx dc = drop_ctx(seed, hyper, p);
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<f32x4*>(x)[q];
    reinterpret_cast<f32x4*>(x)[q] = v * dropout_quad(dc, (uint64_t)q);
  }
}

struct AttnArgs {
  const float* qkv;     // [L*B, 3D] time-major rows: q | k | v
  float* out;           // [L*B, D] context (forward output; read by the backward for delta = dO . O)
  float* probs;         // [B, H, L, L] softmax probabilities BEFORE dropout (row i, column j; 0 beyond the causal limit)
  const float* dout;    // backward: gradient of `out`
  float* dscore;        // backward scratch [B, H, L, L]: dS * scale
  float* dqkv;          // backward output [L*B, 3D]
  const float* hyper;
  const unsigned char* kmask;   // [B, L] 1 = key may be attended to (nullable: all keys; src_key_padding_mask of the reference)
  uint64_t seed;
  float drop_p, scale;
  int B, L, D, H, dh, causal;
};

__device__ __forceinline__ float attn_keep(const DropCtx& dc, bool drop, long idx) { return drop ? dropout_one(dc, (uint64_t)idx) : 1.0f; }

// one wave per query row i of one (batch b, head h): scores over the keys j <= i (causal) or all keys
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  __shared__ float qs[4][kAttnMaxDh];
  __shared__ float ps[4][64 * kAttnMaxChunks];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int i = blockIdx.y * 4 + wave;
  if (i >= p.L) return;   // (whole wave; no block-wide barrier below)
  const int B = p.B, L = p.L, D = p.D, dh = p.dh;
  const long rs = 3L * D;
  const float* q = p.qkv + ((long)i * B + b) * rs + h * dh;
  for (int d = lane; d < dh; d += 64) qs[wave][d] = q[d] * p.scale;
  const int nk = p.causal ? i + 1 : L;
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  float sc[kAttnMaxChunks];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < kAttnMaxChunks; ++c) {
    const int j = c * 64 + lane;
    float s = -INFINITY;
    if (c * 64 < nk && j < nk && (!p.kmask || p.kmask[(long)b * L + j])) {
      const float* k = p.qkv + ((long)j * B + b) * rs + D + h * dh;
      float a = 0.f;
      for (int d = 0; d < dh; d += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(k + d);
        a += qs[wave][d] * kv[0] + qs[wave][d + 1] * kv[1] + qs[wave][d + 2] * kv[2] + qs[wave][d + 3] * kv[3];
      }
      s = a;
    }
    sc[c] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < kAttnMaxChunks; ++c) {
    sc[c] = (sc[c] == -INFINITY) ? 0.f : expf(sc[c] - m);
    sum += sc[c];
  }
  const float tot = wave_sum(sum);
  const float inv = tot > 0.f ? 1.0f / tot : 0.f;   // (a query whose keys are all masked -- an all-PAD graph: zero row, not NaN)
  const long prow = (((long)b * p.H + h) * L + i) * L;
#pragma unroll
  for (int c = 0; c < kAttnMaxChunks; ++c) {
    const int j = c * 64 + lane;
    if (j < L) {
      const float pr = sc[c] * inv;
      p.probs[prow + j] = pr;
      ps[wave][j] = pr * attn_keep(dc, drop, prow + j);
    }
  }
  // context: column d = lane + 64 k of this head
  float acc[kAttnMaxDh / 64] = {};
  for (int j = 0; j < nk; ++j) {
    const float pj = ps[wave][j];
    const float* v = p.qkv + ((long)j * B + b) * rs + 2 * D + h * dh;
#pragma unroll
    for (int k = 0; k < kAttnMaxDh / 64; ++k) {
      const int d = lane + 64 * k;
      if (d < dh) acc[k] += pj * v[d];
    }
  }
  float* o = p.out + ((long)i * B + b) * D + h * dh;
#pragma unroll
  for (int k = 0; k < kAttnMaxDh / 64; ++k) {
    const int d = lane + 64 * k;
    if (d < dh) o[d] = acc[k];
  }
}

// backward, query side: dS[i, :] (kept for the key side) and dQ[i].  With M the dropout keep-scale and Pm = P o M:
//   dPm[i, j] = dO[i] . V[j],  delta[i] = sum_j Pm[i, j] dPm[i, j] = dO[i] . O[i],  dS = P o (M o dPm - delta)
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs p) {
  __shared__ float dos[4][kAttnMaxDh];
  __shared__ float dss[4][64 * kAttnMaxChunks];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int i = blockIdx.y * 4 + wave;
  if (i >= p.L) return;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh;
  const long rs = 3L * D;
  const float* dO = p.dout + ((long)i * B + b) * D + h * dh;
  const float* O = p.out + ((long)i * B + b) * D + h * dh;
  float dl = 0.f;
  for (int d = lane; d < dh; d += 64) {
    const float g = dO[d];
    dos[wave][d] = g;
    dl += g * O[d];
  }
  const float delta = wave_sum(dl);
  const int nk = p.causal ? i + 1 : L;
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  const long prow = (((long)b * p.H + h) * L + i) * L;
  for (int c = 0; c * 64 < L; ++c) {
    const int j = c * 64 + lane;
    if (j >= L) continue;
    float ds = 0.f;
    if (j < nk && (!p.kmask || p.kmask[(long)b * L + j])) {
      const float* v = p.qkv + ((long)j * B + b) * rs + 2 * D + h * dh;
      float a = 0.f;
      for (int d = 0; d < dh; d += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(v + d);
        a += dos[wave][d] * vv[0] + dos[wave][d + 1] * vv[1] + dos[wave][d + 2] * vv[2] + dos[wave][d + 3] * vv[3];
      }
      ds = p.probs[prow + j] * (a * attn_keep(dc, drop, prow + j) - delta) * p.scale;
    }
    p.dscore[prow + j] = ds;
    dss[wave][j] = ds;
  }
  float acc[kAttnMaxDh / 64] = {};
  for (int j = 0; j < nk; ++j) {
    const float dj = dss[wave][j];
    const float* k = p.qkv + ((long)j * B + b) * rs + D + h * dh;
#pragma unroll
    for (int kk = 0; kk < kAttnMaxDh / 64; ++kk) {
      const int d = lane + 64 * kk;
      if (d < dh) acc[kk] += dj * k[d];
    }
  }
  float* dq = p.dqkv + ((long)i * B + b) * rs + h * dh;
#pragma unroll
  for (int kk = 0; kk < kAttnMaxDh / 64; ++kk) {
    const int d = lane + 64 * kk;
    if (d < dh) dq[d] = acc[kk];
  }
}

// backward, key side: dK[j] = sum_i dS[i, j] Q[i] (dS already carries the 1/sqrt(dh) scale), dV[j] = sum_i Pm[i, j] dO[i]
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnArgs p) {
  __shared__ float dsc[4][64 * kAttnMaxChunks];
  __shared__ float pmc[4][64 * kAttnMaxChunks];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int j = blockIdx.y * 4 + wave;
  if (j >= p.L) return;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh;
  const long rs = 3L * D;
  const int i0 = p.causal ? j : 0;
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  const long base = ((long)b * p.H + h) * L * L;
  for (int i = i0 + lane; i < L; i += 64) {
    const long idx = base + (long)i * L + j;
    dsc[wave][i] = p.dscore[idx];
    pmc[wave][i] = p.probs[idx] * attn_keep(dc, drop, idx);
  }
  float ak[kAttnMaxDh / 64] = {}, av[kAttnMaxDh / 64] = {};
  for (int i = i0; i < L; ++i) {
    const float ds = dsc[wave][i], pm = pmc[wave][i];
    const float* q = p.qkv + ((long)i * B + b) * rs + h * dh;
    const float* dO = p.dout + ((long)i * B + b) * D + h * dh;
#pragma unroll
    for (int kk = 0; kk < kAttnMaxDh / 64; ++kk) {
      const int d = lane + 64 * kk;
      if (d < dh) { ak[kk] += ds * q[d]; av[kk] += pm * dO[d]; }
    }
  }
  float* dk = p.dqkv + ((long)j * B + b) * rs + D + h * dh;
  float* dv = p.dqkv + ((long)j * B + b) * rs + 2 * D + h * dh;
#pragma unroll
  for (int kk = 0; kk < kAttnMaxDh / 64; ++kk) {
    const int d = lane + 64 * kk;
    if (d < dh) { dk[d] = ak[kk]; dv[d] = av[kk]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// SHORT sequences (L <= 16: the syn-* datasets' 10 decoder steps, t-SAIL's 3 encoder triples): ONE WAVE per (batch, head)
// does the whole head -- Q, K, V (backward: + dO) of its L rows staged once in LDS (row stride dh + 1), the L x L scores one
// (query, key) pair per lane, softmax one query per lane, context / dQ / dK / dV one head column per lane.  The kernels above
// run one wave per QUERY ROW and re-read every K / V row of the head from global memory per query: at L = 10 that is 40 960
// waves per attention and was 0.75 ms of a 3.9-ms t-ARK step.  Same arithmetic, same dropout indices (prow + j), same
// probabilities array.  Only wave-level synchronisation: a wave's LDS accesses execute in program order.
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnArgs p, int per_wave_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem_as[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * (blockDim.x >> 6) + wave;
  if (pair >= p.B * p.H) return;   // (whole wave; no block-wide barrier below)
  const int b = pair / p.H, h = pair % p.H;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh, DS = dh + 1, LS = L + 1;
  const long rs = 3L * D;
  float* Qs = smem_as + (long)wave * per_wave_floats;
  float* Ks = Qs + L * DS;
  float* Vs = Ks + L * DS;
  float* Ss = Vs + L * DS;   // [L][L + 1]: scores, then dropped probabilities
  for (int t = 0; t < L; ++t) {
    const float* row = p.qkv + ((long)t * B + b) * rs + h * dh;
    for (int d = lane; d < dh; d += 64) {
      Qs[t * DS + d] = row[d] * p.scale;
      Ks[t * DS + d] = row[D + d];
      Vs[t * DS + d] = row[2 * D + d];
    }
  }
  lds_wave_sync();
  for (int idx = lane; idx < L * L; idx += 64) {
    const int i = idx / L, j = idx - i * L;
    float sc = -INFINITY;
    if ((!p.causal || j <= i) && (!p.kmask || p.kmask[(long)b * L + j])) {
      float a = 0.f;
      for (int d = 0; d < dh; ++d) a += Qs[i * DS + d] * Ks[j * DS + d];
      sc = a;
    }
    Ss[i * LS + j] = sc;
  }
  lds_wave_sync();
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  if (lane < L) {
    const int i = lane;
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, Ss[i * LS + j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      const float s0 = Ss[i * LS + j];
      sum += (s0 == -INFINITY) ? 0.f : expf(s0 - m);
    }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;   // (all keys masked: zero row, not NaN)
    const long prow = (((long)b * p.H + h) * L + i) * L;
    for (int j = 0; j < L; ++j) {
      const float s0 = Ss[i * LS + j];
      const float pr = ((s0 == -INFINITY) ? 0.f : expf(s0 - m)) * inv;
      p.probs[prow + j] = pr;
      Ss[i * LS + j] = pr * attn_keep(dc, drop, prow + j);
    }
  }
  lds_wave_sync();
  for (int i = 0; i < L; ++i) {
    float* o = p.out + ((long)i * B + b) * D + h * dh;
    const int nk = p.causal ? i + 1 : L;
    for (int d = lane; d < dh; d += 64) {
      float acc = 0.f;
      for (int j = 0; j < nk; ++j) acc += Ss[i * LS + j] * Vs[j * DS + d];
      o[d] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(AttnArgs p, int per_wave_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem_as[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * (blockDim.x >> 6) + wave;
  if (pair >= p.B * p.H) return;
  const int b = pair / p.H, h = pair % p.H;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh, DS = dh + 1, LS = L + 1;
  const long rs = 3L * D;
  float* Qs = smem_as + (long)wave * per_wave_floats;
  float* Ks = Qs + L * DS;
  float* Vs = Ks + L * DS;
  float* Gs = Vs + L * DS;    // dO
  float* dS = Gs + L * DS;    // [L][L + 1]: dS * scale
  float* Pm = dS + L * LS;    // [L][L + 1]: dropped probabilities
  float* dl = Pm + L * LS;    // [L]: delta
  for (int t = 0; t < L; ++t) {
    const float* row = p.qkv + ((long)t * B + b) * rs + h * dh;
    const float* g = p.dout + ((long)t * B + b) * D + h * dh;
    const float* o = p.out + ((long)t * B + b) * D + h * dh;
    float part = 0.f;
    for (int d = lane; d < dh; d += 64) {
      Qs[t * DS + d] = row[d];
      Ks[t * DS + d] = row[D + d];
      Vs[t * DS + d] = row[2 * D + d];
      const float gv = g[d];
      Gs[t * DS + d] = gv;
      part += gv * o[d];
    }
    part = wave_sum(part);
    if (lane == 0) dl[t] = part;
  }
  lds_wave_sync();
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  for (int idx = lane; idx < L * L; idx += 64) {
    const int i = idx / L, j = idx - i * L;
    float ds = 0.f, pm = 0.f;
    if ((!p.causal || j <= i) && (!p.kmask || p.kmask[(long)b * L + j])) {
      float a = 0.f;
      for (int d = 0; d < dh; ++d) a += Gs[i * DS + d] * Vs[j * DS + d];
      const long pidx = (((long)b * p.H + h) * L + i) * L + j;
      const float pr = p.probs[pidx], keep = attn_keep(dc, drop, pidx);
      ds = pr * (a * keep - dl[i]) * p.scale;
      pm = pr * keep;
    }
    dS[i * LS + j] = ds;
    Pm[i * LS + j] = pm;
  }
  lds_wave_sync();
  for (int t = 0; t < L; ++t) {
    float* dq = p.dqkv + ((long)t * B + b) * rs + h * dh;
    for (int d = lane; d < dh; d += 64) {
      float aq = 0.f, ak = 0.f, av = 0.f;
      for (int u = 0; u < L; ++u) {
        aq += dS[t * LS + u] * Ks[u * DS + d];     // dQ[t] = sum_j dS[t, j] K[j]
        ak += dS[u * LS + t] * Qs[u * DS + d];     // dK[t] = sum_i dS[i, t] Q[i]
        av += Pm[u * LS + t] * Gs[u * DS + d];     // dV[t] = sum_i Pm[i, t] dO[i]
      }
      dq[d] = aq;
      dq[D + d] = ak;
      dq[2 * D + d] = av;
    }
  }
}

// waves per workgroup of the short-sequence kernels (0: not eligible): per-wave LDS of the BACKWARD (the larger), <= 64 KB in all
static int attn_small_waves(const AttnArgs& p, int* fwd_floats, int* bwd_floats) {
  if (p.L > 16) return 0;
  const int f = 3 * p.L * (p.dh + 1) + p.L * (p.L + 1);
  const int bw = 4 * p.L * (p.dh + 1) + 2 * p.L * (p.L + 1) + p.L;
  *fwd_floats = f;
  *bwd_floats = bw;
  const int per = bw * 4;
  if (per > 64 * 1024) return 0;
  return per * 4 <= 64 * 1024 ? 4 : per * 2 <= 64 * 1024 ? 2 : 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// t-SAIL encoder input (AutoRegEncoder.forward, reference models.py:80-86): x[(t, b)] = [E[h] | R[r] | E[t]] of triple t of
// graph b (rows time-major over the TRIPLE index), kmask[b, t] = (r != pad_rid).  One wave per row.
__global__ __launch_bounds__(256) void triple_gather_kernel(const int64_t* __restrict__ triples, const float* __restrict__ E,
                                                            const float* __restrict__ R, float* __restrict__ x,
                                                            unsigned char* __restrict__ kmask, int B, int T, int D, long pad_rid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T * B) return;
  const int t = row / B, b = row % B;
  const int64_t* tr = triples + ((long)b * T + t) * 3;
  const long h = tr[0], r = tr[1], tl = tr[2];
  float* o = x + (long)row * 3 * D;
  for (int c = lane; c < D; c += 64) {
    o[c] = E[h * D + c];
    o[D + c] = R[r * D + c];
    o[2 * D + c] = E[tl * D + c];
  }
  if (lane == 0 && kmask) kmask[(long)b * T + t] = (pad_rid < 0 || r != pad_rid) ? 1 : 0;
}

// its backward: dE[h] += dx[:, 0:D], dR[r] += dx[:, D:2D], dE[t] += dx[:, 2D:3D]; padding rows (nn.Embedding padding_idx) get none
__global__ __launch_bounds__(256) void triple_scatter_kernel(const int64_t* __restrict__ triples, const float* __restrict__ dx,
                                                             float* __restrict__ dE, float* __restrict__ dR, int B, int T, int D,
                                                             long pad_eid, long pad_rid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T * B) return;
  const int t = row / B, b = row % B;
  const int64_t* tr = triples + ((long)b * T + t) * 3;
  const long h = tr[0], r = tr[1], tl = tr[2];
  const float* g = dx + (long)row * 3 * D;
  for (int c = lane; c < D; c += 64) {
    if (h != pad_eid) atomicAdd(&dE[h * D + c], g[c]);
    if (r != pad_rid) atomicAdd(&dR[r * D + c], g[D + c]);
    if (tl != pad_eid) atomicAdd(&dE[tl * D + c], g[2 * D + c]);
  }
}

// masked mean over the sequence axis of time-major rows: g[b] = sum_t m[b,t] x[(t,b)] / max(1, sum_t m[b,t])  (models.py:86-91)
// Workgroup (b, 64 columns): the four waves take t = w, w + 4, ... each (with one workgroup per graph and one thread walking
// all T rows of a column the 16 graphs x 212 triples of wd-articles took 600 us for 20 MB), partials meet in LDS in wave order.
__global__ __launch_bounds__(256) void seq_pool_fwd_kernel(const float* __restrict__ x, const unsigned char* __restrict__ kmask,
                                                           float* __restrict__ g, float* __restrict__ inv_cnt, int B, int T, int W) {
  __shared__ float part[4][64];
  __shared__ int cnts[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  int cnt = 0;
  float a = 0.f;
  for (int t = wave; t < T; t += 4) {
    const bool on = !kmask || kmask[(long)b * T + t];
    cnt += on ? 1 : 0;
    if (on && c < W) a += x[((long)t * B + b) * W + c];
  }
  part[wave][lane] = a;
  if (lane == 0) cnts[wave] = cnt;
  __syncthreads();
  if (wave != 0) return;
  const float ic = 1.0f / (float)max(cnts[0] + cnts[1] + cnts[2] + cnts[3], 1);
  if (c < W) g[(long)b * W + c] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) * ic;
  if (lane == 0 && blockIdx.y == 0) inv_cnt[b] = ic;
}

__global__ __launch_bounds__(256) void seq_pool_bwd_kernel(const float* __restrict__ dg, const unsigned char* __restrict__ kmask,
                                                           const float* __restrict__ inv_cnt, float* __restrict__ dx, int B, int T, int W) {
  const long n = (long)T * B * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / W;
    const int c = (int)(i % W), t = (int)(row / B), b = (int)(row % B);
    dx[i] = (!kmask || kmask[(long)b * T + t]) ? dg[(long)b * W + c] * inv_cnt[b] : 0.f;
  }
}

// Cross-attention of t-SAIL's decoder (nn.TransformerDecoderLayer.multihead_attn over a memory that is L copies of ONE row,
// reference models.py:112): all L scores of a query are equal, the softmax is uniform and the context is the value row
// itself -- times c[t, b, h] = (#keys the attention dropout keeps) / (L (1 - p)) in training mode, exactly what dropping
// uniform probabilities does.  ctx[(t, b), h*dh + d] = c[t, b, h] * v[b, h*dh + d]; the query / key projections do not reach
// the output (their gradients are exactly zero).  One wave per (t, b) row.
__global__ __launch_bounds__(256) void xattn_bcast_fwd_kernel(const float* __restrict__ v, float* __restrict__ ctx,
                                                              float* __restrict__ cscale, int B, int L, int D, int H, float p,
                                                              uint64_t seed, const float* __restrict__ hyper) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L * B) return;
  const int t = row / B, b = row % B, dh = D / H;
  for (int h = 0; h < H; ++h) {
    float c = 1.0f;
    if (p > 0.f) {
      const DropCtx dc = drop_ctx(seed, hyper, p);
      const long base = (((long)b * H + h) * L + t) * L;
      float kept = 0.f;
      for (int j = lane; j < L; j += 64) kept += dropout_one(dc, (uint64_t)(base + j));   // keep-scale 1/(1-p) or 0
      c = wave_sum(kept) / (float)L;
    }
    if (lane == 0) cscale[(long)row * H + h] = c;
    for (int d = lane; d < dh; d += 64) ctx[(long)row * D + h * dh + d] = c * v[(long)b * D + h * dh + d];
  }
}

// dv[b, h*dh + d] = sum_t c[t, b, h] * dctx[(t, b), h*dh + d]
__global__ __launch_bounds__(256) void xattn_bcast_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ cscale,
                                                              float* __restrict__ dv, int B, int L, int D, int H) {
  __shared__ float part[4][64];   // workgroup (b, 64 columns), the time axis split over the waves as in seq_pool_fwd_kernel
  const int b = blockIdx.x, dh = D / H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const int h = min(c, D - 1) / dh;
  float a = 0.f;
  if (c < D)
    for (int t = wave; t < L; t += 4) a += cscale[((long)t * B + b) * H + h] * dctx[((long)t * B + b) * D + c];
  part[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < D) dv[(long)b * D + c] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// the same for WIDE rows (D > 1536: t-SAIL's 3 * d_model encoder at d_model = 1024): keeping g and xhat of a row in
// registers would take 4 * D / 64 of them, so the row is walked twice (the second walk re-reads what the first left in L2)
// and only the two per-column sums stay resident
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 float* __restrict__ ds, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int rows, int D, int rows_per_wg) {
  __shared__ float red[2][64 * CPL];   // (the four waves add into it one after the other: same lane -> same columns)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[CPL], ab[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) ag[k] = ab[k] = 0.f;
  const int r0 = blockIdx.x * rows_per_wg;
  for (int row = r0 + wave; row < min(rows, r0 + rows_per_wg); row += 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < D) {
        const float g = dy[(long)row * D + c] * gamma[c];
        s1 += g;
        s2 += g * (s[(long)row * D + c] - mean) * rstd;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < D) {
        const float dyv = dy[(long)row * D + c];
        const float xh = (s[(long)row * D + c] - mean) * rstd;
        ds[(long)row * D + c] = rstd * (dyv * gamma[c] - s1 - xh * s2);
        ag[k] += dyv * xh;
        ab[k] += dyv;
      }
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        if (w == 0) { red[0][lane + 64 * k] = ag[k]; red[1][lane + 64 * k] = ab[k]; }
        else { red[0][lane + 64 * k] += ag[k]; red[1][lane + 64 * k] += ab[k]; }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(&dgamma[c], red[0][c]);
    atomicAdd(&dbeta[c], red[1][c]);
  }
}

// ONE pass where the backward of a sublayer made four (copy, dropout, bias column sum, 16-bit cast of the product operand):
//   v = x * keep-scale (drop_p > 0; the mask of `seed` and the current draw, element index row * cols + col as
//       ark_dropout_apply on a buffer of that shape);   x_out (nullable, may alias x) = v;   out16 = cast(v);
//   colsum[col] += sum_rows v  (nullable).
// 4 waves per workgroup walk the rows of a 256-column slab, lane -> 4 consecutive columns (one dropout hash per quad).
template <int PREC>
__global__ __launch_bounds__(256) void prep16_kernel(const float* __restrict__ x, float* x_out, void* __restrict__ out16_,
                                                     float* __restrict__ colsum, int rows, int cols, int rows_per_wg, float drop_p,
                                                     uint64_t seed, const float* __restrict__ hyper) {
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  typedef h_t h4_t __attribute__((ext_vector_type(4)));
  __shared__ f32x4 red[4][64];
  h_t* out16 = reinterpret_cast<h_t*>(out16_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  const bool drop = drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(seed, hyper, drop_p);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (int r = r0 + wave; r < r1; r += 4) {
      const long i = (long)r * cols + c;
      f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      if (drop) v = v * dropout_quad(dc, (uint64_t)i >> 2);
      if (x_out) *reinterpret_cast<f32x4*>(x_out + i) = v;
      *reinterpret_cast<h4_t*>(out16 + i) = h4_t{PT::cvt(v[0]), PT::cvt(v[1]), PT::cvt(v[2]), PT::cvt(v[3])};
      acc += v;
    }
  }
  if (!colsum) return;   // (uniform)
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < cols) {
    const f32x4 t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&colsum[c + e], t[e]);
  }
}

static int attn_check(const AttnArgs& p) {
  if (p.B <= 0 || p.L <= 0 || p.D <= 0 || p.H <= 0 || p.D % p.H != 0) return ARK_ERR_ARG;
  if (p.L > 64 * kAttnMaxChunks || p.dh > kAttnMaxDh || p.dh % 4 != 0) return ARK_ERR_SHAPE;
  if (p.drop_p < 0.f || p.drop_p >= 1.f || (p.drop_p > 0.f && !p.hyper)) return ARK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(p.qkv) & 15) != 0) return ARK_ERR_ALIGN;
  return 0;
}

}  // namespace ark

extern "C" int ark_layernorm_fwd_drop(const float* x, const float* res, const float* gamma, const float* beta, float* s_out, float* y,
                                      float* stats, int rows, int D, float eps, float drop_p, uint64_t seed, const float* hyper,
                                      void* stream) {
  if (!x || !gamma || !beta || !y || !stats || rows <= 0 || D <= 0) return ARK_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!hyper || !res))) return ARK_ERR_ARG;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
#define ARK_LN_FWD(NV) hipLaunchKernelGGL((ark::layernorm_fwd_vec_kernel<NV>), grid, dim3(256), 0, st, x, res, gamma, beta, s_out, y, stats, rows, D, eps, drop_p, seed, hyper)
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(s_out) |
                    reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
  if (al && D % 256 == 0 && D <= 3072) {
    switch (D / 256) {
      case 1: ARK_LN_FWD(1); break;
      case 2: ARK_LN_FWD(2); break;
      case 3: ARK_LN_FWD(3); break;
      case 4: ARK_LN_FWD(4); break;
      case 6: ARK_LN_FWD(6); break;
      case 8: ARK_LN_FWD(8); break;
      case 12: ARK_LN_FWD(12); break;
      default: hipLaunchKernelGGL(ark::layernorm_fwd_kernel, grid, dim3(256), 0, st, x, res, gamma, beta, s_out, y, stats, rows, D, eps, drop_p, seed, hyper);
    }
  } else {
    hipLaunchKernelGGL(ark::layernorm_fwd_kernel, grid, dim3(256), 0, st, x, res, gamma, beta, s_out, y, stats, rows, D, eps, drop_p, seed, hyper);
  }
#undef ARK_LN_FWD
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* s_out, float* y,
                                 float* stats, int rows, int D, float eps, void* stream) {
  return ark_layernorm_fwd_drop(x, res, gamma, beta, s_out, y, stats, rows, D, eps, 0.f, 0, nullptr, stream);
}

extern "C" int ark_layernorm_bwd(const float* dy, const float* s, const float* stats, const float* gamma, float* ds, float* dgamma,
                                 float* dbeta, int rows, int D, void* stream) {
  using namespace ark;
  if (!dy || !s || !stats || !gamma || !ds || !dgamma || !dbeta || rows <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D > 64 * 48) return ARK_ERR_SHAPE;   // (3 * d_model at d_model = 1024: the reference's syn-types / syn-tipr YAML)
  int rpw = (rows + 1023) / 1024;          // up to ~1024 workgroups (t-ARK step: 3.35 / 3.27 / 3.25 ms at 256 / 512 / 1024), whole groups of 8 rows
  rpw = (rpw + 7) / 8 * 8;
  const unsigned grid = (unsigned)((rows + rpw - 1) / rpw);
  hipStream_t st = (hipStream_t)stream;
#define ARK_LN_BWD(C) hipLaunchKernelGGL((layernorm_bwd_kernel<C>), dim3(grid), dim3(256), 0, st, dy, s, stats, gamma, ds, dgamma, dbeta, rows, D, rpw)
  if (D <= 64) ARK_LN_BWD(1);
  else if (D <= 128) ARK_LN_BWD(2);
  else if (D <= 256) ARK_LN_BWD(4);
  else if (D <= 512) ARK_LN_BWD(8);
  else if (D <= 1024) ARK_LN_BWD(16);
  else if (D <= 1536) ARK_LN_BWD(24);
  else hipLaunchKernelGGL((layernorm_bwd_wide_kernel<48>), dim3(grid), dim3(256), 0, st, dy, s, stats, gamma, ds, dgamma, dbeta, rows, D, rpw);
#undef ARK_LN_BWD
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_dropout_apply(float* x, int64_t n, float p, uint64_t seed, const float* hyper, void* stream) {
  using namespace ark;
  if (!x || !hyper || n <= 0 || p <= 0.f || p >= 1.f) return ARK_ERR_ARG;
  if (n % 4 != 0) return ARK_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return ARK_ERR_ALIGN;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)(n / 4), seed, hyper, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_prep16(int prec, const float* x, float* x_out, void* out16, float* colsum, int rows, int cols, float drop_p,
                          uint64_t seed, const float* hyper, void* stream) {
  using namespace ark;
  if (!x || !out16 || rows <= 0 || cols <= 0) return ARK_ERR_ARG;
  if (cols % 4 != 0) return ARK_ERR_SHAPE;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !hyper)) return ARK_ERR_ARG;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x_out)) & 15) || (reinterpret_cast<uintptr_t>(out16) & 7)) return ARK_ERR_ALIGN;
  int rpw = (rows + 159) / 160;   // ~160 row chunks x the 256-column slabs
  rpw = (rpw + 3) / 4 * 4;
  const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)((rows + rpw - 1) / rpw));
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F16) hipLaunchKernelGGL(prep16_kernel<PREC_F16>, grid, dim3(256), 0, st, x, x_out, out16, colsum, rows, cols, rpw, drop_p, seed, hyper);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(prep16_kernel<PREC_BF16>, grid, dim3(256), 0, st, x, x_out, out16, colsum, rows, cols, rpw, drop_p, seed, hyper);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_attn_fwd(const float* qkv, float* out, float* probs, const unsigned char* kmask, int B, int L, int D, int n_heads,
                            int causal, float drop_p, uint64_t seed, const float* hyper, void* stream) {
  using namespace ark;
  if (!qkv || !out || !probs) return ARK_ERR_ARG;
  AttnArgs p{qkv, out, probs, nullptr, nullptr, nullptr, hyper, kmask, seed, drop_p, 0.f, B, L, D, n_heads, n_heads > 0 ? D / n_heads : 0, causal};
  int rc = attn_check(p);
  if (rc) return rc;
  p.scale = 1.0f / sqrtf((float)p.dh);
  int ff = 0, bf = 0;
  const int nw = attn_small_waves(p, &ff, &bf);
  if (nw > 0) {   // short sequences: one wave per (batch, head)
    hipLaunchKernelGGL(attn_small_fwd_kernel, dim3((unsigned)((B * n_heads + nw - 1) / nw)), dim3(64 * nw), (size_t)nw * ff * sizeof(float),
                       (hipStream_t)stream, p, ff);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * n_heads), (unsigned)((L + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_attn_bwd(const float* qkv, const float* out, const float* probs, const float* dout, float* dscore, float* dqkv,
                            const unsigned char* kmask, int B, int L, int D, int n_heads, int causal, float drop_p, uint64_t seed,
                            const float* hyper, void* stream) {
  using namespace ark;
  if (!qkv || !out || !probs || !dout || !dscore || !dqkv) return ARK_ERR_ARG;
  AttnArgs p{qkv, const_cast<float*>(out), const_cast<float*>(probs), dout, dscore, dqkv, hyper, kmask, seed, drop_p, 0.f, B, L, D, n_heads,
             n_heads > 0 ? D / n_heads : 0, causal};
  int rc = attn_check(p);
  if (rc) return rc;
  p.scale = 1.0f / sqrtf((float)p.dh);
  int ff = 0, bf = 0;
  const int nw = attn_small_waves(p, &ff, &bf);
  if (nw > 0) {   // short sequences: one wave per (batch, head); `dscore` is not used
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3((unsigned)((B * n_heads + nw - 1) / nw)), dim3(64 * nw), (size_t)nw * bf * sizeof(float),
                       (hipStream_t)stream, p, bf);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((unsigned)(B * n_heads), (unsigned)((L + 3) / 4));
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_triple_gather(const int64_t* triples, const float* E, const float* R, float* x, unsigned char* kmask, int B, int T,
                                 int D, int64_t pad_rid, void* stream) {
  if (!triples || !E || !R || !x || B <= 0 || T <= 0 || D <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::triple_gather_kernel, dim3((unsigned)((T * B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, triples, E, R, x,
                     kmask, B, T, D, (long)pad_rid);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_triple_scatter(const int64_t* triples, const float* dx, float* dE, float* dR, int B, int T, int D, int64_t pad_eid,
                                  int64_t pad_rid, void* stream) {
  if (!triples || !dx || !dE || !dR || B <= 0 || T <= 0 || D <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::triple_scatter_kernel, dim3((unsigned)((T * B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, triples, dx, dE,
                     dR, B, T, D, (long)pad_eid, (long)pad_rid);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_seq_pool_fwd(const float* x, const unsigned char* kmask, float* g, float* inv_cnt, int B, int T, int W, void* stream) {
  if (!x || !g || !inv_cnt || B <= 0 || T <= 0 || W <= 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::seq_pool_fwd_kernel, dim3((unsigned)B, (unsigned)((W + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, kmask, g, inv_cnt, B, T, W);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_seq_pool_bwd(const float* dg, const unsigned char* kmask, const float* inv_cnt, float* dx, int B, int T, int W,
                                void* stream) {
  if (!dg || !inv_cnt || !dx || B <= 0 || T <= 0 || W <= 0) return ARK_ERR_ARG;
  long blocks = ((long)T * B * W + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ark::seq_pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dg, kmask, inv_cnt, dx, B, T, W);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_xattn_bcast_fwd(const float* v, float* ctx, float* cscale, int B, int L, int D, int n_heads, float drop_p,
                                   uint64_t seed, const float* hyper, void* stream) {
  if (!v || !ctx || !cscale || B <= 0 || L <= 0 || D <= 0 || n_heads <= 0 || D % n_heads != 0) return ARK_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !hyper)) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::xattn_bcast_fwd_kernel, dim3((unsigned)((L * B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, v, ctx, cscale, B,
                     L, D, n_heads, drop_p, seed, hyper);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_xattn_bcast_bwd(const float* dctx, const float* cscale, float* dv, int B, int L, int D, int n_heads, void* stream) {
  if (!dctx || !cscale || !dv || B <= 0 || L <= 0 || D <= 0 || n_heads <= 0 || D % n_heads != 0) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::xattn_bcast_bwd_kernel, dim3((unsigned)B, (unsigned)((D + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dctx, cscale, dv, B, L, D, n_heads);
  ARK_LAUNCH_CHECK();
  return 0;
}
