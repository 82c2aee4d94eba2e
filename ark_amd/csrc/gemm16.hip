// C[M,N] = A16[M,K] * B16[N,K]^T on the LDS-DMA ring engine (dma_core.h): both operands are 16-bit,
// K-contiguous, row-major; fp32 accumulate; fused bias / mask epilogue; C either row-major or in
// the MFMA-tile-native order the GRU cell epilogues consume (tile_native_off).
// Serves the time-batched GRU input products, the tied vocabulary projection and the
// input-gradient products (against transposed 16-bit weight shadows).  Reference ops replaced:
// nn.GRU input projection (kgvae/model/models.py:121-127), nn.Linear `out` (:128,142) and their
// autograd input-gradients.
#include "dma_core.h"
// Diagnostic build only (-DARK_STAMPS, tools/wpk_stamps.py): 100-MHz real-time stamps per WAVE of the wave-private
// K-slice kernel -- entry, ring primed, first slice landed, main loop done, rings free, tiles summed, stores drained.
#ifdef ARK_STAMPS
namespace ark { __device__ unsigned long long ark_wpk_stamp_buf[1024 * 8 * 8]; }
#define WPK_STAMP(i)                                                                                         \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                        \
      ark::ark_wpk_stamp_buf[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#endif
#include "wpk_core.h"
#include "../../include/ark_amd.h"

namespace ark {

struct Gemm16Args {
  const void* A; const void* B; float* C; const float* bias; const float* aux;
  long lda, ldb, ldc;
  int M, N, K, epi, c_tiled, tiles_n;
  void* c16a; void* c16b; int prec_a, prec_b;   // optional row-major 16-bit copies of the result (ld = ldc)
  float* colsum;   // optional: colsum[col] += sum over rows of the value written to C (bias gradient of a dpre output)
};

// What the epilogue of one 16 x 16 accumulator tile READS per lane (MFMA C/D order: this lane holds rows row0 .. row0+3 of
// column col): the aux / old-C values of its four elements and its column's bias.  The kernel gathers these for ALL of a
// lane's tiles in front of its first store (round 5): read inside the per-element loop, each load sat behind the previous
// element's store -- which may alias it as far as the compiler knows -- and the epilogue was a chain of TM x TN x 4 memory
// round trips (s_waitcnt vmcnt(0) after every load), longer than the product itself for the encoder's K = 512 input gradients.
struct G16Pre { f32x4 x; float bias; };
__device__ __forceinline__ G16Pre g16_preload(const Gemm16Args& p, int row0, int col) {
  G16Pre q{f32x4{0.f, 0.f, 0.f, 0.f}, 0.f};
  const int M = p.M, N = p.N, epi = p.epi;
  if (row0 >= M || col >= N) return q;
  if (epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_RELU || epi == ARK_EPI_BIAS_GELU) q.bias = p.bias[col];
  if (p.c_tiled) {
    if (epi == ARK_EPI_MUL_AUX) q.x = *reinterpret_cast<const f32x4*>(p.aux + tile_native_off(row0, col, (int)p.ldc));
    return q;
  }
  const float* src = (epi == ARK_EPI_MUL_AUX || epi == ARK_EPI_MUL_DGELU || epi == ARK_EPI_MUL_RELU) ? p.aux : epi == ARK_EPI_ADD ? p.C : nullptr;
  if (src) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (row0 + i < M) q.x[i] = src[(long)(row0 + i) * p.ldc + col];
  }
  return q;
}
// epilogue of that tile from the gathered values; returns the lane's contribution to the column sum of what it wrote
// (bias gradient of a dpre output)
__device__ __forceinline__ float g16_epilogue(const Gemm16Args& p, f32x4 v, int row0, int col, const G16Pre& q) {
  const int M = p.M, N = p.N;
  if (row0 >= M || col >= N) return 0.f;   // (the column-sum shuffles run outside: all lanes take part)
  if (p.epi == ARK_EPI_BIAS || p.epi == ARK_EPI_BIAS_RELU) v += q.bias;
  if (p.c_tiled) {  // M % 16 == 0 and ldc % 16 == 0 (checked on the host): the quad is whole
    const long o = tile_native_off(row0, col, (int)p.ldc);
    if (p.epi == ARK_EPI_MUL_AUX) v *= q.x;
    *reinterpret_cast<f32x4*>(p.C + o) = v;
    return 0.f;
  }
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (row0 + i >= M) break;
    const long o = (long)(row0 + i) * p.ldc + col;
    float x = v[i];
    if (p.epi == ARK_EPI_MUL_AUX) x *= q.x[i];
    if (p.epi == ARK_EPI_MUL_DGELU) x *= dgelu_fast(q.x[i]);
    if (p.epi == ARK_EPI_BIAS_RELU) x = fmaxf(x, 0.f);
    if (p.epi == ARK_EPI_MUL_RELU) x = q.x[i] > 0.f ? x : 0.f;
    if (p.epi == ARK_EPI_ADD) x += q.x[i];   // (each element belongs to exactly one lane of one workgroup)
    if (p.epi == ARK_EPI_BIAS_GELU) {
      x += q.bias;
      p.C[o] = x;            // pre-activation (fp32, kept for the backward pass)
      x = gelu_fast(x);      // the 16-bit copies carry the activation
    } else {
      if (p.C) p.C[o] = x;
      cs += x;
    }
    if (p.c16a) put16(p.c16a, o, x, p.prec_a);
    if (p.c16b) put16(p.c16b, o, x, p.prec_b);
  }
  return cs;
}
// the four lane groups of a wave hold different rows of the same column: fold them, one atomic per column per wave
__device__ __forceinline__ void g16_colsum(const Gemm16Args& p, float cs, int col) {
  cs += __shfl_xor(cs, 16, 64);
  cs += __shfl_xor(cs, 32, 64);
  if ((threadIdx.x & 63) < 16 && col < p.N) atomicAdd(&p.colsum[col], cs);
}

template <int PREC, int BM, int BN, int NBUF>
__global__ __launch_bounds__(256) void gemm16_kernel(Gemm16Args p) {
  ARK_CHAIN_PRIO();
  using G = DmaTile<PREC, BM, BN, NBUF, 2, 2>;
  using h_t = typename G::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;
  const int M = p.M, N = p.N;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, reinterpret_cast<const h_t*>(p.A), p.lda, [=](int r) -> long { return (long)min(m0 + r, M - 1); },
         reinterpret_cast<const h_t*>(p.B), p.ldb, [=](int r) -> long { return (long)min(n0 + r, N - 1); }, p.K, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float csum[G::TN];
#pragma unroll
  for (int tn = 0; tn < G::TN; ++tn) csum[tn] = 0.f;
  G16Pre pre[G::TM][G::TN];
#pragma unroll
  for (int tm = 0; tm < G::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::TN; ++tn)
      pre[tm][tn] = g16_preload(p, m0 + wm * G::WTM + tm * 16 + 4 * (lane >> 4), n0 + wn * G::WTN + tn * 16 + (lane & 15));
#pragma unroll
  for (int tm = 0; tm < G::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::TN; ++tn)
      csum[tn] += g16_epilogue(p, acc[tm][tn], m0 + wm * G::WTM + tm * 16 + 4 * (lane >> 4), n0 + wn * G::WTN + tn * 16 + (lane & 15),
                               pre[tm][tn]);
  if (p.colsum && !p.c_tiled) {
#pragma unroll
    for (int tn = 0; tn < G::TN; ++tn) g16_colsum(p, csum[tn], n0 + wn * G::WTN + tn * 16 + (lane & 15));
  }
}

// 16-bit quad store (run-time element type)
__device__ __forceinline__ void put16x4(void* base, long idx, f32x4 v, int prec) {
  if (prec == PREC_F16) {
    using PT = PrecTraits<PREC_F16>;
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(base) + idx) = f16x4{PT::cvt(v[0]), PT::cvt(v[1]), PT::cvt(v[2]), PT::cvt(v[3])};
  } else {
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
}

// The same product on the wave-private K-slice engine (wpk_core.h): one workgroup per CU, BM x BN tile, every wave runs
// the whole tile over its own K-slices, partial tiles summed through LDS and written in row-major quads;
// M % BM == 0, N % BN == 0, K % 64 == 0, row-major C (nullable when a 16-bit copy is asked for: an input gradient that
// is only read as the next product's operand), ldc % 4 == 0, 16-byte aligned C / aux / bias, 8-byte aligned copies.
template <int PREC, int BM, int BN, int NW, int NSLOT, bool CLAMP = false>
__global__ __launch_bounds__(64 * NW) void gemm16_wpk_kernel(Gemm16Args p) {
  ARK_CHAIN_PRIO();
  using G = WpkNT<PREC, BM, BN, NW, NSLOT>;
  using h_t = typename G::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_m = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  wpk_tile_of(lid, tiles_m, p.tiles_n, 8, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  WPK_STAMP(0);
  f32x4 acc[G::TM][G::TN];
#pragma unroll
  for (int tm = 0; tm < G::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  G::template run<CLAMP>(acc, reinterpret_cast<const h_t*>(p.A) + (long)m0 * p.lda, p.lda,
                         reinterpret_cast<const h_t*>(p.B) + (long)n0 * p.ldb, p.ldb, p.K, 0, smem, min(BM, p.M - m0), min(BN, p.N - n0));
  float* cs_lds = G::colsum_lds(smem);
  const int epi = p.epi;
  G::reduce_rows(acc, smem, [&](int row, int col, f32x4 v) {
    // (clamped tiles: a quad beyond M / N -- N % 4 == 0: inside or outside as a whole -- touches no memory, but its lanes stay in
    //  the column-sum shuffles below with zeros: an early return would leave their partners reading dead registers)
    const bool valid = !CLAMP || (m0 + row < p.M && n0 + col < p.N);
    const long o = valid ? (long)(m0 + row) * p.ldc + n0 + col : 0;
    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid && (epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_RELU || epi == ARK_EPI_BIAS_GELU)) v += *reinterpret_cast<const f32x4*>(p.bias + n0 + col);
    if (valid && epi == ARK_EPI_MUL_AUX) v *= *reinterpret_cast<const f32x4*>(p.aux + o);
    if (valid && epi == ARK_EPI_MUL_DGELU) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= dgelu_fast(a[e]);
    }
    if (epi == ARK_EPI_BIAS_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (valid && epi == ARK_EPI_MUL_RELU) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
    }
    if (valid && epi == ARK_EPI_ADD) v += *reinterpret_cast<const f32x4*>(p.C + o);
    if (valid && p.C) *reinterpret_cast<f32x4*>(p.C + o) = v;   // BIAS_GELU: the pre-activation
    if (epi == ARK_EPI_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);   // the 16-bit copies carry the activation
    } else if (p.colsum) {
      // the 8 lanes l, l + 8, ... of a wave hold the same four columns of 8 different rows: fold them, then one LDS add
      // per column and wave
      f32x4 c = v;
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] += __shfl_xor(c[e], o, 64);
      if ((threadIdx.x & 63) < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&cs_lds[col + e], c[e]);
      }
    }
    if (valid && p.c16a) put16x4(p.c16a, o, v, p.prec_a);
    if (valid && p.c16b) put16x4(p.c16b, o, v, p.prec_b);
  });
  if (p.colsum) {
    __syncthreads();
    if ((int)threadIdx.x < BN && n0 + (int)threadIdx.x < p.N) atomicAdd(&p.colsum[n0 + threadIdx.x], cs_lds[threadIdx.x]);
  }
#ifdef ARK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (diagnostic build: the stores have left the wave)
  WPK_STAMP(6);
#endif
}

template <class K>
static void allow_lds16(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

#ifndef ARK_G16_NBUF
#define ARK_G16_NBUF 2
#endif
#ifndef ARK_G16_TILE
#define ARK_G16_TILE 0
#endif
constexpr int g_g16_nbuf = ARK_G16_NBUF;     // ring depth (2 -> 64 KB LDS at 128x128, two workgroups per CU)
constexpr int g_g16_force64 = 0;  // 1: always use 64x64 tiles

template <int PREC, int BM, int BN, int NBUF>
static void launch16_cfg(Gemm16Args p, hipStream_t st) {
  using G = DmaTile<PREC, BM, BN, NBUF, 2, 2>;
  static bool once = (allow_lds16(gemm16_kernel<PREC, BM, BN, NBUF>, G::LDS_BYTES), true); (void)once;
  p.tiles_n = (p.N + BN - 1) / BN;
  const long tiles = (long)((p.M + BM - 1) / BM) * p.tiles_n;
  hipLaunchKernelGGL((gemm16_kernel<PREC, BM, BN, NBUF>), dim3((unsigned)tiles), dim3(256), G::LDS_BYTES, st, p);
}

constexpr int g_g16_tile = ARK_G16_TILE;   // 0: heuristic below, 1: 64x64, 2: 128x64, 3: 128x128, 4: 32x64, 5: 64x32

#ifndef ARK_WPK_WAVES
#define ARK_WPK_WAVES 8
#endif
template <int PREC, int BM, int BN>
static void launch16_wpk(Gemm16Args p, hipStream_t st) {
  constexpr int NW = ARK_WPK_WAVES, NSLOT = NW == 8 ? 1 : 2;   // 160 KB of ring either way (64 x 96 tile)
  using G = WpkNT<PREC, BM, BN, NW, NSLOT>;
  static bool once = (allow_lds16(gemm16_wpk_kernel<PREC, BM, BN, NW, NSLOT>, G::LDS_BYTES), true); (void)once;
  p.tiles_n = p.N / BN;
  const long tiles = (long)(p.M / BM) * p.tiles_n;
  hipLaunchKernelGGL((gemm16_wpk_kernel<PREC, BM, BN, NW, NSLOT>), dim3((unsigned)tiles), dim3(64 * NW), G::LDS_BYTES, st, p);
}

// few tiles, deep K (the latent heads: [1024 x 2Z] over K = 3D = 1536 -- 32 tiles whose 24 ring stages are a dependent chain
// of ~0.5 us each): 32 x 64 tiles whose eight waves take three K-slices each; rows / columns beyond M / N are clamped
template <int PREC>
static void launch16_wpk_deep(Gemm16Args p, hipStream_t st) {
  using G = WpkNT<PREC, 32, 64, 8, 1>;
  static bool once = (allow_lds16(gemm16_wpk_kernel<PREC, 32, 64, 8, 1, true>, G::LDS_BYTES), true); (void)once;
  p.tiles_n = (p.N + 63) / 64;
  const long tiles = (long)((p.M + 31) / 32) * p.tiles_n;
  hipLaunchKernelGGL((gemm16_wpk_kernel<PREC, 32, 64, 8, 1, true>), dim3((unsigned)tiles), dim3(512), G::LDS_BYTES, st, p);
}
static bool wpk_deep(const Gemm16Args& p) {
  if (p.K % 64 != 0 || p.K < 1024 || p.c_tiled || p.ldc % 4 != 0 || p.N % 4 != 0) return false;
  if (((uintptr_t)p.C | (uintptr_t)p.aux | (uintptr_t)p.bias) & 15) return false;
  if (((uintptr_t)p.c16a | (uintptr_t)p.c16b) & 7) return false;
  return (long)((p.M + 31) / 32) * ((p.N + 63) / 64) <= 64;
}

// engine: 0 = choose, 1 = the shared ring (dma_core.h), 2 = wave-private K-slices (wpk_core.h; ARK_ERR_SHAPE if the
// shape does not fit it).  WPK wants one tile per CU: it is chosen when its 64 x 96 (or 32 x 96) tiling gives between
// half a chip and two chips of workgroups and K is deep enough for every wave to own at least two slices.
static int wpk_rows(const Gemm16Args& p) {
  if (p.N % 96 != 0 || p.K % 64 != 0 || p.K < 512 || p.c_tiled || p.ldc % 4 != 0) return 0;
  if (((uintptr_t)p.C | (uintptr_t)p.aux | (uintptr_t)p.bias) & 15) return 0;
  if (((uintptr_t)p.c16a | (uintptr_t)p.c16b) & 7) return 0;
  const long tn = p.N / 96;
#ifdef ARK_WPK_PREFER64
  if (p.M % 64 == 0 && (p.M / 64) * tn >= 128 && (p.M / 64) * tn <= 512) return 64;
#else
  // 64-row tiles where they fill at least three quarters of the chip, else 32-row tiles (B = 256 x 3 D = 3072: 128 tiles of
  // 64 rows would leave half the CUs idle; 256 tiles of 32 rows stream 786 instead of 983 KB each)
  if (p.M % 64 == 0 && (p.M / 64) * tn >= 192 && (p.M / 64) * tn <= 512) return 64;
#endif
  if (p.M % 32 == 0 && (p.M / 32) * tn >= 128 && (p.M / 32) * tn <= 512) return 32;
  if (p.M % 64 == 0 && (p.M / 64) * tn >= 128 && (p.M / 64) * tn <= 512) return 64;
  return 0;
}

template <int PREC>
static int launch16(Gemm16Args p, int engine, hipStream_t st) {
  const int bm = wpk_rows(p);
  // the few-tiles / deep-K flavour is taken only where a caller asks for it (engine 2 or 3): it sums K in another order than
  // the ring, and a model whose later layers amplify 1e-6 of the forward (the Transformer feed-forward: a random 1e-6 on
  // linear2's output moves linear1's gradients by 1e-2 through ReLU / LayerNorm, measured) would no longer compare against
  // its register-staged twin at that test's 5e-3
  const bool deep = !bm && (engine == 2 || engine == 3) && wpk_deep(p);
  if (engine == 2 && !bm && !deep) return ARK_ERR_SHAPE;
  if (engine == 3) engine = 0;
#ifdef ARK_G16_NO_WPK   // (A/B builds: the library's own choice stays on the shared ring)
  if (engine == 0) engine = 1;
#endif
  if (bm && engine != 1) {
    if (bm == 64) launch16_wpk<PREC, 64, 96>(p, st); else launch16_wpk<PREC, 32, 96>(p, st);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  if (deep && engine != 1) {
    launch16_wpk_deep<PREC>(p, st);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  int tile = g_g16_tile;
  if (tile == 0) {
    // these products are latency-bound, not MFMA-bound: what counts is how many workgroups a CU can keep
    // resident.  64x64 tiles unless that leaves fewer than two workgroups per CU -- then 32x64 tiles, 24 KB of LDS each
    const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    tile = (g_g16_force64 || t64 >= 512) ? 1 : 4;
  }
  if (tile == 4) {   // small tiles: more workgroups resident per CU for short-M products
    if (g_g16_nbuf == 4) launch16_cfg<PREC, 32, 64, 4>(p, st); else launch16_cfg<PREC, 32, 64, 2>(p, st);
  } else if (tile == 5) {
    if (g_g16_nbuf == 4) launch16_cfg<PREC, 64, 32, 4>(p, st); else launch16_cfg<PREC, 64, 32, 2>(p, st);
  } else if (tile == 3) {
    if (g_g16_nbuf == 3) launch16_cfg<PREC, 128, 128, 3>(p, st); else launch16_cfg<PREC, 128, 128, 2>(p, st);
  } else if (tile == 2) {
    if (g_g16_nbuf == 4) launch16_cfg<PREC, 128, 64, 4>(p, st);
    else if (g_g16_nbuf == 3) launch16_cfg<PREC, 128, 64, 3>(p, st);
    else launch16_cfg<PREC, 128, 64, 2>(p, st);
  } else {
    if (g_g16_nbuf == 4) launch16_cfg<PREC, 64, 64, 4>(p, st); else launch16_cfg<PREC, 64, 64, 2>(p, st);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

// ---- decoder input: x16[(t,b), :] = cast(W_tok[seq[b,t]] (+ W_pos[t])) in one or two 16-bit types ----
template <int PA, int PB2>
__global__ __launch_bounds__(256) void tok_gather16_kernel(const int64_t* __restrict__ seq, long ld_seq,
                                                           const float* __restrict__ Wt, const float* __restrict__ Wp,
                                                           void* xa_, void* xb_, int B, int L, int D, float* hyper_tick) {
  using HA = typename PrecTraits<PA>::h_t;
  // a training forward draws fresh dropout masks: bump the draw counter the GRU cells hash (they run after this
  // kernel in stream order; the previous step's backward ran before it)
  if (hyper_tick && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t*>(hyper_tick)[kHpDropStep] += 1u;
  using HB = typename PrecTraits<PB2>::h_t;
  typedef HA ha4 __attribute__((ext_vector_type(4)));
  typedef HB hb4 __attribute__((ext_vector_type(4)));
  const int D4 = D >> 2;
  const long total = (long)B * L * D4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const long row = i / D4;
    const int t = (int)(row / B), b = (int)(row % B);
    const long tok = seq[(long)b * ld_seq + t];
    f32x4 v = *reinterpret_cast<const f32x4*>(Wt + tok * D + 4 * d4);
    if (Wp) v += *reinterpret_cast<const f32x4*>(Wp + (long)t * D + 4 * d4);
    reinterpret_cast<ha4*>(xa_)[row * D4 + d4] = ha4{PrecTraits<PA>::cvt(v[0]), PrecTraits<PA>::cvt(v[1]), PrecTraits<PA>::cvt(v[2]), PrecTraits<PA>::cvt(v[3])};
    if (xb_) reinterpret_cast<hb4*>(xb_)[row * D4 + d4] = hb4{PrecTraits<PB2>::cvt(v[0]), PrecTraits<PB2>::cvt(v[1]), PrecTraits<PB2>::cvt(v[2]), PrecTraits<PB2>::cvt(v[3])};
  }
}

// ---- per-token sums of a 16-bit panel: S[v, c] += sum over rows (t, b) with seq[b, t] == v of X16[(t, b), c] ----
// With S = the token sums of layer 0's gate-gradient panel, two exact identities replace a [B*L, D] input-gradient
// product + scatter and a whole weight-gradient product (small vocabularies, no position embedding):
//   dW_tok  += onehot^T dX0 = S W_ih_0            (autograd embedding_backward of models.py:138)
//   dW_ih_0 += dgi_0^T X0   = S^T W_tok           (X0 rows ARE rows of W_tok)
// A tiny vocabulary means MANY rows per token (every row of step 0 is BOS; a relation slot has 3 values), so adding
// rows into a shared table with atomics serialises.  Instead the rows of each chunk are counting-sorted by token ONCE
// (token_sort_kernel: two passes of LDS integer atomics over <= a few hundred rows), and every (token, column) sum is
// then a plain loop over that token's row list -- the only atomics left are one per table entry.
__global__ __launch_bounds__(256) void token_sort_kernel(const int64_t* __restrict__ seq, long ld_seq, int B, int R, int Vp,
                                                         int rows_per_chunk, short* __restrict__ perm, short* __restrict__ ptok,
                                                         int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem_sort[];
  int* cnt = reinterpret_cast<int*>(smem_sort);        // [Vp + 1] counts, then exclusive starts
  int* fill = cnt + Vp + 1;                            // [Vp]
  short* tk = reinterpret_cast<short*>(fill + Vp);     // [rows_per_chunk]
  const int r0 = blockIdx.x * rows_per_chunk, nr = min(R, r0 + rows_per_chunk) - r0;
  for (int i = threadIdx.x; i <= Vp; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < nr; i += 256) {
    const int r = r0 + i, t = r / B, b = r % B;
    int tok = (int)seq[(long)b * ld_seq + t];
    if (tok < 0 || tok >= Vp) tok = Vp;                // out-of-range ids are dropped (list Vp is never summed)
    tk[i] = (short)tok;
    if (tok < Vp) atomicAdd(&cnt[tok], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {                              // Vp <= 256: a serial exclusive scan is a few hundred cycles
    int run = 0;
    for (int v = 0; v < Vp; ++v) { const int c = cnt[v]; cnt[v] = run; fill[v] = run; run += c; }
    cnt[Vp] = run;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr; i += 256) {
    const int tok = tk[i];
    if (tok < Vp) {
      const int pos = atomicAdd(&fill[tok], 1);
      perm[r0 + pos] = (short)i;
      ptok[r0 + pos] = (short)tok;
    }
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = cnt[Vp];   // rows of this chunk that carry a token in range
}

template <int PREC>
__global__ __launch_bounds__(256) void token_sums16_kernel(const short* __restrict__ perm, const short* __restrict__ ptok,
                                                           const int* __restrict__ counts, const void* X_, long ldx,
                                                           float* __restrict__ S, long lds_, int Vp, int rows_per_chunk) {
  using H = typename PrecTraits<PREC>::h_t;
  typedef H h4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem_ts[];
  float* tab = reinterpret_cast<float*>(smem_ts);   // [Vp][64]
  const H* X = reinterpret_cast<const H*>(X_);
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * rows_per_chunk, n = counts[blockIdx.y];
  for (int i = threadIdx.x; i < Vp * 64; i += 256) tab[i] = 0.f;
  __syncthreads();
  // 16 row lanes x 16 column quads.  A row lane walks a CONTIGUOUS piece of the token-sorted row list, so it meets a
  // handful of distinct tokens: it sums in registers and touches the LDS table only when the token changes.
  const int rl = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  const int per = (n + 15) / 16, lo = min(n, rl * per), hi = min(n, lo + per);
  const short* pm = perm + r0;
  const short* pt = ptok + r0;
  int cur = -1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto flush = [&]() {
    if (cur >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&tab[cur * 64 + c4 + e], acc[e]);
    }
  };
  for (int k = lo; k < hi; k += 4) {
    h4v x[4];
    int tk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // four independent loads in flight per thread
      const int kk = min(k + u, hi - 1);
      tk[u] = pt[kk];
      x[u] = *reinterpret_cast<const h4v*>(X + (long)(r0 + pm[kk]) * ldx + c0 + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k + u >= hi) break;
      if (tk[u] != cur) {
        flush();
        cur = tk[u];
        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (float)x[u][e];
    }
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < Vp * 64; i += 256) {
    const float v = tab[i];
    if (v != 0.f) atomicAdd(&S[(long)(i >> 6) * lds_ + c0 + (i & 63)], v);
  }
}

// out[n] = sum_m X16[m, n]   (bias gradients from the 16-bit gate-gradient panels)
template <int PREC>
__global__ __launch_bounds__(256) void colsum16_kernel(const void* X_, long ld, float* __restrict__ out, int M, int N, int rows_per_wg) {
  using H = typename PrecTraits<PREC>::h_t;
  const H* X = reinterpret_cast<const H*>(X_);
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
  float s = 0.f;
  if (col < N)
    for (int r = m0 + wave; r < m1; r += 4) s += (float)X[(long)r * ld + col];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) atomicAdd(&out[col], red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// the same sums with 16-byte loads: a lane owns 8 columns, a wave reads 1 KB of a row per instruction and keeps four rows in
// flight (round 5: the kernel above issues one 2-byte load per lane and iteration and waits for it -- 15 us for the 42 MB
// gate-gradient panel of syn-paths).  N % 8 == 0, ld % 8 == 0, 16-byte aligned X.
template <int PREC>
__global__ __launch_bounds__(256) void colsum16_wide_kernel(const void* X_, long ld, float* __restrict__ out, int M, int N, int rows_per_wg) {
  using PT = PrecTraits<PREC>;
  using H = typename PT::h_t;
  using H8 = typename PT::h8;
  const H* X = reinterpret_cast<const H*>(X_);
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int m0 = blockIdx.y * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
  constexpr int RF = 8;   // rows in flight per wave
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    for (int r = m0 + wave; r < m1; r += 4 * RF) {
      H8 v[RF];
#pragma unroll
      for (int u = 0; u < RF; ++u) {
        const int rr = min(r + 4 * u, m1 - 1);   // (a row past the end re-reads the last one and is not added)
        v[u] = *reinterpret_cast<const H8*>(X + (long)rr * ld + col);
      }
#pragma unroll
      for (int u = 0; u < RF; ++u)
        if (r + 4 * u < m1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int cc = blockIdx.x * 512 + c;
    if (cc < N) atomicAdd(&out[cc], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ x, void* out_, long n) {
  using H = typename PrecTraits<PREC>::h_t;
  H* out = reinterpret_cast<H*>(out_);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = PrecTraits<PREC>::cvt(x[i]);
}

template <int PREC>
__global__ __launch_bounds__(256) void uncast16_kernel(const void* x_, float* __restrict__ out, long n) {
  using H = typename PrecTraits<PREC>::h_t;
  const H* x = reinterpret_cast<const H*>(x_);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)x[i];
}

// row-major fp32 [rows, ld] -> tile-native fp32 (rows % 16 == 0, ld % 16 == 0)
__global__ __launch_bounds__(256) void to_tiled_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int ld) {
  const long n4 = (long)rows * ld / 4;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    // q indexes tile-native float4s: decode (row0, col)
    const long tile = q >> 6;
    const int lane = (int)(q & 63);
    const int tr = (int)(tile / (ld >> 4)), tc = (int)(tile % (ld >> 4));
    const int row0 = tr * 16 + 4 * (lane >> 4), col = tc * 16 + (lane & 15);
    f32x4 v = {x[(long)row0 * ld + col], x[(long)(row0 + 1) * ld + col], x[(long)(row0 + 2) * ld + col], x[(long)(row0 + 3) * ld + col]};
    reinterpret_cast<f32x4*>(out)[q] = v;
  }
}

// the inverse: tile-native fp32 -> row-major [rows, ld] (the fused vocabulary CE writes dY tile-native; the Transformer
// backward reads gradients row-major)
__global__ __launch_bounds__(256) void from_tiled_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int ld) {
  const long n4 = (long)rows * ld / 4;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    const long tile = q >> 6;
    const int lane = (int)(q & 63);
    const int tr = (int)(tile / (ld >> 4)), tc = (int)(tile % (ld >> 4));
    const int row0 = tr * 16 + 4 * (lane >> 4), col = tc * 16 + (lane & 15);
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(long)(row0 + i) * ld + col] = v[i];
  }
}

}  // namespace ark

static int gemm16_impl(int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                       int64_t ldc, const float* bias, const float* aux, int M, int N, int K, int c_tiled, void* c16a,
                       void* c16b, int prec_b, float* colsum, int engine, void* stream) {
  using namespace ark;
  if (!A16 || !B16 || M <= 0 || N <= 0 || K <= 0) return ARK_ERR_ARG;
  // C may be NULL when a 16-bit copy is asked for (the product is only read as the next product's 16-bit operand), except
  // where C itself carries information the copies do not (BIAS_GELU: the pre-activation; ADD: the running sum)
  if (!C && (!c16a || c_tiled || epi == ARK_EPI_BIAS_GELU || epi == ARK_EPI_ADD)) return ARK_ERR_ARG;
  if (K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0) return ARK_ERR_SHAPE;
  if (((uintptr_t)A16 | (uintptr_t)B16) & 15) return ARK_ERR_ALIGN;
  if (epi < ARK_EPI_NONE || epi > ARK_EPI_ADD) return ARK_ERR_ARG;
  if (epi == ARK_EPI_ADD && (c_tiled || c16a || c16b || colsum)) return ARK_ERR_ARG;
  if ((epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_GELU || epi == ARK_EPI_BIAS_RELU) && !bias) return ARK_ERR_ARG;
  if ((epi == ARK_EPI_MUL_AUX || epi == ARK_EPI_MUL_DGELU || epi == ARK_EPI_MUL_RELU) && !aux) return ARK_ERR_ARG;
  if (c_tiled && (epi == ARK_EPI_BIAS_RELU || epi == ARK_EPI_MUL_RELU)) return ARK_ERR_ARG;
  if (c_tiled && (M % 16 != 0 || ldc % 16 != 0 || N > ldc)) return ARK_ERR_SHAPE;
  if (c_tiled && (epi == ARK_EPI_BIAS_GELU || epi == ARK_EPI_MUL_DGELU || c16a || c16b || colsum)) return ARK_ERR_ARG;
  if (colsum && epi == ARK_EPI_BIAS_GELU) return ARK_ERR_ARG;
  if (c16b && prec_b != PREC_F16 && prec_b != PREC_BF16) return ARK_ERR_ARG;
  Gemm16Args p{A16, B16, C, bias, aux, (long)lda, (long)ldb, (long)ldc, M, N, K, epi, c_tiled ? 1 : 0, 0, c16a, c16b, prec, prec_b, colsum};
  if (engine < 0 || engine > 3) return ARK_ERR_ARG;
  if (engine == 2 && (((uintptr_t)A16 | (uintptr_t)B16) & 15)) return ARK_ERR_ALIGN;
  if (prec == PREC_F16) return launch16<PREC_F16>(p, engine, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch16<PREC_BF16>(p, engine, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_gemm16(int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                          int64_t ldc, const float* bias, const float* aux, int M, int N, int K, int c_tiled,
                          void* stream) {
  return gemm16_impl(prec, epi, A16, lda, B16, ldb, C, ldc, bias, aux, M, N, K, c_tiled, nullptr, nullptr, 0, nullptr, 0, stream);
}

// same product with row-major 16-bit copies of the result written by the epilogue: c16a in `prec`,
// c16b (nullable) in `prec_b`.  BIAS_GELU: C = pre-activation (fp32), copies = gelu(C);
// MUL_DGELU: C = acc * gelu'(aux), copies = C.  colsum (nullable, not with BIAS_GELU): colsum[col] += sum_rows C.
extern "C" int ark_gemm16_ex(int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                             int64_t ldc, const float* bias, const float* aux, void* c16a, void* c16b, int prec_b,
                             float* colsum, int M, int N, int K, void* stream) {
  return gemm16_impl(prec, epi, A16, lda, B16, ldb, C, ldc, bias, aux, M, N, K, 0, c16a, c16b, prec_b, colsum, 0, stream);
}

// ark_gemm16_ex on a chosen engine (tests, A/B timing): 0 = the library's choice, 1 = shared ring, 2 = wave-private
// K-slices (ARK_ERR_SHAPE unless M % 32 == 0, N % 96 == 0, K % 64 == 0, K >= 512 and the tiling fills half a chip, or the
// product is a few tiles over K >= 1024), 3 = the library's choice INCLUDING the few-tiles / deep-K flavour
extern "C" int ark_gemm16_engine(int engine, int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb,
                                 float* C, int64_t ldc, const float* bias, const float* aux, void* c16a, void* c16b,
                                 int prec_b, float* colsum, int M, int N, int K, void* stream) {
  return gemm16_impl(prec, epi, A16, lda, B16, ldb, C, ldc, bias, aux, M, N, K, 0, c16a, c16b, prec_b, colsum, engine, stream);
}

extern "C" int ark_tok_gather16(int prec_a, int prec_b, const int64_t* seq, int64_t ld_seq, const float* w_tok,
                                const float* w_pos, void* x16a, void* x16b, int B, int L, int D, float* hyper_tick, void* stream) {
  using namespace ark;
  if (!seq || !w_tok || !x16a || B <= 0 || L <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 4 != 0) return ARK_ERR_SHAPE;
  long total = (long)B * L * (D / 4);
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipStream_t st = (hipStream_t)stream;
#define ARK_TG(PA, PB2) hipLaunchKernelGGL((tok_gather16_kernel<PA, PB2>), dim3(grid), dim3(256), 0, st, seq, (long)ld_seq, w_tok, w_pos, x16a, x16b, B, L, D, hyper_tick)
  if (prec_a == PREC_F16 && prec_b == PREC_BF16) ARK_TG(PREC_F16, PREC_BF16);
  else if (prec_a == PREC_F16 && prec_b == PREC_F16) ARK_TG(PREC_F16, PREC_F16);
  else if (prec_a == PREC_BF16 && prec_b == PREC_BF16) ARK_TG(PREC_BF16, PREC_BF16);
  else return ARK_ERR_ARG;
#undef ARK_TG
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_token_sums16(int prec, const int64_t* seq, int64_t ld_seq, const void* x16, int64_t ldx, float* S,
                                int64_t ld_s, void* scratch, int64_t scratch_bytes, int B, int L, int Vp, int n_cols,
                                void* stream) {
  using namespace ark;
  if (!seq || !x16 || !S || !scratch || B <= 0 || L <= 0 || Vp <= 0 || n_cols <= 0) return ARK_ERR_ARG;
  if (n_cols % 64 != 0 || ldx % 4 != 0 || (long)Vp * 64 * 4 > 64 * 1024) return ARK_ERR_SHAPE;
  const int R = B * L;
  int chunks = 512 / (n_cols / 64);   // ~512 workgroups in the sum kernel
  if (chunks < 1) chunks = 1;
  int rows_per_chunk = ((R + chunks - 1) / chunks + 15) / 16 * 16;
  if (rows_per_chunk < 64) rows_per_chunk = 64;
  if (rows_per_chunk > 8192) rows_per_chunk = 8192;   // (row-in-chunk indices are 16-bit; LDS holds one per row)
  chunks = (R + rows_per_chunk - 1) / rows_per_chunk;
  const size_t lst = ((size_t)R * sizeof(short) + 15) / 16 * 16;
  if ((size_t)scratch_bytes < 2 * lst + (size_t)chunks * sizeof(int)) return ARK_ERR_ARG;
  short* perm = reinterpret_cast<short*>(scratch);
  short* ptok = reinterpret_cast<short*>(reinterpret_cast<char*>(scratch) + lst);
  int* counts = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + 2 * lst);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(2 * Vp + 1) * sizeof(int) + (size_t)rows_per_chunk * sizeof(short);
  hipLaunchKernelGGL(token_sort_kernel, dim3(chunks), dim3(256), lds, st, seq, (long)ld_seq, B, R, Vp, rows_per_chunk, perm, ptok, counts);
  dim3 grid(n_cols / 64, chunks);
  const size_t tlds = (size_t)Vp * 64 * sizeof(float);
  if (prec == PREC_F16) hipLaunchKernelGGL(token_sums16_kernel<PREC_F16>, grid, dim3(256), tlds, st, perm, ptok, counts, x16, (long)ldx, S, (long)ld_s, Vp, rows_per_chunk);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(token_sums16_kernel<PREC_BF16>, grid, dim3(256), tlds, st, perm, ptok, counts, x16, (long)ldx, S, (long)ld_s, Vp, rows_per_chunk);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_colsum16(int prec, const void* x16, int64_t ld, float* out, int M, int N, int accumulate,
                            void* stream) {
  using namespace ark;
  if (!x16 || !out || M <= 0 || N <= 0) return ARK_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, st);
    if (e != hipSuccess) return (int)e;
  }
  if (prec != PREC_F16 && prec != PREC_BF16) return ARK_ERR_ARG;
  if ((N + 7) / 8 * 8 <= ld && ld % 8 == 0 && ((uintptr_t)x16 & 15) == 0) {   // (a lane's 8 columns may reach into the row's padding)
    int rows_per_wg = 256;
    const int col_tiles = (N + 511) / 512;
    // (10 240 x 2 048: 15.1 / 12.8 / 14.4 / 21.7 us at 32 / 64 / 128 / 256 rows per workgroup -- 640 workgroups; a NARROW panel
    //  -- the [10 240, 45] dlogits of syn-paths -- keeps >= 32 KB per workgroup, 40 workgroups instead of 640 ending in one atomic
    //  per column each: 15.3 -> 12.8 us, most of which is the fixed cost of a stand-alone launch over 1.3 MB)
    const long row_bytes = 2L * (N < 512 ? (N + 7) / 8 * 8 : 512);
    while (rows_per_wg > 16 && (long)col_tiles * ((M + rows_per_wg - 1) / rows_per_wg) < 600 && (rows_per_wg / 2) * row_bytes >= 32768)
      rows_per_wg >>= 1;
    dim3 grid(col_tiles, (M + rows_per_wg - 1) / rows_per_wg);
    if (prec == PREC_F16) hipLaunchKernelGGL(colsum16_wide_kernel<PREC_F16>, grid, dim3(256), 0, st, x16, (long)ld, out, M, N, rows_per_wg);
    else hipLaunchKernelGGL(colsum16_wide_kernel<PREC_BF16>, grid, dim3(256), 0, st, x16, (long)ld, out, M, N, rows_per_wg);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  int rows_per_wg = 128;
  const int col_tiles = (N + 63) / 64;
  while (rows_per_wg > 16 && (long)col_tiles * ((M + rows_per_wg - 1) / rows_per_wg) < 512) rows_per_wg >>= 1;
  dim3 grid(col_tiles, (M + rows_per_wg - 1) / rows_per_wg);
  if (prec == PREC_F16) hipLaunchKernelGGL(colsum16_kernel<PREC_F16>, grid, dim3(256), 0, st, x16, (long)ld, out, M, N, rows_per_wg);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(colsum16_kernel<PREC_BF16>, grid, dim3(256), 0, st, x16, (long)ld, out, M, N, rows_per_wg);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_cast16(int prec, const float* x, void* out, int64_t n, void* stream) {
  using namespace ark;
  if (!x || !out || n <= 0) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (prec == PREC_F16) hipLaunchKernelGGL(cast16_kernel<PREC_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(cast16_kernel<PREC_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_uncast16(int prec, const void* x16, float* out, int64_t n, void* stream) {
  using namespace ark;
  if (!x16 || !out || n <= 0) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (prec == PREC_F16) hipLaunchKernelGGL(uncast16_kernel<PREC_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x16, out, (long)n);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(uncast16_kernel<PREC_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x16, out, (long)n);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_to_tiled(const float* x, float* out, int rows, int ld, void* stream) {
  using namespace ark;
  if (!x || !out || rows <= 0 || ld <= 0) return ARK_ERR_ARG;
  if (rows % 16 != 0 || ld % 16 != 0) return ARK_ERR_SHAPE;
  long n4 = (long)rows * ld / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(to_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, rows, ld);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_from_tiled(const float* x, float* out, int rows, int ld, void* stream) {
  using namespace ark;
  if (!x || !out || rows <= 0 || ld <= 0) return ARK_ERR_ARG;
  if (rows % 16 != 0 || ld % 16 != 0) return ARK_ERR_SHAPE;
  long n4 = (long)rows * ld / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(from_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, rows, ld);
  ARK_LAUNCH_CHECK();
  return 0;
}

#ifdef ARK_STAMPS
extern "C" int ark_debug_wpk_stamps(unsigned long long* host, int n_blocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ark::ark_wpk_stamp_buf), sizeof(unsigned long long) * 64 * (size_t)n_blocks);
}
#endif

namespace ark {
// h0 = tanh(z Wz^T + bz) written once for every GRU layer in every layout the LDS-DMA path reads:
// row-major fp32 (z-projection backward), tile-native fp32 state, row-major 16-bit operand copies
struct ZprojOut { float* y_t[8]; void* y16a[8]; void* y16b[8]; int n; };
template <int PA, int PB2>
__global__ __launch_bounds__(256) void zproj_fwd_v2_kernel(const float* __restrict__ z, const float* __restrict__ Wz,
                                                           const float* __restrict__ bz, float* __restrict__ h0, ZprojOut o,
                                                           int B, int Z, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * D) return;
  const int b = (int)(i / D), d = (int)(i % D);
  float a = bz[d];
  for (int j = 0; j < Z; ++j) a += z[(long)b * Z + j] * Wz[(long)d * Z + j];
  const float h = tanhf(a);
  h0[i] = h;
  const long ot = tile_native_off(b, d, D);
  for (int l = 0; l < o.n; ++l) {
    o.y_t[l][ot] = h;
    reinterpret_cast<typename PrecTraits<PA>::h_t*>(o.y16a[l])[i] = PrecTraits<PA>::cvt(h);
    if (o.y16b[l]) reinterpret_cast<typename PrecTraits<PB2>::h_t*>(o.y16b[l])[i] = PrecTraits<PB2>::cvt(h);
  }
}
}  // namespace ark

namespace ark {
// reparameterisation + z-projection in ONE launch (they were a single-workgroup kernel and a per-element kernel on the
// dependent chain between the encoder heads and the first GRU diagonal): a workgroup = one batch row x 256 hidden
// units; its first Z threads form mu, clamped logv, z = mu + eps * exp(logv / 2) and the row's KL terms (written by the
// row's first workgroup; the per-row sums are added up in fixed order by ark_loss_finalize: deterministic), every
// thread then forms h0[b, d] = tanh(bz[d] + z . Wz[d, :]) in all layouts the GRU path reads.  Rows >= n_valid (padding
// of a ragged batch) get z = 0 and no KL term.  Reference: models.py:61-63, 139, 199-200.
template <int PA, int PB2>
__global__ __launch_bounds__(256) void latent_zproj_fwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                                               float* __restrict__ mu, float* __restrict__ logv,
                                                               float* __restrict__ z, float* __restrict__ kl_rows,
                                                               const float* __restrict__ Wz, const float* __restrict__ bz,
                                                               float* __restrict__ h0, ZprojOut o, int B, int n_valid, int Z,
                                                               int D, int DC) {
  ARK_CHAIN_PRIO();
  __shared__ float zs[128], ts[128];
  const int b = blockIdx.x / DC, dc = blockIdx.x % DC, tid = threadIdx.x;
  if (tid < Z) {
    float zz = 0.f, term = 0.f;
    if (b < n_valid) {
      const float m = head[(long)b * 2 * Z + tid];
      const float lv = fminf(fmaxf(head[(long)b * 2 * Z + Z + tid], -10.0f), 10.0f);
      zz = m + (eps ? eps[(long)b * Z + tid] : 0.f) * expf(0.5f * lv);
      term = 1.0f + lv - m * m - expf(lv);
      if (dc == 0) {
        mu[(long)b * Z + tid] = m;
        logv[(long)b * Z + tid] = lv;
        z[(long)b * Z + tid] = zz;
      }
    }
    zs[tid] = zz;
    ts[tid] = term;
  }
  __syncthreads();
  if (dc == 0 && tid == 0 && b < n_valid) {
    float s = 0.f;
    for (int j = 0; j < Z; ++j) s += ts[j];
    kl_rows[b] = s;
  }
  const int d = dc * 256 + tid;
  if (d >= D) return;
  float a = bz[d];
  for (int j = 0; j < Z; ++j) a += zs[j] * Wz[(long)d * Z + j];
  const float h = tanhf(a);
  const long i = (long)b * D + d;
  h0[i] = h;
  const long ot = tile_native_off(b, d, D);
  for (int l = 0; l < o.n; ++l) {
    o.y_t[l][ot] = h;
    reinterpret_cast<typename PrecTraits<PA>::h_t*>(o.y16a[l])[i] = PrecTraits<PA>::cvt(h);
    if (o.y16b[l]) reinterpret_cast<typename PrecTraits<PB2>::h_t*>(o.y16b[l])[i] = PrecTraits<PB2>::cvt(h);
  }
}
}  // namespace ark

extern "C" int ark_latent_zproj_fwd(int prec_a, int prec_b, const float* head, const float* eps, float* mu, float* logv,
                                    float* z, float* kl_rows, const float* w_z, const float* b_z, float* h0, int n_layers,
                                    float* const* y_t, void* const* y16a, void* const* y16b, int B, int n_valid, int Z, int D,
                                    void* stream) {
  using namespace ark;
  if (!head || !mu || !logv || !z || !kl_rows || !w_z || !b_z || !h0 || !y_t || !y16a || n_layers <= 0 || n_layers > 8 ||
      B <= 0 || n_valid <= 0 || n_valid > B || Z <= 0 || D <= 0)
    return ARK_ERR_ARG;
  if (B % 16 != 0 || D % 16 != 0 || Z > 128) return ARK_ERR_SHAPE;
  ZprojOut o{};
  o.n = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    if (!y_t[l] || !y16a[l]) return ARK_ERR_ARG;
    o.y_t[l] = y_t[l]; o.y16a[l] = y16a[l]; o.y16b[l] = y16b ? y16b[l] : nullptr;
  }
  const int DC = (D + 255) / 256;
  dim3 grid((unsigned)((long)B * DC));
  hipStream_t st = (hipStream_t)stream;
#define ARK_LZ(PA, PB2) hipLaunchKernelGGL((latent_zproj_fwd_kernel<PA, PB2>), grid, dim3(256), 0, st, head, eps, mu, logv, z, kl_rows, w_z, b_z, h0, o, B, n_valid, Z, D, DC)
  if (prec_a == PREC_F16 && prec_b == PREC_BF16) ARK_LZ(PREC_F16, PREC_BF16);
  else if (prec_a == PREC_F16 && prec_b == PREC_F16) ARK_LZ(PREC_F16, PREC_F16);
  else if (prec_a == PREC_BF16 && prec_b == PREC_BF16) ARK_LZ(PREC_BF16, PREC_BF16);
  else return ARK_ERR_ARG;
#undef ARK_LZ
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_zproj_fwd_v2(int prec_a, int prec_b, const float* z, const float* w_z, const float* b_z, float* h0,
                                int n_layers, float* const* y_t, void* const* y16a, void* const* y16b, int B, int Z, int D,
                                void* stream) {
  using namespace ark;
  if (!z || !w_z || !b_z || !h0 || !y_t || !y16a || n_layers <= 0 || n_layers > 8 || B <= 0 || Z <= 0 || D <= 0) return ARK_ERR_ARG;
  if (B % 16 != 0 || D % 16 != 0) return ARK_ERR_SHAPE;
  ZprojOut o{};
  o.n = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    if (!y_t[l] || !y16a[l]) return ARK_ERR_ARG;
    o.y_t[l] = y_t[l]; o.y16a[l] = y16a[l]; o.y16b[l] = y16b ? y16b[l] : nullptr;
  }
  const long n = (long)B * D;
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define ARK_ZP(PA, PB2) hipLaunchKernelGGL((zproj_fwd_v2_kernel<PA, PB2>), grid, dim3(256), 0, st, z, w_z, b_z, h0, o, B, Z, D)
  if (prec_a == PREC_F16 && prec_b == PREC_BF16) ARK_ZP(PREC_F16, PREC_BF16);
  else if (prec_a == PREC_F16 && prec_b == PREC_F16) ARK_ZP(PREC_F16, PREC_F16);
  else if (prec_a == PREC_BF16 && prec_b == PREC_BF16) ARK_ZP(PREC_BF16, PREC_BF16);
  else return ARK_ERR_ARG;
#undef ARK_ZP
  ARK_LAUNCH_CHECK();
  return 0;
}
