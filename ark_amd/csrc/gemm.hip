// Generic dense contraction C[M,N] = op(A)[M,K] * op(B)[N,K]^T with fused epilogues.
// One kernel family serves every "plain" product on the SAIL/ARK path; see gemm_core.h for the
// tile engine.  Reference ops replaced: nn.Linear forward/backward in
// kgvae/model/models.py:36,43-44,128 and the time-batched nn.GRU input products (:121-127).
#include "gemm_core.h"
#include "../../include/ark_amd.h"

namespace ark {

// (speed choices are compile-time constants: the library keeps no mutable state, include/ark_amd.h)
constexpr bool g_split_k_enabled = true;   // long-K products reduce over workgroups with fp32 atomics
constexpr bool g_wgrad_tile128 = false;    // 128x128 tiles + deeper split-K for weight-gradient products: measured slower

struct GemmArgs {
  const float* A; const float* B; float* C; float* C2; const float* bias; const float* aux;
  long lda, ldb, ldc;
  int M, N, K;
  int epi, accumulate;
  int tiles_n;
  int split_k, k_chunk;  // split_k > 1: blockIdx.y owns K range [y*k_chunk, ...) and adds atomically into zeroed C
};

template <int PREC, int ALAY, int BLAY, int BM, int BN, int ASRC16 = 0, int BSRC16 = 0>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  using G = GemmTile<PREC, ALAY, BLAY, BM, BN, 2, 2, ASRC16, BSRC16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = p.M, N = p.N;
  f32x4 acc[G::TM][G::TN];
  // split-K: this workgroup reduces k in [kb, ke); operands are offset along their reduction index
  const int kb = blockIdx.y * p.k_chunk;
  const int ke = min(p.K, kb + p.k_chunk);
  // (16-bit sources are half as wide: offsets are in elements of the operand's own type)
  const char* Ab = reinterpret_cast<const char*>(p.A) + (ALAY == LAY_KMAJ ? (long)kb : (long)kb * p.lda) * (ASRC16 ? 2 : 4);
  const char* Bb = reinterpret_cast<const char*>(p.B) + (BLAY == LAY_KMAJ ? (long)kb : (long)kb * p.ldb) * (BSRC16 ? 2 : 4);
  G::run(acc, Ab, p.lda, [=](int r) -> long { return (m0 + r < M) ? (long)(m0 + r) : -1L; },
         Bb, p.ldb, [=](int r) -> long { return (n0 + r < N) ? (long)(n0 + r) : -1L; }, ke - kb, smem);

  if (p.split_k > 1) {  // EPI_NONE only (checked on the host); C was zeroed on the stream
    G::for_each(acc, [&](int r, int c, float v) {
      const int row = m0 + r, col = n0 + c;
      if (row < M && col < N) atomicAdd(&p.C[(long)row * p.ldc + col], v);
    });
    return;
  }
  const int epi = p.epi;
  G::for_each(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (row >= M || col >= N) return;
    const long o = (long)row * p.ldc + col;
    if (epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_GELU || epi == ARK_EPI_BIAS_RELU) v += p.bias[col];
    if (epi == ARK_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
    if (epi == ARK_EPI_MUL_RELU) v = p.aux[o] > 0.f ? v : 0.f;
    if (epi == ARK_EPI_BIAS_GELU) {
      p.C[o] = v;               // pre-activation, kept for the backward pass
      p.C2[o] = gelu_erf(v);    // activation
      return;
    }
    if (epi == ARK_EPI_MUL_DGELU) v *= dgelu_erf(p.aux[o]);
    if (epi == ARK_EPI_MUL_AUX) v *= p.aux[o];
    if (p.accumulate) v += p.C[o];
    p.C[o] = v;
  });
}

template <int PREC, int ALAY, int BLAY, int ASRC16 = 0, int BSRC16 = 0>
static int launch_gemm(GemmArgs p, hipStream_t st) {
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  constexpr int BK = PrecTraits<PREC>::BK;
  p.split_k = 1;
  p.k_chunk = p.K;
  const bool long_k = p.epi == ARK_EPI_NONE && g_split_k_enabled && p.K >= 16 * BK;
  if (t128 >= 192 || (g_wgrad_tile128 && long_k && t128 >= 16)) {
    p.tiles_n = (p.N + 127) / 128;
    if (t128 < 192) {
      int split = 1;
      while (t128 * split < 768 && p.K / (split * 2) >= 4 * BK && split < 64) split *= 2;
      if (split > 1) {
        p.k_chunk = ((p.K + split - 1) / split + BK - 1) / BK * BK;
        p.split_k = (p.K + p.k_chunk - 1) / p.k_chunk;
        if (!p.accumulate) {
          hipError_t e = (p.ldc == p.N) ? hipMemsetAsync(p.C, 0, sizeof(float) * (size_t)p.M * p.N, st)
                                        : hipMemset2DAsync(p.C, sizeof(float) * p.ldc, 0, sizeof(float) * p.N, p.M, st);
          if (e != hipSuccess) return (int)e;
        }
      }
    }
    using G = GemmTile<PREC, ALAY, BLAY, 128, 128, 2, 2, ASRC16, BSRC16>;
    hipLaunchKernelGGL((gemm_kernel<PREC, ALAY, BLAY, 128, 128, ASRC16, BSRC16>), dim3((unsigned)t128, (unsigned)p.split_k), dim3(256), G::LDS_BYTES, st, p);
  } else {
    p.tiles_n = (p.N + 63) / 64;
    const long t64 = (long)((p.M + 63) / 64) * p.tiles_n;
    // Long-K products with few output tiles (weight gradients: K = B*L) are latency-bound at
    // <= 1 workgroup per CU; split K across workgroups until ~4 workgroups per CU are resident,
    // keeping >= 4 K-steps per split, and reduce with fp32 atomics into the zeroed output.
    if (p.epi == ARK_EPI_NONE && g_split_k_enabled) {
      int split = 1;
      while (t64 * split < 1024 && p.K / (split * 2) >= 4 * BK && split < 64) split *= 2;
      if (split > 1) {
        p.split_k = split;
        p.k_chunk = ((p.K + split - 1) / split + BK - 1) / BK * BK;
        p.split_k = (p.K + p.k_chunk - 1) / p.k_chunk;
      }
    }
    if (p.split_k > 1 && !p.accumulate) {
      // rows of C are ldc apart; zero the [M, N] window (dense when ldc == N)
      hipError_t e = (p.ldc == p.N) ? hipMemsetAsync(p.C, 0, sizeof(float) * (size_t)p.M * p.N, st)
                                    : hipMemset2DAsync(p.C, sizeof(float) * p.ldc, 0, sizeof(float) * p.N, p.M, st);
      if (e != hipSuccess) return (int)e;
    }
    using G = GemmTile<PREC, ALAY, BLAY, 64, 64, 2, 2, ASRC16, BSRC16>;
    hipLaunchKernelGGL((gemm_kernel<PREC, ALAY, BLAY, 64, 64, ASRC16, BSRC16>), dim3((unsigned)t64, (unsigned)p.split_k), dim3(256),
                       G::LDS_BYTES, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC>
static int dispatch_lay(int a_lay, int b_lay, const GemmArgs& p, hipStream_t st) {
  if (a_lay == LAY_KMAJ && b_lay == LAY_KMAJ) return launch_gemm<PREC, LAY_KMAJ, LAY_KMAJ>(p, st);
  if (a_lay == LAY_KMAJ && b_lay == LAY_MMAJ) return launch_gemm<PREC, LAY_KMAJ, LAY_MMAJ>(p, st);
  if (a_lay == LAY_MMAJ && b_lay == LAY_KMAJ) return launch_gemm<PREC, LAY_MMAJ, LAY_KMAJ>(p, st);
  if (a_lay == LAY_MMAJ && b_lay == LAY_MMAJ) return launch_gemm<PREC, LAY_MMAJ, LAY_MMAJ>(p, st);
  return ARK_ERR_ARG;
}

}  // namespace ark

extern "C" int ark_gemm(int prec, int a_lay, int b_lay, int epi, const float* A, int64_t lda, const float* B,
                        int64_t ldb, float* C, int64_t ldc, float* C2, const float* bias, const float* aux,
                        int M, int N, int K, int accumulate, void* stream) {
  using namespace ark;
  if (M <= 0 || N <= 0 || K < 0 || !A || !B || !C) return ARK_ERR_ARG;
  if ((epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_GELU || epi == ARK_EPI_BIAS_RELU) && !bias) return ARK_ERR_ARG;
  if (epi == ARK_EPI_BIAS_GELU && !C2) return ARK_ERR_ARG;
  if ((epi == ARK_EPI_MUL_DGELU || epi == ARK_EPI_MUL_AUX || epi == ARK_EPI_MUL_RELU) && !aux) return ARK_ERR_ARG;
  if (epi < 0 || epi > ARK_EPI_MUL_RELU) return ARK_ERR_ARG;
  GemmArgs p{A, B, C, C2, bias, aux, (long)lda, (long)ldb, (long)ldc, M, N, K, epi, accumulate, 0, 1, K};
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F32) return dispatch_lay<PREC_F32>(a_lay, b_lay, p, st);
  if (prec == PREC_BF16) return dispatch_lay<PREC_BF16>(a_lay, b_lay, p, st);
  if (prec == PREC_F16) return dispatch_lay<PREC_F16>(a_lay, b_lay, p, st);
  return ARK_ERR_ARG;
}

// Weight-gradient products C[M,N] = sum_k A[k, M] * B[k, N] (both operands reduction-major) where
// one or both operands are stored in 16 bits (the type `prec` computes in).  Same kernel family as
// ark_gemm; a_is16 / b_is16 select the storage of each operand (0: fp32, 1: 16-bit).
extern "C" int ark_gemm_wgrad(int prec, const void* A, int a_is16, int64_t lda, const void* B, int b_is16, int64_t ldb,
                              float* C, int64_t ldc, int M, int N, int K, int accumulate, void* stream) {
  using namespace ark;
  if (M <= 0 || N <= 0 || K < 0 || !A || !B || !C) return ARK_ERR_ARG;
  if (prec != PREC_BF16 && prec != PREC_F16) return ARK_ERR_ARG;
  GemmArgs p{reinterpret_cast<const float*>(A), reinterpret_cast<const float*>(B), C, nullptr, nullptr, nullptr,
             (long)lda, (long)ldb, (long)ldc, M, N, K, ARK_EPI_NONE, accumulate ? 1 : 0, 0, 1, K};
  hipStream_t st = (hipStream_t)stream;
#define ARK_WG(P)                                                                                         \
  if (a_is16 && b_is16) return launch_gemm<P, LAY_MMAJ, LAY_MMAJ, 1, 1>(p, st);                          \
  if (!a_is16 && b_is16) return launch_gemm<P, LAY_MMAJ, LAY_MMAJ, 0, 1>(p, st);                         \
  if (!a_is16 && !b_is16) return launch_gemm<P, LAY_MMAJ, LAY_MMAJ, 0, 0>(p, st);                        \
  return ARK_ERR_ARG;
  if (prec == PREC_BF16) { ARK_WG(PREC_BF16) }
  ARK_WG(PREC_F16)
#undef ARK_WG
}
