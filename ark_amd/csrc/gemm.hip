// Generic dense contraction C[M,N] = op(A)[M,K] * op(B)[N,K]^T with fused epilogues.
// One kernel family serves every "plain" product on the SAIL/ARK path; see gemm_core.h for the
// tile engine.  Reference ops replaced: nn.Linear forward/backward in
// kgvae/model/models.py:36,43-44,128 and the time-batched nn.GRU input products (:121-127).
#include "gemm_core.h"
#include "../../include/ark_amd.h"

namespace ark {

struct GemmArgs {
  const float* A; const float* B; float* C; float* C2; const float* bias; const float* aux;
  long lda, ldb, ldc;
  int M, N, K;
  int epi, accumulate;
  int tiles_n;
};

template <int PREC, int ALAY, int BLAY, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  using G = GemmTile<PREC, ALAY, BLAY, BM, BN, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = p.M, N = p.N;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, p.A, p.lda, [=](int r) -> long { return (m0 + r < M) ? (long)(m0 + r) : -1L; },
         p.B, p.ldb, [=](int r) -> long { return (n0 + r < N) ? (long)(n0 + r) : -1L; }, p.K, smem);

  const int epi = p.epi;
  G::for_each(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (row >= M || col >= N) return;
    const long o = (long)row * p.ldc + col;
    if (epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_GELU) v += p.bias[col];
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

template <int PREC, int ALAY, int BLAY>
static int launch_gemm(GemmArgs p, hipStream_t st) {
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (t128 >= 192) {
    p.tiles_n = (p.N + 127) / 128;
    using G = GemmTile<PREC, ALAY, BLAY, 128, 128, 2, 2>;
    hipLaunchKernelGGL((gemm_kernel<PREC, ALAY, BLAY, 128, 128>), dim3((unsigned)t128), dim3(256), G::LDS_BYTES, st, p);
  } else {
    p.tiles_n = (p.N + 63) / 64;
    const long t64 = (long)((p.M + 63) / 64) * p.tiles_n;
    using G = GemmTile<PREC, ALAY, BLAY, 64, 64, 2, 2>;
    hipLaunchKernelGGL((gemm_kernel<PREC, ALAY, BLAY, 64, 64>), dim3((unsigned)t64), dim3(256), G::LDS_BYTES, st, p);
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
  if ((epi == ARK_EPI_BIAS || epi == ARK_EPI_BIAS_GELU) && !bias) return ARK_ERR_ARG;
  if (epi == ARK_EPI_BIAS_GELU && !C2) return ARK_ERR_ARG;
  if ((epi == ARK_EPI_MUL_DGELU || epi == ARK_EPI_MUL_AUX) && !aux) return ARK_ERR_ARG;
  if (epi < 0 || epi > ARK_EPI_MUL_AUX) return ARK_ERR_ARG;
  GemmArgs p{A, B, C, C2, bias, aux, (long)lda, (long)ldb, (long)ldc, M, N, K, epi, accumulate, 0};
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F32) return dispatch_lay<PREC_F32>(a_lay, b_lay, p, st);
  if (prec == PREC_BF16) return dispatch_lay<PREC_BF16>(a_lay, b_lay, p, st);
  if (prec == PREC_F16) return dispatch_lay<PREC_F16>(a_lay, b_lay, p, st);
  return ARK_ERR_ARG;
}
