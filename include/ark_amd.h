/* ark_amd C-ABI: hand-written gfx950 (MI355X) kernels for the SAIL / ARK training hot path.
 *
 * The reference (thiviyanT/ARK) has no FFI; its boundary is the Python module API of
 * kgvae/model/models.py.  Every entry point below replaces the stock-torch op sequence cited
 * next to it.  Conventions:
 *   - all pointers are DEVICE pointers owned by the caller (torch-allocated); kernels never
 *     allocate or free; fp32 buffers are dense row-major unless a leading dimension is given;
 *   - integer ids are int64 (torch.long), exactly what GraphSeqDataset yields
 *     (kgvae/model/utils.py:131-146);
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *   - return 0 on success, a positive hipError_t if a launch failed, a negative ARK_ERR_* for
 *     bad arguments.  Calls are asynchronous and stateless (thread-safe).
 *   - decoder activations are TIME-MAJOR: row (t, b) = t * B + b.
 */
#ifndef ARK_AMD_H
#define ARK_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ARK_PREC_F32 0  /* v_mfma_f32_16x16x4_f32, exact fp32 products (parity / decode)   */
#define ARK_PREC_BF16 1 /* v_mfma_f32_16x16x32_bf16, operands rounded to bf16, fp32 accum  */
#define ARK_LAY_KMAJ 0  /* element (row,k) at base[row*ld + k] */
#define ARK_LAY_MMAJ 1  /* element (row,k) at base[k*ld + row] */

#define ARK_EPI_NONE 0      /* C = acc                                   */
#define ARK_EPI_BIAS 1      /* C = acc + bias[col]                       */
#define ARK_EPI_BIAS_GELU 2 /* C = acc + bias (pre-act), C2 = gelu_erf(C) */
#define ARK_EPI_MUL_DGELU 3 /* C = acc * gelu_erf'(aux[row,col])         */
#define ARK_EPI_MUL_AUX 4   /* C = acc * aux[row,col]                    */

int ark_version(void);

/* C[M,N] (+)= A[M,K] * B[N,K]^T with a fused epilogue; replaces nn.Linear forward / backward
 * (reference kgvae/model/models.py:36,43-44,120,128) and the time-batched nn.GRU input
 * products (models.py:121-127).  A/B layouts select which index is contiguous. */
int ark_gemm(int prec, int a_lay, int b_lay, int epi, const float* A, int64_t lda, const float* B, int64_t ldb,
             float* C, int64_t ldc, float* C2, const float* bias, const float* aux, int M, int N, int K,
             int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif
