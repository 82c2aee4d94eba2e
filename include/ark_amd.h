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
 *     bad arguments.  Calls are asynchronous and STATELESS (thread-safe): the library has no mutable globals;
 *     the few speed-only choices that are not compile-time constants travel in per-call structs
 *     (ArkDiagTuning, ArkWgradTuning; NULL = measured defaults).
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
#define ARK_PREC_F16 2  /* v_mfma_f32_16x16x32_f16, operands rounded to fp16 (saturating), fp32 accum */
#define ARK_LAY_KMAJ 0  /* element (row,k) at base[row*ld + k] */
#define ARK_LAY_MMAJ 1  /* element (row,k) at base[k*ld + row] */

#define ARK_EPI_NONE 0      /* C = acc                                   */
#define ARK_EPI_BIAS 1      /* C = acc + bias[col]                       */
#define ARK_EPI_BIAS_GELU 2 /* C = acc + bias (pre-act), C2 = gelu_erf(C) */
#define ARK_EPI_MUL_DGELU 3 /* C = acc * gelu_erf'(aux[row,col])         */
#define ARK_EPI_MUL_AUX 4   /* C = acc * aux[row,col]                    */
#define ARK_EPI_BIAS_RELU 5 /* C = max(acc + bias[col], 0)   (Transformer feed-forward, nn.TransformerEncoderLayer default activation) */
#define ARK_EPI_MUL_RELU 6  /* C = aux[row,col] > 0 ? acc : 0 (its backward: aux = the activation) */
#define ARK_EPI_ADD 7       /* C += acc  (ark_gemm16 only, row-major C: a gradient that joins a residual's)  */

int ark_version(void);

/* C[M,N] (+)= A[M,K] * B[N,K]^T with a fused epilogue; replaces nn.Linear forward / backward
 * (reference kgvae/model/models.py:36,43-44,120,128) and the time-batched nn.GRU input
 * products (models.py:121-127).  A/B layouts select which index is contiguous. */
int ark_gemm(int prec, int a_lay, int b_lay, int epi, const float* A, int64_t lda, const float* B, int64_t ldb,
             float* C, int64_t ldc, float* C2, const float* bias, const float* aux, int M, int N, int K,
             int accumulate, void* stream);


/* ---- device-resident step scalars ("hyper" array, ARK_HP_COUNT floats) -------------------------
 * Everything that changes between steps lives in device memory so a captured hipGraph of the
 * whole train step can be replayed; the host (or ark_adam_tick / ark_count_targets) updates it. */
#define ARK_HP_LR 0            /* Adam learning rate (CosineAnnealingLR value, ablation_study.py:577) */
#define ARK_HP_BETA 1          /* KL weight b of `ce + b*kl` (ablation_study.py:71,590-591)            */
#define ARK_HP_KL_NORM 2       /* 1 / (B_global * Z): kl_mean is a mean over all B*Z elements          */
#define ARK_HP_CE_INV_COUNT 3  /* 1 / (#non-PAD targets in the GLOBAL batch)                           */
#define ARK_HP_CE_COUNT 4      /* #non-PAD targets                                                     */
#define ARK_HP_ADAM_STEP 5     /* optimiser step count t (float)                                       */
#define ARK_HP_ADAM_BC1 6      /* 1 - b1^t                                                             */
#define ARK_HP_ADAM_BC2 7      /* 1 - b2^t                                                             */
#define ARK_HP_ADAM_B1 8
#define ARK_HP_ADAM_B2 9
#define ARK_HP_ADAM_EPS 10
#define ARK_HP_GRAD_SCALE 11   /* multiplies every gradient inside Adam (1 for summed DP gradients)    */
#define ARK_HP_DROP_STEP 12    /* uint32 (bit pattern in the float slot): dropout draw counter, bumped by
                                  ark_tok_gather / ark_tok_gather16 once per TRAINING forward; the GRU cells hash
                                  (seed, this counter, element index) -> forward and backward of one step agree,
                                  consecutive forwards differ (reference: nn.GRU(dropout=p), models.py:121-127)  */
#define ARK_HP_NOISE_STEP 13   /* uint32: latent-noise draw counter, bumped by every ark_normal_fill launch        */
#define ARK_HP_COUNT 16

#define ARK_TOK_PAD 0 /* special_tokens["PAD"], the ignore_index of the reference's cross-entropy */

/* ---- GRU recurrence (reference: nn.GRU inside AutoRegDecoderGRU / DecoderOnlyGRU,
 *      kgvae/model/models.py:121-127,141 and :329-335,344) ------------------------------------ */
/* one timestep of one layer: h_out = GRUCell(gi (= x W_ih^T + b_ih, precomputed), h_prev).
 * save_* ([B,D] each, nullable together) keep r, z, n and (W_hn h + b_hn) for the backward pass;
 * h_drop (nullable) additionally receives h_out * drop_mask (inter-layer dropout). */
int ark_gru_cell_fwd(int prec, const float* h_prev, const float* w_hh, const float* b_hh, const float* gi,
                     float* h_out, float* h_drop, const float* drop_mask, float* save_r, float* save_z,
                     float* save_n, float* save_hn, int B, int D, void* stream);
/* BPTT for one timestep: dh_t = dy_t + carry + dgh_next W_hh (first != 0: last timestep, no
 * successor), then gate derivatives -> dgi_t, dgh_t [B,3D]; carry <- dh_t * z_t (in place). */
int ark_gru_cell_bwd(int prec, const float* dgh_next, const float* w_hh, const float* dy, float* carry,
                     const float* save_r, const float* save_z, const float* save_n, const float* save_hn,
                     const float* h_prev, float* dgi, float* dgh, int B, int D, int first, void* stream);
/* gradient wrt the layer's initial state: dh0 (+)= dgh_0 W_hh + carry */
int ark_gru_h0_bwd(int prec, const float* dgh0, const float* w_hh, const float* carry, float* dh0, int accumulate,
                   int B, int D, void* stream);

/* ---- LDS-DMA / 16-bit-operand path (precision 16-bit, d_model % 64 == 0, batch % 16 == 0) --------
 * Layout contract: "16" buffers are row-major 16-bit copies in the named precision's type;
 * "_t" buffers are fp32 (saves: fp16) in the 16x16 MFMA-tile-native order
 *   off(row,col,ld) = ((row>>4)*(ld>>4) + (col>>4))*256 + (((row>>2)&3)*16 + (col&15))*4 + (row&3). */
/* Layer-diagonal forward step: up to ARK_DIAG_MAX_ROLES independent GRU cells -- cell (layer l, step
 * d-l) for every layer of one anti-diagonal d of the (layer, time) grid -- in ONE launch.  Each role
 * computes its input projection itself (x W_ih^T + b_ih; no gi buffer, no per-layer input GEMM), so
 * a stacked GRU of n layers over L steps is L+n-1 dependent launches instead of n*L + n.
 * Same math as ark_gru_cell_fwd on 16-bit operands (reference: nn.GRU, kgvae/model/models.py:121-127). */
#define ARK_DIAG_MAX_ROLES 4
typedef struct {
  const void* x16;       /* [B,D] row-major, forward type: this step's layer input                   */
  const void* h_prev16;  /* [B,D] row-major, forward type                                            */
  const void* w_ih16;    /* [3D,D] shadows, forward type                                             */
  const void* w_hh16;
  const float* b_ih;     /* [3D]                                                                     */
  const float* b_hh;
  const float* y_prev_t; /* tile-native fp32 [B,D]                                                   */
  float* y_out_t;
  void* y16a;            /* row-major 16-bit copies of h: forward type / backward type (nullable)    */
  void* y16b;
  void* yd16a;           /* h * dropout mask (drop_p > 0 only)                                       */
  void* yd16b;
  void* save_r;          /* tile-native fp16 saves, nullable together                                */
  void* save_z;
  void* save_n;
  void* save_hn;
  uint64_t drop_seed;
  int64_t drop_base;
  float drop_p;
  int pad_;
  /* Small vocabularies (layer 0 only): the input projection of a token is a ROW of the table x_tab[V, 3D] =
   * W_tok16 W_ih16^T (fp32 sums of the same 16-bit products, formed once per optimiser step), so the role adds
   * x_tab[x_tok[row]] to its gate pre-activations instead of streaming x16 W_ih^T through the ring: x16 / w_ih16 are then
   * unused (may be NULL) and half of the role's operand stream is gone.  x_tok: int32 token id of each of the B rows. */
  const float* x_tab;    /* nullable */
  const int* x_tok;      /* [B], required with x_tab */
} ArkGruDiagRole;
/* speed-only tile / ring choices of the two diagonal kernels; passed per call (NULL = the measured defaults of
 * ark_diag_tuning_default), so the library keeps no mutable state */
typedef struct {
  int fwd_rows;      /* 32 | 64 */
  int fwd_ki;        /* 64-wide k-images per ring stage: 1 | 2                                  */
  int fwd_nbuf;      /* ring slots: 2 | 4                                                       */
  int fwd_xcd;       /* XCD-aware tile order: 0 | 1                                             */
  int fwd_units;     /* hidden units per forward workgroup: 0 = by grid size | 16 (2 waves) | 32 (4) | 64 (8) */
  int bwd_rows;      /* 32 | 64                                                                 */
  int bwd_ki;        /* 1 | 2                                                                   */
  int bwd_nbuf;      /* 2 | 4                                                                   */
  int bwd_xcd_rows;  /* row-tile classes per XCD octet: 1 (plain order) | 2 | 4 | 8             */
  int bwd_cols;      /* output columns per backward workgroup: 0 = by grid size | 32 | 64        */
} ArkDiagTuning;
void ark_diag_tuning_default(ArkDiagTuning* t);
int ark_gru_diag_fwd(int prec, int prec_b, int n_roles, const ArkGruDiagRole* roles, const float* hyper, int B, int D,
                     const ArkDiagTuning* tuning, void* stream);
/* Layer-diagonal BPTT step: cells (layer l, step t) of one anti-diagonal of the backward order in ONE
 * launch.  A role forms dh_t = carry + dgh_{t+1} W_hh (its own layer) + dy_t, where dy_t is given
 * (top layer: gradient of the vocabulary projection) or computed in place as
 * dropout_mask(l,t) * (dgi_t of the layer above) W_ih(above) -- so the per-layer input-gradient GEMM
 * disappears.  Gate gradients go out as ONE row-major 16-bit panel per (layer, step):
 *   dg16[b, 0:4D] = [ dr | dz | dn | dn*r ]    (ld = 4D)
 * dgi = columns [0,3D) (weight gradient of W_ih, input gradient of the layer below), dgh = columns
 * [0,2D) + [3D,4D) (weight gradient of W_hh, recurrence): the two [B,3D] panels of round 1 shared two thirds.
 * `carry_t` is per layer.  Reference: autograd of nn.GRU, kgvae/model/models.py:121-127. */
typedef struct {
  const void* dgi_up16;    /* [B, ld 4D] backward type: gate-gradient panel of layer l+1 at step t (NULL: top layer) */
  const void* w_ihT_up16;  /* [D,3D] W_ih^T shadow of layer l+1 (NULL: top layer)                             */
  const void* dg_next16;   /* [B, ld 4D] this layer's panel at step t+1 (ignored when first != 0)             */
  const void* w_hhT16;     /* [D,3D] W_hh^T shadow                                                            */
  const float* dy_t;       /* tile-native fp32 [B,D], top layer only (NULL otherwise)                         */
  float* carry_t;          /* tile-native fp32 [B,D], this layer's dh_t * z_t carry (in/out)                  */
  const void* save_r;      /* tile-native fp16 saves of the forward pass                                      */
  const void* save_z;
  const void* save_n;
  const void* save_hn;
  const float* y_prev_t;   /* tile-native fp32 h_{t-1}                                                        */
  void* dg16;              /* row-major [B, ld 4D] output panel, backward type                                */
  float* db_ih;            /* [3D] += column sums of [dr|dz|dn] (nullable together)                           */
  float* db_hh;            /* [3D] += column sums of [dr|dz|dn*r]                                             */
  float* dh0;              /* non-NULL: "initial state" role -- only dh0[B,D] (row-major, pre-zeroed) +=      */
                           /* carry + dgh_next W_hh (dg_next16 = the layer's step-0 panel); saves etc. unused */
  uint64_t drop_seed;      /* dropout of THIS layer's output (applied to the gradient arriving from above)    */
  int64_t drop_base;       /* element index of this timestep in the layer's [B*L, D] space (% 4 == 0)         */
  float drop_p;
  int first;               /* last timestep: no successor                                                     */
} ArkGruDiagBwdRole;
int ark_gru_diag_bwd(int prec, int n_roles, const ArkGruDiagBwdRole* roles, const float* hyper, int B, int D,
                     const ArkDiagTuning* tuning, void* stream);
/* Persistent forward sweep for small batches of long sequences (csrc/gru_sweep.hip): the WHOLE stacked-GRU recurrence of
 * L steps as ONE launch of n_layers * (B/16) * (D/16) co-resident workgroups, each keeping its 16 units' weight rows in
 * registers and passing the hidden state on through `exch` with write-through stores, one counter per (layer, step,
 * row block) and bounded spins (two 16-row tiles per workgroup where that makes the grid fit the chip).  Inputs and outputs are those of L + n - 1 ark_gru_diag_fwd launches: time-major arrays
 * (row = t*B + b), slot 0 of y_t / y16a holding the initial state.  D in {128, 256, 512}, B % 16 == 0; returns
 * ARK_ERR_SHAPE when the grid cannot be co-resident (then use the diagonal launches).
 * `sync` protocol: zero the workspace ONCE (allocation, or ark_gru_sweep_sync_reset); the launches that share it keep
 * monotone counters and an epoch word, so no call clears anything.  After a launch sync[0] != 0 means a workgroup gave up
 * waiting (sync[1] = its id << 12 | step): the outputs of that launch AND of every later launch on the workspace are
 * invalid -- the word is sticky, later sweeps leave at once -- until the caller has read it and called
 * ark_gru_sweep_sync_reset.  The launch is a plain one: the caller must not run anything beside it that keeps a CU from
 * offering 96 KB of LDS to one of its workgroups for longer than the 2-s spin bound (the grid needs n_layers *
 * (B/16) * (D/16) / row_tiles CUs).
 * Reference: nn.GRU forward, kgvae/model/models.py:121-127. */
#define ARK_SWEEP_MAX_LAYERS 4
typedef struct {
  const void* w_ih16;    /* [3D,D] shadows, forward type                                              */
  const void* w_hh16;
  const float* b_ih;     /* [3D]                                                                      */
  const float* b_hh;
  float* y_t;            /* tile-native fp32 [(L+1)*B, D]; slot 0 = initial state (read), 1..L written */
  void* y16a;            /* row-major forward type [(L+1)*B, D]; slot 0 = initial state (read)        */
  void* y16b;            /* row-major backward type [(L+1)*B, D] (nullable)                           */
  void* yd16a;           /* [L*B, D] h * dropout mask (drop_p > 0 only)                               */
  void* yd16b;           /* nullable                                                                  */
  void* save_r;          /* tile-native fp16 [L*B, D] gate saves, nullable together                   */
  void* save_z;
  void* save_n;
  void* save_hn;
  uint64_t drop_seed;
  float drop_p;
  int pad_;
} ArkGruSweepLayer;
typedef struct {
  ArkGruSweepLayer layer[ARK_SWEEP_MAX_LAYERS];
  const void* x0_16;     /* [L*B, D] row-major forward type: layer 0's inputs                         */
  void* exch;            /* workspace, ark_gru_sweep_exch_bytes() bytes                               */
  unsigned* sync;        /* workspace, ark_gru_sweep_sync_words() words (zeroed ONCE by the caller)   */
  const float* hyper;
  int n_layers, B, D, L;
  int t0;                /* index of this launch's first step inside the whole sequence (a sequence swept in several launches
                          * passes arrays advanced by t0 steps): only the dropout hash offsets use it                    */
  int wg_slices;         /* unit slices per physical workgroup: 0 / 1 = one (the default), 2 = two (512 threads, half the CUs) */
} ArkGruSweep;
/* 16-row tiles per workgroup the sweeps would use on the current device: 1 or 2; 0 = the grid cannot be co-resident */
int ark_gru_sweep_row_tiles(int n_layers, int B, int D);
/* CUs (= physical workgroups) the forward (backward = 0) / backward (1) sweep of this shape holds; 0 = does not fit */
int ark_gru_sweep_cus(int n_layers, int B, int D, int backward, int wg_slices);
long ark_gru_sweep_exch_bytes(int n_layers, int B, int D, int L);
long ark_gru_sweep_sync_words(int n_layers, int B, int L);
int ark_gru_sweep_fwd(int prec, int prec_b, const ArkGruSweep* sweep, void* stream);
/* zero a sync workspace (error word, epoch, counters) on the stream: after allocation and after a reported failure */
int ark_gru_sweep_sync_reset(unsigned* sync, long words, void* stream);
/* The backward pass of the same recurrence (BPTT: autograd of nn.GRU) as ONE launch: what L + n - 1 ark_gru_diag_bwd
 * launches and the initial-state roles compute.  Writes the gate-gradient panels dg16[l] = [dr | dz | dn | dn*r]
 * (row-major [L*B, 4D], backward type), adds the bias gradients and (dh0 non-NULL) the initial-state gradient
 * sum over layers of (carry + dgh_0 W_hh) into the pre-zeroed row-major dh0[B, D]. */
typedef struct {
  const void* w_hhT16;     /* [D,3D] W_hh^T shadow, backward type                                       */
  const void* w_ihT_up16;  /* [D,3D] W_ih^T shadow of layer l+1 (NULL: top layer)                       */
  const void* save_r;      /* tile-native fp16 saves of the forward pass [L*B, D]                       */
  const void* save_z;
  const void* save_n;
  const void* save_hn;
  const float* y_t;        /* tile-native fp32 [(L+1)*B, D]: slot t = h_{t-1}                           */
  void* dg16;              /* row-major [L*B, 4D] output panels, backward type                          */
  float* db_ih;            /* [3D] += (nullable together)                                               */
  float* db_hh;
  uint64_t drop_seed;      /* dropout of THIS layer's output                                            */
  float drop_p;
  int pad_;
} ArkGruSweepBwdLayer;
typedef struct {
  ArkGruSweepBwdLayer layer[ARK_SWEEP_MAX_LAYERS];
  const float* dy_t;       /* tile-native fp32 [L*B, D]: gradient of the loss wrt the top layer's outputs */
  float* dh0;              /* row-major fp32 [B, D], += (nullable: no initial-state gradient wanted)    */
  void* exch;              /* workspace, ark_gru_sweep_bwd_exch_bytes() bytes                           */
  unsigned* sync;          /* workspace, ark_gru_sweep_sync_words() words (zeroed ONCE by the caller)   */
  const float* hyper;
  int n_layers, B, D, L;
  int wg_slices;           /* as in ArkGruSweep */
  int pad_;
} ArkGruSweepBwd;
long ark_gru_sweep_bwd_exch_bytes(int n_layers, int B, int D, int L);
int ark_gru_sweep_bwd(int prec, const ArkGruSweepBwd* sweep, void* stream);
/* up to 12 jobs in one launch: dst[i] = cast(src[i] [R,C]) in prec[i]; dstT[i] = cast(src[i]^T) in precT[i],
 * rows of dstT ldT[i] apart (ldT NULL or 0: dense, = R) */
int ark_weight_shadows(int n_jobs, const float* const* src, void* const* dst, void* const* dstT, const int* R,
                       const int* C, const int* prec, const int* precT, const int* ldT, void* stream);
/* C[M,N] = A16[M,K] B16[N,K]^T (+bias | *aux), C row-major or tile-native (c_tiled) */
int ark_gemm16(int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
               const float* bias, const float* aux, int M, int N, int K, int c_tiled, void* stream);
/* ark_gemm16 with row-major 16-bit copies of the result (c16a in `prec`, c16b nullable in `prec_b`);
 * BIAS_GELU: C = pre-activation, copies = gelu(C); MUL_DGELU: C = acc * gelu'(aux), copies = C;
 * colsum (nullable, not with BIAS_GELU): colsum[col] += sum over rows of C -- the bias gradient when C is a
 * pre-activation gradient (autograd of nn.Linear, models.py:36) */
int ark_gemm16_ex(int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                  const float* bias, const float* aux, void* c16a, void* c16b, int prec_b, float* colsum, int M, int N,
                  int K, void* stream);
/* ark_gemm16_ex on a chosen engine (parity tests, A/B timing): 0 = the library's choice (what ark_gemm16 / _ex do),
 * 1 = the shared two-barrier ring (csrc/dma_core.h), 2 = wave-private K-slices (csrc/wpk_core.h: one 64 x 96 or 32 x 96
 * tile per CU, every wave streams its own K-slices, no barrier in the main loop; ARK_ERR_SHAPE unless M % 32 == 0,
 * N % 96 == 0, K % 64 == 0, K >= 512 and the tiling gives 128 .. 512 workgroups -- or the product is at most 64 tiles of
 * 32 x 64 over K >= 1024: the latent heads), 3 = the library's choice INCLUDING that few-tiles / deep-K flavour (which
 * `ark_gemm16` / `_ex` never pick by themselves: another summation order).  Same products as nn.Linear forward / input
 * gradient of the encoder MLP and the heads (kgvae/model/models.py:32-44,60-61). */
int ark_gemm16_engine(int engine, int prec, int epi, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                      int64_t ldc, const float* bias, const float* aux, void* c16a, void* c16b, int prec_b, float* colsum,
                      int M, int N, int K, void* stream);
/* C[M,N] = sum_k A[k,M] B[k,N]: weight gradients with fp32 or 16-bit stored operands */
int ark_gemm_wgrad(int prec, const void* A, int a_is16, int64_t lda, const void* B, int b_is16, int64_t ldb, float* C,
                   int64_t ldc, int M, int N, int K, int accumulate, void* stream);
/* same product on the LDS-DMA ring with ds_read_b64_tr_b16 fragment reads; C += (accumulates);
 * needs M, N, K multiples of 64 and both operands 16-bit */
/* speed-only choices of the weight-gradient kernel, passed per call (NULL = ark_wgrad_tuning_default) */
typedef struct {
  int tile;        /* 64 | 128 (square output tile; 128 needs M, N multiples of 128)                         */
  int nbuf;        /* LDS-DMA ring slots: 2..4                                                               */
  int target_wgs;  /* split K over workgroups (fp32 atomics) until at least this many workgroups exist        */
  int balance;     /* 257..511 whole-K tiles: 256 whole tiles + the rest cut into k-slices, 1.x tiles per CU  */
  int waves;       /* 8 | 4 waves per workgroup (128 x 128 tiles: wave tile 64 x 32 | 64 x 64)                  */
} ArkWgradTuning;
void ark_wgrad_tuning_default(ArkWgradTuning* t);
int ark_wgrad16(int prec, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc, int M,
                int N, int K, const ArkWgradTuning* tuning, void* stream);
/* ark_wgrad16 for an A operand column-padded to a tile multiple (M) while C has only m_valid rows */
int ark_wgrad16_rows(int prec, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc, int M,
                     int m_valid, int N, int K, const ArkWgradTuning* tuning, void* stream);
#define ARK_WGRAD_MAX_GROUP 12
int ark_wgrad16_group(int prec, int n, const void* const* A16, const int64_t* lda, const void* const* B16,
                      const int64_t* ldb, float* const* C, const int64_t* ldc, const int* M, const int* N, const int* K,
                      const ArkWgradTuning* tuning, void* stream);
/* hyper_tick (nullable): training forward -> ++hyper[ARK_HP_DROP_STEP] (fresh dropout masks for this step) */
int ark_tok_gather16(int prec_a, int prec_b, const int64_t* seq, int64_t ld_seq, const float* w_tok,
                     const float* w_pos, void* x16a, void* x16b, int B, int L, int D, float* hyper_tick, void* stream);
/* h0 = tanh(z_proj(z)) for all layers in every layout of the LDS-DMA path, one launch */
int ark_zproj_fwd_v2(int prec_a, int prec_b, const float* z, const float* w_z, const float* b_z, float* h0, int n_layers,
                     float* const* y_t, void* const* y16a, void* const* y16b, int B, int Z, int D, void* stream);
/* reparameterisation + z-projection in one launch (fast path): mu, clamped logv, z = mu + eps * exp(logv / 2) for rows
 * < n_valid (rows beyond: z = 0, nothing written), kl_rows[b] = sum_j (1 + logv - mu^2 - exp(logv)), and
 * h0 = tanh(z_proj(z)) for all layers in every layout of the LDS-DMA path.  B % 16 == 0, Z <= 128.
 * Replaces ark_latent_fwd + ark_zproj_fwd_v2 (models.py:61-63,139,199-200). */
int ark_latent_zproj_fwd(int prec_a, int prec_b, const float* head, const float* eps, float* mu, float* logv, float* z,
                         float* kl_rows, const float* w_z, const float* b_z, float* h0, int n_layers, float* const* y_t,
                         void* const* y16a, void* const* y16b, int B, int n_valid, int Z, int D, void* stream);
/* per-token sums of a 16-bit panel: S[v, 0:n_cols] += sum over rows (t, b) with seq[b, t] == v of x16[(t, b), 0:n_cols]
 * (rows time-major, v < Vp, Vp * 256 B <= 64 KB of LDS, n_cols % 64 == 0; ids outside [0, Vp) are skipped).  `scratch`
 * (caller-owned, >= 4*B*L + 32 + 4*ceil(B*L/64) bytes) holds the per-chunk row lists sorted by token.  With x16 = layer 0's
 * gate-gradient panel this turns the embedding-gradient scatter (autograd embedding_backward of models.py:138) into
 * dW_tok += S W_ih_0, and layer 0's input weight gradient into dW_ih_0 += S^T W_tok. */
int ark_token_sums16(int prec, const int64_t* seq, int64_t ld_seq, const void* x16, int64_t ldx, float* S, int64_t ld_s,
                     void* scratch, int64_t scratch_bytes, int B, int L, int Vp, int n_cols, void* stream);
int ark_colsum16(int prec, const void* x16, int64_t ld, float* out, int M, int N, int accumulate, void* stream);
int ark_cast16(int prec, const float* x, void* out, int64_t n, void* stream);
/* one pass for what the backward of a Transformer sublayer made four passes of: v = x * dropout keep-scale (drop_p > 0: the
 * mask ark_dropout_apply(seed) draws on a [rows, cols] buffer), x_out (nullable, may alias x) = v, out16 = cast(v) in `prec`,
 * colsum[col] += sum over rows of v (nullable).  cols % 4 == 0. */
int ark_prep16(int prec, const float* x, float* x_out, void* out16, float* colsum, int rows, int cols, float drop_p, uint64_t seed,
               const float* hyper, void* stream);
/* out[i] = (float)x16[i]: data-parallel gradient buckets reduced in 16 bits go back into the fp32 gradient buffer */
int ark_uncast16(int prec, const void* x16, float* out, int64_t n, void* stream);
int ark_to_tiled(const float* x, float* out, int rows, int ld, void* stream);
/* the inverse: tile-native fp32 -> row-major [rows, ld] (rows % 16 == 0, ld % 16 == 0) */
int ark_from_tiled(const float* x, float* out, int rows, int ld, void* stream);

/* ---- embeddings (reference: models.py:47-58 encoder gather+concat+masked mean; :138,:343
 *      decoder token / position lookup; autograd embedding_backward) --------------------------- */
int ark_enc_pool_fwd(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt, int B, int T,
                     int D, int64_t pad_rid /* -1: none */, void* stream);
/* ark_enc_pool_fwd that also writes 16-bit copies of g for the MLP products */
int ark_enc_pool_fwd16(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt, void* g16a,
                       int prec_a, void* g16b, int prec_b, int B, int T, int D, int64_t pad_rid, void* stream);
int ark_enc_pool_bwd(const int64_t* triples, const float* dg, const float* inv_cnt, float* dE, float* dR, int B,
                     int T, int D, int n_ent, int n_rel, int64_t pad_eid, int64_t pad_rid, void* stream);
int ark_tok_gather(const int64_t* seq, int64_t ld_seq, const float* w_tok, const float* w_pos /* nullable */,
                   float* x, int B, int L, int D, float* hyper_tick /* nullable, as ark_tok_gather16 */, void* stream);
/* encoder pool (ark_enc_pool_fwd16) + decoder token gather (ark_tok_gather16, no position table) in ONE launch: the two
 * first kernels of a SAIL step only need the batch indices.  x16a/x16b[(t, b), :] = cast(W_tok[seq[b, t]]), t < L. */
/* tok_tm (nullable): int32 [L*B] time-major token ids, tok_tm[t*B + b] = seq[b, t] (the x_tok rows of ArkGruDiagRole);
 * with tok_tm given x16a / x16b may be NULL: the embedding rows are then not gathered at all. */
int ark_pool_gather_fwd16(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt, void* g16a,
                          int prec_a, void* g16b, int prec_b, int B, int T, int D, int64_t pad_rid, const int64_t* seq,
                          int64_t ld_seq, const float* w_tok, void* x16a, void* x16b, int L, int D_dec, int* tok_tm,
                          float* hyper_tick, void* stream);
/* the token half of that launch on its own: tok_tm[t*B + b] = seq[b, t] (+ the dropout tick) */
int ark_tok_time_major(const int64_t* seq, int64_t ld_seq, int* tok_tm, int B, int L, float* hyper_tick, void* stream);
int ark_tok_scatter(const int64_t* seq, int64_t ld_seq, const float* dx, float* d_w_tok, int B, int L, int D,
                    int vocab, void* stream);

/* ---- latent head / ELBO (reference: models.py:61-63,139,199-200; ablation_study.py:65-71) ------ */
int ark_latent_fwd(const float* head /* [B,2Z] = mu | raw logv */, const float* eps /* [B,Z], nullable = 0 */,
                   float* mu, float* logv, float* z, float* kl_out, int B, int Z, void* stream);
int ark_latent_bwd(const float* dz, const float* head, const float* eps, const float* hyper, float* dhead, int B,
                   int Z, void* stream);
int ark_zproj_fwd(const float* z, const float* w_z, const float* b_z, float* h0, int64_t copy_stride, int n_copies,
                  int B, int Z, int D, void* stream);
int ark_zproj_bwd(float* dh0, const float* h0, const float* z, const float* w_z, float* dz, float* d_w_z,
                  float* d_b_z, int B, int Z, int D, int accumulate, void* stream);
/* the batch reductions of ark_zproj_bwd alone (dzp = dh0 * (1 - h0^2) already formed) */
int ark_zproj_bwd_dw(const float* dzp, const float* z, float* d_w_z, float* d_b_z, int B, int Z, int D, int accumulate,
                     void* stream);
/* Per-row latent backward in one launch (Z <= 128): dh0 -> dzp (in place) -> dz -> dhead[B,2Z] (KL + reparameterisation
 * backward of models.py:61-63,199-200, plus optional external [dmu | dlogv]) -> dA[B,H] = (dhead W_head) * gelu'(pre)
 * and its 16-bit copy.  The reductions over the batch (ark_zproj_bwd_dw, db_head, dW_head) are separate. */
int ark_latent_chain_bwd(float* dh0, const float* h0, const float* w_z, const float* head, const float* eps,
                         const float* hyper, const float* ext_dhead /* nullable */, const float* w_head /* [2Z,H] */,
                         const float* pre /* [B,H] */, float* dhead, float* dA, void* dA16, int prec16,
                         float* dA_colsum /* nullable: [H] += column sums of dA = bias gradient of the last MLP layer */,
                         int B, int Z, int D, int H, void* stream);
/* the batch reductions behind ark_latent_chain_bwd in ONE launch, all outputs accumulate (+=):
 * dWz [D,Z] / dbz [D] from dzp [B,D] (what ark_latent_chain_bwd leaves in dh0) and z; dW_head [2Z,H] / db_head [2Z]
 * from dhead [B,2Z] and the 16-bit copy act16 [B,H] (type prec16) of the last encoder activation (models.py:43-44,139) */
int ark_latent_reduce_bwd(const float* dzp, const float* z, float* d_w_z, float* d_b_z, const float* dhead, const void* act16,
                          int prec16, float* d_w_head, float* d_b_head, int B, int Z, int D, int H, void* stream);
int ark_count_targets(const int64_t* seq, int64_t ld_seq, int B, int L, float* hyper, void* stream);
/* rows are time-major (t,b); target of row (t,b) is seq[b, t+1]; dlogits may alias logits or be NULL */
/* dlogits16 (nullable): [B*L, ld16] copy of dlogits in 16 bits (prec16), zero beyond V; may be the ONLY gradient
 * output (dlogits NULL): the fp32 [B*L,V] round trip is then skipped */
int ark_ce_fwd_bwd(float* logits, int64_t ld, const int64_t* seq, int64_t ld_seq, const float* hyper,
                   float* row_loss, float* dlogits, void* dlogits16, int prec16, int64_t ld16, int B, int L, int V,
                   void* stream);
/* Tied vocabulary projection FUSED with the cross-entropy, for large vocabularies: the [B*L, V] logits and their
 * gradient are never written (reference: logits = out(y), kgvae/model/models.py:128-134,142; F.cross_entropy with
 * ignore_index = PAD, kgvae/experiments/ablation_study.py:65-69; and their autograd).  Rows are time-major
 * (t, b); target of row (t, b) is seq[b, t + 1].  Y16 [B*L, D] / W16 [V, D] are 16-bit row-major in `prec`;
 * D in {64, 128, 256, 512}.
 *   ark_vocab_ce_fwd: row_loss[r] = lse_r - logit_r[target] (0 for PAD targets), lse[r], and (dY_t non-NULL,
 *     B*L % 16 == 0) dY = (softmax_r W_tok - W_tok[target]) * hyper[CE_INV_COUNT] in tile-native fp32.
 *   ark_vocab_ce_dw:  dW[V, D] += dlogits^T Y and db[V] += colsum(dlogits), dlogits recomputed from lse. */
int ark_vocab_ce_fwd(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq, int64_t ld_seq,
                     const float* hyper, float* row_loss, float* lse, float* dY_t /* nullable */, int B, int L, int V, int D,
                     void* stream);
/* ark_vocab_ce_fwd with dY for FEW rows: the vocabulary is swept in ark_vocab_ce_fwd_splits() parts by different
 * workgroups (the 64-row blocks alone would leave CUs idle: 160 at wd-articles B = 16, 20 on one of its 8 data-parallel
 * ranks) and a second small launch merges the partial softmax statistics and products.  `ws` holds splits * (B*L*D + 4*B*L) floats (unused and
 * may be NULL where ark_vocab_ce_fwd_splits() returns 1). */
/* cu_budget: the CUs this launch may fill (0 or 256 = the whole chip).  A 512-thread workgroup of this kernel owns a CU's
 * whole register file; the engine passes the CUs a persistent GRU sweep running BESIDE the launch leaves free, so the grid
 * can never keep one of the sweep's co-resident workgroups off the chip. */
int ark_vocab_ce_fwd_splits(int R, int V, int D, int cu_budget);
int ark_vocab_ce_fwd_ws(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq, int64_t ld_seq,
                        const float* hyper, float* row_loss, float* lse, float* dY_t, float* ws, int64_t ws_floats, int B,
                        int L, int V, int D, int cu_budget, void* stream);
int ark_vocab_ce_dw(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq, int64_t ld_seq,
                    const float* hyper, const float* lse, float* dW, float* db, int B, int L, int V, int D, void* stream);
/* out4 = {loss = ce + beta*kl, ce, kl, sum of token losses}; kl nullable (ARK) */
int ark_loss_finalize(const float* row_loss, int n_rows, const float* kl, const float* hyper, float* out4,
                      void* stream);
/* the same with kl = kl_scale * sum(kl_rows[0:n_kl]): per-row KL terms of ark_latent_zproj_fwd, kl_scale = -0.5 / (rows * Z);
 * summed in a fixed order (deterministic) */
int ark_loss_finalize_rows(const float* row_loss, int n_rows, const float* kl_rows, int n_kl, float kl_scale,
                           const float* hyper, float* out4, void* stream);
int ark_argmax_rows(const float* x, int64_t ld, int64_t* out, int rows, int V, void* stream);

/* ---- Transformer variant t-ARK (reference: DecoderOnlyTransformer, kgvae/model/models.py:349-366 = stock
 *      nn.TransformerEncoderLayer stack: post-norm, ReLU feed-forward, causal mask).  Rows are time-major (t, b);
 *      the dense products run on ark_gemm (exact fp32) or, in the 16-bit precisions, on ark_gemm16 / ark_wgrad16 from
 *      16-bit operand copies (ARK_EPI_BIAS / ARK_EPI_BIAS_RELU / ARK_EPI_MUL_RELU on both). ----------------------------- */
/* y = LayerNorm(x + res) * gamma + beta (res nullable); s_out (nullable) = x + res; stats[row] = (mean, rstd) */
int ark_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* s_out, float* y,
                      float* stats, int rows, int D, float eps, void* stream);
/* the same with the sublayer output's dropout applied on the way in: `res` is the output BEFORE dropout, the keep-scale of
 * element row * D + col of the current draw (the mask ark_dropout_apply(seed) would draw on that buffer) multiplies it */
int ark_layernorm_fwd_drop(const float* x, const float* res, const float* gamma, const float* beta, float* s_out, float* y,
                           float* stats, int rows, int D, float eps, float drop_p, uint64_t seed, const float* hyper, void* stream);
/* ds = dL/d(x + res) from dy, the saved sum s and stats; dgamma / dbeta ACCUMULATE (+=) */
int ark_layernorm_bwd(const float* dy, const float* s, const float* stats, const float* gamma, float* ds, float* dgamma,
                      float* dbeta, int rows, int D, void* stream);
/* x[i] *= keep-scale of element i for the current dropout draw (hyper[ARK_HP_DROP_STEP]) under `seed`: applied to an
 * activation in the forward pass and to its gradient in the backward pass (n % 4 == 0) */
int ark_dropout_apply(float* x, int64_t n, float p, uint64_t seed, const float* hyper, void* stream);
/* multi-head scaled-dot-product attention of nn.MultiheadAttention on packed qkv rows [L*B, 3D] = [q | k | v]:
 * out [L*B, D] = (softmax(q k^T / sqrt(dh) + causal mask) o dropout) v per (batch, head); probs [B, H, L, L] keeps the
 * probabilities BEFORE dropout for the backward pass (the mask is a counter hash, regenerated there).  L <= 640, dh <= 256 */
int ark_attn_fwd(const float* qkv, float* out, float* probs, const unsigned char* kmask /* [B, L] 1 = key visible, nullable */,
                 int B, int L, int D, int n_heads, int causal, float drop_p, uint64_t seed, const float* hyper, void* stream);
/* dqkv [L*B, 3D] from dout; dscore [B, H, L, L] is scratch */
int ark_attn_bwd(const float* qkv, const float* out, const float* probs, const float* dout, float* dscore, float* dqkv,
                 const unsigned char* kmask, int B, int L, int D, int n_heads, int causal, float drop_p, uint64_t seed,
                 const float* hyper, void* stream);
/* The same attention on the matrix cores, flash style (csrc/attn_mfma.hip): scores and probabilities never exist in memory.
 * Forward: out and lse [B, H, Lp] (log2-domain log-sum-exp per query; Lp = L rounded up to 64: ark_attn_flash_stat_floats()
 * floats); backward: dqkv from dout, recomputing the probabilities twice (query side, key side); `delta` is scratch of the
 * same size as lse.  Operands are converted to `prec` (ARK_PREC_F16 / ARK_PREC_BF16) while staged; statistics and
 * accumulators fp32; the dropout masks are those of ark_attn_fwd / ark_attn_bwd under the same seed.  Any L; dh % 32 == 0,
 * dh <= 384.  Reference: F.scaled_dot_product_attention inside the stock Transformer layers, models.py:73-74, 104-105, 355-356 */
long ark_attn_flash_stat_floats(int B, int L, int n_heads);
int ark_attn_flash_fwd(int prec, const float* qkv, float* out, float* lse, const unsigned char* kmask, int B, int L, int D,
                       int n_heads, int causal, float drop_p, uint64_t seed, const float* hyper, void* stream);
int ark_attn_flash_bwd(int prec, const float* qkv, const float* out, const float* lse, const float* dout, float* delta,
                       float* dqkv, const unsigned char* kmask, int B, int L, int D, int n_heads, int causal, float drop_p,
                       uint64_t seed, const float* hyper, void* stream);
/* ---- t-SAIL (reference: AutoRegEncoder / AutoRegDecoder, kgvae/model/models.py:66-114) ----------------------------------
 * encoder input rows (t, b) over the TRIPLE index: x = [E[h] | R[r] | E[t]], kmask[b, t] = (r != pad_rid) (nullable) */
int ark_triple_gather(const int64_t* triples, const float* E, const float* R, float* x, unsigned char* kmask, int B, int T, int D,
                      int64_t pad_rid /* < 0: none */, void* stream);
int ark_triple_scatter(const int64_t* triples, const float* dx, float* dE, float* dR, int B, int T, int D, int64_t pad_eid,
                       int64_t pad_rid, void* stream);
/* masked mean over the sequence axis of time-major rows [T*B, W] -> g [B, W]; inv_cnt[b] = 1 / max(1, #visible) */
int ark_seq_pool_fwd(const float* x, const unsigned char* kmask, float* g, float* inv_cnt, int B, int T, int W, void* stream);
int ark_seq_pool_bwd(const float* dg, const unsigned char* kmask, const float* inv_cnt, float* dx, int B, int T, int W, void* stream);
/* cross-attention over a memory of L IDENTICAL rows (z_proj(z) repeated, models.py:112): uniform softmax, so
 * ctx[(t, b), h] = c[t, b, h] * v[b, h] with c = 1 (no dropout) or (#kept keys) / (L (1 - p)); cscale [L*B, H] keeps c */
int ark_xattn_bcast_fwd(const float* v, float* ctx, float* cscale, int B, int L, int D, int n_heads, float drop_p, uint64_t seed,
                        const float* hyper, void* stream);
int ark_xattn_bcast_bwd(const float* dctx, const float* cscale, float* dv, int B, int L, int D, int n_heads, void* stream);
/* ark_latent_fwd / ark_latent_bwd with the [-10, 10] clamp of logv switchable (clamp = 0: t-SAIL, models.py:93) */
int ark_latent_fwd_ex(const float* head, const float* eps, float* mu, float* logv, float* z, float* kl_out, int B, int Z,
                      int clamp, void* stream);
int ark_latent_bwd_ex(const float* dz, const float* head, const float* eps, const float* hyper, float* dhead, int B, int Z,
                      int clamp, void* stream);

/* out[0:n] = N(0,1) draws for the reparameterisation noise (reference: torch.randn_like(mu), kgvae/model/models.py:63):
 * counter-based -- element i of draw number hyper[ARK_HP_NOISE_STEP] under `seed` is a pure function of (seed, draw, i)
 * (two 32-bit hashes -> Box-Muller) -- so a captured graph replays fresh noise every step and ranks with different seeds
 * draw independent streams.  The launch bumps the draw counter.  One workgroup (n is B * d_latent). */
int ark_normal_fill(float* out, int64_t n, uint64_t seed, float* hyper, void* stream);
/* zero `nbytes` bytes (multiple of 4, 16-byte aligned start): a plain kernel, ordered like every other launch of the stream */
int ark_zero(void* ptr, int64_t nbytes, void* stream);
/* diagnostics: buf[slot] = the device's 100-MHz real-time counter once everything queued before on `stream` has run
 * (a one-thread launch; tools/step_stamps.py captures one behind every launch of a train step) */
int ark_stamp(unsigned long long* buf, int slot, void* stream);
/* y[0:n] += a * x[0:n] */
int ark_axpy(float* y, const float* x, int64_t n, float a, void* stream);
/* device-to-device copy of `nbytes` bytes (multiple of 4, 16-byte aligned ends): a plain kernel */
int ark_copy(void* dst, const void* src, int64_t nbytes, void* stream);
/* ---- optimiser and reductions (reference: optim.Adam, ablation_study.py:571,76) ---------------- */
int ark_adam_tick(float* hyper, void* stream);
int ark_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream);
/* ark_adam_step over a job list, with the 16-bit weight shadows (ark_weight_shadows) of every matrix job written
 * from the updated values in the same pass: job i is a linear range [off, off + C) of the flat buffers (R == 0)
 * or a matrix [R, C] at `off` with optional plain (dst, prec) / transposed (dstT [C, ldT], precT) shadows. */
#define ARK_ADAM_MAX_JOBS 48
int ark_adam_step_shadows(float* p, const float* g, float* m, float* v, int n_jobs, const int64_t* off, const int* R,
                          const int* C, void* const* dst, void* const* dstT, const int* prec, const int* precT,
                          const int* ldT, const float* hyper, void* stream);
/* the same with the gradient read from a bf16 buffer (element i of g16 = the gradient of flat element i): the all-reduced
 * 16-bit transport copy of a data-parallel bucket feeds the update directly */
int ark_adam_step_shadows_g16(float* p, const void* g16, float* m, float* v, int n_jobs, const int64_t* off, const int* R,
                              const int* C, void* const* dst, void* const* dstT, const int* prec, const int* precT,
                              const int* ldT, const float* hyper, void* stream);
/* accumulate != 0: add into `out` (caller zeroed it, e.g. the whole gradient buffer at once) */
int ark_colsum(const float* x, int64_t ld, int64_t batch_stride_in, float* out, int64_t batch_stride_out, int M,
               int N, int n_batch, int accumulate, void* stream);
/* mask[i] = keep-scale of element i for the current dropout draw (hyper[ARK_HP_DROP_STEP]); n % 4 == 0 */
int ark_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const float* hyper, void* stream);
int ark_mul(const float* a, const float* b, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
