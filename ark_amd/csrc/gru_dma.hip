// GRU recurrence, LDS-DMA edition (engine: dma_core.h).  Same math as gru.hip; what changes is
// where every byte lives so that a timestep is one short, fully coalesced launch:
//   * recurrent operands are 16-bit row-major copies (h_{t-1} written by the previous cell,
//     W_hh / W_hh^T shadows refreshed after each optimiser step) streamed by LDS-DMA;
//   * everything only epilogues touch (fp32 state h, gi, gate saves, dy, carry) is stored in the
//     16x16 MFMA-tile-native order (tile_native_off): one lane's 4 rows = one 16-byte access,
//     one wave's tile = one contiguous 1-KB line;
//   * row-major 16-bit outputs that later products read (h, dgi, dgh) are assembled in LDS and
//     written as whole 64/128-byte row segments.
// Used when the precision is 16-bit, d_model % 64 == 0 and batch % 16 == 0; gru.hip remains the
// exact-fp32 / odd-shape path.  Reference op replaced: torch.nn.GRU (kgvae/model/models.py:121-127).
#include "dma_core.h"
#include "../../include/ark_amd.h"

namespace ark {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld_half4(const _Float16* p) {
  const half4_t h = *reinterpret_cast<const half4_t*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ f32x4 cv_half4(half4_t h) { return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
__device__ __forceinline__ void st_half4(_Float16* p, f32x4 v) {
  *reinterpret_cast<half4_t*>(p) = half4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}

struct GruFwdDmaArgs {
  const void* h_prev16; const void* w_hh16;      // DMA operands (prec type), row-major [B,D] / [3D,D]
  const float* y_prev; const float* b_hh; const float* gi;   // tile-native fp32 [B,D], [3D], tile-native [B,3D]
  float* y_out;                                  // tile-native fp32 [B,D]
  void* y16a; void* y16b;                        // row-major 16-bit copies of h: forward type / backward type (nullable)
  void* yd16a; void* yd16b;                      // dropped copies h*mask of the layer output, nullable
  float drop_p; uint64_t drop_seed; long drop_base; const float* hyper;   // in-kernel dropout (drop_p > 0)
  _Float16* sr; _Float16* sz; _Float16* sn; _Float16* shn;   // tile-native fp16 saves, nullable
  int B, D;
  int dbg;   // timing ablations (results invalid when != 0): 1 no main loop, 2 no epilogue stores, 4 no prefetch loads
};

template <int PREC, int PRECB, int NBUF, int KI, int BM, int NW>
__global__ __launch_bounds__(64 * NW) void gru_cell_fwd_dma_kernel(GruFwdDmaArgs p) {
  constexpr int BU = 32, BN = 3 * BU;
  using G = DmaTile<PREC, BM, BN, NBUF, NW / 2, 2, KI>;  // wave tile (2*BM/NW) x 48 (16 units x 3 gates)
  constexpr int TM = G::TM;
  using h_t = typename G::h_t;
  using hb_t = typename PrecTraits<PRECB>::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int units_tiles = p.D / BU;
  const int m0 = (blockIdx.x / units_tiles) * BM, u0 = (blockIdx.x % units_tiles) * BU;
  const int B = p.B, D = p.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ul = wn * 16 + (lane & 15);           // unit within the workgroup tile
  const int u = u0 + ul;
  // Epilogue operands are requested BEFORE the recurrent product: they are older than every LDS-DMA
  // op in the vmcnt queue, so they land underneath the main loop instead of after it.
  int rl[TM];
  long o[TM];
  f32x4 gr[TM], gz[TM], gn[TM], hp[TM], mk[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    rl[tm] = wm * G::WTM + tm * 16 + 4 * (lane >> 4);   // first of this lane's 4 rows within the tile
    const int rowc = min(m0 + rl[tm], B - 4);            // B % 16 == 0: clamped quads stay in bounds
    const long og = tile_native_off(rowc, u, 3 * D);
    o[tm] = tile_native_off(rowc, u, D);
    gr[tm] = *reinterpret_cast<const f32x4*>(p.gi + og);
    gz[tm] = *reinterpret_cast<const f32x4*>(p.gi + og + (long)(D >> 4) * 256);
    gn[tm] = *reinterpret_cast<const f32x4*>(p.gi + og + (long)(D >> 4) * 512);
    hp[tm] = *reinterpret_cast<const f32x4*>(p.y_prev + o[tm]);
    mk[tm] = f32x4{1.f, 1.f, 1.f, 1.f};
  }
  const bool drop = p.drop_p > 0.f;
  if (drop) {   // same hash stream as ark_dropout_mask over the layer's tile-native [B*L, D] index space
    const uint64_t step = (uint64_t)p.hyper[ARK_HP_ADAM_STEP];
    const float ks = 1.0f / (1.0f - p.drop_p);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        mk[tm][i] = dropout_keep_scale(p.drop_seed, step, (uint64_t)(p.drop_base + o[tm] + i), p.drop_p, ks);
  }
  const float bhr = p.b_hh[u], bhz = p.b_hh[D + u], bhn = p.b_hh[2 * D + u];

  f32x4 acc[G::TM][G::TN];
  G::run(acc, reinterpret_cast<const h_t*>(p.h_prev16), D, [=](int r) -> long { return (long)min(m0 + r, B - 1); },
         reinterpret_cast<const h_t*>(p.w_hh16), D,
         [=](int j) -> long { return (long)((j >> 4) % 3) * D + u0 + (j / 48) * 16 + (j & 15); }, (p.dbg & 1) ? 0 : D, smem);

  // the LDS ring is free now; reuse it to assemble row-major 16-bit rows [BM][32+pad]
  __syncthreads();
  constexpr int TS = 40;  // row stride in elements (80 B: 16-B aligned rows, spreads banks)
  constexpr int ARR = BM * TS * 2;
  h_t* ta = reinterpret_cast<h_t*>(smem);                 // h        (forward type)
  hb_t* tb = reinterpret_cast<hb_t*>(smem + ARR);         // h        (backward type)
  h_t* tda = reinterpret_cast<h_t*>(smem + 2 * ARR);      // h*mask
  hb_t* tdb = reinterpret_cast<hb_t*>(smem + 3 * ARR);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (m0 + rl[tm] >= B) continue;
    f32x4 r, z, n, hn, h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = sigmoidf_(gr[tm][i] + acc[tm][0][i] + bhr);
      z[i] = sigmoidf_(gz[tm][i] + acc[tm][1][i] + bhz);
      hn[i] = acc[tm][2][i] + bhn;
      n[i] = tanhf(gn[tm][i] + r[i] * hn[i]);
      h[i] = (1.0f - z[i]) * n[i] + z[i] * hp[tm][i];
    }
    if (!(p.dbg & 2)) *reinterpret_cast<f32x4*>(p.y_out + o[tm]) = h;
    if (p.sr && !(p.dbg & 2)) {
      st_half4(p.sr + o[tm], r); st_half4(p.sz + o[tm], z); st_half4(p.sn + o[tm], n); st_half4(p.shn + o[tm], hn);
    }
    const f32x4 hd = h * mk[tm];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ta[(rl[tm] + i) * TS + ul] = G::PT::cvt(h[i]);
      if (p.y16b) tb[(rl[tm] + i) * TS + ul] = PrecTraits<PRECB>::cvt(h[i]);
      if (drop) {
        tda[(rl[tm] + i) * TS + ul] = G::PT::cvt(hd[i]);
        tdb[(rl[tm] + i) * TS + ul] = PrecTraits<PRECB>::cvt(hd[i]);
      }
    }
  }
  __syncthreads();
  // BM rows x 64 B per array: thread t -> row t/4, 16-byte chunk t%4
  const int t = threadIdx.x;
  if (t < BM * 4 && !(p.dbg & 2)) {
    const int rr = t >> 2, ch = t & 3;
    const int row = m0 + rr;
    if (row < B) {
      const long go = (long)row * D + u0 + ch * 8;
      *reinterpret_cast<uint4*>(reinterpret_cast<h_t*>(p.y16a) + go) = *reinterpret_cast<const uint4*>(ta + rr * TS + ch * 8);
      if (p.y16b) *reinterpret_cast<uint4*>(reinterpret_cast<hb_t*>(p.y16b) + go) = *reinterpret_cast<const uint4*>(tb + rr * TS + ch * 8);
      if (drop) {
        *reinterpret_cast<uint4*>(reinterpret_cast<h_t*>(p.yd16a) + go) = *reinterpret_cast<const uint4*>(tda + rr * TS + ch * 8);
        if (p.yd16b) *reinterpret_cast<uint4*>(reinterpret_cast<hb_t*>(p.yd16b) + go) = *reinterpret_cast<const uint4*>(tdb + rr * TS + ch * 8);
      }
    }
  }
}

struct GruBwdDmaArgs {
  const void* dgh_next16; const void* w_hhT16;   // DMA operands: row-major [B,3D] / [D,3D] (prec type)
  const float* dy; float* carry;                 // tile-native fp32 [B,D]
  const _Float16* sr; const _Float16* sz; const _Float16* sn; const _Float16* shn;   // tile-native fp16
  const float* y_prev;                           // tile-native fp32 [B,D]
  void* dgi16; void* dgh16;                      // row-major [B,3D] (prec type)
  float* db_ih; float* db_hh;                    // [3D] bias gradients, accumulated with atomics (nullable)
  float* dh0; int dh0_accumulate;                // row-major fp32 [B,D] (final_ only)
  int B, D, first, final_;
};

template <int PREC, int NBUF, int KI, int BN>
__global__ __launch_bounds__(256) void gru_cell_bwd_dma_kernel(GruBwdDmaArgs p) {
  constexpr int BM = 32;
  using G = DmaTile<PREC, BM, BN, NBUF, 2, 2, KI>;   // wave tile 16 x (BN/2)
  constexpr int WN = BN / 2;
  using h_t = typename G::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = p.D / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int B = p.B, D = p.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int rl = wm * 16 + 4 * (lane >> 4);
  const int row0 = m0 + rl;
  // epilogue operands first (older than the LDS-DMA ops -> they land underneath the main loop)
  const int rowc = min(row0, B - 4);
  f32x4 pc[G::TN], pdy[G::TN], phy[G::TN];
  half4_t psr[G::TN], psz[G::TN], psn[G::TN], phn[G::TN];
#pragma unroll
  for (int tn = 0; tn < G::TN; ++tn) {
    const long o = tile_native_off(rowc, n0 + wn * WN + tn * 16 + (lane & 15), D);
    pc[tn] = p.first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.carry + o);
    pdy[tn] = (p.dy && !p.final_) ? *reinterpret_cast<const f32x4*>(p.dy + o) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (!p.final_) {
      phy[tn] = *reinterpret_cast<const f32x4*>(p.y_prev + o);
      psr[tn] = *reinterpret_cast<const half4_t*>(p.sr + o);
      psz[tn] = *reinterpret_cast<const half4_t*>(p.sz + o);
      psn[tn] = *reinterpret_cast<const half4_t*>(p.sn + o);
      phn[tn] = *reinterpret_cast<const half4_t*>(p.shn + o);
    }
  }

  f32x4 acc[G::TM][G::TN];
  G::run(acc, reinterpret_cast<const h_t*>(p.dgh_next16), 3L * D, [=](int r) -> long { return (long)min(m0 + r, B - 1); },
         reinterpret_cast<const h_t*>(p.w_hhT16), 3L * D, [=](int r) -> long { return (long)(n0 + r); },
         p.first ? 0 : 3 * D, smem);

  __syncthreads();
  // LDS assembly of the six row-major 16-bit output panels: [gate 0..2][32 rows][BN+8] for dgi and dgh
  constexpr int TS = BN + 8;
  h_t* tgi = reinterpret_cast<h_t*>(smem);
  h_t* tgh = tgi + 3 * 32 * TS;
#pragma unroll
  for (int tn = 0; tn < G::TN; ++tn) {
    const int ul = wn * WN + tn * 16 + (lane & 15);
    const int u = n0 + ul;
    if (row0 >= B) continue;
    const long o = tile_native_off(row0, u, D);
    f32x4 dh = acc[0][tn] + pc[tn];
    if (p.final_) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long ro = (long)(row0 + i) * D + u;
        p.dh0[ro] = p.dh0_accumulate ? p.dh0[ro] + dh[i] : dh[i];
      }
      continue;
    }
    dh += pdy[tn];
    const f32x4 r = cv_half4(psr[tn]), z = cv_half4(psz[tn]), n = cv_half4(psn[tn]), hn = cv_half4(phn[tn]);
    const f32x4 hp = phy[tn];
    f32x4 cz;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dn_pre = dh[i] * (1.0f - z[i]) * (1.0f - n[i] * n[i]);
      const float dz_pre = dh[i] * (hp[i] - n[i]) * z[i] * (1.0f - z[i]);
      const float dr_pre = dn_pre * hn[i] * r[i] * (1.0f - r[i]);
      cz[i] = dh[i] * z[i];
      const int ro = (rl + i) * TS + ul;
      tgi[ro] = G::PT::cvt(dr_pre); tgi[32 * TS + ro] = G::PT::cvt(dz_pre); tgi[64 * TS + ro] = G::PT::cvt(dn_pre);
      tgh[ro] = G::PT::cvt(dr_pre); tgh[32 * TS + ro] = G::PT::cvt(dz_pre); tgh[64 * TS + ro] = G::PT::cvt(dn_pre * r[i]);
    }
    *reinterpret_cast<f32x4*>(p.carry + o) = cz;
  }
  if (p.final_) return;
  __syncthreads();
  // 3 gates x 32 rows x (2*BN) B per array: thread t -> (row, 16-byte chunk) for each gate
  const int t = threadIdx.x;
  constexpr int CPR = BN / 8;
  const int rr = t / CPR, ch = t % CPR;
  const int row = m0 + rr;
  if (rr < 32 && row < B) {
    h_t* gi16 = reinterpret_cast<h_t*>(p.dgi16);
    h_t* gh16 = reinterpret_cast<h_t*>(p.dgh16);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const long go = (long)row * 3 * D + (long)g * D + n0 + ch * 8;
      *reinterpret_cast<uint4*>(gi16 + go) = *reinterpret_cast<const uint4*>(tgi + g * 32 * TS + rr * TS + ch * 8);
      *reinterpret_cast<uint4*>(gh16 + go) = *reinterpret_cast<const uint4*>(tgh + g * 32 * TS + rr * TS + ch * 8);
    }
  }
  // bias gradients: column sums of this tile's dgi / dgh panels straight from LDS, one atomic per
  // (gate, unit) per workgroup -- replaces a separate pass over the [B*L, 3D] panels
  if (p.db_ih && t < 3 * BN) {
    const int g = t / BN, ul = t % BN;
    const int nrows = min(32, B - m0);
    float si = 0.f, sh = 0.f;
    for (int r2 = 0; r2 < nrows; ++r2) {
      si += (float)tgi[g * 32 * TS + r2 * TS + ul];
      sh += (float)tgh[g * 32 * TS + r2 * TS + ul];
    }
    atomicAdd(&p.db_ih[(long)g * D + n0 + ul], si);
    atomicAdd(&p.db_hh[(long)g * D + n0 + ul], sh);
  }
}

// ---- 16-bit shadows (plain and transposed) of a list of fp32 matrices, one launch ------------
struct ShadowJob { const float* src; void* dst; void* dstT; int R, C, prec, precT, ldT; };   // dstT[c*ldT + r]
struct ShadowJobs { ShadowJob j[12]; int n; };

__device__ __forceinline__ void shadow_put(void* base, long idx, float v, int prec) {
  if (prec == PREC_F16) reinterpret_cast<_Float16*>(base)[idx] = PrecTraits<PREC_F16>::cvt(v);
  else reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
}

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
      if (jb.dst) shadow_put(jb.dst, (long)r * jb.C + c, v, jb.prec);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (jb.dstT)
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (r < jb.R && c < jb.C) shadow_put(jb.dstT, (long)c * jb.ldT + r, tile[tx][i], jb.precT);
    }
}

// dynamic LDS above 64 KB must be opted into once per kernel
template <class K>
static void allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int g_dma_dbg = 0;
int g_fwd_nbuf = 2, g_bwd_nbuf = 2;   // ring slots
int g_fwd_ki = 2, g_bwd_ki = 2;       // 64-wide k-images per stage (1 | 2)

int g_fwd_bm = 32;   // rows per forward-cell workgroup: 32 | 64 (4 waves) | 128 = 64 rows with 8 waves

template <int PREC, int PRECB, int NBUF, int KI, int BM, int NW>
static void launch_fwd_nb(const GruFwdDmaArgs& p, hipStream_t st) {
  using G = DmaTile<PREC, BM, 96, NBUF, NW / 2, 2, KI>;
  constexpr int MINL = 4 * BM * 40 * 2;
  constexpr int LDS = G::LDS_BYTES > MINL ? G::LDS_BYTES : MINL;
  static bool once = (allow_lds(gru_cell_fwd_dma_kernel<PREC, PRECB, NBUF, KI, BM, NW>, LDS), true); (void)once;
  const unsigned grid = (unsigned)(((p.B + BM - 1) / BM) * (p.D / 32));
  hipLaunchKernelGGL((gru_cell_fwd_dma_kernel<PREC, PRECB, NBUF, KI, BM, NW>), dim3(grid), dim3(64 * NW), LDS, st, p);
}
template <int PREC, int PRECB, int BM>
static void launch_fwd_bm(const GruFwdDmaArgs& p, hipStream_t st) {
  const bool ki2 = g_fwd_ki == 2 && p.D % 128 == 0;
  if (ki2) {
    if (g_fwd_nbuf >= 4 && BM == 32) launch_fwd_nb<PREC, PRECB, (BM == 32 ? 4 : 2), 2, BM, 4>(p, st);
    else launch_fwd_nb<PREC, PRECB, 2, 2, BM, 4>(p, st);
  } else {
    if (g_fwd_nbuf == 8 && BM == 32) launch_fwd_nb<PREC, PRECB, (BM == 32 ? 8 : 4), 1, BM, 4>(p, st);
    else if (g_fwd_nbuf >= 4) launch_fwd_nb<PREC, PRECB, 4, 1, BM, 4>(p, st);
    else launch_fwd_nb<PREC, PRECB, 2, 1, BM, 4>(p, st);
  }
}
template <int PREC, int PRECB>
static int launch_fwd_dma(const GruFwdDmaArgs& p, hipStream_t st) {
  if (g_fwd_bm == 128 && p.D % 128 == 0) {   // 64 rows, 8 waves: W_hh panel loaded once per 64 rows, same waves per CU
    if (g_fwd_nbuf >= 4) launch_fwd_nb<PREC, PRECB, 3, 2, 64, 8>(p, st);   // 3 x 40 KB slots
    else launch_fwd_nb<PREC, PRECB, 2, 2, 64, 8>(p, st);                  // 2 x 40 KB slots
  } else if (g_fwd_bm == 64) launch_fwd_bm<PREC, PRECB, 64>(p, st);
  else launch_fwd_bm<PREC, PRECB, 32>(p, st);
  ARK_LAUNCH_CHECK();
  return 0;
}
int g_bwd_bn = 64;   // hidden units per backward-cell workgroup (64 | 32)

template <int PREC, int NBUF, int KI, int BN>
static void launch_bwd_nb(const GruBwdDmaArgs& p, hipStream_t st) {
  using G = DmaTile<PREC, 32, BN, NBUF, 2, 2, KI>;
  constexpr int MINL = 2 * 3 * 32 * (BN + 8) * 2;
  constexpr int LDS = G::LDS_BYTES > MINL ? G::LDS_BYTES : MINL;
  static bool once = (allow_lds(gru_cell_bwd_dma_kernel<PREC, NBUF, KI, BN>, LDS), true); (void)once;
  const unsigned grid = (unsigned)(((p.B + 31) / 32) * (p.D / BN));
  hipLaunchKernelGGL((gru_cell_bwd_dma_kernel<PREC, NBUF, KI, BN>), dim3(grid), dim3(256), LDS, st, p);
}
template <int PREC, int BN>
static void launch_bwd_bn(const GruBwdDmaArgs& p, hipStream_t st) {
  const bool ki2 = g_bwd_ki == 2 && (3 * p.D) % 128 == 0;
  if (ki2) {
    if (g_bwd_nbuf >= 4) launch_bwd_nb<PREC, 4, 2, BN>(p, st);
    else launch_bwd_nb<PREC, 2, 2, BN>(p, st);
  } else {
    if (g_bwd_nbuf >= 4) launch_bwd_nb<PREC, 4, 1, BN>(p, st);
    else launch_bwd_nb<PREC, 2, 1, BN>(p, st);
  }
}
template <int PREC>
static int launch_bwd_dma(const GruBwdDmaArgs& p, hipStream_t st) {
  if (g_bwd_bn == 32) launch_bwd_bn<PREC, 32>(p, st);
  else launch_bwd_bn<PREC, 64>(p, st);
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

// speed-only knobs: ring depth (2|4|8) of the forward / backward LDS-DMA cell kernels
extern "C" int ark_set_dma_bwd_units(int bn) {
  if (bn != 32 && bn != 64) return ARK_ERR_ARG;
  ark::g_bwd_bn = bn;
  return 0;
}

extern "C" int ark_set_dma_fwd_rows(int bm) {
  if (bm != 32 && bm != 64 && bm != 128) return ARK_ERR_ARG;
  ark::g_fwd_bm = bm;
  return 0;
}

extern "C" int ark_set_dma_stage(int fwd_ki, int bwd_ki) {
  if ((fwd_ki != 1 && fwd_ki != 2) || (bwd_ki != 1 && bwd_ki != 2)) return ARK_ERR_ARG;
  ark::g_fwd_ki = fwd_ki;
  ark::g_bwd_ki = bwd_ki;
  return 0;
}

extern "C" int ark_set_dma_ring(int fwd_nbuf, int bwd_nbuf) {
  auto ok = [](int v) { return v == 2 || v == 4 || v == 8; };   // 8 applies to the forward cell only
  if (!ok(fwd_nbuf) || !ok(bwd_nbuf)) return ARK_ERR_ARG;
  ark::g_fwd_nbuf = fwd_nbuf;
  ark::g_bwd_nbuf = bwd_nbuf;
  return 0;
}

extern "C" int ark_gru_cell_fwd_dma(int prec, int prec_b, const void* h_prev16, const void* w_hh16, const float* y_prev_t,
                                    const float* b_hh, const float* gi_t, float* y_out_t, void* y16a, void* y16b,
                                    void* yd16a, void* yd16b, float drop_p, uint64_t drop_seed, int64_t drop_base,
                                    const float* hyper, void* save_r, void* save_z, void* save_n, void* save_hn, int B,
                                    int D, void* stream) {
  using namespace ark;
  if (!h_prev16 || !w_hh16 || !y_prev_t || !b_hh || !gi_t || !y_out_t || !y16a || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0 || B % 16 != 0) return ARK_ERR_SHAPE;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!yd16a || !hyper))) return ARK_ERR_ARG;
  if (save_r && (!save_z || !save_n || !save_hn)) return ARK_ERR_ARG;
  GruFwdDmaArgs p{h_prev16, w_hh16, y_prev_t, b_hh, gi_t, y_out_t, y16a, y16b, yd16a, yd16b, drop_p, drop_seed, (long)drop_base, hyper,
                  (_Float16*)save_r, (_Float16*)save_z, (_Float16*)save_n, (_Float16*)save_hn, B, D, g_dma_dbg};
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F16 && prec_b == PREC_BF16) return launch_fwd_dma<PREC_F16, PREC_BF16>(p, st);
  if (prec == PREC_F16 && prec_b == PREC_F16) return launch_fwd_dma<PREC_F16, PREC_F16>(p, st);
  if (prec == PREC_BF16 && prec_b == PREC_BF16) return launch_fwd_dma<PREC_BF16, PREC_BF16>(p, st);
  return ARK_ERR_ARG;
}

extern "C" int ark_gru_cell_bwd_dma(int prec, const void* dgh_next16, const void* w_hhT16, const float* dy_t, float* carry_t,
                                    const void* save_r, const void* save_z, const void* save_n, const void* save_hn,
                                    const float* y_prev_t, void* dgi16, void* dgh16, float* db_ih, float* db_hh, int B,
                                    int D, int first, void* stream) {
  using namespace ark;
  if (!w_hhT16 || !carry_t || !save_r || !save_z || !save_n || !save_hn || !y_prev_t || !dgi16 || !dgh16 || B <= 0 || D <= 0)
    return ARK_ERR_ARG;
  if (!first && !dgh_next16) return ARK_ERR_ARG;
  if ((db_ih == nullptr) != (db_hh == nullptr)) return ARK_ERR_ARG;
  if (D % 64 != 0 || B % 16 != 0) return ARK_ERR_SHAPE;
  GruBwdDmaArgs p{dgh_next16 ? dgh_next16 : dgh16, w_hhT16, dy_t, carry_t, (const _Float16*)save_r, (const _Float16*)save_z,
                  (const _Float16*)save_n, (const _Float16*)save_hn, y_prev_t, dgi16, dgh16, db_ih, db_hh, nullptr, 0, B, D,
                  first ? 1 : 0, 0};
  if (prec == PREC_F16) return launch_bwd_dma<PREC_F16>(p, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_bwd_dma<PREC_BF16>(p, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_gru_h0_bwd_dma(int prec, const void* dgh0_16, const void* w_hhT16, const float* carry_t, float* dh0,
                                  int accumulate, int B, int D, void* stream) {
  using namespace ark;
  if (!dgh0_16 || !w_hhT16 || !carry_t || !dh0 || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0 || B % 16 != 0) return ARK_ERR_SHAPE;
  GruBwdDmaArgs p{dgh0_16, w_hhT16, nullptr, const_cast<float*>(carry_t), nullptr, nullptr, nullptr, nullptr, nullptr,
                  nullptr, nullptr, nullptr, nullptr, dh0, accumulate ? 1 : 0, B, D, 0, 1};
  if (prec == PREC_F16) return launch_bwd_dma<PREC_F16>(p, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_bwd_dma<PREC_BF16>(p, (hipStream_t)stream);
  return ARK_ERR_ARG;
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

// timing ablations of the forward cell (results invalid while != 0); see tools/cell_timing.py
extern "C" int ark_set_dma_debug(int mask) { ark::g_dma_dbg = mask; return 0; }
