// Layer-diagonal GRU forward and BPTT (engine: dma_core.h, DmaTile::run_segs).
//
// In a stacked GRU cell (l, t) depends on (l, t-1) and (l-1, t) only, so the cells of one
// anti-diagonal d = l + t are independent.  One launch runs them all: workgroup -> (role = layer,
// row tile, unit tile).  Each role forms BOTH products itself -- x_t W_ih^T and h_{t-1} W_hh^T stream
// through one LDS-DMA ring back to back as two K-segments -- so the per-layer input GEMM, its
// [B*L,3D] fp32 `gi` round trip and 2 of every 3 dependent launches disappear.  State is tile-native
// fp32, gate saves are tile-native fp16, the row-major 16-bit copies later products read are
// assembled in LDS, dropout masks come from a counter hash (common.h) in both directions.
// Reference op replaced: torch.nn.GRU (kgvae/model/models.py:121-127) and its autograd backward.
//
// Round-2 rebuild (rocprofv3 SQ counters, profiles/r02_diag_sq_counters.json): the kernels were
// instruction-issue bound, not memory bound -- 2 700 / 1 870 VALU instructions per wave against
// 192 / 160 MFMAs.  Fixed here: one consume loop per K-segment (no accumulator shuffling), hardware
// exp / rcp gate math, one 64-bit hash per 4 mask elements, and the backward cell writes ONE
// [B,4D] panel [dr | dz | dn | dn*r] instead of two [B,3D] panels that shared two thirds.
#include "dma_core.h"
#include "../../include/ark_amd.h"

// cache policy of the forward operands' LDS-DMA loads (experiment switches)
#ifndef ARK_FWD_AUXA
#define ARK_FWD_AUXA 0
#endif
#ifndef ARK_FWD_AUXB
#define ARK_FWD_AUXB 0
#endif

namespace ark {

typedef _Float16 dhalf4_t __attribute__((ext_vector_type(4)));

// epilogue stores / loads of once-touched state: nontemporal in the ARK_FWD_NT_ST experiment build
#ifdef ARK_FWD_NT_ST
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
#else
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { *p = v; }
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return *p; }
#endif

// Diagnostic build only (-DARK_STAMPS, tools/stamp_probe.py): shader-clock stamps per workgroup -- kernel entry, main loop
// start / end, epilogue math end, kernel end -- into a buffer of their own; the shipped library contains none of this.
#ifdef ARK_STAMPS
__device__ unsigned long long ark_stamp_buf[8192 * 8];
#define ARK_STAMP(i)                                                                       \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (threadIdx.x == 0) stamp_[i] = __builtin_amdgcn_s_memtime();                        \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define ARK_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime()
#define ARK_STAMP_FLUSH()                                                                  \
  do {                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                           \
      stamp_[5] = rt0_; stamp_[6] = __builtin_amdgcn_s_memrealtime();                      \
      unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));     \
      stamp_[7] = xcc & 15;                                                                \
      for (int i_ = 0; i_ < 8; ++i_) ark_stamp_buf[blockIdx.x * 8 + i_] = stamp_[i_];      \
    }                                                                                      \
  } while (0)
#else
#define ARK_STAMP(i) do {} while (0)
#define ARK_STAMP_DECL do {} while (0)
#define ARK_STAMP_FLUSH() do {} while (0)
#endif

struct GruDiagArgs {
  ArkGruDiagRole role[ARK_DIAG_MAX_ROLES];
  const float* hyper;
  int n_roles, B, D, xcd_map;
};

// BM rows x BU hidden units (x 3 gates) per workgroup, WGM x (BU/16) waves; wave tile (BM/WGM) x 48 = 16 units x 3 gates.
// MODE: 0 = two-barrier ring (small tiles, several workgroups per CU); 1 = single-barrier ring (run_segs<.., true>: measured,
// no instantiation shipped).  (Round 3's 128 x 64 ping-pong tile -- one 8-wave workgroup per CU, 21.8 us per launch against
// 19.1 -- was removed in round 5 together with its runner.)
template <int PREC, int PRECB, int NBUF, int KI, int BM, int BU, int WGM = 2, int MODE = 0>
__global__ __launch_bounds__(64 * WGM * (BU / 16)) void gru_diag_fwd_kernel(GruDiagArgs p) {
  ARK_CHAIN_PRIO();
  constexpr int BN = 3 * BU, WGN = BU / 16, NTHR = 64 * WGM * WGN;
  using G = DmaTile<PREC, BM, BN, NBUF, WGM, WGN, KI>;
  constexpr int TM = G::TM, TN = G::TN;
  static_assert(TN == 3, "wave tile = 16 units x (r, z, n)");
  using h_t = typename G::h_t;
  using h8 = typename G::h8;
  using hb_t = typename PrecTraits<PRECB>::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ARK_STAMP_DECL;
  ARK_STAMP(0);
  const int B = p.B, D = p.D;
  const int UT = D / BU, MT = (B + BM - 1) / BM;
  int role, mt, ut;
  if (p.xcd_map) {
    // consecutive workgroup ids go to consecutive XCDs (8 private L2s).  Give XCD x the unit tiles
    // = x (mod 4) and row tiles = x/4 (mod 2) of every role, so an L2 holds a quarter of the weight
    // panels and half of the activation rows instead of everything.  (host checks UT%4==0, MT%2==0)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_role = (UT / 4) * (MT / 2);
    role = j / per_role;
    const int k = j % per_role;
    ut = (k % (UT / 4)) * 4 + (xcd & 3);
    mt = (k / (UT / 4)) * 2 + (xcd >> 2);
  } else {
    role = blockIdx.x / (UT * MT);
    const int k = blockIdx.x % (UT * MT);
    mt = k / UT;
    ut = k % UT;
  }
  const ArkGruDiagRole R = p.role[role];   // (by value: the whole role in one burst of scalar loads, one wait)
  const int m0 = mt * BM, u0 = ut * BU;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int ul = wn * 16 + (lane & 15);
  const int u = u0 + ul;
  // epilogue operands first: older than every LDS-DMA op in the vmcnt queue -> they land underneath the
  // products.  (Tall tiles keep only the addresses: 4 x 16 rows of prefetched state would spill.)
  constexpr bool PRE = TM <= 2 || MODE != 0;   // (the 8-wave lone workgroup has 256 registers per lane)
  int rl[TM];
  long o[TM];
  f32x4 hp[PRE ? TM : 1];
  const bool drop = R.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(R.drop_seed, p.hyper, R.drop_p);
  // layer 0 of a small vocabulary: x_t W_ih^T is a ROW of x_tab = W_tok W_ih^T per token -- no K-segment for it (half of
  // this role's stream); the token ids ride with the other early loads, the table rows are read in the epilogue
  const float* xt = R.x_tab;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 tk[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    rl[tm] = wm * G::WTM + tm * 16 + 4 * (lane >> 4);
    const int rowc = min(m0 + rl[tm], B - 4);   // B % 16 == 0: clamped quads stay in bounds
    o[tm] = tile_native_off(rowc, u, D);
    if constexpr (PRE) hp[tm] = ld_stream(reinterpret_cast<const f32x4*>(R.y_prev_t + o[tm]));
    tk[tm] = xt ? *reinterpret_cast<const i32x4*>(R.x_tok + rowc) : i32x4{0, 0, 0, 0};
  }
  const float br = R.b_ih[u] + R.b_hh[u], bz = R.b_ih[D + u] + R.b_hh[D + u];
  const float bin = R.b_ih[2 * D + u], bhn = R.b_hh[2 * D + u];

  // r, z: input + recurrent parts summed; the candidate gate needs W_in x and W_hn h apart
  f32x4 acc[TM][TN + 1];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn <= TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  typename G::template Segs<2> sg{{reinterpret_cast<const h_t*>(R.x16), reinterpret_cast<const h_t*>(R.h_prev16)},
                                  {reinterpret_cast<const h_t*>(R.w_ih16), reinterpret_cast<const h_t*>(R.w_hh16)},
                                  {xt ? 0 : D, D}};
  auto rma = [=](int r) -> long { return (long)min(m0 + r, B - 1); };
  auto rmb = [=](int j) -> long { return (long)((j >> 4) % 3) * D + u0 + (j / 48) * 16 + (j & 15); };
  auto mm = [&](auto seg, const h8 (&a)[TM], const h8 (&b)[TN]) {
    constexpr int NCOL = decltype(seg)::value == 0 ? TN - 1 : TN;   // accumulator of the candidate-gate block
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      acc[tm][0] = G::PT::mfma(a[tm], b[0], acc[tm][0]);
      acc[tm][1] = G::PT::mfma(a[tm], b[1], acc[tm][1]);
      acc[tm][NCOL] = G::PT::mfma(a[tm], b[2], acc[tm][NCOL]);
    }
  };
  ARK_STAMP(1);
  G::template run_segs<2, MODE == 1, ARK_FWD_AUXA, ARK_FWD_AUXB>(sg, (long)D, (long)D, rma, rmb, smem, mm);
  ARK_STAMP(2);

  __syncthreads();   // ring is free: reuse it to assemble row-major 16-bit rows [BM][BU+pad]
  constexpr int TS = BU + 8;   // row stride in elements: 16-B aligned rows, spreads banks
  constexpr int ARR = BM * TS * 2;
  h_t* ta = reinterpret_cast<h_t*>(smem);
  hb_t* tb = reinterpret_cast<hb_t*>(smem + ARR);
  h_t* tda = reinterpret_cast<h_t*>(smem + 2 * ARR);
  hb_t* tdb = reinterpret_cast<hb_t*>(smem + 3 * ARR);
  _Float16* sr = reinterpret_cast<_Float16*>(R.save_r);
  _Float16* sz = reinterpret_cast<_Float16*>(R.save_z);
  _Float16* sn = reinterpret_cast<_Float16*>(R.save_n);
  _Float16* shn = reinterpret_cast<_Float16*>(R.save_hn);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (m0 + rl[tm] >= B) continue;
    f32x4 hpv;
    if constexpr (PRE) hpv = hp[tm];
    else hpv = ld_stream(reinterpret_cast<const f32x4*>(R.y_prev_t + o[tm]));
    if (xt) {   // (block-uniform) the input projection of these four rows' tokens
      f32x4 gx[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* row = xt + (long)tk[tm][i] * 3 * D + u;
        gx[0][i] = row[0];
        gx[1][i] = row[D];
        gx[2][i] = row[2 * D];
      }
      acc[tm][0] += gx[0];
      acc[tm][1] += gx[1];
      acc[tm][2] += gx[2];
    }
    f32x4 r, z, n, hn, h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = fast_sigmoid(acc[tm][0][i] + br);
      z[i] = fast_sigmoid(acc[tm][1][i] + bz);
      hn[i] = acc[tm][3][i] + bhn;
      n[i] = fast_tanh(acc[tm][2][i] + bin + r[i] * hn[i]);
      h[i] = n[i] + z[i] * (hpv[i] - n[i]);   // (1-z) n + z h_prev
    }
    st_stream(reinterpret_cast<f32x4*>(R.y_out_t + o[tm]), h);
    if (sr) {
      st_stream(reinterpret_cast<dhalf4_t*>(sr + o[tm]), dhalf4_t{(_Float16)r[0], (_Float16)r[1], (_Float16)r[2], (_Float16)r[3]});
      st_stream(reinterpret_cast<dhalf4_t*>(sz + o[tm]), dhalf4_t{(_Float16)z[0], (_Float16)z[1], (_Float16)z[2], (_Float16)z[3]});
      st_stream(reinterpret_cast<dhalf4_t*>(sn + o[tm]), dhalf4_t{(_Float16)n[0], (_Float16)n[1], (_Float16)n[2], (_Float16)n[3]});
      st_stream(reinterpret_cast<dhalf4_t*>(shn + o[tm]), dhalf4_t{(_Float16)hn[0], (_Float16)hn[1], (_Float16)hn[2], (_Float16)hn[3]});
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ta[(rl[tm] + i) * TS + ul] = G::PT::cvt(h[i]);
      if (R.y16b) tb[(rl[tm] + i) * TS + ul] = PrecTraits<PRECB>::cvt(h[i]);
    }
    if (drop) {
      const f32x4 hd = h * dropout_quad(dc, (uint64_t)(R.drop_base + o[tm]) >> 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tda[(rl[tm] + i) * TS + ul] = G::PT::cvt(hd[i]);
        tdb[(rl[tm] + i) * TS + ul] = PrecTraits<PRECB>::cvt(hd[i]);
      }
    }
  }
  __syncthreads();
  ARK_STAMP(3);
  // BM rows x (2*BU) B per array: thread t -> row t / CPR, 16-byte chunk t % CPR, NTHR / CPR rows per pass
  const int t = threadIdx.x;
  constexpr int CPR = BU / 8, RPP = NTHR / CPR;
#pragma unroll
  for (int r0 = 0; r0 < BM; r0 += RPP) {
    const int rr = r0 + t / CPR, ch = t % CPR;
    const int row = m0 + rr;
    if (rr < BM && row < B) {
      const long go = (long)row * D + u0 + ch * 8;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(R.y16a) + go), *reinterpret_cast<const u32x4*>(ta + rr * TS + ch * 8));
      if (R.y16b) st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(R.y16b) + go), *reinterpret_cast<const u32x4*>(tb + rr * TS + ch * 8));
      if (drop) {
        st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(R.yd16a) + go), *reinterpret_cast<const u32x4*>(tda + rr * TS + ch * 8));
        if (R.yd16b) st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(R.yd16b) + go), *reinterpret_cast<const u32x4*>(tdb + rr * TS + ch * 8));
      }
    }
  }
#ifdef ARK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (diagnostic build: the stores have left the wave)
#endif
  ARK_STAMP(4);
  ARK_STAMP_FLUSH();
}

template <int PREC, int PRECB, int NBUF, int KI, int BM, int BU, int WGM = 2, int MODE = 0>
static void launch_diag(const GruDiagArgs& p, hipStream_t st) {
  using G = DmaTile<PREC, BM, 3 * BU, NBUF, WGM, BU / 16, KI>;
  constexpr int MINL = 4 * BM * (BU + 8) * 2;
  constexpr int LDS = G::LDS_BYTES > MINL ? G::LDS_BYTES : MINL;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = gru_diag_fwd_kernel<PREC, PRECB, NBUF, KI, BM, BU, WGM, MODE>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  const unsigned grid = (unsigned)(p.n_roles * ((p.B + BM - 1) / BM) * (p.D / BU));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WGM * (BU / 16)), LDS, st, p);
}

template <int PREC, int PRECB>
static int launch_diag_cfg(GruDiagArgs& p, const ArkDiagTuning& tn, hipStream_t st) {
  bool ki2 = tn.fwd_ki == 2 && p.D % 128 == 0;
  int rows = tn.fwd_rows;
  int units = (tn.fwd_units == 64 && p.D % 64 == 0) ? 64 : 32;
  if (rows == 64 && units == 32 && (long)p.n_roles * ((p.B + 63) / 64) * (p.D / 32) < 256) {
    // small batch x width (e.g. B = 256, D = 128): 64-row tiles leave most CUs empty -> 32-row tiles, two k-images per
    // stage (measured on the wd-movies shape: 4.33 -> 4.12 ms/step)
    rows = 32;
    ki2 = p.D % 128 == 0;
    // ... and when even that fills at most 96 CUs (wd-articles B = 16, wd-movies B = 256 x D = 128): 16-unit tiles on
    // 128-thread workgroups, twice the workgroups, each streaming (32 + 48) instead of (32 + 96) rows of K
    if (tn.fwd_units == 0 && (long)p.n_roles * ((p.B + 31) / 32) * (p.D / 32) <= 96) units = 16;
  }
  if (tn.fwd_units == 16) { units = 16; rows = 32; ki2 = tn.fwd_ki == 2 && p.D % 128 == 0; }
  const int MT = (p.B + rows - 1) / rows, UT = p.D / units;
  // (measured and not kept: four ring slots instead of two for grids of at most one workgroup per CU -- wd-movies B=256,
  //  wd-articles B=16 -- 5.9 -> 6.2 us forward, 9.9 -> 13.6 us backward per launch)
  const bool deep = false;
  p.xcd_map = (tn.fwd_xcd && UT % 4 == 0 && MT % 2 == 0) ? 1 : 0;
  if (units == 64) {   // 8 waves
    if (ki2) launch_diag<PREC, PRECB, 2, 2, 64, 64>(p, st);
    else launch_diag<PREC, PRECB, 2, 1, 64, 64>(p, st);
  } else if (units == 16) {   // 2 waves
    // (measured on the wd-articles shape, B = 16, profiles/r03_deep_ring_sweep.txt / r03_helper_waves_sweep.txt: neither a
    //  six-slot ring -- 12 k-images in flight -- nor two extra waves that only issue LDS-DMA move these launches: 7.8 us
    //  per launch either way.  They are made of fixed costs -- launch boundary, first-stage latency, epilogue -- not of
    //  their eight ring stages.)
    if (ki2) launch_diag<PREC, PRECB, 2, 2, 32, 16>(p, st);
    else launch_diag<PREC, PRECB, 2, 1, 32, 16>(p, st);
  } else if (rows == 64) {
    if (ki2) launch_diag<PREC, PRECB, 2, 2, 64, 32>(p, st);
    else if (tn.fwd_nbuf >= 4) launch_diag<PREC, PRECB, 4, 1, 64, 32>(p, st);
    else launch_diag<PREC, PRECB, 2, 1, 64, 32>(p, st);
  } else {
    if (ki2 && deep) launch_diag<PREC, PRECB, 4, 2, 32, 32>(p, st);
    else if (ki2) launch_diag<PREC, PRECB, 2, 2, 32, 32>(p, st);
    else if (tn.fwd_nbuf >= 4 || deep) launch_diag<PREC, PRECB, 4, 1, 32, 32>(p, st);
    else launch_diag<PREC, PRECB, 2, 1, 32, 32>(p, st);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

static bool diag_tuning_ok(const ArkDiagTuning& t) {
  return (t.fwd_rows == 32 || t.fwd_rows == 64) && (t.fwd_ki == 1 || t.fwd_ki == 2) &&
         (t.fwd_nbuf == 2 || t.fwd_nbuf == 4) && (t.fwd_units == 0 || t.fwd_units == 16 || t.fwd_units == 32 || t.fwd_units == 64) &&
         (t.bwd_rows == 32 || t.bwd_rows == 64) && (t.bwd_ki == 1 || t.bwd_ki == 2) && (t.bwd_nbuf == 2 || t.bwd_nbuf == 4) &&
         (t.bwd_xcd_rows == 1 || t.bwd_xcd_rows == 2 || t.bwd_xcd_rows == 4 || t.bwd_xcd_rows == 8) &&
         (t.bwd_cols == 0 || t.bwd_cols == 32 || t.bwd_cols == 64);
}

}  // namespace ark

// measured on MI355X (syn-paths, B=1024): see DESIGN.md section 6
extern "C" void ark_diag_tuning_default(ArkDiagTuning* t) {
  if (!t) return;
  t->fwd_rows = 64; t->fwd_ki = 1; t->fwd_nbuf = 2; t->fwd_xcd = 1; t->fwd_units = 0;
  t->bwd_rows = 32; t->bwd_ki = 2; t->bwd_nbuf = 2; t->bwd_xcd_rows = 4;
  t->bwd_cols = 0;
}

extern "C" int ark_gru_diag_fwd(int prec, int prec_b, int n_roles, const ArkGruDiagRole* roles, const float* hyper, int B, int D,
                                const ArkDiagTuning* tuning, void* stream) {
  using namespace ark;
  if (!roles || n_roles <= 0 || n_roles > ARK_DIAG_MAX_ROLES || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0 || B % 16 != 0) return ARK_ERR_SHAPE;
  ArkDiagTuning tn;
  ark_diag_tuning_default(&tn);
  if (tuning) tn = *tuning;
  if (!diag_tuning_ok(tn)) return ARK_ERR_ARG;
  GruDiagArgs p;
  for (int i = 0; i < n_roles; ++i) {
    const ArkGruDiagRole& r = roles[i];
    if (!r.h_prev16 || !r.w_hh16 || !r.b_ih || !r.b_hh || !r.y_prev_t || !r.y_out_t || !r.y16a) return ARK_ERR_ARG;
    if (r.x_tab ? !r.x_tok : (!r.x16 || !r.w_ih16)) return ARK_ERR_ARG;
    if (r.x_tab && ((uintptr_t)r.x_tok & 15)) return ARK_ERR_ALIGN;   // four token ids per lane in one load
    if (r.drop_p < 0.f || r.drop_p >= 1.f || (r.drop_p > 0.f && (!r.yd16a || !hyper))) return ARK_ERR_ARG;
    if ((r.drop_base & 3) != 0) return ARK_ERR_ALIGN;   // one hash serves a quad of elements
    if (r.save_r && (!r.save_z || !r.save_n || !r.save_hn)) return ARK_ERR_ARG;
    p.role[i] = r;
    if (r.x_tab) {   // (the ring computes lane offsets from these bases even for a K = 0 segment: keep them valid)
      p.role[i].x16 = r.h_prev16;
      p.role[i].w_ih16 = r.w_hh16;
    }
  }
  for (int i = n_roles; i < ARK_DIAG_MAX_ROLES; ++i) p.role[i] = p.role[0];
  p.hyper = hyper;
  p.n_roles = n_roles;
  p.B = B;
  p.D = D;
  p.xcd_map = 0;
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F16 && prec_b == PREC_BF16) return launch_diag_cfg<PREC_F16, PREC_BF16>(p, tn, st);
  if (prec == PREC_F16 && prec_b == PREC_F16) return launch_diag_cfg<PREC_F16, PREC_F16>(p, tn, st);
  if (prec == PREC_BF16 && prec_b == PREC_BF16) return launch_diag_cfg<PREC_BF16, PREC_BF16>(p, tn, st);
  return ARK_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// Layer-diagonal BPTT.  Cell (l, t) needs (l, t+1) [its gate-gradient panel, carry] and (l+1, t) [dgi]; the
// cells of one backward anti-diagonal are independent.  A role streams three K-segments through one ring:
//   dgi(l+1,t)   [B,3D]  x  W_ih(l+1)^T            -> ax  (passes through this layer's output-dropout mask)
//   [dr|dz](l,t+1) [B,2D] x  W_hh(l)^T[:, 0:2D]    -> ah
//   dn*r (l,t+1)  [B,D]   x  W_hh(l)^T[:, 2D:3D]   -> ah
// then runs the gate-derivative epilogue: the row-major 16-bit panel [dr | dz | dn | dn*r] is assembled in
// LDS, bias gradients are column sums of the tile.
namespace ark {

struct GruDiagBwdArgs {
  ArkGruDiagBwdRole role[ARK_DIAG_MAX_ROLES];
  const float* hyper;
  int n_roles, B, D, xcd_m;
};

template <int PREC, int NBUF, int KI, int BM, int BN = 64, int WGM = 2, bool ONEBAR = false>
__global__ __launch_bounds__(128 * WGM) void gru_diag_bwd_kernel(GruDiagBwdArgs p) {
  ARK_CHAIN_PRIO();
  static_assert(BN == 64 || BN == 32, "64 output columns per workgroup, or 32 for grids that would leave most CUs empty");
  using G = DmaTile<PREC, BM, BN, NBUF, WGM, 2, KI>;   // wave tile (BM/WGM) x (BN/2); WGM = 4: 128 x 64 on 8 waves
  constexpr int TM = G::TM, TN = G::TN, WN = BN / 2, NTHR = 128 * WGM;
  using h_t = typename G::h_t;
  using h8 = typename G::h8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int B = p.B, D = p.D;
  const int NT = D / BN, MT = (B + BM - 1) / BM;
  int role, mt, nt;
  if (p.xcd_m > 1) {
    // consecutive workgroup ids land on consecutive XCDs (8 private L2s).  XCD x serves the row tiles
    // = x % xcd_m (mod xcd_m) and the unit tiles = x / xcd_m (mod 8/xcd_m) of every role: its L2 then
    // fetches 1/xcd_m of the gate-gradient rows and xcd_m/8 of the transposed weight panels instead of
    // all rows (PMC: the plain order fetched the [B,3D] panels once per XCD, 124 MB per launch).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int xm = p.xcd_m, xn = 8 / p.xcd_m;
    const int per_role = (MT / xm) * (NT / xn);
    role = j / per_role;
    const int k = j % per_role;
    nt = (k % (NT / xn)) * xn + xcd / xm;
    mt = (k / (NT / xn)) * xm + xcd % xm;
  } else {
    role = blockIdx.x / (NT * MT);
    const int kk = blockIdx.x % (NT * MT);
    mt = kk / NT;
    nt = kk % NT;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const ArkGruDiagBwdRole R = p.role[role];   // (by value: the whole role in one burst of scalar loads, one wait -- by reference its fields came one dependent s_load at a time)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const bool fin = R.dh0 != nullptr;   // initial-state role: dh0 += carry + dgh_0 W_hh, nothing else
  const bool top = R.dgi_up16 == nullptr;
  const bool drop = R.drop_p > 0.f && !top && !fin;
  const _Float16* sr = reinterpret_cast<const _Float16*>(R.save_r);
  const _Float16* sz = reinterpret_cast<const _Float16*>(R.save_z);
  const _Float16* sn = reinterpret_cast<const _Float16*>(R.save_n);
  const _Float16* shn = reinterpret_cast<const _Float16*>(R.save_hn);
  // epilogue operands first (older than the LDS-DMA ops -> they land underneath the products)
  int rl[TM];
  f32x4 pc[TM][TN], pdy[TM][TN], phy[TM][TN];
  dhalf4_t psr[TM][TN], psz[TM][TN], psn[TM][TN], phn[TM][TN];
  DropCtx dc{};
  if (drop) dc = drop_ctx(R.drop_seed, p.hyper, R.drop_p);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    rl[tm] = wm * G::WTM + tm * 16 + 4 * (lane >> 4);
    const int rowc = min(m0 + rl[tm], B - 4);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const long o = tile_native_off(rowc, n0 + wn * WN + tn * 16 + (lane & 15), D);
      pc[tm][tn] = R.first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(R.carry_t + o);
      pdy[tm][tn] = R.dy_t ? *reinterpret_cast<const f32x4*>(R.dy_t + o) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (fin) continue;
      phy[tm][tn] = *reinterpret_cast<const f32x4*>(R.y_prev_t + o);
      psr[tm][tn] = *reinterpret_cast<const dhalf4_t*>(sr + o);
      psz[tm][tn] = *reinterpret_cast<const dhalf4_t*>(sz + o);
      psn[tm][tn] = *reinterpret_cast<const dhalf4_t*>(sn + o);
      phn[tm][tn] = *reinterpret_cast<const dhalf4_t*>(shn + o);
    }
  }

  f32x4 ax[TM][TN], ah[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) ax[tm][tn] = ah[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const h_t* gup = reinterpret_cast<const h_t*>(R.dgi_up16);
  const h_t* gnx = reinterpret_cast<const h_t*>(R.dg_next16);
  const h_t* whT = reinterpret_cast<const h_t*>(R.w_hhT16);
  const bool rec = !R.first;   // a successor step exists (always true for the initial-state role)
  typename G::template Segs<3> sg{{gup, gnx, gnx + 3 * D},
                                  {reinterpret_cast<const h_t*>(R.w_ihT_up16), whT, whT + 2 * D},
                                  {top ? 0 : 3 * D, rec ? 2 * D : 0, rec ? D : 0}};
  G::template run_segs<3, ONEBAR>(
      sg, 4L * D, 3L * D, [=](int r) -> long { return (long)min(m0 + r, B - 1); }, [=](int r) -> long { return (long)(n0 + r); },
      smem, [&](auto seg, const h8 (&a)[TM], const h8 (&b)[TN]) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            if constexpr (decltype(seg)::value == 0) ax[tm][tn] = G::PT::mfma(a[tm], b[tn], ax[tm][tn]);
            else ah[tm][tn] = G::PT::mfma(a[tm], b[tn], ah[tm][tn]);
          }
      });

  if (fin) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row0 = m0 + rl[tm];
      if (row0 >= B) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int u = n0 + wn * WN + tn * 16 + (lane & 15);
        const f32x4 dh = ah[tm][tn] + pc[tm][tn];
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&R.dh0[(long)(row0 + i) * D + u], dh[i]);
      }
    }
    return;   // block-uniform: every wave of this workgroup serves the same role
  }
  __syncthreads();
  // LDS assembly of the row-major 16-bit output panel: [part 0..3 = dr, dz, dn, dn*r][BM rows][BN+8]
  constexpr int TS = BN + 8;
  h_t* tg = reinterpret_cast<h_t*>(smem);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row0 = m0 + rl[tm];
    if (row0 >= B) continue;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int ul = wn * WN + tn * 16 + (lane & 15);
      const long o = tile_native_off(row0, n0 + ul, D);
      f32x4 dh = ah[tm][tn] + pc[tm][tn] + pdy[tm][tn];
      if (drop) dh += ax[tm][tn] * dropout_quad(dc, (uint64_t)(R.drop_base + o) >> 2);
      else dh += ax[tm][tn];
      const f32x4 hp = phy[tm][tn];
      f32x4 cz;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = (float)psr[tm][tn][i], z = (float)psz[tm][tn][i], n = (float)psn[tm][tn][i], hn = (float)phn[tm][tn][i];
        const float dn_pre = dh[i] * (1.0f - z) * (1.0f - n * n);
        const float dz_pre = dh[i] * (hp[i] - n) * z * (1.0f - z);
        const float dr_pre = dn_pre * hn * r * (1.0f - r);
        cz[i] = dh[i] * z;
        const int ro = (rl[tm] + i) * TS + ul;
        tg[ro] = G::PT::cvt(dr_pre);
        tg[BM * TS + ro] = G::PT::cvt(dz_pre);
        tg[2 * BM * TS + ro] = G::PT::cvt(dn_pre);
        tg[3 * BM * TS + ro] = G::PT::cvt(dn_pre * r);
      }
      *reinterpret_cast<f32x4*>(R.carry_t + o) = cz;
    }
  }
  __syncthreads();
  // 4 parts x BM rows x (2 BN) B: thread t -> (row, 16-byte chunk) for each part, NTHR / CPR rows per pass
  const int t = threadIdx.x;
  constexpr int CPR = BN / 8;   // chunks per row: 8 -> 32 rows per 256 threads, 4 -> 64
  constexpr int RPASS = NTHR / CPR;
  h_t* g16 = reinterpret_cast<h_t*>(R.dg16);
#pragma unroll
  for (int r0 = 0; r0 < BM; r0 += RPASS) {
    const int rr = r0 + t / CPR, ch = t % CPR;
    const int row = m0 + rr;
    if (rr < BM && row < B) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long go = (long)row * 4 * D + (long)g * D + n0 + ch * 8;
        *reinterpret_cast<uint4*>(g16 + go) = *reinterpret_cast<const uint4*>(tg + g * BM * TS + rr * TS + ch * 8);
      }
    }
  }
  // bias gradients: column sums of this tile straight from LDS, one atomic per (gate, unit):
  // db_ih = colsum [dr | dz | dn], db_hh = colsum [dr | dz | dn*r]
  constexpr int NSPL = NTHR / (4 * BN) > 0 ? NTHR / (4 * BN) : 1;   // row groups per column (2 on the 8-wave tile)
  if (R.db_ih && t < 4 * BN * NSPL) {
    const int sp = t / (4 * BN), g = (t / BN) & 3, ul = t % BN;   // 4 parts x BN units x NSPL row groups
    const int nrows = min(BM, B - m0);
    const int r_lo = sp * (BM / NSPL), r_hi = min(nrows, (sp + 1) * (BM / NSPL));
    float s = 0.f;
    for (int r2 = r_lo; r2 < r_hi; ++r2) s += (float)tg[g * BM * TS + r2 * TS + ul];
    if (g < 3) atomicAdd(&R.db_ih[(long)g * D + n0 + ul], s);
    if (g < 2) atomicAdd(&R.db_hh[(long)g * D + n0 + ul], s);
    if (g == 3) atomicAdd(&R.db_hh[2L * D + n0 + ul], s);
  }
}

template <int PREC, int NBUF, int KI, int BM, int BN = 64, int WGM = 2, bool ONEBAR = false>
static void launch_diag_bwd(const GruDiagBwdArgs& p, hipStream_t st) {
  using G = DmaTile<PREC, BM, BN, NBUF, WGM, 2, KI>;
  constexpr int MINL = 4 * BM * (BN + 8) * 2;
  constexpr int LDS = G::LDS_BYTES > MINL ? G::LDS_BYTES : MINL;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = gru_diag_bwd_kernel<PREC, NBUF, KI, BM, BN, WGM, ONEBAR>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  const unsigned grid = (unsigned)(p.n_roles * ((p.B + BM - 1) / BM) * (p.D / BN));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WGM), LDS, st, p);
}

template <int PREC>
static int launch_diag_bwd_cfg(GruDiagBwdArgs& p, const ArkDiagTuning& tn, hipStream_t st) {
  const bool ki2 = tn.bwd_ki == 2 && p.D % 128 == 0;   // every K-segment (3D, 2D, D) must be a whole number of stages
  // small batch x width (wd-articles B = 16: 24 workgroups of 32 x 64): a launch is bound by what ONE CU can stream, so
  // halve the column tile and use twice the CUs (each streams (32 + 32) instead of (32 + 64) rows of K)
  const bool narrow = tn.bwd_cols == 32 ||
                      (tn.bwd_cols == 0 && tn.bwd_rows == 32 && (long)p.n_roles * ((p.B + 31) / 32) * (p.D / 64) <= 96);
  {
    const int rows = tn.bwd_rows;
    const int MT = (p.B + rows - 1) / rows, NT = p.D / (narrow ? 32 : 64);
    const int xm = tn.bwd_xcd_rows;
    p.xcd_m = (xm > 1 && MT % xm == 0 && NT % (8 / xm) == 0) ? xm : 1;
  }
  if (narrow) {
    if (ki2) launch_diag_bwd<PREC, 2, 2, 32, 32>(p, st);   // (eight slots / four helper waves: 12.1 -> 11.9 / 13.1 us, not kept)
    else launch_diag_bwd<PREC, 2, 1, 32, 32>(p, st);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  // (measured and not shipped: 128 x 64 backward tiles on 8 waves with the single-barrier ring, one workgroup per CU --
  //  27.5 us (three 48-KB slots) / 30.3 us (five 24-KB slots) per launch against 23.2 for the default 32 x 64 tiles)
  const bool deep = false;   // (see the forward launch)
  if (tn.bwd_rows == 64 && !deep) {
    if (ki2) launch_diag_bwd<PREC, 2, 2, 64>(p, st);
    else if (tn.bwd_nbuf >= 4) launch_diag_bwd<PREC, 4, 1, 64>(p, st);
    else launch_diag_bwd<PREC, 2, 1, 64>(p, st);
  } else {
    if (ki2 && deep) launch_diag_bwd<PREC, 4, 2, 32>(p, st);
    else if (ki2) launch_diag_bwd<PREC, 2, 2, 32>(p, st);
    else if (tn.bwd_nbuf >= 4 || deep) launch_diag_bwd<PREC, 4, 1, 32>(p, st);
    else launch_diag_bwd<PREC, 2, 1, 32>(p, st);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

extern "C" int ark_gru_diag_bwd(int prec, int n_roles, const ArkGruDiagBwdRole* roles, const float* hyper, int B, int D,
                                const ArkDiagTuning* tuning, void* stream) {
  using namespace ark;
  if (!roles || n_roles <= 0 || n_roles > ARK_DIAG_MAX_ROLES || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0 || B % 16 != 0) return ARK_ERR_SHAPE;
  ArkDiagTuning tn;
  ark_diag_tuning_default(&tn);
  if (tuning) tn = *tuning;
  if (!diag_tuning_ok(tn)) return ARK_ERR_ARG;
  GruDiagBwdArgs p;
  for (int i = 0; i < n_roles; ++i) {
    const ArkGruDiagBwdRole& r = roles[i];
    if (!r.w_hhT16 || !r.carry_t) return ARK_ERR_ARG;
    if (r.dh0) {   // initial-state role
      if (!r.dg_next16 || r.first || r.dgi_up16 || r.w_ihT_up16 || r.dy_t) return ARK_ERR_ARG;
      p.role[i] = r;
      continue;
    }
    if (!r.save_r || !r.save_z || !r.save_n || !r.save_hn || !r.y_prev_t || !r.dg16) return ARK_ERR_ARG;
    if (!r.first && !r.dg_next16) return ARK_ERR_ARG;
    if ((r.dgi_up16 == nullptr) != (r.w_ihT_up16 == nullptr)) return ARK_ERR_ARG;
    if ((r.dgi_up16 == nullptr) == (r.dy_t == nullptr)) return ARK_ERR_ARG;   // exactly one source of dy
    if ((r.db_ih == nullptr) != (r.db_hh == nullptr)) return ARK_ERR_ARG;
    if (r.drop_p < 0.f || r.drop_p >= 1.f || (r.drop_p > 0.f && !hyper)) return ARK_ERR_ARG;
    if ((r.drop_base & 3) != 0) return ARK_ERR_ALIGN;
    p.role[i] = r;
  }
  for (int i = n_roles; i < ARK_DIAG_MAX_ROLES; ++i) p.role[i] = roles[0];
  p.hyper = hyper;
  p.n_roles = n_roles;
  p.B = B;
  p.D = D;
  p.xcd_m = 1;
  if (prec == PREC_F16) return launch_diag_bwd_cfg<PREC_F16>(p, tn, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_diag_bwd_cfg<PREC_BF16>(p, tn, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

#ifdef ARK_STAMPS
extern "C" int ark_debug_stamps(unsigned long long* host, int n_blocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ark::ark_stamp_buf), sizeof(unsigned long long) * 8 * (size_t)n_blocks);
}
#endif
