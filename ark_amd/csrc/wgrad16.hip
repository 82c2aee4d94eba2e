// Weight-gradient products on the LDS-DMA ring with hardware-transposed fragment reads:
//     C[M,N] += sum_k A16[k, M] * B16[k, N]          (K = B*L tokens, both operands reduction-MAJOR)
// i.e. dW = dgate^T x activation straight from the row-major 16-bit panels the GRU cells wrote.
// LDS-DMA cannot transpose, so the stage image stays [k][m] (m contiguous, as in memory) and the
// MFMA fragments are fetched with ds_read_b64_tr_b16, which hands each lane one COLUMN (fixed m)
// of a 4(k) x 16(m) block -- two of them give the 8 consecutive k an MFMA 16x16x32 lane needs.
// A 32-byte-slot XOR keyed by k makes the 8 k-rows a half-wave touches land on distinct bank slots;
// it is applied to the LDS-DMA SOURCE address (the DMA write itself is lane-linear).
// Split-K over blockIdx.y with fp32 atomics into the (pre-zeroed / accumulating) gradient buffer.
// Reference op replaced: autograd's weight gradients of nn.GRU / nn.Linear (kgvae/model/models.py:121-128).
#include "dma_core.h"
#include "../../include/ark_amd.h"

namespace ark {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int RB>  // RB = bytes per k-row of the image (128 or 256)
__device__ __forceinline__ int tr_sw(int k) {
  if constexpr (RB == 128) return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);   // 4 slots of 32 B
  else return (k & 3) | (((k >> 3) & 1) << 2);                              // 8 slots of 32 B
}

struct Wgrad16Args {
  const void* A; const void* B; float* C;
  long lda, ldb, ldc;
  int M, N, K, k_chunk, tiles_n, use_atomics;
  int m_valid;   // rows of C that exist (<= M): A may be column-padded to a tile multiple, C is not
};
// several independent products in ONE launch (e.g. dW_ih / dW_hh of every GRU layer at the end of
// the backward pass): enough workgroups to fill the chip without deep split-K and its atomics
constexpr int kMaxGroup = ARK_WGRAD_MAX_GROUP;
struct Wgrad16Group {
  Wgrad16Args p[kMaxGroup];
  int tile_start[kMaxGroup + 1];
  int n;
  // Balanced mode (n_long > 0; 1-D grid): the first n_long tiles are whole-K workgroups, one per CU; every
  // remaining tile is cut into s_short k-slices (fp32 atomics).  288 tiles on 256 CUs would otherwise leave
  // 32 CUs with two whole tiles and everyone else waiting: 256 whole + 32 x 8 slices gives every CU 1.125 tiles.
  int n_long, s_short;
};

// square BT x BT output tile; NWV = 8 waves as 2 (m) x 4 (n), wave tile BT/2 x BT/4, or NWV = 4 waves as 2 x 2, wave tile
// BT/2 x BT/2 (round 5: the two workgroups of a CU then put ONE wave each on a SIMD -- partners that run out of step --
// and a wave multiplies 32 tiles per stage instead of 16 between the same two barriers, reading a third fewer fragments)
template <int PREC, int BT, int NBUF, int NWV_ = 8>
__global__ __launch_bounds__(64 * NWV_) void wgrad16_kernel(Wgrad16Group grp) {
  int gtile = blockIdx.x, slice = blockIdx.y, nslice = gridDim.y;
  if (grp.n_long > 0 && (int)blockIdx.x < grp.n_long) {
    gtile = xcd_remap(blockIdx.x, grp.n_long);   // consecutive tiles share operand panels: keep them on one XCD
  } else if (grp.n_long == 0 && grp.n > 1) {
    gtile = xcd_remap(blockIdx.x, gridDim.x);
  }
  if (grp.n_long > 0 && (int)blockIdx.x >= grp.n_long) {
    const int j = blockIdx.x - grp.n_long;
    gtile = grp.n_long + j / grp.s_short;
    slice = j % grp.s_short;
    nslice = grp.s_short;
  }
  int gi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < grp.n && gtile >= grp.tile_start[i]) gi = i;
  const Wgrad16Args p = grp.p[gi];
  const int bid_raw = gtile - grp.tile_start[gi];
  const int ntiles = grp.tile_start[gi + 1] - grp.tile_start[gi];
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  // Two waves per SIMD: a wave may keep at most 16 LDS reads in flight, and one wave per SIMD left the
  // matrix cores idle during every fragment fetch (measured: 1.2 us per 64-k stage vs 0.46 us of LDS-DMA).
  constexpr int NWV = NWV_;
  static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
  constexpr int RB = BT * 2;                 // bytes per k-row per operand
  constexpr int LPR = RB / 16;               // lanes (16-B chunks) per k-row
  constexpr int RPP = 64 / LPR;              // k-rows per 1-KB LDS-DMA piece
  constexpr int PIECES = 64 / RPP;           // pieces per operand per stage
  constexpr int PPW = PIECES / NWV;          // pieces per wave per operand
  static_assert(PPW >= 1, "tile too small for 8 waves");
  constexpr int OP_BYTES = 64 * RB;
  constexpr int STAGE = 2 * OP_BYTES;
  constexpr int LPS = 2 * PPW;
  constexpr int WGN = NWV / 2;               // waves along n
  constexpr int WTM = BT / 2, WTN = BT / WGN;  // wave tile
  constexpr int TA = WTM / 16, TB = WTN / 16;
  static_assert((NBUF - 1) * LPS <= 63, "vmcnt");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int bid = (grp.n == 1) ? xcd_remap(bid_raw, ntiles) : bid_raw;
  const int m0 = (bid / p.tiles_n) * BT, n0 = (bid % p.tiles_n) * BT;
  const int k_chunk = (grp.n_long > 0) ? ((p.K + nslice - 1) / nslice + 63) / 64 * 64 : p.k_chunk;
  const int kb = slice * k_chunk;
  const int kend = min(p.K, kb + k_chunk);
  const bool atomics = (grp.n_long > 0) ? nslice > 1 : p.use_atomics != 0;
  const int NS = (kend - kb) / 64;

  const h_t* A = reinterpret_cast<const h_t*>(p.A) + (long)kb * p.lda + m0;
  const h_t* Bm = reinterpret_cast<const h_t*>(p.B) + (long)kb * p.ldb + n0;

  // per-lane DMA source offsets (elements) for this wave's pieces; LDS image is lane-linear
  long aoff[PPW], boff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave + NWV * i;
    const int k = piece * RPP + lane / LPR;            // k-row inside the stage
    const int pc = lane % LPR;                          // physical 16-B chunk
    const int c = pc ^ (tr_sw<RB>(k) << 1);             // logical chunk
    aoff[i] = (long)k * p.lda + 8 * c;
    boff[i] = (long)k * p.ldb + 8 * c;
  }
  auto issue = [&](int s) {
    char* base = smem + (s % NBUF) * STAGE;
    const long ka = (long)s * 64 * p.lda, kbb = (long)s * 64 * p.ldb;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + ka + aoff[i]),
                                       (__attribute__((address_space(3))) void*)(base + (wave + NWV * i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bm + kbb + boff[i]),
                                       (__attribute__((address_space(3))) void*)(base + OP_BYTES + (wave + NWV * i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane = 16*g + 4*q + pp  ->  k-row 8g+q (+4), columns 4pp..4pp+3 of the 16-wide block
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  // (inline-asm reads: see lds_tr16 in dma_core.h -- the builtin gets an s_waitcnt vmcnt(0) in front of the first read of every
  //  stage, which drains the ring: 101 us for the GRU group whatever the ring depth or wave layout)
  auto frag = [&](const char* op, int s2, int col0, tr16x4& lo, tr16x4& hi) {
    const int k1 = 32 * s2 + 8 * g + q, k2 = k1 + 4;
    const int cb = (col0 * 2) / 16 + (pp >> 1);         // logical 16-B chunk of this lane's 8 bytes
    const int sub = (pp & 1) * 8;
    lo = lds_tr16(op + k1 * RB + ((cb ^ (tr_sw<RB>(k1) << 1)) << 4) + sub);
    hi = lds_tr16(op + k2 * RB + ((cb ^ (tr_sw<RB>(k2) << 1)) << 4) + sub);
  };
  using h8 = typename PT::h8;

  if (NS > 0) {
    const int pre = NS < NBUF ? NS : NBUF;
    for (int s = 0; s < pre; ++s) issue(s);
    for (int s = 0; s < NS; ++s) {
      const int ahead = (NS - 1 - s) < (NBUF - 1) ? (NS - 1 - s) : (NBUF - 1);
      switch (ahead) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF > 2 ? 2 : 0) * LPS) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF > 3 ? 3 : 0) * LPS) : "memory"); break;
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* opA = smem + (s % NBUF) * STAGE;
      const char* opB = opA + OP_BYTES;
      // both 32-k halves of the stage are requested at once; the second half's reads land under the first half's products
      tr16x4 fa[2][TA][2], fb[2][TB][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int t = 0; t < TA; ++t) frag(opA, s2, wm * WTM + t * 16, fa[s2][t][0], fa[s2][t][1]);
#pragma unroll
        for (int t = 0; t < TB; ++t) frag(opB, s2, wn * WTN + t * 16, fb[s2][t][0], fb[s2][t][1]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 == 0) tr_wait_cnt<2 * (TA + TB)>(); else tr_wait_cnt<0>();
#pragma unroll
        for (int t = 0; t < TA; ++t) { tr_tie(fa[s2][t][0]); tr_tie(fa[s2][t][1]); }
#pragma unroll
        for (int t = 0; t < TB; ++t) { tr_tie(fb[s2][t][0]); tr_tie(fb[s2][t][1]); }
        h8 b[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) b[t] = tr_join<h8>(fb[s2][t][0], fb[s2][t][1]);
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) {
          const h8 a = tr_join<h8>(fa[s2][ta][0], fa[s2][ta][1]);
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) acc[ta][tb] = PT::mfma(a, b[tb], acc[ta][tb]);
        }
        __builtin_amdgcn_sched_barrier(0);   // (the first half's products stay in front of the second wait)
      }
      if (s + NBUF < NS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + NBUF);
      }
    }
  }

#pragma unroll
  for (int ta = 0; ta < TA; ++ta)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * WTM + ta * 16 + 4 * (lane >> 4) + i;
        const int col = n0 + wn * WTN + tb * 16 + (lane & 15);
        if (row >= p.m_valid) continue;
        float* c = p.C + (long)row * p.ldc + col;
        if (atomics) atomicAdd(c, acc[ta][tb][i]);
        else *c += acc[ta][tb][i];
      }
}

template <class K>
static void allow_lds_w(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}


template <int PREC, int BT, int NBUF, int NWV = 8>
static void launch_wg(Wgrad16Group& g, const ArkWgradTuning& tn, hipStream_t st) {
  constexpr int LDS = NBUF * 2 * 64 * BT * 2;
  static bool once = (allow_lds_w(wgrad16_kernel<PREC, BT, NBUF, NWV>, LDS), true); (void)once;
  long tiles = 0;
  int kmax = 0;
  for (int i = 0; i < g.n; ++i) {
    g.p[i].tiles_n = g.p[i].N / BT;
    g.tile_start[i] = (int)tiles;
    tiles += (long)(g.p[i].M / BT) * g.p[i].tiles_n;
    if (g.p[i].K > kmax) kmax = g.p[i].K;
  }
  g.tile_start[g.n] = (int)tiles;
  int split = 1;
  while (tiles * split < tn.target_wgs && kmax / (split * 2) >= 256 && split < 64) split *= 2;
  for (int i = 0; i < g.n; ++i) {
    g.p[i].k_chunk = ((g.p[i].K + split - 1) / split + 63) / 64 * 64;
    g.p[i].use_atomics = split > 1;
  }
  g.n_long = 0;
  g.s_short = 1;
  constexpr int kCUs = 256;
  if (tn.balance && split == 1 && tiles > kCUs && tiles < 2 * kCUs) {
    const int rest = (int)tiles - kCUs;
    const int s = kCUs / rest;   // slices per remaining tile: rest * s <= 256 short workgroups
    if (s >= 2 && kmax / s >= 256) {
      g.n_long = kCUs;
      g.s_short = s;
      hipLaunchKernelGGL((wgrad16_kernel<PREC, BT, NBUF, NWV>), dim3((unsigned)(kCUs + rest * s), 1), dim3(64 * NWV), LDS, st, g);
      return;
    }
  }
  hipLaunchKernelGGL((wgrad16_kernel<PREC, BT, NBUF, NWV>), dim3((unsigned)tiles, (unsigned)split), dim3(64 * NWV), LDS, st, g);
}

template <int PREC>
static int launch_wg_prec(Wgrad16Group& g, const ArkWgradTuning* tuning, hipStream_t st) {
  ArkWgradTuning tn;
  ark_wgrad_tuning_default(&tn);
  if (tuning) tn = *tuning;
  if ((tn.tile != 64 && tn.tile != 128) || tn.nbuf < 2 || tn.nbuf > 4 || tn.target_wgs < 1 || (tn.waves != 4 && tn.waves != 8)) return ARK_ERR_ARG;
  bool ok128 = tn.tile == 128;
  for (int i = 0; i < g.n; ++i) ok128 = ok128 && g.p[i].M % 128 == 0 && g.p[i].N % 128 == 0;
  if (ok128 && tn.waves == 4) {
    if (tn.nbuf == 2) launch_wg<PREC, 128, 2, 4>(g, tn, st);
    else launch_wg<PREC, 128, 3, 4>(g, tn, st);
  } else if (ok128) {
    if (tn.nbuf == 2) launch_wg<PREC, 128, 2>(g, tn, st);
    else if (tn.nbuf == 3) launch_wg<PREC, 128, 3>(g, tn, st);
    else launch_wg<PREC, 128, 4>(g, tn, st);
  } else {
    if (tn.nbuf == 2) launch_wg<PREC, 64, 2>(g, tn, st); else launch_wg<PREC, 64, 4>(g, tn, st);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

// measured defaults (MI355X, syn-paths B=1024): 128x128 tiles, two ring slots, split K until >= 300 workgroups
// (240 whole-K tiles -> 480 half-K: same-box A/B 1.245 -> 1.202 ms/step), even dealing of 257..511 tiles over the CUs
extern "C" void ark_wgrad_tuning_default(ArkWgradTuning* t) {
  if (!t) return;
  t->tile = 128; t->nbuf = 2; t->target_wgs = 300; t->balance = 1; t->waves = 8;
}

static int check_one(const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int M, int N, int K) {
  if (!A16 || !B16 || !C || M <= 0 || N <= 0 || K <= 0) return ARK_ERR_ARG;
  if (M % 64 != 0 || N % 64 != 0 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0) return ARK_ERR_SHAPE;
  if (((uintptr_t)A16 | (uintptr_t)B16) & 15) return ARK_ERR_ALIGN;
  return 0;
}

// C[M,N] += A16[K,M]^T B16[K,N]; requires M % 64 == N % 64 == K % 64 == 0, 16-byte aligned rows.
extern "C" int ark_wgrad16(int prec, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                           int M, int N, int K, const ArkWgradTuning* tuning, void* stream) {
  using namespace ark;
  int rc = check_one(A16, lda, B16, ldb, C, M, N, K);
  if (rc) return rc;
  Wgrad16Group g{};
  g.n = 1;
  g.p[0] = Wgrad16Args{A16, B16, C, (long)lda, (long)ldb, (long)ldc, M, N, K, K, 1, 0, M};
  if (prec == PREC_F16) return launch_wg_prec<PREC_F16>(g, tuning, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_wg_prec<PREC_BF16>(g, tuning, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

// n (<= ARK_WGRAD_MAX_GROUP) independent products of the kind above in ONE launch
extern "C" int ark_wgrad16_group(int prec, int n, const void* const* A16, const int64_t* lda, const void* const* B16,
                                 const int64_t* ldb, float* const* C, const int64_t* ldc, const int* M, const int* N,
                                 const int* K, const ArkWgradTuning* tuning, void* stream) {
  using namespace ark;
  if (n <= 0 || n > kMaxGroup || !A16 || !B16 || !C || !lda || !ldb || !ldc || !M || !N || !K) return ARK_ERR_ARG;
  Wgrad16Group g{};
  g.n = n;
  for (int i = 0; i < n; ++i) {
    int rc = check_one(A16[i], lda[i], B16[i], ldb[i], C[i], M[i], N[i], K[i]);
    if (rc) return rc;
    g.p[i] = Wgrad16Args{A16[i], B16[i], C[i], (long)lda[i], (long)ldb[i], (long)ldc[i], M[i], N[i], K[i], K[i], 1, 0, M[i]};
  }
  if (prec == PREC_F16) return launch_wg_prec<PREC_F16>(g, tuning, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_wg_prec<PREC_BF16>(g, tuning, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

// as ark_wgrad16, for an A operand whose M columns are padded to a tile multiple while C has only m_valid rows
// (e.g. the K-padded 16-bit dlogits against the [V,D] vocabulary gradient): rows >= m_valid are not written
extern "C" int ark_wgrad16_rows(int prec, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                                int M, int m_valid, int N, int K, const ArkWgradTuning* tuning, void* stream) {
  using namespace ark;
  int rc = check_one(A16, lda, B16, ldb, C, M, N, K);
  if (rc) return rc;
  if (m_valid <= 0 || m_valid > M) return ARK_ERR_ARG;
  Wgrad16Group g{};
  g.n = 1;
  g.p[0] = Wgrad16Args{A16, B16, C, (long)lda, (long)ldb, (long)ldc, M, N, K, K, 1, 0, m_valid};
  if (prec == PREC_F16) return launch_wg_prec<PREC_F16>(g, tuning, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_wg_prec<PREC_BF16>(g, tuning, (hipStream_t)stream);
  return ARK_ERR_ARG;
}
