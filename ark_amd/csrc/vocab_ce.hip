// Tied vocabulary projection fused with the token cross-entropy for LARGE vocabularies: the [B*L, V]
// logits (1.7 GB fp32 per step at wd-movies B=256, 2.5 GB at wd-articles B=16) and their gradient
// never exist in memory.  Replaces, for V in the tens of thousands, the sequence
//   logits = y W_tok^T + b_out        (kgvae/model/models.py:128-134,142)
//   F.cross_entropy(ignore_index=PAD) (kgvae/experiments/ablation_study.py:65-69)
//   dY = dlogits W_tok, dW_tok += dlogits^T y, db_out = colsum(dlogits)   (autograd of the above)
// with two launches, both of the flash-attention kind (softmax over the vocabulary axis):
//
//   ark_vocab_ce_fwd  one workgroup per 64 rows; W_tok streams through an LDS-DMA ring in stages of 64 tokens (D = 512) to
//                     256 tokens (D <= 128: several 64-token blocks per stage, one barrier and one running-maximum update
//                     per stage).
//                     S^T = W_tile y^T on the matrix cores (TRANSPOSED on purpose: the accumulator then has the
//                     row index on the lane and the tokens in its registers, so the online max / sum-exp are
//                     per-lane scalars and exp(S^T - m) IS the B fragment of the second product -- no LDS
//                     round trip), U^T += W_tile^T P^T with the W fragments re-read transposed from the same
//                     LDS image (ds_read_b64_tr_b16).  One sweep gives lse, the row loss AND
//                     dY = (U / l - W_tok[target]) / count.
//   ark_vocab_ce_dw   one workgroup per 64 tokens; the rows stream through the ring; S = y W_tile^T is
//                     recomputed (W fragments in registers), G = exp(S + b - lse) - onehot becomes the B
//                     fragment of dW^T += y^T G (y fragments transposed from the ring image), db_out = colsum(G).
//
// 16x16x32 MFMA throughout; the k index of the second product of each kernel is PERMUTED (slot (q, j) of a
// lane's 8-element fragment = accumulator register j of token/row tile j>>2), which is legal because both
// operands use the same permutation.  Both products use the forward operand type (fp16 in mixed precision):
// probabilities are O(1), and the 1/count factor is applied in fp32 at the end.
// 4 GEMM-equivalents instead of 3, ~6.8 GB (wd-movies) / ~10 GB (wd-articles) of HBM traffic less per step.
#include "dma_core.h"
#include "../../include/ark_amd.h"

namespace ark {

typedef short vs16x4 __attribute__((ext_vector_type(4)));
typedef short vs16x8 __attribute__((ext_vector_type(8)));

struct VocabCeArgs {
  const void* Y16;        // [R, D] row-major, forward type: top GRU layer outputs, time-major rows (t, b)
  const void* W16;        // [V, D] row-major, forward type: tied vocabulary weight shadow
  const float* bias;      // [V]
  const int64_t* seq;     // [B, ld_seq]: target of row (t, b) is seq[b, t + 1]
  const float* hyper;
  float* row_loss;        // [R]
  float* lse;             // [R] natural-log sum-exp of the logits (input of the dW kernel)
  float* dY_t;            // [R, D] tile-native fp32 (fwd kernel output, nullable)
  float* dW;              // [V, D] += (dw kernel)
  float* db;              // [V] +=
  long ld_seq;
  int R, B, V;
  float* part_u;          // vocabulary-split forward: [NV][R*D] tile-native unnormalised U, relative to part_s's maximum
  float* part_s;          // [NV][R][4] = {maximum (log2 domain), sum, target logit (log2 domain, 0 if not in this split), -}
};

constexpr int kVcImg = 64 * 128;   // one k-image: 64 rows x 128 B (64 16-bit elements of the reduction index)
// Swizzle of an image row's eight 16-byte chunks: physical chunk = logical chunk ^ vc_key(row).  Bits 1-2 of the row in bits
// 1-2 of the key serve BOTH kinds of fragment read: the 16 lanes of a ds_read_b128 cycle (rows {0-3, 12-15} with chunk ch and
// rows {4-11} with chunk ch + 1) land on 64 different banks, and so do the 32 lanes of a ds_read_b64_tr_b16 cycle (rows
// r .. r + 7, four lanes per row on a 32-byte chunk pair).  Round 5; with the key (row >> 1) & 7 of the other kernels' images
// rows r and r + 2 met on the same chunk pair in the transposed reads: SQ_LDS_BANK_CONFLICT = 30 % of the LDS cycles.
__device__ __forceinline__ int vc_key(int row) { return row & 6; }
constexpr int kVcAux = 256;        // per-wave side data of a stage: 64 dwords

// issue one ring stage: DCH k-images of 64 rows = 8*DCH one-KB pieces, dealt round-robin over the NW waves (every wave
// issues PPW = ceil(8*DCH / NW) instructions so that the counted vmcnt waits are the same for all; a wave whose last
// piece does not exist repeats piece 0's bytes into piece 0 -- same data, harmless)
// `lane` goes through an empty asm first: everything derived from it is then re-computed per call (a handful of VALU
// instructions per piece).  Left to itself the compiler keeps up to sixteen loop-invariant per-piece offsets, spills them in
// the register-bound wide kernels, and each reload waits -- vmcnt(0) -- for the LDS-DMA traffic of the stages in flight.
template <int DCH, int NW, class T, class RM>
__device__ __forceinline__ void vc_issue_stage(const T* src, int D, RM rowmap, char* slot, int wave, int lane) {
  constexpr int NP = 8 * DCH, PPW = (NP + NW - 1) / NW;
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    int piece = wave + NW * i;
    if (piece >= NP) piece = 0;
    const int j = piece >> 3, pr = piece & 7;
    const int row = 8 * pr + (lane >> 3);
    const T* g = src + (long)rowmap(row) * D + 64 * j + 8 * ((lane & 7) ^ vc_key(row));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
  }
}

// The same pieces for a block whose 64 rows all exist (every block but the last of the launch), from per-lane byte offsets
// computed ONCE (models up to D = 256 have the registers): one instruction per piece -- the row clamp, the swizzle and the
// 64-bit address arithmetic of vc_issue_stage were ~13 vector instructions per piece, 120 per wave and stage at D = 128.
template <int DCH, int NW>
struct VcPieces {
  static constexpr int NP = 8 * DCH, PPW = (NP + NW - 1) / NW;
  unsigned off[PPW];
  __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int piece = wave + NW * i;
      if (piece >= NP) piece = 0;
      const int j = piece >> 3, pr = piece & 7;
      const int row = 8 * pr + (lane >> 3);
      off[i] = 2u * (unsigned)(row * (64 * DCH) + 64 * j + 8 * ((lane & 7) ^ vc_key(row)));
    }
  }
  // block = address of the block's first row (wave-uniform)
  __device__ __forceinline__ void issue(const char* block, char* slot, int wave) const {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int piece = wave + NW * i;
      if (piece >= NP) piece = 0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(block + off[i]),
                                       (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
    }
  }
};
template <int DCH> constexpr bool vc_fast_issue() { return DCH <= 4; }

// Both products of both kernels read their LDS operand through inline asm (lds_b128 / lds_tr16, dma_core.h), G fragments
// ahead of the matrix instructions that consume them.  (Round 5.  The builtin transposed read drained the LDS-DMA ring at
// every stage -- the waitcnt pass put vmcnt(0) in front of it -- and the compiler scheduled the plain reads of the first
// product as read, lgkmcnt(0), one MFMA: every fragment paid its full LDS latency.)
//
// First product: S[b][t] += F(b, t, ks) x reg[ks] over the KSTEPS 32-wide steps of the model dimension; F = the plain
// fragment (16 image rows `row0 + 16 t`, chunk 4 (ks & 1) + q of image ks >> 1) of block b.  a_even / a_odd: the lane's
// byte address for even / odd ks in block 0, image 0, t = 0 (the swizzle folds the odd step's chunk bit into bit 6).
template <class PT, int KSTEPS, int TM, int BLK, int G>
__device__ __forceinline__ void vc_first_product(unsigned a_even, unsigned a_odd, const typename PT::h8 (&reg)[KSTEPS], f32x4 (&S)[TM][2]) {
  using h8 = typename PT::h8;
  f32x4 f[2][G];
  lds_pipeline<KSTEPS * TM * 2, G, 1>(
      [&](auto i, auto set, auto slot) {   // item i: ks = i / (2 TM), block (i / 2) % TM, t = i & 1
        constexpr int ks = i / (2 * TM), b = (i / 2) % TM, t = i & 1;
        f[set][slot] = lds_b128<b * BLK + (ks >> 1) * kVcImg + t * 2048>((ks & 1) ? a_odd : a_even);
      },
      [&](auto i, auto set, auto slot) {
        constexpr int ks = i / (2 * TM), b = (i / 2) % TM, t = i & 1;
        tr_tie(f[set][slot]);
        S[b][t] = PT::mfma(__builtin_bit_cast(h8, f[set][slot]), reg[ks], S[b][t]);
      });
}
// Second product: acc[dt] += sum over blocks b of F(b, dt) x bf[b] for the DT 16-wide tiles of the model dimension; F = the
// TRANSPOSED fragment: element j of lane (i16 = lane&15, q = lane>>4) = image[row(q, j)][16*dt + i16] with
// row(q, j) = rbase + 16*(j>>2) + 4q + (j&3) (two ds_read_b64_tr_b16, 16 image rows = 2048 B apart: same swizzle key).
// a4[m]: the lane's byte address for tiles with dt & 3 == m in block 0, image 0.
template <class PT, int DT, int NB, int BLK, int G>
__device__ __forceinline__ void vc_second_product(const unsigned (&a4)[4], const typename PT::h8 (&bf)[NB], f32x4 (&acc)[DT]) {
  using h8 = typename PT::h8;
  tr16x4 f[2][G][2];
  lds_pipeline<DT * NB, G, 2>(
      [&](auto i, auto set, auto slot) {   // item i = (block i / DT, tile i % DT): consecutive products, different accumulators
        constexpr int b = i / DT, dt = i % DT;
        f[set][slot][0] = lds_tr16<b * BLK + (dt >> 2) * kVcImg>(a4[dt & 3]);
        f[set][slot][1] = lds_tr16<b * BLK + (dt >> 2) * kVcImg + 2048>(a4[dt & 3]);
      },
      [&](auto i, auto set, auto slot) {
        constexpr int b = i / DT, dt = i % DT;
        tr_tie(f[set][slot][0]);
        tr_tie(f[set][slot][1]);
        acc[dt] = PT::mfma(tr_join<h8>(f[set][slot][0], f[set][slot][1]), bf[b], acc[dt]);
      });
}
// the lane's fragment addresses relative to a stage image (both kernels: `half` = the 32-row half of a block it works on);
// at(b0): the addresses inside the image at LDS byte address b0
struct VcFragAddr {
  unsigned even, odd, tr[4];
  __device__ __forceinline__ VcFragAddr at(unsigned b0) const {
    VcFragAddr r;
    r.even = even + b0;
    r.odd = odd + b0;
#pragma unroll
    for (int m = 0; m < 4; ++m) r.tr[m] = tr[m] + b0;
    return r;
  }
};
__device__ __forceinline__ VcFragAddr vc_frag_addr(int half, int lane) {
  VcFragAddr r;
  const int c = lane & 15, q = lane >> 4;
  const int row = half * 32 + c;   // first product: image row of fragment t = 0
  r.even = row * 128 + ((q ^ vc_key(row)) << 4);
  r.odd = row * 128 + (((q ^ vc_key(row)) ^ 4) << 4);
  const int rowt = half * 32 + 4 * q + (c >> 2);   // second product
  const int key = vc_key(rowt), c0 = (c & 3) >> 1;
#pragma unroll
  for (int m = 0; m < 4; ++m) r.tr[m] = rowt * 128 + (c & 1) * 8 + (((2 * m + c0) ^ key) << 4);
  return r;
}
// fragments requested ahead: as many as the registers allow (D = 512 keeps 192 of a wave's 256 in accumulators and its row;
// workgroups of more than 8 waves put three waves on a SIMD: 168 registers each)
template <int DT, int NW> constexpr int vc_frag_group() { return DT >= 32 ? 2 : NW > 8 ? (DT >= 16 ? 2 : 4) : (DT >= 16 ? 4 : 8); }

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// ONE barrier per stage (round 4): the barrier that publishes stage s also says every wave has finished reading stage s - 1,
// whose slot is refilled at once (NSLOT - 1 stages in flight) -- no second barrier, no LDS drain in front of it.
// (Measured and not kept: the refill's pieces spread over the first product's k-steps instead of issued as one block behind
// the barrier -- wd-articles forward 1.97 -> 2.02 ms, weight gradient 1.75 -> 1.91 ms: a piece between MFMAs stalls its wave
// with LDS reads queued behind it.)

// A stage is TM blocks of 64 rows of the streamed operand (64 tokens of W_tok forward, 64 rows of y in the weight gradient),
// each DCH k-images.  Round 5: narrow models take SEVERAL blocks per stage -- at D = 128 a 64-row stage is 16 products per
// wave between two barriers, behind a fixed cost per stage (barrier, the cross-lane maximum, waits for fragments that have
// nothing to overlap with) several times as long; with TM blocks a wave has 2 TM independent accumulator chains in the
// first product and one maximum / one barrier per 64 TM rows.  D = 512 keeps TM = 1 (registers: 192 of a wave's 256 hold
// accumulators and its row).
constexpr int vc_tm_cap(int dch) { return dch <= 2 ? 4 : dch <= 4 ? 2 : 1; }
// ring depth by stage size (8 KB per image): small stages afford four slots -- three in flight per workgroup
constexpr int vc_slots(int imgs) { return imgs <= 2 ? 4 : imgs <= 4 ? 3 : 2; }
constexpr int vc_ring_bytes(int dch, int nw, int aux_per_wave, int tm) {
  return vc_slots(dch * tm) * (tm * dch * kVcImg + nw * tm * aux_per_wave);
}
// blocks per stage: the most the cap and the 160 KB of LDS allow for NW waves with `aux_per_wave` bytes of side data per block
constexpr int vc_tm(int dch, int nw, int aux_per_wave) {
  int tm = vc_tm_cap(dch);
  while (tm > 1 && vc_ring_bytes(dch, nw, aux_per_wave, tm) > 160 * 1024) tm /= 2;
  return tm;
}

// wait until all but the youngest `k` stages (LPS LDS-DMA instructions each) of this wave have landed
template <int LPS>
__device__ __forceinline__ void vc_wait_stages(int k) {
  switch (k) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory"); break;
  }
}

// ---------------------------------------------------------------------------------------------------------
// RG groups of 16 rows per workgroup (2*RG waves: RG row groups x 2 token halves), chosen on the host so that the row
// tiles fill the 256 CUs evenly (64-row tiles gave 280 workgroups at wd-movies B=256: two rounds for 1.09 rounds of work)
template <int PREC, int DCH, int RG, bool WITH_DY>
__global__ __launch_bounds__(128 * RG) void vocab_ce_fwd_kernel(VocabCeArgs p) {
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  constexpr int D = 64 * DCH, KSTEPS = D / 32, DT = D / 16;
  constexpr int NW = 2 * RG;
  constexpr int TM = vc_tm(DCH, NW, kVcAux), TS = 64 * TM;   // blocks / tokens per stage
  constexpr int BLOCK = DCH * kVcImg, STAGE = TM * BLOCK, SLOT = STAGE + NW * TM * kVcAux;
  constexpr int LPS = TM * ((8 * DCH + NW - 1) / NW + 1);
  constexpr int NSLOT = vc_slots(DCH * TM);
  static_assert((NSLOT - 1) * LPS <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave % RG, vh = wave / RG;
  const int c = lane & 15, q = lane >> 4;
  const int R = p.R, V = p.V;
  // (row block, vocabulary split) of this workgroup.  Round 5: with NV splits dividing the 8 XCDs, a split's workgroups are
  // dealt to 8 / NV XCDs only (workgroup ids round-robin over the XCDs: id % 8), so an XCD's L2 streams 1 / NV of W_tok per
  // launch instead of all of it -- wd-articles: 4 chunk launches x 8 XCDs x 62 MB = 2.1 GB of fabric reads per step against
  // 62 MB of weights (placement is a speed matter only: any mapping that is a bijection is correct)
  int rbk = blockIdx.x, vs = blockIdx.y;
  const int NV = gridDim.y;
  if (NV > 1 && 8 % NV == 0 && (gridDim.x * NV) % 8 == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, k = 8 / NV, xcd = lin & 7;
    vs = xcd / k;
    rbk = (lin >> 3) * k + xcd % k;
  }
  const int r = rbk * (16 * RG) + rg * 16 + c;
  const int rc = min(r, R - 1);
  const h_t* Y = reinterpret_cast<const h_t*>(p.Y16);
  const h_t* W = reinterpret_cast<const h_t*>(p.W16);
  // this lane's row as the B operand of S^T = W y^T: y[r][32s + 8q .. +7]
  h8 yf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) yf[s] = *reinterpret_cast<const h8*>(Y + (long)rc * D + 32 * s + 8 * q);
  const int t = rc / p.B, b = rc % p.B;
  const long tgt = p.seq[(long)b * p.ld_seq + t + 1];
  const bool live = r < R && tgt != ARK_TOK_PAD;
  // from here on only LDS-DMA is outstanding (counted waits); the loaded registers are used here so that the compiler's own
  // wait for them is not an s_waitcnt vmcnt(0) at their first use inside the loop
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(yf[s]));
  { long t2 = tgt; asm volatile("" : "+v"(t2)); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // the workgroups of blockIdx.y = vs sweep the tiles [s0, s1) of the vocabulary (gridDim.y > 1: few row blocks, e.g. 160 at
  // wd-articles B = 16 -- the splits fill the other CUs; partial results meet in vocab_ce_combine_kernel)
  const int nsteps = (V + TS - 1) / TS;
  const int s0 = (int)((long)nsteps * vs / NV), s1 = (int)((long)nsteps * (vs + 1) / NV);
  VcPieces<DCH, NW> pieces;
  if constexpr (vc_fast_issue<DCH>()) pieces.init(wave, lane);
  const unsigned lane4 = 4u * lane;
  auto issue = [&](int s) {
    char* slot = smem + ((s - s0) % NSLOT) * SLOT;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int v0 = s * TS + 64 * b;
      char* aux = slot + STAGE + (wave * TM + b) * kVcAux;
      if (vc_fast_issue<DCH>() && v0 + 64 <= V) {
        pieces.issue(reinterpret_cast<const char*>(W + (long)v0 * D), slot + b * BLOCK, wave);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.bias + v0) + lane4),
                                         (__attribute__((address_space(3))) void*)aux, 4, 0, 0);
      } else {
        vc_issue_stage<DCH, NW>(W, D, [=](int row) { return min(v0 + row, V - 1); }, slot + b * BLOCK, wave, lane);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + min(v0 + ln, V - 1)),
                                         (__attribute__((address_space(3))) void*)aux, 4, 0, 0);
      }
    }
  };
  for (int s = s0; s < s0 + NSLOT - 1 && s < s1; ++s) issue(s);

  float m2 = -INFINITY, lsum = 0.f, picked = 0.f;   // running max (log2 domain), this LANE's partial sum, target logit
  f32x4 U[WITH_DY ? DT : 1];
  if constexpr (WITH_DY) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) U[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const VcFragAddr fo = vc_frag_addr(vh, lane);
  // the bias of this lane's tokens: 4 consecutive floats per 16-token tile, at (vh * 32 + 16 * tile + 4 q) of the block's 64
  const unsigned bias_off = STAGE + wave * TM * kVcAux + (vh * 32 + 4 * q) * 4;
  for (int s = s0; s < s1; ++s) {
    vc_wait_stages<LPS>(min(NSLOT - 2, s1 - 1 - s));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + NSLOT - 1 < s1) issue(s + NSLOT - 1);   // into the slot of stage s - 1: every wave is past its reads of it
    const unsigned sb = lds_addr(smem + ((s - s0) % NSLOT) * SLOT);
    const VcFragAddr fa = fo.at(sb);
    // this wave's tokens of the stage: the half vh (32 tokens) of each of its TM blocks
    // (requested in front of the first product's fragments, which wait for everything at their end -- where three waves share
    //  a SIMD and its registers, behind it)
    f32x4 bq[TM][2];
    auto request_bias = [&]() {
      static_for<TM>([&](auto b) {
        bq[b][0] = lds_b128<b * kVcAux>(sb + bias_off);
        bq[b][1] = lds_b128<b * kVcAux + 64>(sb + bias_off);
      });
    };
    constexpr bool kBiasEarly = NW <= 8;
    if constexpr (kBiasEarly) request_bias();
    f32x4 S[TM][2];
#pragma unroll
    for (int b = 0; b < TM; ++b) S[b][0] = S[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    vc_first_product<PT, KSTEPS, TM, BLOCK, vc_frag_group<DT, NW>()>(fa.even, fa.odd, yf, S);
    if constexpr (!kBiasEarly) { request_bias(); tr_wait_cnt<0>(); }
    constexpr int NE = 8 * TM;   // logits of this lane's row in the stage; element e = (block e >> 3, j = e & 7)
    auto tok_of = [&](int e) { return 64 * (e >> 3) + vh * 32 + 16 * ((e & 7) >> 2) + 4 * q + (e & 3); };   // token inside the stage
    float sv[NE];   // logit (natural units) until the maximum is known, then the probability relative to it
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      tr_tie(bq[b][0]);
      tr_tie(bq[b][1]);
      const f32x4 t0 = S[b][0] + bq[b][0], t1 = S[b][1] + bq[b][1];
#pragma unroll
      for (int i = 0; i < 4; ++i) { sv[8 * b + i] = t0[i]; sv[8 * b + 4 + i] = t1[i]; }
    }
    // two rare cases, each behind a wave-uniform branch so that the common step pays nothing per element:
    // the last stage reaches past V; some row's target token lives in this stage
    if (s == nsteps - 1) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (s * TS + tok_of(e) >= V) sv[e] = -INFINITY;
    }
    if (__any((int)(tgt / TS) == s)) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if ((long)(s * TS + tok_of(e)) == tgt) picked = sv[e] * kLog2e;   // (log2 domain)
    }
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < NE; ++e) mt = fmaxf(mt, sv[e]);
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m2, mt * kLog2e);
    const float ref = (mn == -INFINITY) ? 0.f : mn;     // (a stage half beyond V has nothing to add)
    const float scale = __builtin_amdgcn_exp2f(m2 - ref);   // m2 = -inf -> 0
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) { sv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], kLog2e, -ref)); ps += sv[e]; }
    lsum = lsum * scale + ps;
    if constexpr (WITH_DY) {
      if (__any(scale != 1.0f)) {   // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) U[dt] *= scale;
      }
      h8 pf[TM];   // (probabilities, <= 1: the plain conversion -- no saturation needed)
#pragma unroll
      for (int e = 0; e < NE; ++e) pf[e >> 3][e & 7] = (h_t)sv[e];
      vc_second_product<PT, DT, TM, BLOCK, vc_frag_group<DT, NW>()>(fa.tr, pf, U);
    }
    m2 = mn;
  }
  // this lane's row sum over the 4 token quarters held by the lanes c, c+16, c+32, c+48
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  picked += __shfl_xor(picked, 16, 64);   // exactly one lane of one wave saw the target (others hold 0)
  picked += __shfl_xor(picked, 32, 64);
  // combine the two token halves (waves w and w+4 share their rows) through the ring space
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* xs = reinterpret_cast<float*>(smem);                  // [RG row groups][DT][64 lanes] f32x4 + stats
  constexpr int UW = (WITH_DY ? DT : 0) * 64 * 4;              // floats per row group
  float* st = xs + RG * UW;                                    // [RG][3][64]
  if (vh == 1) {
    if constexpr (WITH_DY) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(xs + rg * UW + (dt * 64 + lane) * 4) = U[dt];
    }
    st[(rg * 3 + 0) * 64 + lane] = m2;
    st[(rg * 3 + 1) * 64 + lane] = lsum;
    st[(rg * 3 + 2) * 64 + lane] = picked;
  }
  __syncthreads();
  if (vh == 1) return;
  const float mb = st[(rg * 3 + 0) * 64 + lane], lb = st[(rg * 3 + 1) * 64 + lane], pb = st[(rg * 3 + 2) * 64 + lane];
  const float mm = fmaxf(m2, mb);
  const float fa = __builtin_amdgcn_exp2f(m2 - mm), fb = (mb == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mb - mm);
  const float l = lsum * fa + lb * fb;
  const float lse = (mm + __builtin_amdgcn_logf(l)) * kLn2;   // v_log_f32 is log2
  const float tl = (picked + pb) * kLn2;   // the target logit was picked in the log2 domain
  if constexpr (WITH_DY) {
    if (NV > 1) {   // partial results of this vocabulary split, all relative to mm
      if (r >= R) return;
      if (q == 0) {
        float* ps = p.part_s + ((long)vs * R + r) * 4;
        ps[0] = mm;
        ps[1] = l;
        ps[2] = picked + pb;
      }
      float* pu = p.part_u + (long)vs * R * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const f32x4 ub = *reinterpret_cast<const f32x4*>(xs + rg * UW + (dt * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) pu[tile_native_off(r, 16 * dt + 4 * q + i, D)] = U[dt][i] * fa + ub[i] * fb;
      }
      return;
    }
  }
  if (q == 0 && r < R) {
    p.row_loss[r] = live ? (lse - tl) : 0.f;
    p.lse[r] = lse;
  }
  if constexpr (WITH_DY) {
    if (r >= R) return;
    const float sc = live ? p.hyper[ARK_HP_CE_INV_COUNT] : 0.f;
    const float il = 1.0f / l;
    const h_t* wt = W + (live ? tgt : 0) * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 ub = *reinterpret_cast<const f32x4*>(xs + rg * UW + (dt * 64 + lane) * 4);
      typedef h_t h4 __attribute__((ext_vector_type(4)));
      const h4 w4 = *reinterpret_cast<const h4*>(wt + 16 * dt + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g = sc * ((U[dt][i] * fa + ub[i] * fb) * il - (float)w4[i]);
        p.dY_t[tile_native_off(r, 16 * dt + 4 * q + i, D)] = g;
      }
    }
  }
}

// vocabulary-split forward, second launch: one thread per quad (4 consecutive rows x one column) of the tile-native dY;
// a row's splits are merged as in a split-K softmax: M = max M_i, l = sum l_i 2^(M_i - M), U = sum U_i 2^(M_i - M)
template <int PREC>
__global__ __launch_bounds__(256) void vocab_ce_combine_kernel(VocabCeArgs p, int D, int NV) {
  using h_t = typename PrecTraits<PREC>::h_t;
  const long quad = (long)blockIdx.x * 256 + threadIdx.x;
  const long nquad = (long)p.R * D / 4;
  if (quad >= nquad) return;
  const long off = quad * 4;
  const long tile = off >> 8;
  const int within = (int)(off & 255);
  const int row0 = (int)(tile / (D >> 4)) * 16 + ((within >> 6) & 3) * 4;
  const int col = (int)(tile % (D >> 4)) * 16 + ((within >> 2) & 15);
  const int R = p.R;
  const h_t* W = reinterpret_cast<const h_t*>(p.W16);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  float out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = row0 + i;   // (R % 16 == 0: host check)
    float M = -INFINITY;
    for (int v = 0; v < NV; ++v) M = fmaxf(M, p.part_s[((long)v * R + r) * 4]);
    float l = 0.f, pick = 0.f, u = 0.f;
    for (int v = 0; v < NV; ++v) {
      const float* ps = p.part_s + ((long)v * R + r) * 4;
      const float f = __builtin_amdgcn_exp2f(ps[0] - M);
      l += ps[1] * f;
      pick += ps[2];
      u += p.part_u[(long)v * R * D + off + i] * f;
    }
    const int t = r / p.B, b = r % p.B;
    const long tgt = p.seq[(long)b * p.ld_seq + t + 1];
    const bool live = tgt != ARK_TOK_PAD;
    const float lse = (M + __builtin_amdgcn_logf(l)) * kLn2;
    if (col == 0) {
      p.row_loss[r] = live ? (lse - pick * kLn2) : 0.f;
      p.lse[r] = lse;
    }
    const float sc = live ? p.hyper[ARK_HP_CE_INV_COUNT] : 0.f;
    out[i] = sc * (u / l - (float)W[(live ? tgt : 0) * D + col]);
  }
  (void)acc;
  *reinterpret_cast<f32x4*>(p.dY_t + off) = f32x4{out[0], out[1], out[2], out[3]};
}

// ---------------------------------------------------------------------------------------------------------
// VG groups of 16 tokens per workgroup (2*VG waves: VG token groups x 2 row halves)
template <int PREC, int DCH, int VG>
__global__ __launch_bounds__(128 * VG) void vocab_ce_dw_kernel(VocabCeArgs p) {
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  constexpr int D = 64 * DCH, KSTEPS = D / 32, DT = D / 16;
  constexpr int NW = 2 * VG;
  constexpr int TM = vc_tm(DCH, NW, 2 * kVcAux), RS = 64 * TM;   // blocks / rows per stage
  constexpr int BLOCK = DCH * kVcImg, STAGE = TM * BLOCK, SLOT = STAGE + NW * TM * 2 * kVcAux;   // per wave and block: lse[64] + target[64]
  constexpr int LPS = TM * ((8 * DCH + NW - 1) / NW + 2);
  constexpr int NSLOT = vc_slots(DCH * TM);
  static_assert((NSLOT - 1) * LPS <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vg = wave % VG, rh = wave / VG;
  const int c = lane & 15, q = lane >> 4;
  const int R = p.R, V = p.V, B = p.B;
  const int v = blockIdx.x * (16 * VG) + vg * 16 + c;
  const int vc = min(v, V - 1);
  const h_t* Y = reinterpret_cast<const h_t*>(p.Y16);
  const h_t* W = reinterpret_cast<const h_t*>(p.W16);
  // this lane's token as the B operand of S = y W^T: W[v][32s + 8q .. +7]
  h8 wf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) wf[s] = *reinterpret_cast<const h8*>(W + (long)vc * D + 32 * s + 8 * q);
  const float bv = (v < V) ? p.bias[vc] * kLog2e : -INFINITY;   // tokens past V (last tile): exp2(-inf) = 0, no target matches
  // the token the one-hot term compares targets with: PAD rows carry target PAD and no gradient, so the lane that owns the PAD
  // token must never match (tokens past V neither)
  const int vmatch = (v < V && v != ARK_TOK_PAD) ? v : -2;
  // From here on only LDS-DMA is outstanding (counted waits).  The registers loaded above are USED here, so the compiler's
  // own wait for them lands here too: left to their first use inside the loop it would be an s_waitcnt vmcnt(0) per stage,
  // which drains the ring (tools/isa_dma_waits.py).
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(wf[s]));
  { float t = bv; asm volatile("" : "+v"(t)); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int nsteps = (R + RS - 1) / RS;
  VcPieces<DCH, NW> pieces;
  if constexpr (vc_fast_issue<DCH>()) pieces.init(wave, lane);
  const unsigned lane4 = 4u * lane;
  // row -> (t, b) = (row / B, row % B) by a multiply: exact while row * B < 2^32
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)B - 1) / (unsigned)B);
  const bool fast_ok = vc_fast_issue<DCH>() && (unsigned long long)R * (unsigned)B < 0x100000000ull && (unsigned long long)B * p.ld_seq < 0x20000000ull;
  auto issue = [&](int s) {
    char* slot = smem + (s % NSLOT) * SLOT;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int r0 = s * RS + 64 * b;
      // per-wave side data of the block's 64 rows: lse and the target token (low dword of the int64)
      char* aux = slot + STAGE + (wave * TM + b) * 2 * kVcAux;
      if (fast_ok && r0 + 64 <= R) {
        pieces.issue(reinterpret_cast<const char*>(Y + (long)r0 * D), slot + b * BLOCK, wave);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.lse + r0) + lane4),
                                         (__attribute__((address_space(3))) void*)aux, 4, 0, 0);
        const unsigned rr = (unsigned)r0 + (lane4 >> 2);
        const unsigned t = __umulhi(rr, magic), bb = rr - t * (unsigned)B;
        const unsigned idx = bb * (unsigned)p.ld_seq + t + 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.seq) + 8ull * idx),
                                         (__attribute__((address_space(3))) void*)(aux + kVcAux), 4, 0, 0);
      } else {
        vc_issue_stage<DCH, NW>(Y, D, [=](int row) { return min(r0 + row, R - 1); }, slot + b * BLOCK, wave, lane);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int rr = min(r0 + ln, R - 1);
        const int t = rr / B, bb = rr % B;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lse + rr),
                                         (__attribute__((address_space(3))) void*)aux, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.seq + (long)bb * p.ld_seq + t + 1),
                                         (__attribute__((address_space(3))) void*)(aux + kVcAux), 4, 0, 0);
      }
    }
  };
  for (int s = 0; s < NSLOT - 1 && s < nsteps; ++s) issue(s);

  f32x4 dWt[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dWt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbv = 0.f;
  const VcFragAddr fo = vc_frag_addr(rh, lane);
  // lse / target of this lane's rows: 4 consecutive dwords per 16-row tile, at (rh * 32 + 16 * tile + 4 q) of the block's 64
  const unsigned aux_off = STAGE + wave * TM * 2 * kVcAux + (rh * 32 + 4 * q) * 4;
  constexpr bool kAuxEarly = NW <= 8;   // (three waves on a SIMD: no registers to hold the side data across the first product)
  for (int s = 0; s < nsteps; ++s) {
    vc_wait_stages<LPS>(min(NSLOT - 2, nsteps - 1 - s));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + NSLOT - 1 < nsteps) issue(s + NSLOT - 1);
    const unsigned sb = lds_addr(smem + (s % NSLOT) * SLOT);
    const VcFragAddr fa = fo.at(sb);
    // this wave's rows of the stage: the half rh (32 rows) of each of its TM blocks
    f32x4 lq[TM][2], tq[TM][2];   // lse and targets (as raw dwords) of those rows
    auto request_aux = [&]() {
      static_for<TM>([&](auto b) {
        lq[b][0] = lds_b128<b * 2 * kVcAux>(sb + aux_off);
        lq[b][1] = lds_b128<b * 2 * kVcAux + 64>(sb + aux_off);
        tq[b][0] = lds_b128<b * 2 * kVcAux + kVcAux>(sb + aux_off);
        tq[b][1] = lds_b128<b * 2 * kVcAux + kVcAux + 64>(sb + aux_off);
      });
    };
    if constexpr (kAuxEarly) request_aux();
    f32x4 S[TM][2];
#pragma unroll
    for (int b = 0; b < TM; ++b) S[b][0] = S[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    vc_first_product<PT, KSTEPS, TM, BLOCK, vc_frag_group<DT, NW>()>(fa.even, fa.odd, wf, S);
    if constexpr (!kAuxEarly) { request_aux(); tr_wait_cnt<0>(); }
    constexpr int NE = 8 * TM;   // element e = (block e >> 3, j = e & 7): row 64 (e >> 3) + rloc(e) of the stage
    auto rloc_of = [&](int e) { return rh * 32 + 16 * ((e & 7) >> 2) + 4 * q + (e & 3); };
    h8 gf[TM];
    // per row: lse_r, +inf for rows whose target is PAD (their gradient is zero) and, in the last stage only, for the
    // clamped copies of the last row (whose target is set to -1 so that no token matches it)
    float ls[NE];
    int tg[NE];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        tr_tie(lq[b][h]);
        tr_tie(tq[b][h]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 8 * b + 4 * h + i;
          const float tbits = tq[b][h][i];   // (a copy first: __builtin_bit_cast on the vector ELEMENT read element 0 every time)
          tg[e] = __builtin_bit_cast(int, tbits);
          ls[e] = tg[e] == ARK_TOK_PAD ? INFINITY : lq[b][h][i];
        }
      }
    }
    if (s == nsteps - 1) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (s * RS + 64 * (e >> 3) + rloc_of(e) >= R) { ls[e] = INFINITY; tg[e] = -1; }
    }
    // G = softmax - onehot: exp2((S - lse) log2e + b log2e); the one-hot term is rare (a row's target among this wave's 16
    // tokens) and sits behind a wave-uniform branch
    float g[NE];
    unsigned long long hit = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      g[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[e >> 3][(e & 7) >> 2][e & 3] - ls[e], kLog2e, bv));
      hit |= __ballot(tg[e] == vmatch);
    }
    if (hit) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (tg[e] == vmatch) g[e] -= 1.0f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      dbv += g[e];
      gf[e >> 3][e & 7] = (h_t)g[e];   // (|g| <= 1: the plain conversion)
    }
    vc_second_product<PT, DT, TM, BLOCK, vc_frag_group<DT, NW>()>(fa.tr, gf, dWt);
  }
  dbv += __shfl_xor(dbv, 16, 64);
  dbv += __shfl_xor(dbv, 32, 64);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* xs = reinterpret_cast<float*>(smem);   // [VG token groups][DT][64 lanes] f32x4, then [VG][64] bias partials
  constexpr int UW = DT * 64 * 4;
  float* st = xs + VG * UW;
  if (rh == 1) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(xs + vg * UW + (dt * 64 + lane) * 4) = dWt[dt];
    st[vg * 64 + lane] = dbv;
  }
  __syncthreads();
  if (rh == 1 || v >= V) return;
  const float sc = p.hyper[ARK_HP_CE_INV_COUNT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const f32x4 ub = *reinterpret_cast<const f32x4*>(xs + vg * UW + (dt * 64 + lane) * 4);
    f32x4* o = reinterpret_cast<f32x4*>(p.dW + (long)v * D + 16 * dt + 4 * q);
    *o = *o + (dWt[dt] + ub) * sc;   // (v, d) is owned by exactly one workgroup: a plain read-modify-write
  }
  if (q == 0) p.db[v] += (dbv + st[vg * 64 + lane]) * sc;
}

template <class K>
static void vc_allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int DCH, int NG>
constexpr int vc_lds_bytes(int aux_per_wave) {
  const int ring = vc_ring_bytes(DCH, 2 * NG, aux_per_wave, vc_tm(DCH, 2 * NG, aux_per_wave));
  const int comb = NG * (64 * DCH / 16) * 64 * 16 + NG * 3 * 64 * 4;
  return ring > comb ? ring : comb;
}

template <int PREC, int DCH, int RG>
static int vc_launch_fwd_rg(const VocabCeArgs& p, bool with_dy, hipStream_t st) {
  constexpr int LDS = vc_lds_bytes<DCH, RG>(kVcAux);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  const unsigned grid = (unsigned)((p.R + 16 * RG - 1) / (16 * RG));
  if (with_dy) {
    static bool once = (vc_allow_lds(vocab_ce_fwd_kernel<PREC, DCH, RG, true>, LDS), true); (void)once;
    hipLaunchKernelGGL((vocab_ce_fwd_kernel<PREC, DCH, RG, true>), dim3(grid), dim3(128 * RG), LDS, st, p);
  } else {
    static bool once = (vc_allow_lds(vocab_ce_fwd_kernel<PREC, DCH, RG, false>, LDS), true); (void)once;
    hipLaunchKernelGGL((vocab_ce_fwd_kernel<PREC, DCH, RG, false>), dim3(grid), dim3(128 * RG), LDS, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

// groups of 16 (rows / tokens) per workgroup so that the tiles come out as close below a multiple of 256 CUs as possible
static int vc_groups(int n16, int max_groups) {
  int best = 4;
  double best_eff = 0.0;
  for (int g : {6, 5, 4, 3}) {   // (ties go to the larger tile: fewer sweeps over the streamed operand)
    if (g > max_groups) continue;   // (wide models need the whole register file of a 512-thread workgroup)
    const int wgs = (n16 + g - 1) / g;
    const int rounds = (wgs + 255) / 256;
    const double eff = (double)n16 / ((double)rounds * 256 * g);   // useful fraction of the CU-rounds spent
    if (eff > best_eff + 0.02) { best_eff = eff; best = g; }
  }
  return best;
}

// vocabulary splits of the forward with dY: the count that minimises rounds-of-workgroups x work-per-workgroup on `cus` CUs
// (1 where the 64-row blocks alone fill them); each split keeps at least 32 tiles.  cus = 256, or the CUs a persistent
// sweep running beside this launch leaves free (cu_budget: one 512-thread workgroup owns a CU's whole register file, so
// a grid capped there can never keep a sweep workgroup off the chip).
static int vc_splits(int R, int V, int cus) {
  if (cus <= 0 || cus > 256) cus = 256;
  const int rb = (R + 63) / 64, nsteps = (V + 63) / 64;
  if (rb > cus * 3 / 4) return 1;   // (the row tiles fill the chip; vc_groups balances the last round)
  int best = 1;
  double best_t = (double)((rb + cus - 1) / cus);
  for (int nv = 2; nv <= 16 && nsteps / nv >= 32; ++nv) {
    const double t = (double)((rb * nv + cus - 1) / cus) / nv + 0.02 * nv;   // (+ ramp / combine cost per split)
    if (t < best_t - 1e-9) { best_t = t; best = nv; }
  }
  return best;
}
extern "C" int ark_vocab_ce_fwd_splits(int R, int V, int D, int cu_budget) {
  return (D == 64 || D == 128 || D == 256 || D == 512) && R % 16 == 0 ? vc_splits(R, V, cu_budget) : 1;
}

template <int PREC, int DCH>
static int vc_launch_fwd(const VocabCeArgs& p, bool with_dy, hipStream_t st, int nv = 1) {
  if (with_dy && p.part_u) {   // few row blocks: the vocabulary is split over workgroups, 64-row tiles, one merging launch
    constexpr int LDS = vc_lds_bytes<DCH, 4>(kVcAux);
    static bool once = (vc_allow_lds(vocab_ce_fwd_kernel<PREC, DCH, 4, true>, LDS), true); (void)once;
    hipLaunchKernelGGL((vocab_ce_fwd_kernel<PREC, DCH, 4, true>), dim3((unsigned)((p.R + 63) / 64), (unsigned)nv), dim3(512), LDS, st, p);
    ARK_LAUNCH_CHECK();
    const long nquad = (long)p.R * (64 * DCH) / 4;
    hipLaunchKernelGGL((vocab_ce_combine_kernel<PREC>), dim3((unsigned)((nquad + 255) / 256)), dim3(256), 0, st, p, 64 * DCH, nv);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  // (measured: D = 512 is register-bound -- 256 VGPRs with spills -- and slower with anything but 4 groups: 3.9 -> 7.4 ms at
  //  wd-articles; D = 128 gains 10 % from 5 row groups at wd-movies, 280 -> 224 workgroups)
  switch (DCH >= 8 ? 4 : vc_groups((p.R + 15) / 16, DCH <= 2 ? 6 : 5)) {
    case 3: return vc_launch_fwd_rg<PREC, DCH, 3>(p, with_dy, st);
    case 5: if constexpr (DCH <= 4) return vc_launch_fwd_rg<PREC, DCH, 5>(p, with_dy, st); else break;
    case 6: if constexpr (DCH <= 2) return vc_launch_fwd_rg<PREC, DCH, 6>(p, with_dy, st); else break;
    default: break;
  }
  return vc_launch_fwd_rg<PREC, DCH, 4>(p, with_dy, st);
}

template <int PREC, int DCH, int VG>
static int vc_launch_dw_vg(const VocabCeArgs& p, hipStream_t st) {
  constexpr int LDS = vc_lds_bytes<DCH, VG>(2 * kVcAux);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool once = (vc_allow_lds(vocab_ce_dw_kernel<PREC, DCH, VG>, LDS), true); (void)once;
  hipLaunchKernelGGL((vocab_ce_dw_kernel<PREC, DCH, VG>), dim3((unsigned)((p.V + 16 * VG - 1) / (16 * VG))), dim3(128 * VG), LDS, st, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC, int DCH>
static int vc_launch_dw(const VocabCeArgs& p, hipStream_t st) {
  // (wd-movies, V = 24 101, standalone: 547 / 551 / 924 / 507 us at 3 / 4 / 5 / 6 token groups: one full round of 252 workgroups wins)
  // (D = 256 with 10 waves: 168 registers per wave do not hold two blocks per stage)
  switch (DCH >= 8 ? 4 : vc_groups((p.V + 15) / 16, DCH <= 2 ? 6 : 4)) {
    case 3: return vc_launch_dw_vg<PREC, DCH, 3>(p, st);
    case 5: if constexpr (DCH <= 2) return vc_launch_dw_vg<PREC, DCH, 5>(p, st); else break;
    case 6: if constexpr (DCH <= 2) return vc_launch_dw_vg<PREC, DCH, 6>(p, st); else break;
    default: break;
  }
  return vc_launch_dw_vg<PREC, DCH, 4>(p, st);
}

static int vc_check(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq, const float* hyper, int B,
                    int L, int V, int D) {
  if (!Y16 || !W16 || !bias || !seq || !hyper || B <= 0 || L <= 0 || V <= 0) return ARK_ERR_ARG;
  if (prec != PREC_F16 && prec != PREC_BF16) return ARK_ERR_ARG;
  if (D != 64 && D != 128 && D != 256 && D != 512) return ARK_ERR_SHAPE;
  if (((uintptr_t)Y16 | (uintptr_t)W16) & 15) return ARK_ERR_ALIGN;
  return 0;
}

}  // namespace ark

#define ARK_VC_DISPATCH(FN, ...)                                             \
  do {                                                                       \
    if (prec == PREC_F16) {                                                  \
      if (D == 64) return FN<PREC_F16, 1>(__VA_ARGS__);                      \
      if (D == 128) return FN<PREC_F16, 2>(__VA_ARGS__);                     \
      if (D == 256) return FN<PREC_F16, 4>(__VA_ARGS__);                     \
      return FN<PREC_F16, 8>(__VA_ARGS__);                                   \
    }                                                                        \
    if (D == 64) return FN<PREC_BF16, 1>(__VA_ARGS__);                       \
    if (D == 128) return FN<PREC_BF16, 2>(__VA_ARGS__);                      \
    if (D == 256) return FN<PREC_BF16, 4>(__VA_ARGS__);                      \
    return FN<PREC_BF16, 8>(__VA_ARGS__);                                    \
  } while (0)

extern "C" int ark_vocab_ce_fwd(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq,
                                int64_t ld_seq, const float* hyper, float* row_loss, float* lse, float* dY_t, int B, int L,
                                int V, int D, void* stream) {
  using namespace ark;
  int rc = vc_check(prec, Y16, W16, bias, seq, hyper, B, L, V, D);
  if (rc) return rc;
  if (!row_loss || !lse) return ARK_ERR_ARG;
  if (dY_t && (B * L) % 16 != 0) return ARK_ERR_SHAPE;   // tile-native dY
  VocabCeArgs p{Y16, W16, bias, seq, hyper, row_loss, lse, dY_t, nullptr, nullptr, (long)ld_seq, B * L, B, V};
  const bool with_dy = dY_t != nullptr;
  ARK_VC_DISPATCH(vc_launch_fwd, p, with_dy, (hipStream_t)stream);
}

extern "C" int ark_vocab_ce_fwd_ws(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq,
                                   int64_t ld_seq, const float* hyper, float* row_loss, float* lse, float* dY_t, float* ws,
                                   int64_t ws_floats, int B, int L, int V, int D, int cu_budget, void* stream) {
  using namespace ark;
  int rc = vc_check(prec, Y16, W16, bias, seq, hyper, B, L, V, D);
  if (rc) return rc;
  if (!row_loss || !lse || !dY_t) return ARK_ERR_ARG;
  if ((B * L) % 16 != 0) return ARK_ERR_SHAPE;
  const long R = (long)B * L;
  const int nv = ark_vocab_ce_fwd_splits((int)R, V, D, cu_budget);
  VocabCeArgs p{Y16, W16, bias, seq, hyper, row_loss, lse, dY_t, nullptr, nullptr, (long)ld_seq, B * L, B, V};
  if (nv > 1) {
    if (!ws || ws_floats < (long)nv * (R * D + R * 4)) return ARK_ERR_ARG;
    p.part_u = ws;
    p.part_s = ws + (long)nv * R * D;
  }
  ARK_VC_DISPATCH(vc_launch_fwd, p, true, (hipStream_t)stream, nv);
}

extern "C" int ark_vocab_ce_dw(int prec, const void* Y16, const void* W16, const float* bias, const int64_t* seq, int64_t ld_seq,
                               const float* hyper, const float* lse, float* dW, float* db, int B, int L, int V, int D,
                               void* stream) {
  using namespace ark;
  int rc = vc_check(prec, Y16, W16, bias, seq, hyper, B, L, V, D);
  if (rc) return rc;
  if (!lse || !dW || !db) return ARK_ERR_ARG;
  if ((uintptr_t)dW & 15) return ARK_ERR_ALIGN;
  VocabCeArgs p{Y16, W16, bias, seq, hyper, nullptr, const_cast<float*>(lse), nullptr, dW, db, (long)ld_seq, B * L, B, V};
  ARK_VC_DISPATCH(vc_launch_dw, p, (hipStream_t)stream);
}
