// LDS-DMA ring engine for products whose operands are both 16-bit and K-contiguous (GRU cells,
// input / input-gradient products against plain or transposed weight shadows).  Operands stream
// HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no conversions); NBUF ring slots of
// KI 64-wide k-images each are kept in flight per workgroup with counted `s_waitcnt vmcnt(N)` and raw
// `s_barrier`s, so the next stages keep flying underneath the MFMA block.
//
// Why: the register-staged engine (gemm_core.h) holds at most ~32 KB in flight per CU, and a GRU
// timestep is one short dependent launch -- Little's law capped it at ~8 TB/s of L2->CU traffic
// (profiles/r01_first_path_*).  Measured on MI355X: SMALL rings win (2 slots x 128 k for the cells,
// 2 x 64 k for the GEMMs): what matters is how many workgroups are resident per CU, not ring depth;
// a single-barrier variant with one more slot was slower for the same reason.
//
// LDS image per slot and operand: KI "k-images" of R rows x 128 B, same XOR swizzle and the same
// ds_read_b128 fragment reads as gemm_core.h.  LDS-DMA writes are lane-linear (wave-uniform base +
// lane*16), so the swizzle is applied to the per-lane SOURCE address instead: lane i of the
// wave-instruction that fills rows 8p..8p+7 of an image supplies row 8p + i/8, logical chunk
// (i%8) ^ ((row>>1)&7).
//
// ONE runner, `run_segs`, serves every user: a product is a list of K-SEGMENTS (operand pair + length)
// that stream through the same ring back to back without draining between them; the caller's functor
// says which accumulators a segment's MFMAs go to.  Each segment gets its OWN consume loop: with a
// single loop and a run-time "which accumulator" select, hipcc moved every accumulator between AGPRs
// and VGPRs on every stage (rocprofv3 round 2: 2 700 VALU instructions per wave against 192 MFMAs in
// the forward diagonal kernel; SQ_ACTIVE_INST + SQ_WAIT_INST = 71 % of the wave cycles).
#pragma once
#include <type_traits>
#include <utility>
#include "gemm_core.h"

namespace ark {

// 16x16-tile-native addressing shared by producers and consumers of per-element fp32 state (GRU
// gate saves, gi): element (row, col) of a [rows, ld] logical matrix lives at
//   ((row>>4) * (ld>>4) + (col>>4)) * 256 + (((row>>2)&3)*16 + (col&15)) * 4 + (row&3)
// i.e. exactly the MFMA C/D fragment order: one lane's 4 accumulator rows are one float4, one
// wave's 16x16 tile is one contiguous 1-KB line -> every epilogue access is a full-rate dwordx4.
__device__ __forceinline__ long tile_native_off(int row, int col, int ld) {
  return ((long)(row >> 4) * (ld >> 4) + (col >> 4)) * 256 + ((((row >> 2) & 3) << 4) + (col & 15)) * 4 + (row & 3);
}

// selects between wave-uniform values on the scalar unit (operands are forced into SGPRs)
__device__ __forceinline__ uint64_t s_select64(int cond, uint64_t a, uint64_t b) {
  uint64_t r;
  asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b64 %0, %2, %3" : "=s"(r) : "s"(cond), "s"(a), "s"(b) : "scc");
  return r;
}
__device__ __forceinline__ int s_select32(int cond, int a, int b) {
  int r;
  asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(r) : "s"(cond), "s"(a), "s"(b) : "scc");
  return r;
}

// ds_read_b64_tr_b16 through inline asm, for kernels whose LDS is filled by LDS-DMA.  The compiler's builtin
// (__builtin_amdgcn_ds_read_tr16_b64_*) reaches the waitcnt pass without a memory operand it could prove disjoint from the
// LDS-DMA destinations in flight, so the pass puts `s_waitcnt vmcnt(0)` in front of the first such read of every stage: the
// whole ring is drained, including the stage issued a moment ago, and a ring of any depth runs as if it had one slot
// (tools/isa_dma_waits.py lists these waits; round 5 found them in wgrad16, vocab_ce fwd and vocab_ce dw).
// The asm form hands back registers the compiler believes READY: tr_wait<N>() -- s_waitcnt lgkmcnt(N) and a tie on every
// register it covers -- has to sit between a read and the first use of its result (any use: a register copy counts).
// N = reads issued AFTER the ones waited for that may stay in flight (LDS returns in order; extra LDS operations of the
// compiler's own in the queue only make a counted wait stricter).
typedef short tr16x4 __attribute__((ext_vector_type(4)));
typedef short tr16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned lds_addr(const char* p) {   // byte address inside the workgroup's LDS
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF = 0>
__device__ __forceinline__ tr16x4 lds_tr16(unsigned a) {
  tr16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int OFF = 0>
__device__ __forceinline__ tr16x4 lds_tr16(const char* p) { return lds_tr16<OFF>(lds_addr(p)); }
// a plain 16-byte fragment read in the same un-waited form (for loops that pipeline their fragment reads by hand: left to
// the compiler, a read of a DMA-filled image was issued, waited for -- lgkmcnt(0) -- and used one at a time)
template <int OFF = 0>
__device__ __forceinline__ f32x4 lds_b128(unsigned a) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void tr_wait_cnt() {
  static_assert(N >= 0, "count");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
__device__ __forceinline__ void tr_tie(tr16x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void tr_tie(f32x4& v) { asm volatile("" : "+v"(v)); }
template <class H8>
__device__ __forceinline__ H8 tr_join(tr16x4 lo, tr16x4 hi) {
  const tr16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(H8, v);
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NI un-waited LDS reads-and-uses, the reads G items ahead of their uses in two register sets:
//   req(item, set, slot) issues the RPI asm reads of `item` into slot `slot` of set `set` (all three integral_constants);
//   use(item, set, slot) ties those registers (tr_tie) and consumes them.
template <int NI, int G, int RPI, class Req, class Use>
__device__ __forceinline__ void lds_pipeline(Req&& req, Use&& use) {
  static_assert(NI % G == 0, "items per group");
  constexpr int NG = NI / G;
  static_for<G>([&](auto t) { req(t, std::integral_constant<int, 0>{}, t); });
  static_for<NG>([&](auto g) {
    if constexpr (g + 1 < NG) {
      static_for<G>([&](auto t) { req(std::integral_constant<int, (g + 1) * G + t>{}, std::integral_constant<int, (g + 1) & 1>{}, t); });
      tr_wait_cnt<RPI * G>();
    } else {
      tr_wait_cnt<0>();
    }
    static_for<G>([&](auto t) { use(std::integral_constant<int, g * G + t>{}, std::integral_constant<int, g & 1>{}, t); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int PREC, int BM, int BN, int NBUF, int WGM, int WGN, int KI = 1>
struct DmaTile {
  static_assert(PREC == PREC_F16 || PREC == PREC_BF16, "LDS-DMA engine takes 16-bit operands");
  static_assert(BM % (16 * WGM) == 0 && BN % (16 * WGN) == 0, "tile shape: whole 16 x 16 MFMA tiles per wave");
  static constexpr int NW = WGM * WGN;   // waves per workgroup (4; 2 for narrow tiles; 8 or 16 for 512- / 1024-thread workgroups)
  static_assert(NW == 2 || NW == 4 || NW == 8 || NW == 16, "2, 4, 8 or 16 waves");
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  // one pipeline stage = KI 64-wide k-images of both operands; NBUF stages ride a ring in LDS
  static constexpr int KS = 64 * KI;
  static constexpr int A_IMG = BM * 128, B_IMG = BN * 128;
  static constexpr int A_STAGE = A_IMG * KI, B_STAGE = B_IMG * KI;
  static constexpr int STAGE_BYTES = A_STAGE + B_STAGE;
  static constexpr int LDS_BYTES = NBUF * STAGE_BYTES;
  // a stage of one operand is KI * R/8 one-KB pieces (8 rows x 128 B each), dealt round-robin to the waves
  // Pieces that do not divide evenly (24 weight pieces over 16 waves) leave the high waves one piece short: those waves
  // simply count fewer LDS-DMA instructions per stage in their s_waitcnt (wave-uniform, see lps_of / wait_stage).
  static constexpr int PA = KI * BM / 8, PB = KI * BN / 8;
  static constexpr int NPA = (PA + NW - 1) / NW, NPB = (PB + NW - 1) / NW;   // pieces per stage per wave (at most)
  static constexpr int LPS = NPA + NPB;                // LDS-DMA instructions per stage per wave (at most)
  static constexpr bool EVEN_A = PA % NW == 0, EVEN_B = PB % NW == 0;
  static_assert((NBUF - 1) * LPS <= 63, "in-flight stages must fit the 6-bit vmcnt");
  static_assert(NBUF >= 2 && NBUF <= 8, "ring depth");
  static constexpr int WTM = BM / WGM, WTN = BN / WGN;
  static constexpr int TM = WTM / 16, TN = WTN / 16;

  template <int N>
  static __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  }
  // this wave's LDS-DMA instructions per stage: LPS, or one / two fewer where the pieces run out (wave-uniform)
  static __device__ __forceinline__ int lps_of(int wave) {
    return LPS - ((EVEN_A || wave < PA % NW) ? 0 : 1) - ((EVEN_B || wave < PB % NW) ? 0 : 1);
  }
  // all but the youngest NBUF-1 stages of this wave have landed
  static __device__ __forceinline__ void wait_stage(int lps) {
    if constexpr (EVEN_A && EVEN_B) {
      wait_vmcnt<(NBUF - 1) * LPS>();
    } else {
      if (lps == LPS) wait_vmcnt<(NBUF - 1) * LPS>();
      else if (lps == LPS - 1) wait_vmcnt<(NBUF - 1) * (LPS - 1)>();
      else wait_vmcnt<(NBUF - 1) * (LPS > 2 ? LPS - 2 : 0)>();
    }
  }

  // K-segments of one product: acc(seg) += A[seg] x B[seg]^T over K[seg] (K % KS == 0, 0 allowed);
  // all segments share the leading dimensions and the tile -> memory row maps.
  template <int NSEG>
  struct Segs {
    const h_t* A[NSEG];
    const h_t* B[NSEG];
    int K[NSEG];
  };

  // rma / rmb: tile row -> memory row (must be a valid row; clamp out-of-range rows on the caller side).
  // mm(std::integral_constant<int, seg>, a[TM], b[TN]): the segment's MFMAs for one 32-wide k-step.
  //
  // ONEBAR = false: the two-barrier loop of rounds 1-2 (wait, barrier, consume, barrier, refill the slot just consumed:
  //   NBUF stages in flight).  Right for SMALL tiles, several workgroups per CU: the other workgroups fill the gaps.
  // ONEBAR = true: ONE barrier per stage and the refill issued BEFORE the consume (NBUF - 1 stages in flight): the
  //   barrier that publishes stage s also says every wave has finished reading stage s - 1, whose slot is refilled at
  //   once -- the LDS-DMA queue never runs dry while the waves read and multiply.  For a LONE large workgroup per CU
  //   (128 x 64 tiles on 8 waves), which has no neighbour to hide its own phases behind.
  // AUXA / AUXB: cache-policy bits of the A / B operand's LDS-DMA loads (0 = default, 2 = nt: streamed once, do not keep
  // in L2 -- for the operand that is NOT re-read by later launches, so that the one that is stays resident)
  template <int NSEG, bool ONEBAR = false, int AUXA = 0, int AUXB = 0, class RMA, class RMB, class MM>
  static __device__ __forceinline__ void run_segs(const Segs<NSEG>& sg, long lda, long ldb, RMA rma, RMB rmb, char* lds,
                                                  MM&& mm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    // piece q of an operand's stage: image j = q / (R/8), rows 8p..8p+7 with p = q % (R/8); these are this
    // lane's element offsets inside the operand (the same for every segment)
    long ao[NPA], bo[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int q = min(wave + NW * i, PA - 1), j = q / (BM / 8), pr = q % (BM / 8);
      const int row = 8 * pr + (lane >> 3);
      ao[i] = rma(row) * lda + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int q = min(wave + NW * i, PB - 1), j = q / (BN / 8), pr = q % (BN / 8);
      const int row = 8 * pr + (lane >> 3);
      bo[i] = rmb(row) * ldb + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    // stages per segment, cumulative: segment i owns global stages [cum[i-1], cum[i])
    int cum[NSEG];
    {
      int c = 0;
#pragma unroll
      for (int i = 0; i < NSEG; ++i) { c += sg.K[i] / KS; cum[i] = c; }
    }
    int left = cum[NSEG - 1];             // stages not issued yet
    if (left <= 0) return;
    int remaining = left;                 // stages not consumed yet
    int gi = 0, islot = 0;                // issue cursor: global stage, ring slot (wave-uniform)
    auto issue_next = [&]() {
      // operand bases of the segment that owns stage gi, selected on the SCALAR unit (hipcc if-converted the
      // plain C++ selects into v_cndmask chains on VGPR copies of these wave-uniform pointers)
      uint64_t A = reinterpret_cast<uint64_t>(sg.A[0]), B = reinterpret_cast<uint64_t>(sg.B[0]);
      int start = 0;
#pragma unroll
      for (int i = 1; i < NSEG; ++i) {
        const int in_i = __builtin_amdgcn_readfirstlane(gi >= cum[i - 1] ? 1 : 0);
        A = s_select64(in_i, reinterpret_cast<uint64_t>(sg.A[i]), A);
        B = s_select64(in_i, reinterpret_cast<uint64_t>(sg.B[i]), B);
        start = s_select32(in_i, cum[i - 1], start);
      }
      const int ik = (gi - start) * KS;
      const h_t* Ap = reinterpret_cast<const h_t*>(A) + ik;
      const h_t* Bp = reinterpret_cast<const h_t*>(B) + ik;
      char* base = lds + islot * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < NPA; ++i)   // piece q lands at byte q*1024 of the operand's stage (images are contiguous)
        if (EVEN_A || i + 1 < NPA || wave + NW * i < PA)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ap + ao[i]),
                                           (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, AUXA);
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        if (EVEN_B || i + 1 < NPB || wave + NW * i < PB)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bp + bo[i]),
                                           (__attribute__((address_space(3))) void*)(base + A_STAGE + (wave + NW * i) * 1024), 16, 0, AUXB);
      ++gi;
      islot = (islot + 1 == NBUF) ? 0 : islot + 1;
      --left;
    };
    constexpr int PF = ONEBAR ? NBUF - 1 : NBUF;   // stages issued ahead of the one being consumed
    {
      const int pre = left < PF ? left : PF;
      for (int s = 0; s < pre; ++s) issue_next();
    }
    int cslot = 0;
    const int lps = lps_of(wave);
    static_for<NSEG>([&](auto segc) {
      const int ns = sg.K[decltype(segc)::value] / KS;
      for (int s = 0; s < ns; ++s) {
        if constexpr (ONEBAR) {
          static_assert(!ONEBAR || (EVEN_A && EVEN_B), "single-barrier ring: pieces must deal evenly over the waves");
          static_assert(!ONEBAR || (NBUF >= 3 && NBUF <= 5), "single-barrier ring: 3 to 5 slots");
          // stages in flight behind the one to consume: min(PF - 1, remaining - 1), each LPS instructions of this wave
          const int behind = remaining - 1;
          if (behind >= PF - 1) wait_vmcnt<(PF - 1) * LPS>();
          else if (behind == 2) wait_vmcnt<2 * LPS>();
          else if (behind == 1) wait_vmcnt<LPS>();
          else wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (left > 0) issue_next();   // into the slot of stage s - 1: every wave is past its reads of it
          const char* bufA = lds + cslot * STAGE_BYTES;
          const char* bufB = bufA + A_STAGE;
#pragma unroll
          for (int s2 = 0; s2 < 2 * KI; ++s2) {
            h8 a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
              a[tm] = *reinterpret_cast<const h8*>(bufA + (s2 >> 1) * A_IMG + lds_off(wm * WTM + tm * 16 + lr, 4 * (s2 & 1) + lq));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              b[tn] = *reinterpret_cast<const h8*>(bufB + (s2 >> 1) * B_IMG + lds_off(wn * WTN + tn * 16 + lr, 4 * (s2 & 1) + lq));
            mm(segc, a, b);
          }
          --remaining;
          cslot = (cslot + 1 == NBUF) ? 0 : cslot + 1;
          continue;
        }
        // min(NBUF, remaining) stages are in flight; the oldest must have landed.  (Ring tails deeper than two
        // slots drain completely: exact counting there bought nothing measurable.)
        if (remaining >= NBUF) wait_stage(lps);
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* bufA = lds + cslot * STAGE_BYTES;
        const char* bufB = bufA + A_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2 * KI; ++s2) {
          h8 a[TM], b[TN];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            a[tm] = *reinterpret_cast<const h8*>(bufA + (s2 >> 1) * A_IMG + lds_off(wm * WTM + tm * 16 + lr, 4 * (s2 & 1) + lq));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            b[tn] = *reinterpret_cast<const h8*>(bufB + (s2 >> 1) * B_IMG + lds_off(wn * WTN + tn * 16 + lr, 4 * (s2 & 1) + lq));
          mm(segc, a, b);
        }
        --remaining;
        cslot = (cslot + 1 == NBUF) ? 0 : cslot + 1;
        if (left > 0) {
          // every wave has consumed the ring slot (its fragment reads are complete) -> refill it
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          issue_next();
        }
      }
    });
  }

  // plain product: acc = A x B^T over K
  template <class RMA, class RMB>
  static __device__ __forceinline__ void run(f32x4 (&acc)[TM][TN], const h_t* A, long lda, RMA rma, const h_t* B,
                                             long ldb, RMB rmb, int K, char* lds) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    Segs<1> sg{{A}, {B}, {K}};
    run_segs<1>(sg, lda, ldb, rma, rmb, lds, [&](auto, const h8 (&a)[TM], const h8 (&b)[TN]) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = PT::mfma(a[tm], b[tn], acc[tm][tn]);
    });
  }
};

}  // namespace ark
