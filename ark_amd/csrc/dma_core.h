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
#pragma once
#include <type_traits>
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

template <int PREC, int BM, int BN, int NBUF, int WGM, int WGN, int KI = 1>
struct DmaTile {
  static_assert(PREC == PREC_F16 || PREC == PREC_BF16, "LDS-DMA engine takes 16-bit operands");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile shape");
  static constexpr int NW = WGM * WGN;   // waves per workgroup (4, or 8 for 512-thread workgroups)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  // one pipeline stage = KI 64-wide k-images of both operands; NBUF stages ride a ring in LDS
  static constexpr int KS = 64 * KI;
  static constexpr int A_IMG = BM * 128, B_IMG = BN * 128;
  static constexpr int A_STAGE = A_IMG * KI, B_STAGE = B_IMG * KI;
  static constexpr int STAGE_BYTES = A_STAGE + B_STAGE;
  static constexpr int LDS_BYTES = NBUF * STAGE_BYTES;
  // a stage of one operand is KI * R/8 one-KB pieces (8 rows x 128 B each), dealt round-robin to the waves
  static constexpr int PA = KI * BM / 8, PB = KI * BN / 8;
  static_assert(PA % NW == 0 && PB % NW == 0, "pieces per stage must divide evenly over the waves");
  static constexpr int NPA = PA / NW, NPB = PB / NW;   // pieces per stage per wave
  static constexpr int LPS = NPA + NPB;                // LDS-DMA instructions per stage per wave
  static_assert((NBUF - 1) * LPS <= 63, "in-flight stages must fit the 6-bit vmcnt");
  static_assert(NBUF >= 2 && NBUF <= 8, "ring depth");
  static constexpr int WTM = BM / WGM, WTN = BN / WGN;
  static constexpr int TM = WTM / 16, TN = WTN / 16;

  template <int N>
  static __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  }
  // wait until all but the youngest `stages` stages of this wave's LDS-DMA have landed
  static __device__ __forceinline__ void wait_stages(int stages) {
    switch (stages) {
      case 0: wait_vmcnt<0>(); break;
      case 1: wait_vmcnt<LPS>(); break;
      case 2: wait_vmcnt<(NBUF > 2 ? 2 : 0) * LPS>(); break;
      case 3: wait_vmcnt<(NBUF > 3 ? 3 : 0) * LPS>(); break;
      case 4: wait_vmcnt<(NBUF > 4 ? 4 : 0) * LPS>(); break;
      case 5: wait_vmcnt<(NBUF > 5 ? 5 : 0) * LPS>(); break;
      case 6: wait_vmcnt<(NBUF > 6 ? 6 : 0) * LPS>(); break;
      default: wait_vmcnt<(NBUF > 7 ? 7 : 0) * LPS>(); break;
    }
  }

  // rma / rmb: tile row -> memory row (must be a valid row; clamp out-of-range rows on the caller side)
  template <class RMA, class RMB>
  static __device__ __forceinline__ void run(f32x4 (&acc)[TM][TN], const h_t* A, long lda, RMA rma, const h_t* B,
                                             long ldb, RMB rmb, int K, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;

    // piece q of an operand's stage: image j = q / (R/8), rows 8p..8p+7 with p = q % (R/8)
    const h_t* ap[NPA];
    const h_t* bp[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int q = wave + NW * i, j = q / (BM / 8), pr = q % (BM / 8);
      const int row = 8 * pr + (lane >> 3);
      ap[i] = A + rma(row) * lda + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int q = wave + NW * i, j = q / (BN / 8), pr = q % (BN / 8);
      const int row = 8 * pr + (lane >> 3);
      bp[i] = B + rmb(row) * ldb + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    auto issue = [&](int s) {
      char* base = lds + (s % NBUF) * STAGE_BYTES;
      const int k0 = s * KS;
#pragma unroll
      for (int i = 0; i < NPA; ++i)   // piece q lands at byte q*1024 of the operand's stage (images are contiguous)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ap[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + A_STAGE + (wave + NW * i) * 1024), 16, 0, 0);
    };

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int NS = K / KS;  // host guarantees K % KS == 0
    if (NS <= 0) return;
    const int pre = NS < NBUF ? NS : NBUF;
    for (int s = 0; s < pre; ++s) issue(s);
    for (int s = 0; s < NS; ++s) {
      // stages s+1 .. min(NS, s+NBUF)-1 may stay in flight while stage s is consumed
      const int ahead = (NS - 1 - s) < (NBUF - 1) ? (NS - 1 - s) : (NBUF - 1);
      wait_stages(ahead);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* bufA = lds + (s % NBUF) * STAGE_BYTES;
      const char* bufB = bufA + A_STAGE;
#pragma unroll
      for (int s2 = 0; s2 < 2 * KI; ++s2) {
        typename PT::h8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#if defined(ARK_ABL) && (ARK_ABL & 1)
          a[tm] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f});
#else
          a[tm] = *reinterpret_cast<const typename PT::h8*>(bufA + (s2 >> 1) * A_IMG +
                                                            lds_off(wm * WTM + tm * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#if defined(ARK_ABL) && (ARK_ABL & 2)
          b[tn] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f + tn});
#else
          b[tn] = *reinterpret_cast<const typename PT::h8*>(bufB + (s2 >> 1) * B_IMG +
                                                            lds_off(wn * WTN + tn * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
#if defined(ARK_ABL) && (ARK_ABL & 4)
            acc[tm][tn][0] += (float)a[tm][0] * (float)b[tn][0];
#else
            acc[tm][tn] = PT::mfma(a[tm], b[tn], acc[tm][tn]);
#endif
          }
      }
      if (s + NBUF < NS) {
        // every wave has consumed ring slot s%NBUF (its fragment reads are complete) -> refill it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + NBUF);
      }
    }
  }

  // Two K-segments through ONE ring: acc1 += A1 x B1^T over K1, then acc2 += A2 x B2^T over K2, with the
  // stages of segment 2 already in flight while segment 1 is still being consumed (no drain between
  // them).  Both segments share the tile -> memory row maps.  Used by the layer-diagonal GRU cell
  // (x W_ih^T and h W_hh^T must stay separate for the candidate gate).
  template <class RMA, class RMB>
  static __device__ __forceinline__ void run2(f32x4 (&acc1)[TM][TN], f32x4 (&acc2)[TM][TN], const h_t* A1, const h_t* B1,
                                              int K1, const h_t* A2, const h_t* B2, int K2, long lda, long ldb, RMA rma,
                                              RMB rmb, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    long ao[NPA], bo[NPB];   // element offsets of this lane's piece rows (same in both segments)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int q = wave + NW * i, j = q / (BM / 8), pr = q % (BM / 8);
      const int row = 8 * pr + (lane >> 3);
      ao[i] = rma(row) * lda + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int q = wave + NW * i, j = q / (BN / 8), pr = q % (BN / 8);
      const int row = 8 * pr + (lane >> 3);
      bo[i] = rmb(row) * ldb + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    const int NS1 = K1 / KS, NS = NS1 + K2 / KS;   // host guarantees K1 % KS == 0 && K2 % KS == 0
    auto issue = [&](int s) {
#if defined(ARK_ABL) && (ARK_ABL & 8)
      return;
#endif
      char* base = lds + (s % NBUF) * STAGE_BYTES;
      const bool first = s < NS1;
      const h_t* A = first ? A1 : A2;
      const h_t* B = first ? B1 : B2;
      const int k0 = (first ? s : s - NS1) * KS;
#pragma unroll
      for (int i = 0; i < NPA; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + ao[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + bo[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + A_STAGE + (wave + NW * i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc1[tm][tn] = acc2[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NS <= 0) return;
    const int pre = NS < NBUF ? NS : NBUF;
    for (int s = 0; s < pre; ++s) issue(s);
    auto consume = [&](f32x4 (&acc)[TM][TN], int s) {
      const char* bufA = lds + (s % NBUF) * STAGE_BYTES;
      const char* bufB = bufA + A_STAGE;
#pragma unroll
      for (int s2 = 0; s2 < 2 * KI; ++s2) {
        typename PT::h8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#if defined(ARK_ABL) && (ARK_ABL & 1)
          a[tm] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f});
#else
          a[tm] = *reinterpret_cast<const typename PT::h8*>(bufA + (s2 >> 1) * A_IMG +
                                                            lds_off(wm * WTM + tm * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#if defined(ARK_ABL) && (ARK_ABL & 2)
          b[tn] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f + tn});
#else
          b[tn] = *reinterpret_cast<const typename PT::h8*>(bufB + (s2 >> 1) * B_IMG +
                                                            lds_off(wn * WTN + tn * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
#if defined(ARK_ABL) && (ARK_ABL & 4)
            acc[tm][tn][0] += (float)a[tm][0] * (float)b[tn][0];
#else
            acc[tm][tn] = PT::mfma(a[tm], b[tn], acc[tm][tn]);
#endif
          }
      }
    };
    for (int s = 0; s < NS; ++s) {
      const int ahead = (NS - 1 - s) < (NBUF - 1) ? (NS - 1 - s) : (NBUF - 1);
      wait_stages(ahead);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s < NS1) consume(acc1, s);
      else consume(acc2, s);
      if (s + NBUF < NS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + NBUF);
      }
    }
  }

  // run2 with SHARED accumulators for all but the last 16-column block of the wave tile: segment 1 adds
  // into acc[.][0..TN-1], segment 2 into acc[.][0..TN-2] and acc[.][TN].  The GRU forward cell needs
  // x W_ih^T and h W_hh^T apart for the candidate gate only (last block of the r|z|n wave tile), so
  // it carries 4 instead of 6 accumulator tiles per 16 rows.
  template <class RMA, class RMB>
  static __device__ __forceinline__ void run2_shared(f32x4 (&acc)[TM][TN + 1], const h_t* A1, const h_t* B1,
                                              int K1, const h_t* A2, const h_t* B2, int K2, long lda, long ldb, RMA rma,
                                              RMB rmb, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    long ao[NPA], bo[NPB];   // element offsets of this lane's piece rows (same in both segments)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int q = wave + NW * i, j = q / (BM / 8), pr = q % (BM / 8);
      const int row = 8 * pr + (lane >> 3);
      ao[i] = rma(row) * lda + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int q = wave + NW * i, j = q / (BN / 8), pr = q % (BN / 8);
      const int row = 8 * pr + (lane >> 3);
      bo[i] = rmb(row) * ldb + 64 * j + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    const int NS1 = K1 / KS, NS = NS1 + K2 / KS;   // host guarantees K1 % KS == 0 && K2 % KS == 0
    auto issue = [&](int s) {
#if defined(ARK_ABL) && (ARK_ABL & 8)
      return;
#endif
      char* base = lds + (s % NBUF) * STAGE_BYTES;
      const bool first = s < NS1;
      const h_t* A = first ? A1 : A2;
      const h_t* B = first ? B1 : B2;
      const int k0 = (first ? s : s - NS1) * KS;
#pragma unroll
      for (int i = 0; i < NPA; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + ao[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + bo[i] + k0),
                                         (__attribute__((address_space(3))) void*)(base + A_STAGE + (wave + NW * i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn <= TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(ARK_ABL) && (ARK_ABL & 64)
    return;
#endif
    if (NS <= 0) return;
    const int pre = NS < NBUF ? NS : NBUF;
    for (int s = 0; s < pre; ++s) issue(s);
    auto consume = [&](auto last_col, int s) {   // last_col: accumulator column of the wave tile's last block
      const char* bufA = lds + (s % NBUF) * STAGE_BYTES;
      const char* bufB = bufA + A_STAGE;
#pragma unroll
      for (int s2 = 0; s2 < 2 * KI; ++s2) {
        typename PT::h8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#if defined(ARK_ABL) && (ARK_ABL & 1)
          a[tm] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f});
#else
          a[tm] = *reinterpret_cast<const typename PT::h8*>(bufA + (s2 >> 1) * A_IMG +
                                                            lds_off(wm * WTM + tm * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#if defined(ARK_ABL) && (ARK_ABL & 2)
          b[tn] = __builtin_bit_cast(typename PT::h8, f32x4{1.f + lr, 2.f, 3.f + lq, 4.f + tn});
#else
          b[tn] = *reinterpret_cast<const typename PT::h8*>(bufB + (s2 >> 1) * B_IMG +
                                                            lds_off(wn * WTN + tn * 16 + lr, 4 * (s2 & 1) + lq));
#endif
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            constexpr int LC = decltype(last_col)::value;
            const int c = (tn == TN - 1) ? LC : tn;
#if defined(ARK_ABL) && (ARK_ABL & 4)
            acc[tm][c][0] += (float)a[tm][0] * (float)b[tn][0];
#else
            acc[tm][c] = PT::mfma(a[tm], b[tn], acc[tm][c]);
#endif
          }
      }
    };
    for (int s = 0; s < NS; ++s) {
      const int ahead = (NS - 1 - s) < (NBUF - 1) ? (NS - 1 - s) : (NBUF - 1);
      wait_stages(ahead);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s < NS1) consume(std::integral_constant<int, TN - 1>{}, s);
      else consume(std::integral_constant<int, TN>{}, s);
      if (s + NBUF < NS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + NBUF);
      }
    }
  }
};

}  // namespace ark
