// Wave-private K-slice engine ("WPK") for the chip-filling products of the encoder (and any product of that kind):
// ONE workgroup per CU owns a large output tile, and every one of its waves (eight: two per SIMD; four in the first build)
// runs the WHOLE tile over its own K-slices (slice s belongs to wave s mod NW) through a PRIVATE LDS-DMA ring -- no barrier
// and no cross-wave hand-off anywhere in the main loop; the partial tiles meet in LDS once, at the end.
//
// Why (round 5): the ring engine of dma_core.h shares every stage between the waves of a workgroup, so it needs two
// barriers per 64-wide stage and small tiles (32 x 64 ... 128 x 128 split eight ways) whose phases only overlap across
// three resident workgroups.  For the [1024 x 1536] x [1536 x 1536] encoder products that is 768 workgroups streaming
// 221 MB from L2 into LDS for 12.6 MB of operands (864 KB per CU at ~48 GB/s per CU = 18 us), with three ds_read_b128 per
// two MFMAs.  Here a CU streams each operand byte of its tile exactly once (64 x 96 tile: 480 KB per CU), a wave reads 10
// fragments for 24 MFMAs, 160 KB of LDS-DMA are in flight per CU, and the waves drift apart by themselves: while one
// multiplies, its SIMD partner issues its pieces or waits for data -- the phase overlap three small workgroups gave,
// without their redundant bytes.  Measured: 18.1 -> 13-15 us per product (tools/wpk_stamps.py).
//
// Price: the whole tile's accumulators live in EVERY wave (BM x BN / 64 registers: 96 for 64 x 96), a cross-wave sum
// through LDS at the end (the ring's space is reused), and ONE workgroup per CU (160 KB of LDS): beside a launch that
// leaves less than that free (the weight gradients' two 64-KB workgroups per CU) such a workgroup waits for a whole CU.
#pragma once
#include "dma_core.h"

// diagnostic builds (-DARK_STAMPS, tools/wpk_stamps.py) define WPK_STAMP(i) before including this file: per-wave
// real-time stamps; the shipped library compiles them to nothing
#ifndef WPK_STAMP
#define WPK_STAMP(i) do { } while (0)
#endif

namespace ark {

// ---- "NT" flavour: C[BM, BN] = A[BM, K] x B[BN, K]^T, both operands 16-bit, K-contiguous, row-major -------------------
// Slot image: A image (BM rows x 128 B), then B image (BN rows x 128 B), one 64-wide k-slice; same XOR swizzle and the same
// ds_read_b128 fragment reads as dma_core.h / gemm_core.h (lds_off).
// NW waves (4: one per SIMD, or 8: two per SIMD) x NSLOT ring slots each.  Measured (tools/wpk_stamps.py, 64 x 96 tile,
// K = 1536): with one wave per SIMD a slice costs its wave (fragment reads) + (MFMAs) + (piece issue) one after the other --
// nobody else is on the SIMD -- and the CU takes in 59 GB/s where the bare piece stream of the priming burst reaches 87;
// with two waves per SIMD and ONE slot each (the same 160 KB in flight) a partner fills those gaps.
template <int PREC, int BM, int BN, int NW_, int NSLOT>
struct WpkNT {
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  static constexpr int NW = NW_;
  static constexpr int NT = 64 * NW;                 // threads per workgroup
  static constexpr int TM = BM / 16, TN = BN / 16;
  static constexpr int A_IMG = BM * 128, B_IMG = BN * 128;
  static constexpr int SLOT = A_IMG + B_IMG;
  static constexpr int RING = NSLOT * SLOT;          // per wave
  static constexpr int PA = BM / 8, PB = BN / 8;     // 1-KB pieces per slice
  static constexpr int LPS = PA + PB;
  static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
  static_assert(BM % 16 == 0 && BN % 32 == 0 && (BM / 8) % 2 == 0, "whole MFMA tiles, 32-column groups in the epilogue, even A pieces");
  static_assert(NSLOT >= 1 && NSLOT <= 3 && NSLOT * LPS <= 63, "in-flight pieces must fit the 6-bit vmcnt");

  template <int N>
  static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

  // acc += A[m0 .. m0+BM, :] x B[n0 .. n0+BN, :]^T over this wave's slices of [0, K) (K % 64 == 0); rows must exist.
  // rot: the slice this workgroup starts at (any value; slices are taken modulo K / 64).  The callers pass 0: starting the
  // workgroups of an XCD at different slices (so that one's L2 misses are the others' hits) measured no faster, and with a
  // common start every output row is summed in the SAME order whichever tile it falls into -- a data-parallel shard and the
  // full batch then agree bit for bit in these products, which the two-process tests lean on.
  // a_rows / b_rows (> 0): rows of the tile that exist in memory -- the pieces of the rest re-read the last valid row (their
  // products land in accumulator rows / columns the epilogue drops); 0 = the whole tile exists (no clamping code at all)
  template <bool CLAMP = false>
  static __device__ __forceinline__ void run(f32x4 (&acc)[TM][TN], const h_t* A, long lda, const h_t* B, long ldb, int K,
                                             int rot, char* lds, int a_rows = 0, int b_rows = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    char* ring = lds + wave * RING;
    // piece p of an image = rows 8p .. 8p+7; lane i supplies row 8p + i/8, logical chunk (i%8) ^ ((row>>1)&7); the
    // swizzle term only depends on the parity of p: two lane offsets per operand serve every piece
    int voa[2], vob[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int sw = 8 * ((lane & 7) ^ ((4 * e + (lane >> 4)) & 7));
      voa[e] = (lane >> 3) * (int)lda + sw;
      vob[e] = (lane >> 3) * (int)ldb + sw;
    }
    const int NS = K / 64;
    const int mine = (NS - wave + NW - 1) / NW;        // slices wave, wave + NW, ... of the rotated sequence
    const int rot0 = __builtin_amdgcn_readfirstlane(rot % NS);
    // piece q of slice j: q < PA an A piece, else a B piece (q is a compile-time constant at every call site)
    auto issue_piece = [&](int j, int q) {
      int sl = rot0 + wave + NW * j;
      sl = sl >= NS ? sl - NS : sl;
      const long k0 = (long)sl * 64;
      char* slot = ring + (j % NSLOT) * SLOT;
      if constexpr (CLAMP) {
        const int sw = 8 * ((lane & 7) ^ ((4 * (q & 1) + (lane >> 4)) & 7));   // (PA is even: the parity of q is the piece's own)
        if (q < PA) {
          const int row = min(8 * q + (lane >> 3), a_rows - 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + k0 + (long)row * lda + sw),
                                           (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
        } else {
          const int row = min(8 * (q - PA) + (lane >> 3), b_rows - 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + k0 + (long)row * ldb + sw),
                                           (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
        }
      } else if (q < PA)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + k0 + (long)(8 * q) * lda + voa[q & 1]),
                                         (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + k0 + (long)(8 * (q - PA)) * ldb + vob[(q - PA) & 1]),
                                         (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
    };
    {
      const int pre = mine < NSLOT ? mine : NSLOT;
      for (int j = 0; j < pre; ++j) {
#pragma unroll
        for (int q = 0; q < LPS; ++q) issue_piece(j, q);
      }
    }
    WPK_STAMP(1);
    // Two waves per SIMD have 256 registers each: the tile's accumulators (BM x BN / 64) leave room for ONE 32-wide k-step of
    // fragments at a time -- read, multiply, read, multiply, the refill riding the second half (the partner wave covers
    // the exposed read latency).  One wave per SIMD holds both k-steps and deals the refill over the whole block.
    constexpr bool HALVES = NW == 8;
    constexpr int NMM = (HALVES ? 1 : 2) * TM * TN;        // MFMAs the refill is dealt over
    constexpr int EVERY = NMM / LPS > 0 ? NMM / LPS : 1;   // one LDS-DMA piece in front of every EVERY-th MFMA
    for (int j = 0; j < mine; ++j) {
      // slices j .. min(j + NSLOT, mine) - 1 are in flight; slice j must have landed
      const int behind = mine - 1 - j;
      if (behind >= NSLOT - 1) wait_vmcnt<(NSLOT - 1) * LPS>();
      else if (NSLOT > 2 && behind == 1) wait_vmcnt<LPS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) WPK_STAMP(2);
      const char* bufA = ring + (j % NSLOT) * SLOT;
      const char* bufB = bufA + A_IMG;
      const bool more = j + NSLOT < mine;
      // The refill of this slot is dealt out BETWEEN the MFMAs (a piece costs its wave ~50-100 cycles of issue beside the
      // other waves' pieces: the matrix pipe works through the MFMAs already issued meanwhile); the scheduling barriers
      // keep hipcc from gathering the pieces in front of the block.
      auto mm_block = [&](auto refill, auto s2lo, auto s2n, const h8 (&a)[2][TM], const h8 (&b)[2][TN]) {
        constexpr int N = decltype(s2n)::value * TM * TN;
        static_for<N>([&](auto ic) {
          constexpr int idx = decltype(ic)::value;
          constexpr int s2 = decltype(s2lo)::value + idx / (TM * TN), tm = (idx / TN) % TM, tn = idx % TN;
          if constexpr (decltype(refill)::value && idx % EVERY == 0 && idx / EVERY < LPS) {
            issue_piece(j + NSLOT, idx / EVERY);
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[tm][tn] = PT::mfma(a[s2][tm], b[s2][tn], acc[tm][tn]);
          if constexpr (decltype(refill)::value && (idx + 1) % EVERY == 0 && (idx + 1) / EVERY < LPS) __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (decltype(refill)::value) {   // (pieces that did not get an MFMA slot of their own)
#pragma unroll
          for (int q = (N + EVERY - 1) / EVERY; q < LPS; ++q) issue_piece(j + NSLOT, q);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if constexpr (HALVES) {
        h8 a[2][TM], b[2][TN];   // (only [0] is used: one k-step at a time)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a[0][tm] = *reinterpret_cast<const h8*>(bufA + lds_off(tm * 16 + lr, 4 * s2 + lq));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) b[0][tn] = *reinterpret_cast<const h8*>(bufB + lds_off(tn * 16 + lr, 4 * s2 + lq));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          // after the second k-step's reads every fragment of the slot sits in registers: it may be overwritten
          if (s2 == 1 && more) mm_block(std::true_type{}, I0{}, I1{}, a, b);
          else mm_block(std::false_type{}, I0{}, I1{}, a, b);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        h8 a[2][TM], b[2][TN];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a[s2][tm] = *reinterpret_cast<const h8*>(bufA + lds_off(tm * 16 + lr, 4 * s2 + lq));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) b[s2][tn] = *reinterpret_cast<const h8*>(bufB + lds_off(tn * 16 + lr, 4 * s2 + lq));
        }
        // the slot may be overwritten once every fragment of it sits in registers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) mm_block(std::true_type{}, I0{}, I2{}, a, b); else mm_block(std::false_type{}, I0{}, I2{}, a, b);
      }
    }
    WPK_STAMP(3);
  }

  // Sum of the waves' partial tiles, handed out in ROW-MAJOR quads: f(row, col, v) is called once for every 4 consecutive
  // columns col .. col+3 of tile row `row` (v = their sums), dealt so that 8 consecutive lanes hold 128 contiguous bytes of
  // one output row -- every epilogue access is a 16-byte (fp32) or 8-byte (16-bit) vector access (the fragment-order
  // epilogue of the ring kernels took four times the store instructions and, with a single workgroup on the CU to hide
  // them behind, 5-10 us).  With eight waves the upper four first hand their tiles to the lower four in fragment order
  // (eight row-major partials would not fit LDS); then four partials cross LDS row-major with rows of BN + 4 floats: the four
  // accumulator rows of a lane land in distinct banks, the quad reads are plain ds_read_b128.
  static constexpr int LDP = BN + 4;
  static constexpr int RED_FRAG = TM * TN * 1024;                 // one wave's tile in fragment order
  static constexpr int RED_ROWMAJ = BM * LDP * 4;                 // one wave's tile row-major
  static constexpr int CS_OFF = 4 * RED_ROWMAJ;                   // BN floats of column sums behind the partials
  static constexpr int LDS_BYTES = NW * RING > CS_OFF + BN * 4 ? NW * RING : CS_OFF + BN * 4;
  static_assert(LDS_BYTES <= 160 * 1024 && 4 * RED_FRAG <= LDS_BYTES, "LDS");
  static constexpr int QUADS = BM * (BN / 4);
  static constexpr int NQ = (QUADS + NT - 1) / NT;                // quads per thread (at most)
  template <class F>
  static __device__ __forceinline__ void reduce_rows(f32x4 (&acc)[TM][TN], char* lds, F&& f) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();   // every wave is past its last fragment read: the rings are free
    WPK_STAMP(4);
    if constexpr (NW == 8) {
      char* slab = lds + (wave & 3) * RED_FRAG;
      if (wave >= 4) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) *reinterpret_cast<f32x4*>(slab + ((tm * TN + tn) * 64 + lane) * 16) = acc[tm][tn];
      }
      __syncthreads();
      if (wave < 4) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] += *reinterpret_cast<const f32x4*>(slab + ((tm * TN + tn) * 64 + lane) * 16);
      }
      __syncthreads();   // the slabs are read: their space takes the row-major partials
    }
    if (wave < 4) {
      float* mine = reinterpret_cast<float*>(lds + wave * RED_ROWMAJ);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int i = 0; i < 4; ++i) mine[(tm * 16 + 4 * (lane >> 4) + i) * LDP + tn * 16 + (lane & 15)] = acc[tm][tn][i];
    }
    if (tid < BN) reinterpret_cast<float*>(lds + CS_OFF)[tid] = 0.f;
    __syncthreads();
    WPK_STAMP(5);
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
      const int u = it * NT + tid;
      if (QUADS % NT != 0 && u >= QUADS) break;
      // 8 lanes = 128 contiguous bytes of a row; the 8-lane groups of a wave are 8 consecutive ROWS of one 32-column group
      // (so a wave's column sums are a shuffle fold over its lane groups)
      const int q8 = u & 7, rest = u >> 3;
      const int row = rest % BM, cg = rest / BM;
      const int col = cg * 32 + 4 * q8;
      f32x4 v = *reinterpret_cast<const f32x4*>(lds + (row * LDP + col) * 4);
#pragma unroll
      for (int q = 1; q < 4; ++q) v += *reinterpret_cast<const f32x4*>(lds + q * RED_ROWMAJ + (row * LDP + col) * 4);
      f(row, col, v);
    }
  }
  static __device__ __forceinline__ float* colsum_lds(char* lds) { return reinterpret_cast<float*>(lds + CS_OFF); }
};

// Tile order for a (tiles_m x tiles_n) grid: consecutive logical ids walk GM row tiles down before moving one tile to
// the right, so that the 32 ids an XCD receives from xcd_remap form a (GM x 32/GM) block: its CUs share GM A panels and
// 32/GM B panels in their L2 instead of 2 and 16 (row-major) -- 2.7 instead of 5.1 MB per XCD at 1024 x 1536 x 1536.
__device__ __forceinline__ void wpk_tile_of(int id, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
  const int per = GM * tiles_n;
  const int g = id / per, first = g * GM;
  const int gm = min(GM, tiles_m - first);
  const int r = id - g * per;
  tm = first + r % gm;
  tn = r / gm;
}

}  // namespace ark
