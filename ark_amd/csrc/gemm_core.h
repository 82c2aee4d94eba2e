// LDS-tiled MFMA GEMM main loop for gfx950, shared by every dense contraction on the
// SAIL/ARK training path (encoder MLP, latent heads, GRU input/recurrent products, the tied
// vocabulary projection and all of their backward products).
//
// Design (CDNA4-first, not a warp-tiling port):
//  * 256-thread workgroups = 4 wave64s arranged WGM x WGN; every wave owns a grid of 16x16 MFMA
//    tiles (TM x TN accumulators of 4 VGPRs each, C/D map: col = lane&15, row = 4*(lane>>4)+reg).
//  * Both operands are staged global -> registers -> LDS into ONE canonical image: R rows of
//    128 bytes (32 f32 or 64 bf16 of the reduction index), with the 16-byte chunk index XORed by
//    (row>>1)&7 so that the 16-lane groups of ds_read_b128 hit 16 distinct bank slots.
//  * Operands whose reduction index is the SLOW memory index (weight-gradient and
//    input-gradient products) are transposed in registers on the way in (4k x 4row blocks), so
//    the fragment reads are the same conflict-free ds_read_b128 for every layout.
//  * PREC_F32  : v_mfma_f32_16x16x4_f32 (exact f32 fma chain; parity / decode mode)
//    PREC_BF16 : v_mfma_f32_16x16x32_bf16 with f32 accumulation (operands rounded RNE to bf16
//                while staging; HBM copies stay f32) -- range-safe, used for backward products
//    PREC_F16  : v_mfma_f32_16x16x32_f16, same rate, 3 more mantissa bits -- forward products
//                (activations / weights are O(1); saturating conversion)
//  * Next K-step's global loads are issued right after the LDS image of the current step is
//    complete, so they are in flight underneath the MFMA block (issue-early / write-late).
#pragma once
#include <type_traits>
#include "common.h"

namespace ark {

template <int PREC> struct PrecTraits;
template <> struct PrecTraits<PREC_F32> {
  static constexpr int BK = 32;   // reduction elements per LDS stage (128 B of f32)
  static constexpr int FPC = 1;   // float4 global loads per 16-byte LDS chunk
};
template <> struct PrecTraits<PREC_BF16> {
  static constexpr int BK = 64;   // 128 B of bf16
  static constexpr int FPC = 2;
  using h_t = __bf16; using h8 = bf16x8; using h4 = bf16x4;
  static __device__ __forceinline__ h_t cvt(float x) { return (__bf16)x; }
  static __device__ __forceinline__ f32x4 mfma(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PrecTraits<PREC_F16> {
  static constexpr int BK = 64;   // 128 B of fp16
  static constexpr int FPC = 2;
  using h_t = _Float16; using h8 = f16x8; using h4 = f16x4;
  // saturate instead of overflowing to inf (fp16 max 65504); forward activations/weights only
  static __device__ __forceinline__ h_t cvt(float x) { return (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f); }
  static __device__ __forceinline__ f32x4 mfma(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// store one value as a 16-bit element whose type is chosen at run time (epilogue side outputs)
__device__ __forceinline__ void put16(void* base, long idx, float v, int prec) {
  if (prec == PREC_F16) reinterpret_cast<_Float16*>(base)[idx] = PrecTraits<PREC_F16>::cvt(v);
  else reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
}

// byte offset of 16-byte chunk `chunk` (0..7) of tile row `row` in the swizzled LDS image
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
}

// ---- staging of a K-contiguous operand: element (row,k) at base[row*ld + k] -----------------
template <int PREC, int R>
struct StageK {
  static_assert(R % 32 == 0, "tile rows must be a multiple of 32");
  static constexpr int FPC = PrecTraits<PREC>::FPC;
  static constexpr int NCH = R / 32;  // 16-byte LDS chunks written per thread per stage
  const float* rp[NCH];
  f32x4 v[NCH * FPC];
  int c;
  bool vec_ok;

  template <class RM>
  __device__ __forceinline__ void init(const float* base, long ld, RM rmap, int tid) {
    c = tid & 7;
    vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const long mr = rmap((tid >> 3) + 32 * i);
      rp[i] = mr >= 0 ? base + mr * ld : nullptr;
    }
  }
  __device__ __forceinline__ void load(int k0, int K) {
    const int kb = k0 + c * 4 * FPC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
      for (int f = 0; f < FPC; ++f) {
        const int kk = kb + 4 * f;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (rp[i] != nullptr && kk < K) {
          if (vec_ok && kk + 3 < K) {
            x = *reinterpret_cast<const f32x4*>(rp[i] + kk);
          } else {
            x[0] = rp[i][kk];
            if (kk + 1 < K) x[1] = rp[i][kk + 1];
            if (kk + 2 < K) x[2] = rp[i][kk + 2];
            if (kk + 3 < K) x[3] = rp[i][kk + 3];
          }
        }
        v[i * FPC + f] = x;
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      char* p = lds + lds_off((tid >> 3) + 32 * i, c);
      if constexpr (PREC == PREC_F32) {
        *reinterpret_cast<f32x4*>(p) = v[i];
      } else {
        using PT = PrecTraits<PREC>;
        typename PT::h8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = PT::cvt(v[i * 2][e]);
          h[4 + e] = PT::cvt(v[i * 2 + 1][e]);
        }
        *reinterpret_cast<typename PT::h8*>(p) = h;
      }
    }
  }
};

// ---- staging of a row-contiguous operand: element (row,k) at base[k*ld + row] ----------------
// Each thread moves 4(k) x 4(row) blocks and transposes them in registers, so the LDS image is
// the same K-contiguous one StageK writes.
template <int PREC, int R>
struct StageM {
  static_assert(R % 32 == 0, "tile rows must be a multiple of 32");
  static constexpr int BK = PrecTraits<PREC>::BK;
  static constexpr int RQ = R / 4;
  static constexpr int UNITS = (BK / 4) * RQ;
  static constexpr int NU = (UNITS + 255) / 256;
  const float* cp[NU];
  int nv[NU];
  f32x4 v[NU * 4];
  long ld;
  bool vec_ok;

  template <class RM>
  __device__ __forceinline__ void init(const float* base, long ld_, RM rmap, int tid) {
    ld = ld_;
    vec_ok = ((ld_ & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = tid + 256 * i;
      cp[i] = nullptr;
      nv[i] = 0;
      if (id < UNITS) {
        const int rq = id % RQ;
        const long mr0 = rmap(4 * rq);
        if (mr0 >= 0) {
          int n = 1;
          if (rmap(4 * rq + 1) == mr0 + 1) {
            n = 2;
            if (rmap(4 * rq + 2) == mr0 + 2) {
              n = 3;
              if (rmap(4 * rq + 3) == mr0 + 3) n = 4;
            }
          }
          cp[i] = base + mr0;
          nv[i] = n;
        }
      }
    }
  }
  __device__ __forceinline__ void load(int k0, int K) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = tid_() + 256 * i;
      const int kq = id / RQ;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + 4 * kq + j;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (cp[i] != nullptr && kk < K) {
          const float* p = cp[i] + (long)kk * ld;
          if (vec_ok && nv[i] == 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            x = *reinterpret_cast<const f32x4*>(p);
          } else {
            x[0] = p[0];
            if (nv[i] > 1) x[1] = p[1];
            if (nv[i] > 2) x[2] = p[2];
            if (nv[i] > 3) x[3] = p[3];
          }
        }
        v[i * 4 + j] = x;
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = tid + 256 * i;
      if (id < UNITS) {
        const int kq = id / RQ, rq = id % RQ;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = 4 * rq + rr;
          if constexpr (PREC == PREC_F32) {
            f32x4 t = {v[i * 4 + 0][rr], v[i * 4 + 1][rr], v[i * 4 + 2][rr], v[i * 4 + 3][rr]};
            *reinterpret_cast<f32x4*>(lds + lds_off(row, kq)) = t;
          } else {
            using PT = PrecTraits<PREC>;
            typename PT::h4 h = {PT::cvt(v[i * 4 + 0][rr]), PT::cvt(v[i * 4 + 1][rr]), PT::cvt(v[i * 4 + 2][rr]),
                                 PT::cvt(v[i * 4 + 3][rr])};
            *reinterpret_cast<typename PT::h4*>(lds + lds_off(row, kq >> 1) + (kq & 1) * 8) = h;
          }
        }
      }
    }
  }
  static __device__ __forceinline__ int tid_() { return threadIdx.x; }
};

// ---- row-contiguous operand stored in 16 bits (same type as the MFMA operands): element (row,k) at
// base[k*ld + row].  Units of 4(k) x 8(rows): four 16-byte loads, transposed in registers into
// eight 8-byte LDS writes of the canonical K-contiguous image.
template <int PREC, int R>
struct StageM16 {
  static_assert(R % 32 == 0 && PREC != PREC_F32, "16-bit staging");
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  using h4 = typename PT::h4;
  static constexpr int BK = PT::BK;           // 64
  static constexpr int RO = R / 8;            // row-octets per k-row
  static constexpr int UNITS = (BK / 4) * RO;
  static constexpr int NU = (UNITS + 255) / 256;
  const h_t* cp[NU];
  int nv[NU];
  h8 v[NU * 4];
  long ld;
  bool vec_ok;

  template <class RM>
  __device__ __forceinline__ void init(const void* base_, long ld_, RM rmap, int tid) {
    const h_t* base = reinterpret_cast<const h_t*>(base_);
    ld = ld_;
    vec_ok = ((ld_ & 7) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = tid + 256 * i;
      cp[i] = nullptr;
      nv[i] = 0;
      if (id < UNITS) {
        const int ro = id % RO;
        const long mr0 = rmap(8 * ro);
        if (mr0 >= 0) {
          int n = 1;
          while (n < 8 && rmap(8 * ro + n) == mr0 + n) ++n;
          cp[i] = base + mr0;
          nv[i] = n;
        }
      }
    }
  }
  __device__ __forceinline__ void load(int k0, int K) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = threadIdx.x + 256 * i;
      const int kq = id / RO;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + 4 * kq + j;
        h8 x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (h_t)0.0f;
        if (cp[i] != nullptr && kk < K) {
          const h_t* p = cp[i] + (long)kk * ld;
          if (vec_ok && nv[i] == 8 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            x = *reinterpret_cast<const h8*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e < nv[i]) x[e] = p[e];
          }
        }
        v[i * 4 + j] = x;
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int id = tid + 256 * i;
      if (id < UNITS) {
        const int kq = id / RO, ro = id % RO;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = 8 * ro + rr;
          h4 h = {v[i * 4 + 0][rr], v[i * 4 + 1][rr], v[i * 4 + 2][rr], v[i * 4 + 3][rr]};
          *reinterpret_cast<h4*>(lds + lds_off(row, kq >> 1) + (kq & 1) * 8) = h;
        }
      }
    }
  }
};

template <int PREC, int LAY, int R, int SRC16 = 0> struct StageSel;
template <int PREC, int R> struct StageSel<PREC, LAY_KMAJ, R, 0> { using type = StageK<PREC, R>; };
template <int PREC, int R> struct StageSel<PREC, LAY_MMAJ, R, 0> { using type = StageM<PREC, R>; };
template <int PREC, int R> struct StageSel<PREC, LAY_MMAJ, R, 1> { using type = StageM16<PREC, R>; };

// ---- the tile engine ---------------------------------------------------------------------
template <int PREC, int ALAY, int BLAY, int BM, int BN, int WGM, int WGN, int ASRC16 = 0, int BSRC16 = 0>
struct GemmTile {
  static_assert(WGM * WGN == 4, "256-thread workgroups: 4 waves");
  static constexpr int BK = PrecTraits<PREC>::BK;
  static constexpr int WTM = BM / WGM, WTN = BN / WGN;  // per-wave extent
  static constexpr int TM = WTM / 16, TN = WTN / 16;    // 16x16 MFMA tiles per wave
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");
  static constexpr int LDS_BYTES = (BM + BN) * 128;

  // acc[tm][tn][i] <-> C[row = wm*WTM + tm*16 + 4*(lane>>4) + i][col = wn*WTN + tn*16 + (lane&15)]
  // rma / rmb map a tile row (0..BM-1 / 0..BN-1) to the operand's memory row, or -1 (zero row).
  template <class RMA, class RMB>
  static __device__ __forceinline__ void run(f32x4 (&acc)[TM][TN], const void* A_, long lda, RMA rma,
                                             const void* B_, long ldb, RMB rmb, int K, char* lds) {
    using AP = typename std::conditional<ASRC16 != 0, const void*, const float*>::type;
    using BP = typename std::conditional<BSRC16 != 0, const void*, const float*>::type;
    AP A = reinterpret_cast<AP>(A_);
    BP B = reinterpret_cast<BP>(B_);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    char* ldsA = lds;
    char* ldsB = lds + BM * 128;

    typename StageSel<PREC, ALAY, BM, ASRC16>::type sa;
    typename StageSel<PREC, BLAY, BN, BSRC16>::type sb;
    sa.init(A, lda, rma, tid);
    sb.init(B, ldb, rmb, tid);

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (K <= 0) return;
    sa.load(0, K);
    sb.load(0, K);
    for (int k0 = 0; k0 < K; k0 += BK) {
      sa.store(ldsA, tid);
      sb.store(ldsB, tid);
      __syncthreads();
      if (k0 + BK < K) {  // next stage's loads fly underneath this stage's MFMAs
        sa.load(k0 + BK, K);
        sb.load(k0 + BK, K);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (PREC == PREC_F32) {
          f32x4 a[TM], b[TN];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            a[tm] = *reinterpret_cast<const f32x4*>(ldsA + lds_off(wm * WTM + tm * 16 + lr, 4 * s + lq));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            b[tn] = *reinterpret_cast<const f32x4*>(ldsB + lds_off(wn * WTN + tn * 16 + lr, 4 * s + lq));
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
        } else {
          using PT = PrecTraits<PREC>;
          typename PT::h8 a[TM], b[TN];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            a[tm] = *reinterpret_cast<const typename PT::h8*>(ldsA + lds_off(wm * WTM + tm * 16 + lr, 4 * s + lq));
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            b[tn] = *reinterpret_cast<const typename PT::h8*>(ldsB + lds_off(wn * WTN + tn * 16 + lr, 4 * s + lq));
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = PT::mfma(a[tm], b[tn], acc[tm][tn]);
        }
      }
      __syncthreads();
    }
  }

  // visit every accumulator element owned by this lane: f(tile_row, tile_col, value)
  template <class F>
  static __device__ __forceinline__ void for_each(f32x4 (&acc)[TM][TN], F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          f(wm * WTM + tm * 16 + 4 * (lane >> 4) + i, wn * WTN + tn * 16 + (lane & 15), acc[tm][tn][i]);
  }
};

}  // namespace ark
