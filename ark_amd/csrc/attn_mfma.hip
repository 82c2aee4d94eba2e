// Flash-style multi-head self-attention on the matrix cores, for sequences longer than the one-wave-per-head kernels of
// txf.hip serve (L > 16: the wd-* shapes of the Transformer variants, L up to a few hundred).  Scores and probabilities
// never exist in memory: the forward keeps the online softmax of a 64-query tile in registers and writes the context and
// one log-sum-exp per (batch, head, query); the backward recomputes the probabilities from it, twice --
//
//   ark_attn_flash_fwd       workgroup (b, h, 64 queries), key / value tiles of 64 stream through LDS:
//                            S^T = K_tile Q^T (transposed on purpose, as in vocab_ce.hip: the query is on the lane, the keys
//                            in its registers, so max / sum are per-lane scalars and exp(S^T - m) IS the B operand of the
//                            second product), O^T += V_tile^T P^T with V fragments read transposed (ds_read_b64_tr_b16)
//   ark_attn_flash_bwd  (1)  delta[b, h, i] = dO[i] . O[i]
//                       (2)  dQ: workgroup (b, h, 64 queries), K / V tiles stream: S^T and dP^T = V dO^T on the matrix
//                            cores, dS^T = P^T o (M o dP^T - delta), dQ^T += K_tile^T dS^T (K read plain AND transposed)
//                       (3)  dK, dV: workgroup (b, h, 64 keys), Q / dO tiles stream: S = Q_tile K^T, dP = dO_tile V^T,
//                            dV^T += dO_tile^T (P o M), dK^T += Q_tile^T dS
//
// Operands are converted from the fp32 activations while they are staged (registers -> swizzled 16-bit LDS images, zero
// padded to whole 64-wide k-images: any head width that is a multiple of 32 up to 384); products in the forward type
// (forward) / backward type (backward), softmax statistics, delta and every accumulator in fp32.  Dropout on the
// probabilities uses the SAME counter hash and element index ((b, h, i, j) of a [B, H, L, L] array that is never
// materialised) as the vector-unit kernels of txf.hip, so the two paths draw identical masks.
// Reference ops replaced: F.scaled_dot_product_attention inside nn.TransformerEncoderLayer / nn.TransformerDecoderLayer
// self-attention (kgvae/model/models.py:73-74, 104-105, 355-356) and its autograd.
#include "dma_core.h"
#include "../../include/ark_amd.h"

namespace ark {

typedef short fa16x4 __attribute__((ext_vector_type(4)));
typedef short fa16x8 __attribute__((ext_vector_type(8)));

struct FlashArgs {
  const float* qkv;     // [L*B, 3D] time-major rows (row = t*B + b): q | k | v
  float* out;           // [L*B, D] context (forward output; the backward reads it for delta)
  float* lse;           // [B, H, Lp] log2-domain log-sum-exp of the scaled scores, Lp = L rounded up to 64
  const float* dout;    // backward: gradient of `out`
  float* delta;         // [B, H, Lp] backward scratch: dO . O
  float* dqkv;          // backward output [L*B, 3D]
  const float* hyper;
  const unsigned char* kmask;   // [B, L] 1 = key may be attended to (nullable)
  uint64_t seed;
  float drop_p, scale;
  int B, L, D, H, dh, causal, Lp;
};

constexpr int kFaImg = 64 * 128;   // one k-image: 64 rows x 128 B
constexpr float kFaLog2e = 1.4426950408889634f;

// rows [row0, row0 + 64) of one head's q, k, v or dO (element (t, d) at src[t * stride + d]) -> DCH swizzled 16-bit
// k-images; rows beyond L and columns beyond dh are zero
template <int PREC, int DCH>
__device__ __forceinline__ void fa_stage(const float* src, long stride, int row0, int L, int dh, char* img, int tid) {
  using PT = PrecTraits<PREC>;
  using h8 = typename PT::h8;
  constexpr int CPR = 8 * DCH;
#pragma unroll
  for (int i = 0; i < (64 * CPR) / 256; ++i) {
    const int ch = tid + 256 * i;
    const int row = ch / CPR, cc = ch % CPR;
    const int t = row0 + row;
    h8 v;
    if (t < L && cc * 8 < dh) {
      const float* g = src + (long)t * stride + cc * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(g), b = *reinterpret_cast<const f32x4*>(g + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = PT::cvt(a[e]); v[4 + e] = PT::cvt(b[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = PT::cvt(0.f);
    }
    *reinterpret_cast<h8*>(img + (cc >> 3) * kFaImg + lds_off(row, cc & 7)) = v;
  }
}

// one row of a head as the B operand of a product over the head width: x[32 ks + 8 q .. + 7], zero beyond dh / for a dead row
template <int PREC, int KS>
__device__ __forceinline__ void fa_row_frags(const float* row, bool live, int dh, int q, typename PrecTraits<PREC>::h8 (&f)[KS]) {
  using PT = PrecTraits<PREC>;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = 32 * ks + 8 * q;
    if (live && d0 < dh) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(row + d0), b = *reinterpret_cast<const f32x4*>(row + d0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[ks][e] = PT::cvt(a[e]); f[ks][4 + e] = PT::cvt(b[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[ks][e] = PT::cvt(0.f);
    }
  }
}

// A fragment of a product over the head width: image rows `row` (this lane's), k-step ks
template <class H8>
__device__ __forceinline__ H8 fa_frag(const char* img, int row, int ks, int q) {
  return *reinterpret_cast<const H8*>(img + (ks >> 1) * kFaImg + lds_off(row, 4 * (ks & 1) + q));
}

// transposed A fragment of a product over the tile's ROWS: element j of lane (i16, q) = image[row(q, j)][16 dt + i16],
// row(q, j) = rbase + 16 (j >> 2) + 4 q + (j & 3) -- the k permutation of the accumulator-built B operand (vocab_ce.hip)
template <class H8>
__device__ __forceinline__ H8 fa_tr_frag(const char* img0, int rbase, int dt, int lane) {
  const int i16 = lane & 15, q = lane >> 4;
  const int row = rbase + 4 * q + (i16 >> 2);
  const int chunk = 2 * (dt & 3) + ((i16 & 3) >> 1), within = (i16 & 1) * 8;
  const char* b = img0 + (dt >> 2) * kFaImg;
  const fa16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) fa16x4*)(b + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + within));
  const int row2 = row + 16;
  const fa16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) fa16x4*)(b + row2 * 128 + ((chunk ^ ((row2 >> 1) & 7)) << 4) + within));
  const fa16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(H8, v);
}

// dropout keep-scale of the four consecutive elements idx .. idx + 3 of the (virtual) probability array
__device__ __forceinline__ f32x4 fa_keep4(const DropCtx& dc, long idx) {
  const int off = (int)(idx & 3);
  const f32x4 a = dropout_quad(dc, (uint64_t)idx >> 2);
  if (off == 0) return a;
  const f32x4 b = dropout_quad(dc, ((uint64_t)idx >> 2) + 1);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = off + i;
    r[i] = e == 1 ? a[1] : e == 2 ? a[2] : e == 3 ? a[3] : e == 4 ? b[0] : e == 5 ? b[1] : b[2];
  }
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// forward (MODE 0) and the query side of the backward (MODE 1): workgroup (b, h, 64 queries); wave w owns queries
// 16 w .. 16 w + 15 (lane & 15); key / value tiles of 64 stream through LDS
template <int PREC, int DCH, int MODE>
__global__ __launch_bounds__(256) void attn_flash_q_kernel(FlashArgs p) {
  using PT = PrecTraits<PREC>;
  using h8 = typename PT::h8;
  constexpr int KS = 2 * DCH, DT = 4 * DCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgK = smem;
  char* imgV = smem + DCH * kFaImg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H, qt = blockIdx.y;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh;
  const long rs = 3L * D, stride = (long)B * rs;
  const float* qbase = p.qkv + (long)b * rs + h * dh;      // element (t, d) of q at qbase[t * stride + d]
  const int qi = qt * 64 + 16 * wave + c;
  const bool qlive = qi < L;
  const int qc = qlive ? qi : L - 1;
  h8 yf[KS];
  fa_row_frags<PREC, KS>(qbase + (long)qc * stride, qlive, dh, q, yf);
  h8 dof[MODE == 1 ? KS : 1];
  float lse2 = 0.f, delta = 0.f;
  if constexpr (MODE == 1) {
    fa_row_frags<PREC, KS>(p.dout + ((long)qc * B + b) * D + h * dh, qlive, dh, q, dof);
    lse2 = p.lse[((long)b * p.H + h) * p.Lp + qc];
    delta = p.delta[((long)b * p.H + h) * p.Lp + qc];
  }
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  const float sl2 = p.scale * kFaLog2e;
  const long prow = (((long)b * p.H + h) * L + qc) * L;   // this query's row of the virtual [B, H, L, L] array

  float m2 = -INFINITY, lsum = 0.f;
  f32x4 U[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) U[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = p.causal ? qt + 1 : (L + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    fa_stage<PREC, DCH>(qbase + D, stride, kt * 64, L, dh, imgK, tid);
    fa_stage<PREC, DCH>(qbase + 2 * D, stride, kt * 64, L, dh, imgV, tid);
    __syncthreads();
    f32x4 S[4], G[MODE == 1 ? 4 : 1];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt) {
      S[vt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == 1) G[vt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        S[vt] = PT::mfma(fa_frag<h8>(imgK, 16 * vt + c, ks, q), yf[ks], S[vt]);
        if constexpr (MODE == 1) G[vt] = PT::mfma(fa_frag<h8>(imgV, 16 * vt + c, ks, q), dof[ks], G[vt]);
      }
    // element (vt, i) of this lane: key kt*64 + 16 vt + 4 q + i, query qi
    float sv[16];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt) {
      const int k0 = kt * 64 + 16 * vt + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kj = k0 + i;
        bool ok = kj < L && (!p.causal || kj <= qi);
        if (ok && p.kmask) ok = p.kmask[(long)b * L + kj] != 0;
        sv[vt * 4 + i] = ok ? S[vt][i] * sl2 : -INFINITY;
      }
    }
    float pj[16];
    if constexpr (MODE == 0) {
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) mt = fmaxf(mt, sv[j]);
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m2, mt);
      const float ref = (mn == -INFINITY) ? 0.f : mn;
      const float sc = __builtin_amdgcn_exp2f(m2 - ref);
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { pj[j] = __builtin_amdgcn_exp2f(sv[j] - ref); ps += pj[j]; }
      lsum = lsum * sc + ps;
      if (__any(sc != 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) U[dt] *= sc;
      }
      m2 = mn;
      if (drop) {
#pragma unroll
        for (int vt = 0; vt < 4; ++vt) {
          const f32x4 k4 = fa_keep4(dc, prow + kt * 64 + 16 * vt + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) pj[vt * 4 + i] *= k4[i];
        }
      }
    } else {
      // P from the stored log-sum-exp; dS = P o (M o dP - delta), scaled for the dQ product at the end
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        f32x4 k4 = f32x4{1.f, 1.f, 1.f, 1.f};
        if (drop) k4 = fa_keep4(dc, prow + kt * 64 + 16 * vt + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (a masked key has sv = -inf -> 0; a query whose keys are ALL masked -- an all-PAD graph -- also has lse2 = -inf:
          //  -inf - -inf would be NaN, so the masked case is decided before the subtraction)
          const float sve = sv[vt * 4 + i];
          const float pr = sve == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sve - lse2);
          pj[vt * 4 + i] = pr * (G[vt][i] * k4[i] - delta);
        }
      }
    }
    h8 pf0, pf1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pf0[j] = PT::cvt(pj[j]); pf1[j] = PT::cvt(pj[8 + j]); }
    const char* imgT = MODE == 0 ? imgV : imgK;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      U[dt] = PT::mfma(fa_tr_frag<h8>(imgT, 0, dt, lane), pf0, U[dt]);
      U[dt] = PT::mfma(fa_tr_frag<h8>(imgT, 32, dt, lane), pf1, U[dt]);
    }
  }
  if constexpr (MODE == 0) {
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (!qlive) return;
    const float il = lsum > 0.f ? 1.0f / lsum : 0.f;   // (every key masked: context 0, lse -inf -- not 0 * inf)
    float* o = p.out + ((long)qi * B + b) * D + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      if (16 * dt < dh) *reinterpret_cast<f32x4*>(o + 16 * dt + 4 * q) = U[dt] * il;
    if (q == 0) p.lse[((long)b * p.H + h) * p.Lp + qi] = m2 + __builtin_amdgcn_logf(lsum);   // (v_log_f32 is log2)
  } else {
    if (!qlive) return;
    float* o = p.dqkv + ((long)qi * B + b) * rs + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      if (16 * dt < dh) *reinterpret_cast<f32x4*>(o + 16 * dt + 4 * q) = U[dt] * p.scale;
  }
}

// the key side of the backward: workgroup (b, h, 64 keys); wave w owns keys 16 w .. 16 w + 15; query / dO tiles stream
template <int PREC, int DCH>
__global__ __launch_bounds__(256) void attn_flash_kv_kernel(FlashArgs p) {
  using PT = PrecTraits<PREC>;
  using h8 = typename PT::h8;
  constexpr int KS = 2 * DCH, DT = 4 * DCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgQ = smem;
  char* imgO = smem + DCH * kFaImg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H, kt = blockIdx.y;
  const int B = p.B, L = p.L, D = p.D, dh = p.dh;
  const long rs = 3L * D, stride = (long)B * rs;
  const float* qbase = p.qkv + (long)b * rs + h * dh;
  const int ki = kt * 64 + 16 * wave + c;
  bool klive = ki < L;
  const int kc = klive ? ki : L - 1;
  h8 kf[KS], vf[KS];
  fa_row_frags<PREC, KS>(qbase + D + (long)kc * stride, klive, dh, q, kf);
  fa_row_frags<PREC, KS>(qbase + 2 * D + (long)kc * stride, klive, dh, q, vf);
  if (klive && p.kmask) klive = p.kmask[(long)b * L + ki] != 0;
  const bool drop = p.drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(p.seed, p.hyper, p.drop_p);
  const float sl2 = p.scale * kFaLog2e;
  const float* lsep = p.lse + ((long)b * p.H + h) * p.Lp;
  const float* delp = p.delta + ((long)b * p.H + h) * p.Lp;
  const long pbh = ((long)b * p.H + h) * L;

  f32x4 dK[DT], dV[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nqt = (L + 63) / 64;
  for (int qt = p.causal ? kt : 0; qt < nqt; ++qt) {
    __syncthreads();
    fa_stage<PREC, DCH>(qbase, stride, qt * 64, L, dh, imgQ, tid);
    fa_stage<PREC, DCH>(p.dout + (long)b * D + h * dh, (long)B * D, qt * 64, L, dh, imgO, tid);
    __syncthreads();
    f32x4 S[4], G[4];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt) { S[vt] = f32x4{0.f, 0.f, 0.f, 0.f}; G[vt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        S[vt] = PT::mfma(fa_frag<h8>(imgQ, 16 * vt + c, ks, q), kf[ks], S[vt]);
        G[vt] = PT::mfma(fa_frag<h8>(imgO, 16 * vt + c, ks, q), vf[ks], G[vt]);
      }
    // element (vt, i) of this lane: query qt*64 + 16 vt + 4 q + i, key ki
    float pd[16], ds[16];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt) {
      const int q0 = qt * 64 + 16 * vt + 4 * q;   // (Lp is a multiple of 64: the quad is inside the arrays)
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsep + q0), d4 = *reinterpret_cast<const f32x4*>(delp + q0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qj = q0 + i;
        const bool ok = klive && qj < L && (!p.causal || ki <= qj);
        const float pr = ok ? __builtin_amdgcn_exp2f(S[vt][i] * sl2 - l4[i]) : 0.f;
        const float keep = (drop && ok) ? dropout_one(dc, (uint64_t)((pbh + qj) * L + ki)) : 1.0f;
        pd[vt * 4 + i] = pr * keep;
        ds[vt * 4 + i] = ok ? pr * (G[vt][i] * keep - d4[i]) : 0.f;   // (the padded tail of lse / delta is never written)
      }
    }
    h8 pf0, pf1, sf0, sf1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf0[j] = PT::cvt(pd[j]); pf1[j] = PT::cvt(pd[8 + j]);
      sf0[j] = PT::cvt(ds[j]); sf1[j] = PT::cvt(ds[8 + j]);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dV[dt] = PT::mfma(fa_tr_frag<h8>(imgO, 0, dt, lane), pf0, dV[dt]);
      dV[dt] = PT::mfma(fa_tr_frag<h8>(imgO, 32, dt, lane), pf1, dV[dt]);
      dK[dt] = PT::mfma(fa_tr_frag<h8>(imgQ, 0, dt, lane), sf0, dK[dt]);
      dK[dt] = PT::mfma(fa_tr_frag<h8>(imgQ, 32, dt, lane), sf1, dK[dt]);
    }
  }
  if (ki >= L) return;
  float* o = p.dqkv + ((long)ki * B + b) * rs + D + h * dh;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
    if (16 * dt < dh) {
      *reinterpret_cast<f32x4*>(o + 16 * dt + 4 * q) = dK[dt] * p.scale;
      *reinterpret_cast<f32x4*>(o + D + 16 * dt + 4 * q) = dV[dt];
    }
}

// delta[b, h, i] = dO[i] . O[i] over the head's columns: one wave per (row, head)
__global__ __launch_bounds__(256) void attn_flash_delta_kernel(FlashArgs p) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long n = (long)p.L * p.B * p.H;
  if (wid >= n) return;
  const int h = (int)(wid % p.H);
  const long row = wid / p.H;
  const int t = (int)(row / p.B), b = (int)(row % p.B);
  const float* o = p.out + row * p.D + h * p.dh;
  const float* g = p.dout + row * p.D + h * p.dh;
  float a = 0.f;
  for (int d = lane; d < p.dh; d += 64) a += o[d] * g[d];
  a = wave_sum(a);
  if (lane == 0) p.delta[((long)b * p.H + h) * p.Lp + t] = a;
}

template <class K>
static void fa_allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static int fa_check(const FlashArgs& p) {
  if (p.B <= 0 || p.L <= 0 || p.D <= 0 || p.H <= 0 || p.D % p.H != 0) return ARK_ERR_ARG;
  if (p.dh % 32 != 0 || p.dh > 384) return ARK_ERR_SHAPE;
  if (p.drop_p < 0.f || p.drop_p >= 1.f || (p.drop_p > 0.f && !p.hyper)) return ARK_ERR_ARG;
  if ((double)p.B * p.H * p.L * p.L >= 9.0e18) return ARK_ERR_SHAPE;
  return 0;
}

template <int PREC, int DCH>
static int fa_launch(const FlashArgs& p, int what, hipStream_t st) {
  constexpr int LDS = 2 * DCH * kFaImg;
  const dim3 grid((unsigned)(p.B * p.H), (unsigned)((p.L + 63) / 64));
  if (what == 0) {
    static bool once = (fa_allow_lds(attn_flash_q_kernel<PREC, DCH, 0>, LDS), true); (void)once;
    hipLaunchKernelGGL((attn_flash_q_kernel<PREC, DCH, 0>), grid, dim3(256), LDS, st, p);
  } else {
    static bool once = (fa_allow_lds(attn_flash_q_kernel<PREC, DCH, 1>, LDS), fa_allow_lds(attn_flash_kv_kernel<PREC, DCH>, LDS), true); (void)once;
    const long n = (long)p.L * p.B * p.H;
    hipLaunchKernelGGL(attn_flash_delta_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, p);
    hipLaunchKernelGGL((attn_flash_q_kernel<PREC, DCH, 1>), grid, dim3(256), LDS, st, p);
    hipLaunchKernelGGL((attn_flash_kv_kernel<PREC, DCH>), grid, dim3(256), LDS, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC>
static int fa_dispatch(const FlashArgs& p, int what, hipStream_t st) {
  const int dch = (p.dh + 63) / 64;
  switch (dch) {
    case 1: return fa_launch<PREC, 1>(p, what, st);
    case 2: return fa_launch<PREC, 2>(p, what, st);
    case 3: return fa_launch<PREC, 3>(p, what, st);
    case 4: return fa_launch<PREC, 4>(p, what, st);
    case 5: return fa_launch<PREC, 5>(p, what, st);
    case 6: return fa_launch<PREC, 6>(p, what, st);
    default: return ARK_ERR_SHAPE;
  }
}

}  // namespace ark

extern "C" long ark_attn_flash_stat_floats(int B, int L, int n_heads) { return (long)B * n_heads * ((L + 63) / 64 * 64); }

extern "C" int ark_attn_flash_fwd(int prec, const float* qkv, float* out, float* lse, const unsigned char* kmask, int B, int L, int D,
                                  int n_heads, int causal, float drop_p, uint64_t seed, const float* hyper, void* stream) {
  using namespace ark;
  if (!qkv || !out || !lse) return ARK_ERR_ARG;
  FlashArgs p{qkv, out, lse, nullptr, nullptr, nullptr, hyper, kmask, seed, drop_p, 0.f, B, L, D, n_heads,
              n_heads > 0 ? D / n_heads : 0, causal, (L + 63) / 64 * 64};
  const int rc = fa_check(p);
  if (rc) return rc;
  p.scale = 1.0f / sqrtf((float)p.dh);
  if (prec == PREC_F16) return fa_dispatch<PREC_F16>(p, 0, (hipStream_t)stream);
  if (prec == PREC_BF16) return fa_dispatch<PREC_BF16>(p, 0, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_attn_flash_bwd(int prec, const float* qkv, const float* out, const float* lse, const float* dout, float* delta,
                                  float* dqkv, const unsigned char* kmask, int B, int L, int D, int n_heads, int causal, float drop_p,
                                  uint64_t seed, const float* hyper, void* stream) {
  using namespace ark;
  if (!qkv || !out || !lse || !dout || !delta || !dqkv) return ARK_ERR_ARG;
  FlashArgs p{qkv, const_cast<float*>(out), const_cast<float*>(lse), dout, delta, dqkv, hyper, kmask, seed, drop_p, 0.f, B, L, D,
              n_heads, n_heads > 0 ? D / n_heads : 0, causal, (L + 63) / 64 * 64};
  const int rc = fa_check(p);
  if (rc) return rc;
  p.scale = 1.0f / sqrtf((float)p.dh);
  if (prec == PREC_F16) return fa_dispatch<PREC_F16>(p, 1, (hipStream_t)stream);
  if (prec == PREC_BF16) return fa_dispatch<PREC_BF16>(p, 1, (hipStream_t)stream);
  return ARK_ERR_ARG;
}
