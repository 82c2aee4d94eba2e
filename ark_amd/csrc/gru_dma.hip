// GRU recurrence, LDS-DMA edition (see dma_core.h): same math and epilogues as gru.hip, but the
// recurrent operands arrive as 16-bit copies (h_{t-1} written by the previous cell's epilogue,
// W_hh / W_hh^T shadows refreshed after every optimiser step by ark_gru_weight_shadows), so the
// whole K panel is in flight at once and no VGPRs or conversions are spent on staging.
// Used when d_model % 64 == 0 and the precision is 16-bit; gru.hip remains the exact-fp32 path.
#include "dma_core.h"
#include "../../include/ark_amd.h"

namespace ark {

struct GruFwdDmaArgs {
  const void* h_prev16; const void* w_hh16; const float* h_prev; const float* b_hh; const float* gi;
  float* h_out; void* h_out16; float* h_drop; const float* drop_mask;
  float* sr; float* sz; float* sn; float* shn;
  int B, D;
};

template <int PREC, int KS>
__global__ __launch_bounds__(256) void gru_cell_fwd_dma_kernel(GruFwdDmaArgs p) {
  constexpr int BM = 32, BU = 32, BN = 3 * BU;
  using G = DmaTile<PREC, BM, BN, KS, 2, 2>;  // wave tile 16 x 48 (16 units x 3 gates)
  using h_t = typename G::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int units_tiles = p.D / BU;
  const int m0 = (blockIdx.x / units_tiles) * BM, u0 = (blockIdx.x % units_tiles) * BU;
  const int B = p.B, D = p.D;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, reinterpret_cast<const h_t*>(p.h_prev16), D, [=](int r) -> long { return (long)min(m0 + r, B - 1); },
         reinterpret_cast<const h_t*>(p.w_hh16), D,
         [=](int j) -> long { return (long)((j >> 4) % 3) * D + u0 + (j / 48) * 16 + (j & 15); }, D, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int u = u0 + wn * 16 + (lane & 15);
  const float bhr = p.b_hh[u], bhz = p.b_hh[D + u], bhn = p.b_hh[2 * D + u];
  h_t* h16 = reinterpret_cast<h_t*>(p.h_out16);
#pragma unroll
  for (int tm = 0; tm < G::TM; ++tm) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * G::WTM + tm * 16 + 4 * (lane >> 4) + i;
      if (row >= B) continue;
      const float* gi = p.gi + (long)row * 3 * D;
      const long o = (long)row * D + u;
      const float r = sigmoidf_(gi[u] + acc[tm][0][i] + bhr);
      const float z = sigmoidf_(gi[D + u] + acc[tm][1][i] + bhz);
      const float hn = acc[tm][2][i] + bhn;
      const float n = tanhf(gi[2 * D + u] + r * hn);
      const float hp = p.h_prev[o];
      const float h = (1.0f - z) * n + z * hp;
      p.h_out[o] = h;
      h16[o] = G::PT::cvt(h);
      if (p.h_drop) p.h_drop[o] = h * p.drop_mask[o];
      if (p.sr) { p.sr[o] = r; p.sz[o] = z; p.sn[o] = n; p.shn[o] = hn; }
    }
  }
}

struct GruBwdDmaArgs {
  const void* dgh_next16; const void* w_hhT16; const float* dy; float* carry;
  const float* sr; const float* sz; const float* sn; const float* shn; const float* h_prev;
  float* dgi; float* dgh; void* dgh16;
  float* dh0; int dh0_accumulate;
  int B, D, first, final_;
};

template <int PREC, int KS>
__global__ __launch_bounds__(256) void gru_cell_bwd_dma_kernel(GruBwdDmaArgs p) {
  constexpr int BM = 32, BN = 64;
  using G = DmaTile<PREC, BM, BN, KS, 2, 2>;
  using h_t = typename G::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = p.D / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int B = p.B, D = p.D;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, reinterpret_cast<const h_t*>(p.dgh_next16), 3L * D, [=](int r) -> long { return (long)min(m0 + r, B - 1); },
         reinterpret_cast<const h_t*>(p.w_hhT16), 3L * D, [=](int r) -> long { return (long)(n0 + r); },
         p.first ? 0 : 3 * D, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  h_t* dgh16 = reinterpret_cast<h_t*>(p.dgh16);
#pragma unroll
  for (int tm = 0; tm < G::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::TN; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * G::WTM + tm * 16 + 4 * (lane >> 4) + i;
        const int u = n0 + wn * G::WTN + tn * 16 + (lane & 15);
        if (row >= B) continue;
        const long o = (long)row * D + u;
        float dh = acc[tm][tn][i];
        if (!p.first) dh += p.carry[o];
        if (p.final_) {
          p.dh0[o] = p.dh0_accumulate ? p.dh0[o] + dh : dh;
          continue;
        }
        if (p.dy) dh += p.dy[o];
        const float r = p.sr[o], z = p.sz[o], n = p.sn[o], hn = p.shn[o], hp = p.h_prev[o];
        const float dn_pre = dh * (1.0f - z) * (1.0f - n * n);
        const float dz_pre = dh * (hp - n) * z * (1.0f - z);
        const float dr_pre = dn_pre * hn * r * (1.0f - r);
        p.carry[o] = dh * z;
        const long g = (long)row * 3 * D + u;
        p.dgi[g] = dr_pre; p.dgi[g + D] = dz_pre; p.dgi[g + 2 * D] = dn_pre;
        p.dgh[g] = dr_pre; p.dgh[g + D] = dz_pre; p.dgh[g + 2 * D] = dn_pre * r;
        dgh16[g] = G::PT::cvt(dr_pre); dgh16[g + D] = G::PT::cvt(dz_pre); dgh16[g + 2 * D] = G::PT::cvt(dn_pre * r);
      }
}

// 16-bit shadows of the recurrent weights of every layer in one launch:
//   w16 [n_layers][3D][D]  (forward-cell B operand)   and   wT16 [n_layers][D][3D] (backward-cell B operand)
// 32x32 tiles transposed through LDS so both outputs are written in full lines.
template <int PF, int PB>
__global__ __launch_bounds__(256) void weight_shadow_kernel(const float* __restrict__ w, long layer_stride, void* w16_, void* wT16_,
                                                            int D) {
  using HF = typename PrecTraits<PF>::h_t;
  using HB = typename PrecTraits<PB>::h_t;
  __shared__ float tile[32][33];
  const int l = blockIdx.z;
  const float* src = w + l * layer_stride;
  HF* w16 = reinterpret_cast<HF*>(w16_) + (long)l * 3 * D * D;
  HB* wT16 = reinterpret_cast<HB*>(wT16_) + (long)l * 3 * D * D;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;  // rows of W (3D), cols of W (D)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const float v = src[(long)(r0 + i) * D + c0 + tx];
    tile[i][tx] = v;
    w16[(long)(r0 + i) * D + c0 + tx] = PrecTraits<PF>::cvt(v);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) wT16[(long)(c0 + i) * 3 * D + r0 + tx] = PrecTraits<PB>::cvt(tile[tx][i]);
}

template <int PREC>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ x, void* out_, long n) {
  using H = typename PrecTraits<PREC>::h_t;
  H* out = reinterpret_cast<H*>(out_);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = PrecTraits<PREC>::cvt(x[i]);
}

// dynamic LDS above 64 KB must be opted into once per kernel
template <class K>
static void allow_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int PREC>
static int launch_fwd_dma(const GruFwdDmaArgs& p, hipStream_t st) {
  const unsigned grid = (unsigned)(((p.B + 31) / 32) * (p.D / 32));
  if (p.D % 256 == 0) {
    using G = DmaTile<PREC, 32, 96, 256, 2, 2>;
    static bool once = (allow_lds(gru_cell_fwd_dma_kernel<PREC, 256>, G::LDS_BYTES), true); (void)once;
    hipLaunchKernelGGL((gru_cell_fwd_dma_kernel<PREC, 256>), dim3(grid), dim3(256), G::LDS_BYTES, st, p);
  } else if (p.D % 128 == 0) {
    using G = DmaTile<PREC, 32, 96, 128, 2, 2>;
    static bool once = (allow_lds(gru_cell_fwd_dma_kernel<PREC, 128>, G::LDS_BYTES), true); (void)once;
    hipLaunchKernelGGL((gru_cell_fwd_dma_kernel<PREC, 128>), dim3(grid), dim3(256), G::LDS_BYTES, st, p);
  } else {
    using G = DmaTile<PREC, 32, 96, 64, 2, 2>;
    hipLaunchKernelGGL((gru_cell_fwd_dma_kernel<PREC, 64>), dim3(grid), dim3(256), G::LDS_BYTES, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC>
static int launch_bwd_dma(const GruBwdDmaArgs& p, hipStream_t st) {
  const unsigned grid = (unsigned)(((p.B + 31) / 32) * (p.D / 64));
  if (p.D % 128 == 0) {  // 3D % 384 == 0
    using G = DmaTile<PREC, 32, 64, 384, 2, 2>;
    static bool once = (allow_lds(gru_cell_bwd_dma_kernel<PREC, 384>, G::LDS_BYTES), true); (void)once;
    hipLaunchKernelGGL((gru_cell_bwd_dma_kernel<PREC, 384>), dim3(grid), dim3(256), G::LDS_BYTES, st, p);
  } else {               // D % 64 == 0 -> 3D % 192 == 0
    using G = DmaTile<PREC, 32, 64, 192, 2, 2>;
    static bool once = (allow_lds(gru_cell_bwd_dma_kernel<PREC, 192>, G::LDS_BYTES), true); (void)once;
    hipLaunchKernelGGL((gru_cell_bwd_dma_kernel<PREC, 192>), dim3(grid), dim3(256), G::LDS_BYTES, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

extern "C" int ark_gru_cell_fwd_dma(int prec, const void* h_prev16, const void* w_hh16, const float* h_prev,
                                    const float* b_hh, const float* gi, float* h_out, void* h_out16, float* h_drop,
                                    const float* drop_mask, float* save_r, float* save_z, float* save_n, float* save_hn,
                                    int B, int D, void* stream) {
  using namespace ark;
  if (!h_prev16 || !w_hh16 || !h_prev || !b_hh || !gi || !h_out || !h_out16 || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0) return ARK_ERR_SHAPE;
  if (h_drop && !drop_mask) return ARK_ERR_ARG;
  if (save_r && (!save_z || !save_n || !save_hn)) return ARK_ERR_ARG;
  GruFwdDmaArgs p{h_prev16, w_hh16, h_prev, b_hh, gi, h_out, h_out16, h_drop, drop_mask, save_r, save_z, save_n, save_hn, B, D};
  if (prec == PREC_F16) return launch_fwd_dma<PREC_F16>(p, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_fwd_dma<PREC_BF16>(p, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_gru_cell_bwd_dma(int prec, const void* dgh_next16, const void* w_hhT16, const float* dy, float* carry,
                                    const float* save_r, const float* save_z, const float* save_n, const float* save_hn,
                                    const float* h_prev, float* dgi, float* dgh, void* dgh16, int B, int D, int first,
                                    void* stream) {
  using namespace ark;
  if (!w_hhT16 || !carry || !save_r || !save_z || !save_n || !save_hn || !h_prev || !dgi || !dgh || !dgh16 || B <= 0 || D <= 0)
    return ARK_ERR_ARG;
  if (!first && !dgh_next16) return ARK_ERR_ARG;
  if (D % 64 != 0) return ARK_ERR_SHAPE;
  GruBwdDmaArgs p{dgh_next16 ? dgh_next16 : dgh16, w_hhT16, dy, carry, save_r, save_z, save_n, save_hn, h_prev, dgi, dgh,
                  dgh16, nullptr, 0, B, D, first ? 1 : 0, 0};
  if (prec == PREC_F16) return launch_bwd_dma<PREC_F16>(p, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_bwd_dma<PREC_BF16>(p, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_gru_h0_bwd_dma(int prec, const void* dgh0_16, const void* w_hhT16, const float* carry, float* dh0,
                                  int accumulate, int B, int D, void* stream) {
  using namespace ark;
  if (!dgh0_16 || !w_hhT16 || !carry || !dh0 || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 64 != 0) return ARK_ERR_SHAPE;
  GruBwdDmaArgs p{dgh0_16, w_hhT16, nullptr, const_cast<float*>(carry), nullptr, nullptr, nullptr, nullptr, nullptr,
                  nullptr, nullptr, nullptr, dh0, accumulate ? 1 : 0, B, D, 0, 1};
  if (prec == PREC_F16) return launch_bwd_dma<PREC_F16>(p, (hipStream_t)stream);
  if (prec == PREC_BF16) return launch_bwd_dma<PREC_BF16>(p, (hipStream_t)stream);
  return ARK_ERR_ARG;
}

extern "C" int ark_gru_weight_shadows(int prec_fwd, int prec_bwd, const float* w_hh_l0, int64_t layer_stride, void* w16,
                                      void* wT16, int D, int n_layers, void* stream) {
  using namespace ark;
  if (!w_hh_l0 || !w16 || !wT16 || D <= 0 || n_layers <= 0) return ARK_ERR_ARG;
  if (D % 32 != 0) return ARK_ERR_SHAPE;
  dim3 grid(D / 32, 3 * D / 32, n_layers);
  hipStream_t st = (hipStream_t)stream;
#define ARK_WS(PF, PB) hipLaunchKernelGGL((weight_shadow_kernel<PF, PB>), grid, dim3(256), 0, st, w_hh_l0, (long)layer_stride, w16, wT16, D)
  if (prec_fwd == PREC_F16 && prec_bwd == PREC_BF16) ARK_WS(PREC_F16, PREC_BF16);
  else if (prec_fwd == PREC_F16 && prec_bwd == PREC_F16) ARK_WS(PREC_F16, PREC_F16);
  else if (prec_fwd == PREC_BF16 && prec_bwd == PREC_BF16) ARK_WS(PREC_BF16, PREC_BF16);
  else if (prec_fwd == PREC_BF16 && prec_bwd == PREC_F16) ARK_WS(PREC_BF16, PREC_F16);
  else return ARK_ERR_ARG;
#undef ARK_WS
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_cast16(int prec, const float* x, void* out, int64_t n, void* stream) {
  using namespace ark;
  if (!x || !out || n <= 0) return ARK_ERR_ARG;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (prec == PREC_F16) hipLaunchKernelGGL(cast16_kernel<PREC_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  else if (prec == PREC_BF16) hipLaunchKernelGGL(cast16_kernel<PREC_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}
