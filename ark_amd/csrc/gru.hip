// GRU recurrence kernels: one launch per (layer, timestep), the recurrent product on MFMA with
// the whole cell update fused into the epilogue.  Replaces the per-step body of torch.nn.GRU
// (reference kgvae/model/models.py:121-127,141 forward; autograd's BPTT for backward).
//
//   forward :  gh = h_{t-1} W_hh^T (+ b_hh)            [B,D]x[D,3D]   MFMA
//              r = s(gi_r + gh_r)  z = s(gi_z + gh_z)  n = tanh(gi_n + r * gh_n)
//              h_t = (1-z) n + z h_{t-1}
//   backward:  dh_t = dy_t + z_{t+1} dh_{t+1} + dgh_{t+1} W_hh       [B,3D]x[3D,D]   MFMA
//              then the gate derivatives of step t, written as dgi_t / dgh_t for the
//              time-batched weight-gradient products.
//
// The forward tile interleaves the three gates along N (tile column j -> gate (j/16)%3), so one
// lane owns r, z and n of the same hidden unit and the cell update needs no cross-lane traffic.
#include "gemm_core.h"
#include "../../include/ark_amd.h"

namespace ark {

struct GruFwdArgs {
  const float* h_prev; const float* w_hh; const float* b_hh; const float* gi;
  float* h_out; float* h_drop; const float* drop_mask;
  float* sr; float* sz; float* sn; float* shn;
  int B, D;
};

template <int PREC, int BM>
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(GruFwdArgs p) {
  constexpr int BU = 32, BN = 3 * BU;
  using G = GemmTile<PREC, LAY_KMAJ, LAY_KMAJ, BM, BN, 2, 2>;  // wave tile 32 x 48 (16 units x 3 gates)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int units_tiles = p.D / BU;
  const int m0 = (blockIdx.x / units_tiles) * BM, u0 = (blockIdx.x % units_tiles) * BU;
  const int B = p.B, D = p.D;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, p.h_prev, D, [=](int r) -> long { return (m0 + r < B) ? (long)(m0 + r) : -1L; },
         p.w_hh, D, [=](int j) -> long { return (long)((j >> 4) % 3) * D + u0 + (j / 48) * 16 + (j & 15); },
         D, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int u = u0 + wn * 16 + (lane & 15);
  const float bhr = p.b_hh[u], bhz = p.b_hh[D + u], bhn = p.b_hh[2 * D + u];
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
      if (p.h_drop) p.h_drop[o] = h * p.drop_mask[o];
      if (p.sr) { p.sr[o] = r; p.sz[o] = z; p.sn[o] = n; p.shn[o] = hn; }
    }
  }
}

struct GruBwdArgs {
  const float* dgh_next; const float* w_hh; const float* dy; float* carry;
  const float* sr; const float* sz; const float* sn; const float* shn; const float* h_prev;
  float* dgi; float* dgh;
  float* dh0; int dh0_accumulate;
  int B, D, first, final_;
};

// final_ == 0: regular step t.   final_ == 1: only dh0 = dgh_0 W_hh + carry (gradient wrt the initial state).
template <int PREC, int BM, int BN>
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(GruBwdArgs p) {
  using G = GemmTile<PREC, LAY_KMAJ, LAY_MMAJ, BM, BN, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.D + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int B = p.B, D = p.D;
  f32x4 acc[G::TM][G::TN];
  G::run(acc, p.dgh_next, 3L * D, [=](int r) -> long { return (m0 + r < B) ? (long)(m0 + r) : -1L; },
         p.w_hh, D, [=](int r) -> long { return (n0 + r < D) ? (long)(n0 + r) : -1L; },
         p.first ? 0 : 3 * D, smem);

  G::for_each(acc, [&](int r_, int c_, float a) {
    const int row = m0 + r_, u = n0 + c_;
    if (row >= B || u >= D) return;
    const long o = (long)row * D + u;
    float dh = a;
    if (!p.first) dh += p.carry[o];
    if (p.final_) {
      p.dh0[o] = p.dh0_accumulate ? p.dh0[o] + dh : dh;
      return;
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
  });
}

}  // namespace ark

namespace ark {
constexpr int g_fwd_bm = 32;      // rows per forward-cell workgroup (64 | 32)
constexpr int g_bwd_tile = 2;     // backward-cell tile: 0 = 64x64, 1 = 32x64, 2 = 32x32

template <int PREC, int BM>
static void launch_fwd(const GruFwdArgs& p, hipStream_t st) {
  const unsigned grid = (unsigned)(((p.B + BM - 1) / BM) * (p.D / 32));
  hipLaunchKernelGGL((gru_cell_fwd_kernel<PREC, BM>), dim3(grid), dim3(256), (BM + 96) * 128, st, p);
}
template <int PREC>
static void launch_fwd_prec(const GruFwdArgs& p, hipStream_t st) {
  if (g_fwd_bm == 64) launch_fwd<PREC, 64>(p, st); else launch_fwd<PREC, 32>(p, st);
}
template <int PREC, int BM, int BN>
static void launch_bwd(const GruBwdArgs& p, hipStream_t st) {
  const unsigned grid = (unsigned)(((p.B + BM - 1) / BM) * ((p.D + BN - 1) / BN));
  hipLaunchKernelGGL((gru_cell_bwd_kernel<PREC, BM, BN>), dim3(grid), dim3(256), (BM + BN) * 128, st, p);
}
template <int PREC>
static void launch_bwd_prec(const GruBwdArgs& p, hipStream_t st) {
  if (g_bwd_tile == 0) launch_bwd<PREC, 64, 64>(p, st);
  else if (g_bwd_tile == 1) launch_bwd<PREC, 32, 64>(p, st);
  else launch_bwd<PREC, 32, 32>(p, st);
}
static int launch_bwd_any(int prec, const GruBwdArgs& p, hipStream_t st) {
  if (prec == PREC_F32) launch_bwd_prec<PREC_F32>(p, st);
  else if (prec == PREC_BF16) launch_bwd_prec<PREC_BF16>(p, st);
  else if (prec == PREC_F16) launch_bwd_prec<PREC_F16>(p, st);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}
}  // namespace ark

extern "C" int ark_gru_cell_fwd(int prec, const float* h_prev, const float* w_hh, const float* b_hh, const float* gi,
                                float* h_out, float* h_drop, const float* drop_mask, float* save_r, float* save_z,
                                float* save_n, float* save_hn, int B, int D, void* stream) {
  using namespace ark;
  if (!h_prev || !w_hh || !b_hh || !gi || !h_out || B <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 32 != 0) return ARK_ERR_SHAPE;
  if (h_drop && !drop_mask) return ARK_ERR_ARG;
  if (save_r && (!save_z || !save_n || !save_hn)) return ARK_ERR_ARG;
  GruFwdArgs p{h_prev, w_hh, b_hh, gi, h_out, h_drop, drop_mask, save_r, save_z, save_n, save_hn, B, D};
  hipStream_t st = (hipStream_t)stream;
  if (prec == PREC_F32) launch_fwd_prec<PREC_F32>(p, st);
  else if (prec == PREC_BF16) launch_fwd_prec<PREC_BF16>(p, st);
  else if (prec == PREC_F16) launch_fwd_prec<PREC_F16>(p, st);
  else return ARK_ERR_ARG;
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_gru_cell_bwd(int prec, const float* dgh_next, const float* w_hh, const float* dy, float* carry,
                                const float* save_r, const float* save_z, const float* save_n, const float* save_hn,
                                const float* h_prev, float* dgi, float* dgh, int B, int D, int first, void* stream) {
  using namespace ark;
  if (!w_hh || !carry || !save_r || !save_z || !save_n || !save_hn || !h_prev || !dgi || !dgh || B <= 0 || D <= 0)
    return ARK_ERR_ARG;
  if (!first && !dgh_next) return ARK_ERR_ARG;
  GruBwdArgs p{dgh_next ? dgh_next : dgh, w_hh, dy, carry, save_r, save_z, save_n, save_hn, h_prev, dgi, dgh,
               nullptr, 0, B, D, first ? 1 : 0, 0};
  return launch_bwd_any(prec, p, (hipStream_t)stream);
}

extern "C" int ark_gru_h0_bwd(int prec, const float* dgh0, const float* w_hh, const float* carry, float* dh0,
                              int accumulate, int B, int D, void* stream) {
  using namespace ark;
  if (!dgh0 || !w_hh || !carry || !dh0 || B <= 0 || D <= 0) return ARK_ERR_ARG;
  GruBwdArgs p{dgh0, w_hh, nullptr, const_cast<float*>(carry), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
               nullptr, dh0, accumulate ? 1 : 0, B, D, 0, 1};
  return launch_bwd_any(prec, p, (hipStream_t)stream);
}
