// Persistent GRU sweep for SMALL batches of LONG sequences (wd-articles: 16 graphs x 637 tokens, wd-movies: 256 x 71).
//
// There the layer-diagonal launches of gru_diag.hip are all fixed cost: a [16 x 1024] x [1024 x 1536] cell is ~0.3 us of
// MFMA work inside a 7.7-us launch, and a step is ~1 300 such launches in a row.  This file runs the whole recurrence of
// a sequence as ONE launch: workgroup (layer l, row block rb, unit slice s) owns 16 hidden units of 16 batch rows for
// every timestep, keeps ITS rows of W_ih and W_hh in registers for the whole sweep (96 VGPRs at D = 512), and the
// hidden state moves between workgroups through an exchange buffer in global memory:
//
//   producer (wave 0 of the cell's workgroup): 16-B write-through (`sc1`) stores of the new state slice, every 128-B
//     line whole, `s_waitcnt vmcnt(0)`, then ONE agent-scope atomic add on the counter of (l, t, rb);
//   consumer: wave 0 polls that counter with `sc1` loads (relaxed, s_sleep between polls), a workgroup barrier, then
//     EVERY load of handed-off bytes is a `buffer_load_dwordx4 ... sc1` straight into MFMA A fragments
//     (MI355X_MICROARCH.md, visibility section, valid forms table row 1: no L1 acquire needed).
//
// Each (l, t) has its own exchange region and counter, so nothing is ever reused inside a launch.  The counters are
// MONOTONE over the launches that share a `sync` workspace: launch number e (the epoch word of the header, read by every
// workgroup when it starts and bumped by the last workgroup to leave) waits for counter >= (e + 1) * NS, so no launch has
// to zero anything in front of a sweep (round 3 did: inside a captured step that 60-KB fill queued behind whatever held
// the CUs).  Every spin is bounded: after ~2 s without progress (or when another workgroup has given up) a workgroup
// sets the error word and leaves, so a scheduling accident ends in an error code, not in a hung GPU.  The error word is
// STICKY: no kernel ever clears it, every later sweep on the workspace leaves at once, and only ark_gru_sweep_sync_reset
// (host, after the error has been read) makes the workspace usable again.  All workgroups must be co-resident: the host
// proves that with the occupancy API (96 KB of dynamic LDS pin one workgroup per CU); the launch itself is a plain one
// inside and outside stream capture (round 3 used hipLaunchCooperativeKernel outside capture: it added nothing to the
// residency proof and rocprofv3 crashed in its exit handlers after such a launch).
//
// Outputs are exactly those of the diagonal kernels (tile-native fp32 state, tile-native fp16 gate saves, row-major
// 16-bit copies, dropout-applied copies with the same counter-hash masks), so the projection, the cross-entropy and the
// backward pass do not know which forward ran.  Reference op replaced: torch.nn.GRU (kgvae/model/models.py:121-127).
#include "sweep_sync.h"
#include "../../include/ark_amd.h"

namespace ark {

struct GruSweepArgs {
  ArkGruSweep a;
};

__global__ void sweep_reset_kernel(unsigned* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// Diagnostic build only (-DARK_SWEEP_STAMPS, tools/sweep_stamps.py): 100-MHz real-time ticks wave 0 of every forward
// workgroup spends in each phase of a step, summed over the sweep, into a buffer of their own (no such code is shipped).
#ifdef ARK_SWEEP_STAMPS
__device__ unsigned long long ark_sweep_stamp_buf[512 * 8];
#define SW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (wave == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); sacc_[i] += n_ - last_; last_ = n_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SW_STAMP(i) do {} while (0)
#endif

// 4 waves: wave w forms the partial products of K-steps [w*KSW, (w+1)*KSW) of BOTH operands (x_t W_ih^T, h_{t-1} W_hh^T);
// wave 0 adds the partials, does the gate math and owns the state.  D = 128 * KSW; MT 16-row tiles per workgroup (the
// weights in registers serve all of them: wd-movies, B = 256 x D = 128, fits the chip with MT = 2).
// WS: logical workgroups (unit slices) per PHYSICAL workgroup of 256 * WS threads.  WS = 2 packs two slices onto one CU
// (waves 0-3 / 4-7: two per SIMD), so the sweep of wd-articles holds 48 CUs instead of 96 and the vocabulary CE beside it
// gets 208.  Measured: NOT a win (wd-articles 7.13 -> 8.01 ms/step, wd-movies 1.93 -> 2.00): the two slices lengthen each
// other's memory round trips, which is all a sweep step is made of.  Kept selectable (`wg_slices = 2`), default 1.
template <int PREC, int PRECB, int KSW, int MT, int WS>
__global__ __launch_bounds__(256 * WS) void gru_sweep_fwd_kernel(GruSweepArgs pa) {
  using PT = PrecTraits<PREC>;
  using PB = PrecTraits<PRECB>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  using hb_t = typename PB::h_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ArkGruSweep& p = pa.a;
  constexpr int TS = 24;                                   // row stride of the 16 x 16 transposition tiles (halves)
  constexpr int TILE = 16 * TS * 2;                        // bytes of one such tile
  constexpr int PARTB = 3 * MT * 4 * 64 * 16;
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // which of the WS slices of this physical workgroup
  constexpr int SUBB = PARTB + MT * 4 * TILE;              // LDS of one slice
  f32x4* part = reinterpret_cast<f32x4*>(smem + sub * SUBB);   // [3 waves][MT][4 accumulators][64 lanes]
  char* tiles = smem + sub * SUBB + PARTB;                 // [MT][h fwd type | h bwd type | h*mask fwd | h*mask bwd]
  int* lflag = reinterpret_cast<int*>(smem + WS * SUBB);

  const int D = p.D, B = p.B, L = p.L;
  const int NS = D >> 4, RBW = B / (16 * MT);
  const int wg = blockIdx.x * WS + sub;                    // logical workgroup
  const int l = wg / (NS * RBW), rem = wg - l * (NS * RBW);
  const int rbw = rem / NS, s = rem - rbw * NS;
  const ArkGruSweepLayer& Ly = p.layer[l];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);   // K-quarter inside the slice
  const int r = lane & 15, kg = lane >> 4;
  const int u = s * 16 + r;   // B-operand column / accumulator column of this lane

  // this workgroup's weight rows, as MFMA B fragments, for the whole sweep
  h8 wx[3][KSW], wh[3][KSW];
  {
    const h_t* wi = reinterpret_cast<const h_t*>(Ly.w_ih16);
    const h_t* wr = reinterpret_cast<const h_t*>(Ly.w_hh16);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int j = 0; j < KSW; ++j) {
        const long o = (long)(g * D + u) * D + (wave * KSW + j) * 32 + kg * 8;
        wx[g][j] = *reinterpret_cast<const h8*>(wi + o);
        wh[g][j] = *reinterpret_cast<const h8*>(wr + o);
      }
  }
  const float br = Ly.b_ih[u] + Ly.b_hh[u], bz = Ly.b_ih[D + u] + Ly.b_hh[D + u];
  const float bin = Ly.b_ih[2 * D + u], bhn = Ly.b_hh[2 * D + u];
  const long slot = (long)B * D;   // elements per timestep
  long tile_off[MT];               // tile-native offset of this lane's quad inside a slot
  f32x4 hprev[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    tile_off[m] = (long)((rbw * MT + m) * NS + s) * 256 + lane * 4;
    hprev[m] = *reinterpret_cast<const f32x4*>(Ly.y_t + tile_off[m]);   // slot 0 = initial state
  }
  const bool drop = Ly.drop_p > 0.f;
  const bool below_drop = l > 0 && p.layer[l - 1].drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(Ly.drop_seed, p.hyper, Ly.drop_p);

  const unsigned RG = (unsigned)(slot * 2);   // bytes of one exchange region == of one row-major 16-bit slot
  const __amdgpu_buffer_rsrc_t rx = sweep_rsrc(p.x0_16, RG * (unsigned)L);
  const __amdgpu_buffer_rsrc_t rh0 = sweep_rsrc(Ly.y16a, RG);
  const __amdgpu_buffer_rsrc_t rex = sweep_rsrc(p.exch, RG * (unsigned)(2 * L * p.n_layers));
  int voff_rm[MT][KSW], voff_ex[MT][KSW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const int ks = wave * KSW + j, rb = rbw * MT + m;
      voff_rm[m][j] = ((rb * 16 + r) * D + ks * 32 + kg * 8) * 2;
      voff_ex[m][j] = ((rb * NS + ks * 2 + (kg >> 1)) * 16 + r) * 32 + (kg & 1) * 16;
    }
  unsigned* sync = p.sync;
  unsigned* cnt = p.sync + kSweepSyncHdr;
  const unsigned need = sweep_enter(sync, lflag, (unsigned)NS);
  __syncthreads();
  const bool poisoned = *lflag != 0;   // (uniform) an earlier sweep on this workspace failed: touch nothing

  u32x4 xn[MT][KSW];   // layer 0: the inputs of the coming step
  if (l == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < KSW; ++j) xn[m][j] = ld_sc1(rx, voff_rm[m][j], 0);
  }
#ifdef ARK_SWEEP_STAMPS
  unsigned long long sacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = 0; t < L && !poisoned; ++t) {
    SW_STAMP(7);   // bulk stores of the previous step + loop overhead
    if (wave == 0) {
      bool ok = true;
      const unsigned code = (unsigned)(wg << 12 | (t & 4095));
      unsigned* cown = cnt + (((long)l * L + (t - 1)) * RBW + rbw) * kSweepCntStride;
      unsigned* cbel = cnt + (((long)(l - 1) * L + t) * RBW + rbw) * kSweepCntStride;
      if (t > 0 && l > 0) ok = sweep_wait2(cown, cbel, need, sync, code);
      else if (t > 0) ok = sweep_wait(cown, need, sync, code);
      else if (l > 0) ok = sweep_wait(cbel, need, sync, code | 0x80000000u);
      if (!ok && lane == 0) *lflag = 1;
    }
    SW_STAMP(0);   // waiting for the two counters
    __syncthreads();
    if (*lflag) break;   // uniform: every wave reads the same word behind the barrier
    SW_STAMP(1);   // barrier

    // the handed-off fragments first: they are what the step waits for.  Layer 0's inputs do not depend on the recurrence:
    // they were requested a step ago, and the next step's request goes out BEHIND the state loads (loads return in order:
    // an input load in front of them would hold them back -- it cost 0.5 us per step of the layer that paces the sweep)
    u32x4 xa[MT][KSW], ha[MT][KSW];
    if (l > 0) {
      const int so = (int)((unsigned)(((l - 1) * L + t) * 2 + (below_drop ? 1 : 0)) * RG);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < KSW; ++j) xa[m][j] = ld_sc1(rex, voff_ex[m][j], so);
    }
    {
      const bool ex = t > 0;
      const int so = ex ? (int)((unsigned)((l * L + (t - 1)) * 2) * RG) : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < KSW; ++j) ha[m][j] = ex ? ld_sc1(rex, voff_ex[m][j], so) : ld_sc1(rh0, voff_rm[m][j], so);
    }
    if (l == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < KSW; ++j) xa[m][j] = xn[m][j];
      if (t + 1 < L) {
        const int so = (int)((unsigned)(t + 1) * RG);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < KSW; ++j) xn[m][j] = ld_sc1(rx, voff_rm[m][j], so);
      }
    }
#ifdef ARK_SWEEP_STAMPS
    if (l == 0 && t + 1 < L) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * KSW) : "memory");   // (all but the prefetch)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SW_STAMP(2);   // handed-off fragments have landed
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[m][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KSW; ++j) {
        const h8 a = __builtin_bit_cast(h8, xa[m][j]);
        acc[m][0] = PT::mfma(a, wx[0][j], acc[m][0]);
        acc[m][1] = PT::mfma(a, wx[1][j], acc[m][1]);
        acc[m][2] = PT::mfma(a, wx[2][j], acc[m][2]);
      }
#pragma unroll
      for (int j = 0; j < KSW; ++j) {
        const h8 a = __builtin_bit_cast(h8, ha[m][j]);
        acc[m][0] = PT::mfma(a, wh[0][j], acc[m][0]);
        acc[m][1] = PT::mfma(a, wh[1][j], acc[m][1]);
        acc[m][3] = PT::mfma(a, wh[2][j], acc[m][3]);
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int a = 0; a < 4; ++a) part[(((wave - 1) * MT + m) * 4 + a) * 64 + lane] = acc[m][a];
    }
    __syncthreads();
    SW_STAMP(3);   // products + partial sums in LDS + barrier
    if (wave == 0) {
      const int hl = lane & 31, row = hl >> 1, half = hl & 1;
      const bool hi = lane >= 32;
      f32x4 rr[MT], zz[MT], nn[MT], hn[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[m][a] += part[((0 * MT + m) * 4 + a) * 64 + lane] + part[((1 * MT + m) * 4 + a) * 64 + lane] + part[((2 * MT + m) * 4 + a) * 64 + lane];
        f32x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rr[m][i] = fast_sigmoid(acc[m][0][i] + br);
          zz[m][i] = fast_sigmoid(acc[m][1][i] + bz);
          hn[m][i] = acc[m][3][i] + bhn;
          nn[m][i] = fast_tanh(acc[m][2][i] + bin + rr[m][i] * hn[m][i]);
          h[i] = nn[m][i] + zz[m][i] * (hprev[m][i] - nn[m][i]);   // (1-z) n + z h_prev
        }
        hprev[m] = h;
        h_t* ta = reinterpret_cast<h_t*>(tiles + (m * 4 + 0) * TILE);
        hb_t* tb = reinterpret_cast<hb_t*>(tiles + (m * 4 + 1) * TILE);
        h_t* tda = reinterpret_cast<h_t*>(tiles + (m * 4 + 2) * TILE);
        hb_t* tdb = reinterpret_cast<hb_t*>(tiles + (m * 4 + 3) * TILE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ta[(4 * kg + i) * TS + r] = PT::cvt(h[i]);
          tb[(4 * kg + i) * TS + r] = PB::cvt(h[i]);
        }
        if (drop) {
          const f32x4 hd = h * dropout_quad(dc, (uint64_t)((long)(p.t0 + t) * slot + tile_off[m]) >> 2);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            tda[(4 * kg + i) * TS + r] = PT::cvt(hd[i]);
            tdb[(4 * kg + i) * TS + r] = PB::cvt(hd[i]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      SW_STAMP(4);   // reduction + gate math + transposition tiles
      // hand the slices over: lanes 0-31 the state (own layer, next step), lanes 32-63 the masked copy (layer above)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const h_t* src = reinterpret_cast<const h_t*>(tiles + (m * 4 + (hi ? 2 : 0)) * TILE);
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + row * TS + half * 8);
        const int so = (int)((unsigned)((l * L + t) * 2 + (hi ? 1 : 0)) * RG);
        if (!hi || drop) st_sc1(v, rex, ((rbw * MT + m) * NS + s) * 512 + hl * 16, so);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SW_STAMP(5);   // write-through stores drained
      if (lane == 0) __hip_atomic_fetch_add(cnt + (((long)l * L + t) * RBW + rbw) * kSweepCntStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // everything below is read only after the launch.  (Handing these stores to the idle waves through a second LDS
      // buffer was measured: 1.96 -> 2.07 ms per wd-articles sweep -- the extra LDS writes cost wave 0 more than the stores.)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const long o = (long)t * slot + tile_off[m];   // element offset inside the [L*B, D] arrays
        *reinterpret_cast<f32x4*>(Ly.y_t + o + slot) = hprev[m];
        if (Ly.save_r) {
          *reinterpret_cast<shalf4_t*>(reinterpret_cast<_Float16*>(Ly.save_r) + o) = shalf4_t{(_Float16)rr[m][0], (_Float16)rr[m][1], (_Float16)rr[m][2], (_Float16)rr[m][3]};
          *reinterpret_cast<shalf4_t*>(reinterpret_cast<_Float16*>(Ly.save_z) + o) = shalf4_t{(_Float16)zz[m][0], (_Float16)zz[m][1], (_Float16)zz[m][2], (_Float16)zz[m][3]};
          *reinterpret_cast<shalf4_t*>(reinterpret_cast<_Float16*>(Ly.save_n) + o) = shalf4_t{(_Float16)nn[m][0], (_Float16)nn[m][1], (_Float16)nn[m][2], (_Float16)nn[m][3]};
          *reinterpret_cast<shalf4_t*>(reinterpret_cast<_Float16*>(Ly.save_hn) + o) = shalf4_t{(_Float16)hn[m][0], (_Float16)hn[m][1], (_Float16)hn[m][2], (_Float16)hn[m][3]};
        }
        const h_t* ta = reinterpret_cast<const h_t*>(tiles + (m * 4 + 0) * TILE);
        const hb_t* tb = reinterpret_cast<const hb_t*>(tiles + (m * 4 + 1) * TILE);
        const h_t* tda = reinterpret_cast<const h_t*>(tiles + (m * 4 + 2) * TILE);
        const hb_t* tdb = reinterpret_cast<const hb_t*>(tiles + (m * 4 + 3) * TILE);
        const long go = ((long)t * B + (rbw * MT + m) * 16 + row) * D + s * 16 + half * 8;   // row-major element offset, slot t
        if (!hi) {
          *reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(Ly.y16a) + go + slot) = *reinterpret_cast<const u32x4*>(ta + row * TS + half * 8);
          if (drop) *reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(Ly.yd16a) + go) = *reinterpret_cast<const u32x4*>(tda + row * TS + half * 8);
        } else {
          if (Ly.y16b) *reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(Ly.y16b) + go + slot) = *reinterpret_cast<const u32x4*>(tb + row * TS + half * 8);
          if (drop && Ly.yd16b) *reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(Ly.yd16b) + go) = *reinterpret_cast<const u32x4*>(tdb + row * TS + half * 8);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#ifdef ARK_SWEEP_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 512)
    for (int i_ = 0; i_ < 8; ++i_) ark_sweep_stamp_buf[blockIdx.x * 8 + i_] = sacc_[i_];
#endif
  sweep_leave(sync);
}

template <class Kern, class Args>
static int launch_persistent(Kern kern, const Args& p, unsigned grid, int ws, hipStream_t st);

// `grid` counts LOGICAL workgroups (slices); two of them share a physical workgroup unless the caller asks for one per CU
template <int PREC, int PRECB, int KSW, int MT>
static int launch_sweep_fwd(const GruSweepArgs& p, unsigned grid, hipStream_t st) {
  // (two slices per workgroup = two waves per SIMD = 256 registers per lane: D = 512 with two row tiles does not fit them)
  if (p.a.wg_slices != 2 || grid % 2 != 0 || (KSW == 4 && MT == 2)) {
    auto kern = gru_sweep_fwd_kernel<PREC, PRECB, KSW, MT, 1>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLds), true);
    (void)once;
    return launch_persistent(kern, p, grid, 1, st);
  }
  if constexpr (!(KSW == 4 && MT == 2)) {
    auto kern = gru_sweep_fwd_kernel<PREC, PRECB, KSW, MT, 2>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLds), true);
    (void)once;
    return launch_persistent(kern, p, grid / 2, 2, st);
  }
  return ARK_ERR_SHAPE;
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward sweep (BPTT of the same recurrence, autograd of nn.GRU): workgroup (l, rb, s) walks t = L-1 .. 0 for its 16
// units of 16 rows.  Per step it needs the gate-gradient panels [dr | dz | dn | dn*r] of its own layer at t+1 (recurrence,
// K = 3D against W_hh^T) and of the layer above at t (input gradient, K = 3D against W_ih(l+1)^T, through this layer's
// output-dropout mask); both arrive through the exchange buffer exactly as the hidden state does in the forward sweep.
// The carry dh*z and the bias-gradient column sums stay in wave 0's registers for the whole sweep; after t = 0 the
// workgroup adds its slice of the initial-state gradient (carry + dgh_0 W_hh) to dh0.  Math and outputs are those of
// gru_diag_bwd_kernel (gru_diag.hip).
struct GruSweepBwdArgs {
  ArkGruSweepBwd a;
};

template <int PREC, int KSW, int MT, int WS>
__global__ __launch_bounds__(256 * WS) void gru_sweep_bwd_kernel(GruSweepBwdArgs pa) {
  using PT = PrecTraits<PREC>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  constexpr int KT = 3 * KSW;   // K-steps of 32 per wave and product (3D / 32 / 4 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ArkGruSweepBwd& p = pa.a;
  constexpr int TS = 24;
  constexpr int PARTB = 3 * MT * 2 * 64 * 16;
  constexpr int TILE4 = 4 * 16 * TS * 2;                   // bytes of the four part tiles of one row tile
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // (WS slices per physical workgroup: see the forward kernel)
  constexpr int SUBB = PARTB + MT * TILE4;
  f32x4* part = reinterpret_cast<f32x4*>(smem + sub * SUBB);   // [3 waves][MT][2 accumulators][64 lanes]
  char* tiles = smem + sub * SUBB + PARTB;                 // [MT][4 parts][16 rows][TS]
  int* lflag = reinterpret_cast<int*>(smem + WS * SUBB);

  const int D = p.D, B = p.B, L = p.L, n = p.n_layers;
  const int NS = D >> 4, RBW = B / (16 * MT);
  const int wg = blockIdx.x * WS + sub;
  // the top layer starts the backward wavefront: give it the lowest workgroup ids
  const int l = n - 1 - wg / (NS * RBW), rem = wg % (NS * RBW);
  const int rbw = rem / NS, s = rem - rbw * NS;
  const ArkGruSweepBwdLayer& Ly = p.layer[l];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
  const int r = lane & 15, kg = lane >> 4;
  const int u = s * 16 + r;
  const bool top = l == n - 1;

  h8 wo[KT], wu[KT];   // rows u of W_hh^T (own recurrence) and of W_ih(l+1)^T (gradient from above)
  {
    const h_t* wh = reinterpret_cast<const h_t*>(Ly.w_hhT16);
    const h_t* wi = reinterpret_cast<const h_t*>(Ly.w_ihT_up16);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const long o = (long)u * 3 * D + (wave * KT + j) * 32 + kg * 8;
      wo[j] = *reinterpret_cast<const h8*>(wh + o);
      if (!top) wu[j] = *reinterpret_cast<const h8*>(wi + o);
      else wu[j] = wo[j];
    }
  }
  const long slot = (long)B * D;
  long tile_off[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) tile_off[m] = (long)((rbw * MT + m) * NS + s) * 256 + lane * 4;
  const bool drop = Ly.drop_p > 0.f && !top;
  DropCtx dc{};
  if (drop) dc = drop_ctx(Ly.drop_seed, p.hyper, Ly.drop_p);
  const unsigned RG = (unsigned)(slot * 8);   // bytes of one (layer, step) exchange region: [B, 4D] 16-bit
  const __amdgpu_buffer_rsrc_t rex = sweep_rsrc(p.exch, RG * (unsigned)(L * n));
  int voff_o[KT], voff_u[KT];   // row tile 0; tile m adds m * 4 * NS * 512 bytes
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int k = (wave * KT + j) * 32 + kg * 8;
    {
      const int c = k < 2 * D ? k : k + D;      // W_hh columns: [dr | dz] then dn*r
      const int pt = c / D, un = c - pt * D;
      voff_o[j] = ((((rbw * MT * 4 + pt) * NS + (un >> 4)) * 16 + r) * 16 + (un & 8)) * 2;
    }
    {
      const int pt = k / D, un = k - pt * D;    // W_ih columns: [dr | dz | dn]
      voff_u[j] = ((((rbw * MT * 4 + pt) * NS + (un >> 4)) * 16 + r) * 16 + (un & 8)) * 2;
    }
  }
  const int mstride = 4 * NS * 512;
  unsigned* sync = p.sync;
  unsigned* cnt = p.sync + kSweepSyncHdr;
  const unsigned need = sweep_enter(sync, lflag, (unsigned)NS);
  __syncthreads();
  const bool poisoned = *lflag != 0;

  f32x4 carry[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) carry[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of the ROUNDED dr, dz, dn, dn*r (this lane's rows)
  const _Float16* sr = reinterpret_cast<const _Float16*>(Ly.save_r);
  const _Float16* sz = reinterpret_cast<const _Float16*>(Ly.save_z);
  const _Float16* sn = reinterpret_cast<const _Float16*>(Ly.save_n);
  const _Float16* shn = reinterpret_cast<const _Float16*>(Ly.save_hn);
  const bool fin_step = p.dh0 != nullptr;

  for (int t = L - 1; t >= (fin_step ? -1 : 0) && !poisoned; --t) {
    const bool fin = t < 0;          // initial-state step: dh0 += carry + dgh_0 W_hh
    const bool rec = t < L - 1;      // a successor step exists
    const bool up = !top && !fin;
    // epilogue operands first (wave 0): none of them depends on the recurrence, they land during the wait
    f32x4 hp[MT], dy[MT];
    shalf4_t qr[MT], qz[MT], qn[MT], qhn[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      hp[m] = dy[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      qr[m] = qz[m] = qn[m] = qhn[m] = shalf4_t{};
      if (wave == 0 && !fin) {
        const long o = (long)t * slot + tile_off[m];
        hp[m] = *reinterpret_cast<const f32x4*>(Ly.y_t + o);
        if (top) dy[m] = *reinterpret_cast<const f32x4*>(p.dy_t + o);
        qr[m] = *reinterpret_cast<const shalf4_t*>(sr + o);
        qz[m] = *reinterpret_cast<const shalf4_t*>(sz + o);
        qn[m] = *reinterpret_cast<const shalf4_t*>(sn + o);
        qhn[m] = *reinterpret_cast<const shalf4_t*>(shn + o);
      }
    }
    if (wave == 0) {
      bool ok = true;
      const unsigned code = (unsigned)(wg << 12 | ((t + 1) & 4095));
      unsigned* cown = cnt + (((long)l * L + (t + 1)) * RBW + rbw) * kSweepCntStride;
      unsigned* cabv = cnt + (((long)(l + 1) * L + t) * RBW + rbw) * kSweepCntStride;
      if (rec && up) ok = sweep_wait2(cown, cabv, need, sync, code);
      else if (rec) ok = sweep_wait(cown, need, sync, code);
      else if (up) ok = sweep_wait(cabv, need, sync, code | 0x80000000u);
      if (!ok && lane == 0) *lflag = 1;
    }
    __syncthreads();
    if (*lflag) break;

    f32x4 ah[MT], ax[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      ah[m] = ax[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 ao[KT], au[KT];
      if (rec) {
        const int so = (int)((unsigned)(l * L + (t + 1)) * RG) + m * mstride;
#pragma unroll
        for (int j = 0; j < KT; ++j) ao[j] = ld_sc1(rex, voff_o[j], so);
      }
      if (up) {
        const int so = (int)((unsigned)((l + 1) * L + t) * RG) + m * mstride;
#pragma unroll
        for (int j = 0; j < KT; ++j) au[j] = ld_sc1(rex, voff_u[j], so);
      }
      if (rec) {
#pragma unroll
        for (int j = 0; j < KT; ++j) ah[m] = PT::mfma(__builtin_bit_cast(h8, ao[j]), wo[j], ah[m]);
      }
      if (up) {
#pragma unroll
        for (int j = 0; j < KT; ++j) ax[m] = PT::mfma(__builtin_bit_cast(h8, au[j]), wu[j], ax[m]);
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        part[(((wave - 1) * MT + m) * 2 + 0) * 64 + lane] = ah[m];
        part[(((wave - 1) * MT + m) * 2 + 1) * 64 + lane] = ax[m];
      }
    }
    __syncthreads();
    if (wave == 0) {
      const int hl = lane & 31, row = hl >> 1, half = hl & 1, hi = lane >> 5;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] += part[((0 * MT + m) * 2 + 0) * 64 + lane] + part[((1 * MT + m) * 2 + 0) * 64 + lane] + part[((2 * MT + m) * 2 + 0) * 64 + lane];
        ax[m] += part[((0 * MT + m) * 2 + 1) * 64 + lane] + part[((1 * MT + m) * 2 + 1) * 64 + lane] + part[((2 * MT + m) * 2 + 1) * 64 + lane];
      }
      if (fin) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 dh = ah[m] + carry[m];
#pragma unroll
          for (int i = 0; i < 4; ++i) atomicAdd(&p.dh0[(long)((rbw * MT + m) * 16 + 4 * kg + i) * D + u], dh[i]);
        }
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const long o = (long)t * slot + tile_off[m];
          f32x4 dh = ah[m] + carry[m] + dy[m];
          if (drop) dh += ax[m] * dropout_quad(dc, (uint64_t)o >> 2);
          else dh += ax[m];
          h_t* tg = reinterpret_cast<h_t*>(tiles + m * TILE4);
          f32x4 cz;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float rr = (float)qr[m][i], zz = (float)qz[m][i], nn = (float)qn[m][i], hn = (float)qhn[m][i];
            const float dn_pre = dh[i] * (1.0f - zz) * (1.0f - nn * nn);
            const float dz_pre = dh[i] * (hp[m][i] - nn) * zz * (1.0f - zz);
            const float dr_pre = dn_pre * hn * rr * (1.0f - rr);
            cz[i] = dh[i] * zz;
            const int ro = (4 * kg + i) * TS + r;
            const h_t v0 = PT::cvt(dr_pre), v1 = PT::cvt(dz_pre), v2 = PT::cvt(dn_pre), v3 = PT::cvt(dn_pre * rr);
            tg[ro] = v0;
            tg[16 * TS + ro] = v1;
            tg[2 * 16 * TS + ro] = v2;
            tg[3 * 16 * TS + ro] = v3;
            bs[0] += (float)v0;
            bs[1] += (float)v1;
            bs[2] += (float)v2;
            bs[3] += (float)v3;
          }
          carry[m] = cz;
        }
        __builtin_amdgcn_wave_barrier();
        // hand the 512-byte slices over (two store instructions per row tile, every 128-B line whole), then the row-major copy
        u32x4 v01[MT], v23[MT];
        const int so = (int)((unsigned)(l * L + t) * RG);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const h_t* tg = reinterpret_cast<const h_t*>(tiles + m * TILE4);
          v01[m] = *reinterpret_cast<const u32x4*>(tg + (hi * 16 + row) * TS + half * 8);
          v23[m] = *reinterpret_cast<const u32x4*>(tg + ((2 + hi) * 16 + row) * TS + half * 8);
          st_sc1(v01[m], rex, (((rbw * MT + m) * 4 + hi) * NS + s) * 512 + hl * 16, so);
          st_sc1(v23[m], rex, (((rbw * MT + m) * 4 + 2 + hi) * NS + s) * 512 + hl * 16, so);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt + (((long)l * L + t) * RBW + rbw) * kSweepCntStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          h_t* g16 = reinterpret_cast<h_t*>(Ly.dg16) + ((long)t * B + (rbw * MT + m) * 16 + row) * 4 * D + s * 16 + half * 8;
          *reinterpret_cast<u32x4*>(g16 + (long)hi * D) = v01[m];
          *reinterpret_cast<u32x4*>(g16 + (long)(2 + hi) * D) = v23[m];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // bias gradients: db_ih = colsum [dr | dz | dn], db_hh = colsum [dr | dz | dn*r], one atomic per (gate, unit)
  if (wave == 0 && Ly.db_ih && !poisoned) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bs[g] += __shfl_xor(bs[g], 16, 64);
      bs[g] += __shfl_xor(bs[g], 32, 64);
    }
    if (lane < 16) {
      atomicAdd(&Ly.db_ih[u], bs[0]);
      atomicAdd(&Ly.db_ih[D + u], bs[1]);
      atomicAdd(&Ly.db_ih[2 * D + u], bs[2]);
      atomicAdd(&Ly.db_hh[u], bs[0]);
      atomicAdd(&Ly.db_hh[D + u], bs[1]);
      atomicAdd(&Ly.db_hh[2 * D + u], bs[3]);
    }
  }
  sweep_leave(sync);
}

template <class Kern, class Args>
static int launch_persistent(Kern kern, const Args& p, unsigned grid, int ws, hipStream_t st) {
  // every workgroup spins on others: all of them must be resident at once (`grid` physical workgroups of 256 * ws threads)
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256 * ws, kSweepLds) != hipSuccess || per_cu < 1) return ARK_ERR_SHAPE;
  if ((long)grid > (long)cus * per_cu) return ARK_ERR_SHAPE;
  // a plain launch, captured or not: residency is what the occupancy query above proves (nothing else may hold more
  // than 64 KB of LDS on the CUs this grid needs: the caller's business, see ark_amd.h); a failure is bounded and sticky
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * ws), kSweepLds, st, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC, int KSW, int MT>
static int launch_sweep_bwd(const GruSweepBwdArgs& p, unsigned grid, hipStream_t st) {
  // (the backward keeps two weight panels AND two fragment sets in registers: at D = 512 that needs the 512 registers of a
  //  lone wave per SIMD -- 58-126 spilled registers with two slices per workgroup -- so it stays one slice per workgroup)
  if (p.a.wg_slices != 2 || grid % 2 != 0 || KSW == 4) {
    auto kern = gru_sweep_bwd_kernel<PREC, KSW, MT, 1>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLds), true);
    (void)once;
    return launch_persistent(kern, p, grid, 1, st);
  }
  if constexpr (KSW != 4) {
    auto kern = gru_sweep_bwd_kernel<PREC, KSW, MT, 2>;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLds), true);
    (void)once;
    return launch_persistent(kern, p, grid / 2, 2, st);
  }
  return ARK_ERR_SHAPE;
}

}  // namespace ark

// 16-row tiles per workgroup: 1 where n * (B/16) * (D/16) workgroups fit the chip one per CU, else 2, else 0 (no fit)
static int sweep_row_tiles(int n_layers, int B, int D) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (n_layers <= 0 || n_layers > ARK_SWEEP_MAX_LAYERS || B <= 0 || B % 16 != 0 || (D != 128 && D != 256 && D != 512)) return 0;
  const long wgs = (long)n_layers * (B / 16) * (D / 16);
  if (wgs <= cus) return 1;
  if (B % 32 == 0 && wgs / 2 <= cus) return 2;
  return 0;
}
extern "C" int ark_gru_sweep_row_tiles(int n_layers, int B, int D) { return sweep_row_tiles(n_layers, B, D); }
// CUs (= physical workgroups) the forward (backward = 1) sweep of this shape holds with `wg_slices` slices per workgroup
extern "C" int ark_gru_sweep_cus(int n_layers, int B, int D, int backward, int wg_slices) {
  const int mt = sweep_row_tiles(n_layers, B, D);
  if (mt == 0) return 0;
  const int logical = n_layers * (B / 16) * (D / 16) / mt;
  const bool two = wg_slices == 2 && logical % 2 == 0 && (backward ? D != 512 : !(D == 512 && mt == 2));
  return two ? logical / 2 : logical;
}

extern "C" long ark_gru_sweep_exch_bytes(int n_layers, int B, int D, int L) { return 2L * n_layers * L * B * D * 2; }
extern "C" long ark_gru_sweep_sync_words(int n_layers, int B, int L) { return ark::kSweepSyncHdr + (long)n_layers * L * (B / 16) * ark::kSweepCntStride; }

extern "C" int ark_gru_sweep_fwd(int prec, int prec_b, const ArkGruSweep* a, void* stream) {
  using namespace ark;
  if (!a || a->n_layers <= 0 || a->n_layers > ARK_SWEEP_MAX_LAYERS || a->B <= 0 || a->D <= 0 || a->L <= 0) return ARK_ERR_ARG;
  if (!a->x0_16 || !a->exch || !a->sync || a->t0 < 0) return ARK_ERR_ARG;
  const int D = a->D, B = a->B, L = a->L, n = a->n_layers;
  if (B % 16 != 0 || (D != 128 && D != 256 && D != 512)) return ARK_ERR_SHAPE;
  if (L > 4095 || 2.0 * n * L * B * D * 2 >= 2147483648.0) return ARK_ERR_SHAPE;   // 32-bit buffer offsets
  const int mt = sweep_row_tiles(n, B, D);
  if (mt == 0) return ARK_ERR_SHAPE;   // the workgroups could not all be resident: use the diagonal launches
  for (int l = 0; l < n; ++l) {
    const ArkGruSweepLayer& y = a->layer[l];
    if (!y.w_ih16 || !y.w_hh16 || !y.b_ih || !y.b_hh || !y.y_t || !y.y16a) return ARK_ERR_ARG;
    if (y.drop_p < 0.f || y.drop_p >= 1.f || (y.drop_p > 0.f && (!y.yd16a || !a->hyper))) return ARK_ERR_ARG;
    if (y.save_r && (!y.save_z || !y.save_n || !y.save_hn)) return ARK_ERR_ARG;
  }
  GruSweepArgs p;
  p.a = *a;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(n * (D / 16) * (B / 16) / mt);
#define ARK_SWEEP_GO(PF, PBK)                                                                                   \
  do {                                                                                                          \
    if (D == 512) return mt == 1 ? launch_sweep_fwd<PF, PBK, 4, 1>(p, grid, st) : launch_sweep_fwd<PF, PBK, 4, 2>(p, grid, st); \
    if (D == 256) return mt == 1 ? launch_sweep_fwd<PF, PBK, 2, 1>(p, grid, st) : launch_sweep_fwd<PF, PBK, 2, 2>(p, grid, st); \
    return mt == 1 ? launch_sweep_fwd<PF, PBK, 1, 1>(p, grid, st) : launch_sweep_fwd<PF, PBK, 1, 2>(p, grid, st);               \
  } while (0)
  if (prec == PREC_F16 && prec_b == PREC_BF16) ARK_SWEEP_GO(PREC_F16, PREC_BF16);
  if (prec == PREC_F16 && prec_b == PREC_F16) ARK_SWEEP_GO(PREC_F16, PREC_F16);
  if (prec == PREC_BF16 && prec_b == PREC_BF16) ARK_SWEEP_GO(PREC_BF16, PREC_BF16);
#undef ARK_SWEEP_GO
  return ARK_ERR_ARG;
}

extern "C" long ark_gru_sweep_bwd_exch_bytes(int n_layers, int B, int D, int L) { return (long)n_layers * L * B * 4 * D * 2; }

extern "C" int ark_gru_sweep_sync_reset(unsigned* sync, long words, void* stream) {
  if (!sync || words < ark::kSweepSyncHdr) return ARK_ERR_ARG;
  hipLaunchKernelGGL(ark::sweep_reset_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sync, words);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_gru_sweep_bwd(int prec, const ArkGruSweepBwd* a, void* stream) {
  using namespace ark;
  if (!a || a->n_layers <= 0 || a->n_layers > ARK_SWEEP_MAX_LAYERS || a->B <= 0 || a->D <= 0 || a->L <= 0) return ARK_ERR_ARG;
  if (!a->dy_t || !a->exch || !a->sync) return ARK_ERR_ARG;
  const int D = a->D, B = a->B, L = a->L, n = a->n_layers;
  if (B % 16 != 0 || (D != 128 && D != 256 && D != 512)) return ARK_ERR_SHAPE;
  if (L > 4094 || 8.0 * n * L * B * D >= 2147483648.0) return ARK_ERR_SHAPE;   // 32-bit buffer offsets
  const int mt = sweep_row_tiles(n, B, D);
  if (mt == 0) return ARK_ERR_SHAPE;
  for (int l = 0; l < n; ++l) {
    const ArkGruSweepBwdLayer& y = a->layer[l];
    if (!y.w_hhT16 || (l < n - 1 && !y.w_ihT_up16) || !y.save_r || !y.save_z || !y.save_n || !y.save_hn || !y.y_t || !y.dg16)
      return ARK_ERR_ARG;
    if ((y.db_ih == nullptr) != (y.db_hh == nullptr)) return ARK_ERR_ARG;
    if (y.drop_p < 0.f || y.drop_p >= 1.f || (y.drop_p > 0.f && !a->hyper)) return ARK_ERR_ARG;
  }
  GruSweepBwdArgs p;
  p.a = *a;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(n * (D / 16) * (B / 16) / mt);
#define ARK_SWEEP_GO(PB)                                                                                          \
  do {                                                                                                            \
    if (D == 512) return mt == 1 ? launch_sweep_bwd<PB, 4, 1>(p, grid, st) : launch_sweep_bwd<PB, 4, 2>(p, grid, st); \
    if (D == 256) return mt == 1 ? launch_sweep_bwd<PB, 2, 1>(p, grid, st) : launch_sweep_bwd<PB, 2, 2>(p, grid, st); \
    return mt == 1 ? launch_sweep_bwd<PB, 1, 1>(p, grid, st) : launch_sweep_bwd<PB, 1, 2>(p, grid, st);               \
  } while (0)
  if (prec == PREC_BF16) ARK_SWEEP_GO(PREC_BF16);
  if (prec == PREC_F16) ARK_SWEEP_GO(PREC_F16);
#undef ARK_SWEEP_GO
  return ARK_ERR_ARG;
}

#ifdef ARK_SWEEP_STAMPS
extern "C" int ark_debug_sweep_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ark::ark_sweep_stamp_buf), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
