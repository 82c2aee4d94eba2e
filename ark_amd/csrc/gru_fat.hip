// Weights-stationary forward recurrence of a FULL batch (syn-paths: 1024 graphs x 10 steps x 3 layers, D = 512).
//
// The layer-diagonal launches (gru_diag.hip) re-stream every weight panel through LDS for every cell: with three 64 x 32
// tiles per CU a forward diagonal moves 819-983 KB per CU through the ~100 GB/s L2 -> LDS path and takes 18.5 us for 3.9 us
// of MFMA work; three rounds of tile / ring / residency tuning never got past 0.84 of that path's ceiling, because the limit
// is BYTES PER CU, and two thirds of those bytes are weights that never change during the step.  The 9.4 MB of 16-bit GRU
// weights fit the chip's register files (128 MB).  So here, as in gru_sweep.hip but for many rows:
//
//   workgroup (layer l, unit slice s, row group g): US = 32 hidden units (16 at D = 1024) x 3 gates x K = 2D of W_ih / W_hh
//   = 196 KB live in the registers of its 4 waves (K split four ways, 192 VGPRs each, one wave per SIMD) for the WHOLE
//   recurrence; per step it walks the 16-row tiles of its row group: the tile's x / h fragments come straight from global
//   memory into MFMA A operands (each wave loads only ITS K-quarter: no redundant operand traffic, no LDS staging), the four
//   K-partials meet in LDS, all 256 threads do the gate math, the new state goes to the next step's consumers through the
//   exchange buffer of gru_sweep.hip (write-through stores + one monotone counter per (layer, step, row subgroup)).
//   Per CU and step: 13 tiles x 32 KB of activations = 416 KB instead of 819-983 KB, and no launch boundary, ring prologue
//   or drain between the steps.
//
// Schedule.  A row group is walked as TWO subgroups with counters of their own: while the workgroups of a (layer, row group)
// -- 16 of them, one per unit slice, all waiting on each other -- finish subgroup B of step t, the hand-off of subgroup A is
// already complete, so the counter round trip hides behind the other half's arithmetic.  Inside a subgroup the tiles are
// software-pipelined: fragments of tile k + 2 are in flight while tile k + 1 multiplies and tile k's gate math runs.
// Workgroup ids are dealt so that the 16 slices of a (layer, row group) share an XCD (ids are dealt round-robin over the 8
// XCDs): the recurrent hand-off then stays inside one L2 (speed only; nothing relies on it).
//
// Payload LOADS are plain (L2-cached) loads: every exchange line is written once per launch and read only after its counter
// says so, and a launch starts with clean caches, so no stale copy can exist; the 16 consumers of a line then share one
// fetch.  Stores are write-through (`sc1`) and drained before the counter moves, exactly as in gru_sweep.hip.
// Outputs are those of the diagonal launches (tile-native fp32 state, fp16 gate saves, row-major 16-bit copies, dropout-
// applied copies from the same counter hash), so CE, backward and tests do not know which forward ran.
// Reference op replaced: torch.nn.GRU forward (kgvae/model/models.py:121-127, 141).
#include "sweep_sync.h"
#include "../../include/ark_amd.h"

#ifndef ARK_FAT_LD_AUX
#define ARK_FAT_LD_AUX 0   // cache policy of the payload loads: 0 = plain (see above), 16 = sc1 (debugging aid)
#endif

namespace ark {

struct GruFatArgs {
  ArkGruSweep a;
  int NG;          // row groups
  int NSUB;        // subgroups a row group is walked as (counters of their own): 2 in the register-staged kernel
  int xcd_pairs;   // 1: the NS slices of a (layer, row group) pair share an XCD
};

// Diagnostic build only (-DARK_FAT_STAMPS, tools/fat_stamps.py): 100-MHz ticks thread 0 of every workgroup spends per phase,
// summed over the launch: [0] waiting for the counters, [1] first fragments + first multiply, [2] the tile loop,
// [3] last copies + store drain + publish, [4] tiles walked
#ifdef ARK_FAT_STAMPS
__device__ unsigned long long ark_fat_stamp_buf[512 * 8 * 5];
#define FAT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); facc_[i] += n_ - flast_; flast_ = n_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FAT_STAMP(i) do {} while (0)
#endif

constexpr int kFatMaxTiles = 16;   // 16-row tiles per workgroup (both subgroups): bounds the LDS state array

// Payload loads are ordinary (compiler-tracked) buffer loads, and hipcc's wait insertion cannot count them across the tile
// loop (loads two tiles ahead, a varying number of stores in between): it waits for `vmcnt(0)` in front of every multiply.
// In the first build that wait sat right behind the write-through stores of the previous tile (1.5 us per tile, 287 us per
// forward against 228 for the diagonal launches); the multiply now heads the iteration.  Tried and dropped: hand-issued
// loads (inline asm into AGPR tuples + counted waits: the register allocator parks an asm output in a scratch tuple and
// copies it home at once, i.e. before the data has landed); the walk fully unrolled (exact counts in straight-line code,
// but with all 256 VGPRs in use the unrolled body spills and the scheduler sinks the prefetch next to its use).
__device__ __forceinline__ u32x4 ld_pay(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, ARK_FAT_LD_AUX);
}

// D = 128 * KSW; UT 16-unit tiles per workgroup (US = 16 * UT units); 4 waves, wave w owns K-steps [w*KSW, (w+1)*KSW) of
// both products
template <int PREC, int PRECB, int KSW, int UT>
__global__ __launch_bounds__(256) void gru_fat_fwd_kernel(GruFatArgs pa) {
  using PT = PrecTraits<PREC>;
  using PB = PrecTraits<PRECB>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  using hb_t = typename PB::h_t;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ArkGruSweep& p = pa.a;
  constexpr int D = 128 * KSW, US = 16 * UT, NS = D / US, NS16 = D / 16;
  constexpr int TS = US + 8;                                  // row stride of the transposition tiles (halves)
  constexpr int PART_B = 2 * 4 * 4 * UT * 64 * 16;            // [2 buffers][4 waves][4 accumulators][UT][64 lanes] f32x4
  constexpr int STATE_B = kFatMaxTiles * UT * 64 * 16;        // [local tile][UT][64 lanes] f32x4: fp32 state of this workgroup's rows
  constexpr int TILE_B = 16 * TS * 2;                         // one 16-row transposition tile
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  f32x4* state = reinterpret_cast<f32x4*>(smem + PART_B);
  char* tiles = smem + PART_B + STATE_B;                      // [2 buffers][4 arrays: h fwd | h*mask fwd | h bwd | h*mask bwd]
  int* lflag = reinterpret_cast<int*>(smem + PART_B + STATE_B + 2 * 4 * TILE_B);

  const int B = p.B, L = p.L, n = p.n_layers, NG = pa.NG;
  const int RT = B >> 4, NSG = 2 * NG, NP = n * NG;
  unsigned* sync = p.sync;
  unsigned* cnt = p.sync + kSweepSyncHdr;
  // workgroup -> (pair = layer * NG + row group, unit slice)
  int pair, s;
  if (pa.xcd_pairs) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    pair = xcd + 8 * (j / NS);
    s = j % NS;
  } else {
    pair = blockIdx.x / NS;
    s = blockIdx.x % NS;
  }
  const unsigned need = sweep_enter(sync, lflag, (unsigned)NS);
  __syncthreads();
  if (pair >= NP || *lflag != 0) {   // (block-uniform) a filler id of the XCD map, or a poisoned workspace: nothing to do
    sweep_leave(sync);
    return;
  }
  const int l = pair / NG, g = pair - l * NG;
  const ArkGruSweepLayer& Ly = p.layer[l];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kg = lane >> 4;

  // this workgroup's weight rows as MFMA B fragments, for the whole recurrence
  h8 wx[3][UT][KSW], wh[3][UT][KSW];
  {
    const h_t* wi = reinterpret_cast<const h_t*>(Ly.w_ih16);
    const h_t* wr = reinterpret_cast<const h_t*>(Ly.w_hh16);
#pragma unroll
    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
          const long o = (long)(gt * D + s * US + ut * 16 + r) * D + (wave * KSW + j) * 32 + kg * 8;
          wx[gt][ut][j] = *reinterpret_cast<const h8*>(wi + o);
          wh[gt][ut][j] = *reinterpret_cast<const h8*>(wr + o);
        }
  }
  // epilogue mapping: thread -> (unit tile eut, MFMA lane ln, row half hf): rows 4*(ln>>4) + 2*hf + {0, 1}, unit ln & 15
  const int eq = tid >> 1, hf = tid & 1;
  const int eut = eq >> 6, ln = eq & 63;
  const bool eact = eut < UT;                       // (UT = 1: the upper half of the workgroup has no element)
  const int eu = s * US + (eact ? eut : 0) * 16 + (ln & 15);
  const float br = Ly.b_ih[eu] + Ly.b_hh[eu], bz = Ly.b_ih[D + eu] + Ly.b_hh[D + eu];
  const float bin = Ly.b_ih[2 * D + eu], bhn = Ly.b_hh[2 * D + eu];
  const bool drop = Ly.drop_p > 0.f;
  const bool below_drop = l > 0 && p.layer[l - 1].drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(Ly.drop_seed, p.hyper, Ly.drop_p);
  const long slot = (long)B * D;                    // elements per timestep
  const unsigned RG = (unsigned)(slot * 2);         // bytes of one exchange region == of one row-major 16-bit slot
  const __amdgpu_buffer_rsrc_t rx = sweep_rsrc(p.x0_16, RG * (unsigned)L);
  const __amdgpu_buffer_rsrc_t rh0 = sweep_rsrc(Ly.y16a, RG);
  const __amdgpu_buffer_rsrc_t rex = sweep_rsrc(p.exch, RG * (unsigned)(2 * L * n));
  // per-lane fragment offsets inside a 16-row tile (tile rt adds rt * 16 * D * 2 bytes in both layouts)
  int voff_rm[KSW], voff_ex[KSW];
#pragma unroll
  for (int j = 0; j < KSW; ++j) {
    const int ks = wave * KSW + j;
    voff_rm[j] = (r * D + ks * 32 + kg * 8) * 2;
    voff_ex[j] = ((ks * 2 + (kg >> 1)) * 16 + r) * 32 + (kg & 1) * 16;
  }
  const int tile_bytes = 16 * D * 2;
  const int g_rt0 = (2 * g) * RT / NSG;             // first tile of this workgroup (local tile index = rt - g_rt0)

  // fp32 state of the workgroup's rows: slot 0 of the tile-native state array
  {
    const int g_rt1 = (2 * g + 2) * RT / NSG;
    if (eact)
      for (int rt = g_rt0; rt < g_rt1; ++rt) {
        const long o = (long)(rt * NS16 + s * UT + eut) * 256 + ln * 4 + 2 * hf;
        reinterpret_cast<f32x2*>(state + ((rt - g_rt0) * UT + eut) * 64 + ln)[hf] = *reinterpret_cast<const f32x2*>(Ly.y_t + o);
      }
  }
  __syncthreads();

  auto load_tile = [&](int t, int rt, u32x4 (&xa)[KSW], u32x4 (&ha)[KSW]) {
    const int to = rt * tile_bytes;
    if (l == 0) {
      const int so = (int)((unsigned)t * RG) + to;
#pragma unroll
      for (int j = 0; j < KSW; ++j) xa[j] = ld_pay(rx, voff_rm[j], so);
    } else {
      const int so = (int)((unsigned)(((l - 1) * L + t) * 2 + (below_drop ? 1 : 0)) * RG) + to;
#pragma unroll
      for (int j = 0; j < KSW; ++j) xa[j] = ld_pay(rex, voff_ex[j], so);
    }
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < KSW; ++j) ha[j] = ld_pay(rh0, voff_rm[j], to);
    } else {
      const int so = (int)((unsigned)((l * L + (t - 1)) * 2) * RG) + to;
#pragma unroll
      for (int j = 0; j < KSW; ++j) ha[j] = ld_pay(rex, voff_ex[j], so);
    }
  };
  auto multiply = [&](const u32x4 (&xa)[KSW], const u32x4 (&ha)[KSW], f32x4 (&acc)[4][UT]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) acc[a][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const h8 a = __builtin_bit_cast(h8, xa[j]);
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) {
        acc[0][ut] = PT::mfma(a, wx[0][ut][j], acc[0][ut]);
        acc[1][ut] = PT::mfma(a, wx[1][ut][j], acc[1][ut]);
        acc[2][ut] = PT::mfma(a, wx[2][ut][j], acc[2][ut]);
      }
    }
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const h8 a = __builtin_bit_cast(h8, ha[j]);
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) {
        acc[0][ut] = PT::mfma(a, wh[0][ut][j], acc[0][ut]);
        acc[1][ut] = PT::mfma(a, wh[1][ut][j], acc[1][ut]);
        acc[3][ut] = PT::mfma(a, wh[2][ut][j], acc[3][ut]);
      }
    }
  };
  auto put_part = [&](int buf, const f32x4 (&acc)[4][UT]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) part[(((buf * 4 + wave) * 4 + a) * UT + ut) * 64 + lane] = acc[a][ut];
  };
  // gate math of tile rt (partials in part[buf]) by all threads; leaves the 16-bit copies in the transposition tiles [buf]
  // (The stores of this function are UNCONDITIONAL for UT = 2 on purpose: hipcc can only let a multiply wait for ITS fragments
  //  -- `vmcnt(N)` with N = what was provably issued after them -- if it can prove how many stores lie in between; with the gate
  //  saves behind `if (save_r)` it proved none and every tile waited for the write-through stores issued just before it.)
  auto epilogue = [&](int t, int rt, int buf) {
    if constexpr (UT == 1) {
      if (!eact) return;
    }
    f32x2 sum[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      sum[a] = f32x2{0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) sum[a] += reinterpret_cast<const f32x2*>(part + (((buf * 4 + w) * 4 + a) * UT + eut) * 64 + ln)[hf];
    }
    f32x2* st = reinterpret_cast<f32x2*>(state + ((rt - g_rt0) * UT + eut) * 64 + ln) + hf;
    const f32x2 hp = *st;
    f32x2 rr, zz, nn, hn, h;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rr[i] = fast_sigmoid(sum[0][i] + br);
      zz[i] = fast_sigmoid(sum[1][i] + bz);
      hn[i] = sum[3][i] + bhn;
      nn[i] = fast_tanh(sum[2][i] + bin + rr[i] * hn[i]);
      h[i] = nn[i] + zz[i] * (hp[i] - nn[i]);   // (1-z) n + z h_prev
    }
    *st = h;
    const long o = (long)(rt * NS16 + s * UT + eut) * 256 + ln * 4 + 2 * hf;   // tile-native offset inside a slot
    *reinterpret_cast<f32x2*>(Ly.y_t + (long)(t + 1) * slot + o) = h;
    {
      const long so = (long)t * slot + o;   // (the host requires the four save arrays)
      *reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(Ly.save_r) + so) = half2_t{(_Float16)rr[0], (_Float16)rr[1]};
      *reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(Ly.save_z) + so) = half2_t{(_Float16)zz[0], (_Float16)zz[1]};
      *reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(Ly.save_n) + so) = half2_t{(_Float16)nn[0], (_Float16)nn[1]};
      *reinterpret_cast<half2_t*>(reinterpret_cast<_Float16*>(Ly.save_hn) + so) = half2_t{(_Float16)hn[0], (_Float16)hn[1]};
    }
    char* tb = tiles + buf * 4 * TILE_B;
    const int row = 4 * (ln >> 4) + 2 * hf, col = eut * 16 + (ln & 15);
    h_t* ta = reinterpret_cast<h_t*>(tb);
    hb_t* tbw = reinterpret_cast<hb_t*>(tb + 2 * TILE_B);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ta[(row + i) * TS + col] = PT::cvt(h[i]);
      tbw[(row + i) * TS + col] = PB::cvt(h[i]);
    }
    if (drop) {
      const f32x4 m4 = dropout_quad(dc, (uint64_t)((long)(p.t0 + t) * slot + (o - 2 * hf)) >> 2);
      const float m0 = hf ? m4[2] : m4[0], m1 = hf ? m4[3] : m4[1];
      h_t* tda = reinterpret_cast<h_t*>(tb + TILE_B);
      hb_t* tdb = reinterpret_cast<hb_t*>(tb + 3 * TILE_B);
      tda[row * TS + col] = PT::cvt(h[0] * m0);
      tda[(row + 1) * TS + col] = PT::cvt(h[1] * m1);
      tdb[row * TS + col] = PB::cvt(h[0] * m0);
      tdb[(row + 1) * TS + col] = PB::cvt(h[1] * m1);
    }
  };
  // the 16-bit copies of tile rt leave the transposition tiles [buf]: wave 0 the state (own layer's next step: exchange copy
  // 0, and the row-major forward-type array), wave 1 the dropout-applied copy (layer above: exchange copy 1, and its row-major
  // array), waves 2 / 3 the two backward-type arrays; every 128-byte line of the exchange buffer is written whole
  auto store_tile = [&](int t, int rt, int buf) {
    if (lane >= 32 * UT) return;
    const int sut = lane >> 5, srow = (lane & 31) >> 1, half = lane & 1;
    const char* tb = tiles + (buf * 4 + wave) * TILE_B;
    const u32x4 v = *reinterpret_cast<const u32x4*>(tb + (srow * TS + sut * 16 + half * 8) * 2);
    const long go = ((long)t * B + rt * 16 + srow) * D + s * US + sut * 16 + half * 8;   // row-major element offset, slot t
    const int xo = ((rt * NS16 + s * UT + sut) * 16 + srow) * 32 + half * 16;            // exchange byte offset inside a region
    if (wave == 0) {
      st_sc1(v, rex, xo, (int)((unsigned)((l * L + t) * 2) * RG));
      *reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(Ly.y16a) + go + slot) = v;
    } else if (wave == 1) {
      if (drop) {
        st_sc1(v, rex, xo, (int)((unsigned)((l * L + t) * 2 + 1) * RG));
        *reinterpret_cast<u32x4*>(reinterpret_cast<h_t*>(Ly.yd16a) + go) = v;
      }
    } else if (wave == 2) {
      if (Ly.y16b) *reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(Ly.y16b) + go + slot) = v;
    } else {
      if (drop && Ly.yd16b) *reinterpret_cast<u32x4*>(reinterpret_cast<hb_t*>(Ly.yd16b) + go) = v;
    }
  };

#ifdef ARK_FAT_STAMPS
  unsigned long long facc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flast_ = __builtin_amdgcn_s_memrealtime();
#endif
  bool dead = false;
  for (int t = 0; t < L && !dead; ++t) {
    for (int hs = 0; hs < 2; ++hs) {
      FAT_STAMP(5);
      const int q = 2 * g + hs;
      const int r0 = q * RT / NSG, r1 = (q + 1) * RT / NSG, nt = r1 - r0;
      if (wave == 0) {
        bool ok = true;
        const unsigned code = (unsigned)(blockIdx.x << 12 | (t & 2047) << 1 | hs);
        unsigned* cown = cnt + (((long)l * L + (t - 1)) * NSG + q) * kSweepCntStride;
        unsigned* cbel = cnt + (((long)(l - 1) * L + t) * NSG + q) * kSweepCntStride;
        if (t > 0 && l > 0) ok = sweep_wait2(cown, cbel, need, sync, code);
        else if (t > 0) ok = sweep_wait(cown, need, sync, code);
        else if (l > 0) ok = sweep_wait(cbel, need, sync, code | 0x80000000u);
        if (!ok && lane == 0) *lflag = 1;
      }
      __syncthreads();
      if (*lflag) { dead = true; break; }   // uniform: every wave reads the same word behind the barrier
      FAT_STAMP(0);

      // fragments two tiles ahead of the multiply, the multiply one tile ahead of the gate math.  Two fragment buffers take
      // turns (the tile loop is unrolled by two so that neither is ever copied: a copy would wait for the load in flight).
      // hipcc cannot count the loads across this loop and waits for `vmcnt(0)` in front of every multiply (see the note on
      // ld_pay), so the multiply comes FIRST in an iteration: what is in flight then was issued a whole tile ago.
      u32x4 xaP[KSW], haP[KSW], xaQ[KSW], haQ[KSW];
      f32x4 acc[4][UT];
      load_tile(t, r0, xaP, haP);
      if (nt > 1) load_tile(t, r0 + 1, xaQ, haQ);
      multiply(xaP, haP, acc);
      put_part(0, acc);
      if (nt > 2) load_tile(t, r0 + 2, xaP, haP);
      __syncthreads();
      FAT_STAMP(1);
#define ARK_FAT_ITER(K, XA, HA)                                        \
      {                                                                \
        const int k_ = (K);                                            \
        if (k_ + 1 < nt) multiply(XA, HA, acc);                        \
        if (k_ + 3 < nt) load_tile(t, r0 + k_ + 3, XA, HA);            \
        if (k_ > 0) store_tile(t, r0 + k_ - 1, (k_ - 1) & 1);          \
        epilogue(t, r0 + k_, k_ & 1);                                  \
        if (k_ + 1 < nt) put_part((k_ + 1) & 1, acc);                  \
        __syncthreads();                                               \
      }
      for (int k = 0; k < nt; k += 2) {
        ARK_FAT_ITER(k, xaQ, haQ)
        if (k + 1 < nt) ARK_FAT_ITER(k + 1, xaP, haP)
      }
#undef ARK_FAT_ITER
      FAT_STAMP(2);
#ifdef ARK_FAT_STAMPS
      if (tid == 0) facc_[4] += nt;
#endif
      store_tile(t, r1 - 1, (nt - 1) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have landed
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(cnt + (((long)l * L + t) * NSG + q) * kSweepCntStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      FAT_STAMP(3);
    }
  }
#ifdef ARK_FAT_STAMPS
  if (tid == 0 && blockIdx.x < 512)
    for (int i_ = 0; i_ < 8; ++i_) ark_fat_stamp_buf[blockIdx.x * 8 + i_] = facc_[i_];
#endif
  sweep_leave(sync);
}

// ---------------------------------------------------------------------------------------------------------------------------
// D = 512: the same decomposition with LDS as the landing zone of the activation stream and ROLE-SPECIALISED waves
// (round 4, second and third build).
//
// What the register-staged kernel above measured (tools/fat_stamps.py): 1.53 us per 16-row tile for 0.37 us of MFMA work:
// hipcc waits for `vmcnt(0)` in front of every multiply, so one tile's worth of loads is ever in flight and every tile pays
// the whole memory latency.  Second build: fragments by LDS-DMA (`global_load_lds_dwordx4`, lane-linear 1-KB pieces that ARE
// the MFMA A fragments) into a ring of tile slots, waited for with COUNTED `s_waitcnt vmcnt(N)` -- the data then arrived in
// time (0.05 us of wait per tile), but with ONE wave per SIMD doing everything a tile took 2.4 us: a lone wave issues one
// instruction per ~4 cycles, an LDS-DMA piece costs it 60-185 cycles of issue (MI355X_MICROARCH.md, cycle constants), and
// issue + gate math + multiply + copies + bookkeeping ran one after the other (tools/fat_ring_stamps.py: 0.78 + 0.64 + 0.30 +
// 0.30 + 0.45 us).  Third build (this one): eight waves, two per SIMD --
//   * waves 4-7 ("matrix waves") hold the weights (x product: W_ih rows of unit tile 0 / 1, h product: W_hh; K = 512 = 192
//     VGPRs of B fragments each, no K-split) and do nothing but: barrier, 16 fragment reads, 48 MFMAs, accumulators to LDS;
//   * waves 0-3 ("helpers", one beside each matrix wave) do everything else: LDS-DMA of the next tiles (8-9 pieces each),
//     the counter poll and the verdict (wave 0), the gate math of the tile multiplied in the last iteration in MFMA
//     accumulator layout (waves 0 / 1: one f32x4 per lane = one tile-native quad, 16-byte stores), the 16-bit copies and
//     the counter bump (waves 2 / 3).
// A counted wait needs the exact number of vector-memory instructions a wave issues behind the one it waits for, so a
// helper's iteration is ONE straight-line body whose instruction count never varies: what has nothing to do (a bubble: the
// next tile's producers have not published yet; pipeline head and tail) points its loads at a dummy LDS slot and its stores
// out of range of their buffer descriptors, where the hardware drops them.  Validity travels down the pipeline (issue ->
// multiply R-1 iterations later -> gate math -> copies -> publish) as bits of a shift register.  Nothing in the loop loads
// from global memory into registers (such a load would make hipcc drain the DMA queue): the fp32 state of a tile comes
// through a ring of its own, the hand-off counters are polled by LDS-DMA and read after the same counted wait as the data.
// The tile stream runs ACROSS steps and subgroups (NSUB = 3 subgroups per row group): while the producers of subgroup A's
// next step publish, the tiles of B and C stream; a counter is bumped two iterations after the last copies of its subgroup
// were issued (the counted wait of that iteration covers them), so no wave ever drains its queue.
constexpr int kFatRing = 3;
#ifdef ARK_FAT_STAMPS
#define RING_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); rst_[i] += n_ - rlast_; rlast_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RING_STAMP(i) do {} while (0)
#endif

// an LDS word that other waves (or an LDS-DMA) wrote: read through an explicit LDS pointer (a `volatile` generic pointer
// would become a FLAT load, which counts on vmcnt and would drain the DMA queue); the `asm` waits around the call sites
// keep the compiler from caching it across iterations
__device__ __forceinline__ int lds_word(const void* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) char*)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_set_word(void* p, int v) {
  *reinterpret_cast<__attribute__((address_space(3))) int*>((__attribute__((address_space(3))) char*)p) = v;
}

struct FatCur { int t, sub, k, n; };   // a pipeline stage's place in the workgroup's tile stream (n = tiles passed)

template <int PREC, int PRECB>
__global__ __launch_bounds__(512) void gru_fat_ring_kernel(GruFatArgs pa) {
  using PT = PrecTraits<PREC>;
  using PB = PrecTraits<PRECB>;
  using h_t = typename PT::h_t;
  using h8 = typename PT::h8;
  using hb_t = typename PB::h_t;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  constexpr int D = 512, UT = 2, US = 32, NS = D / US, NS16 = D / 16, R = kFatRing, TS = US + 8;
  constexpr int SLOT_B = 32768, OP_B = 16384, TILE_B = 16 * TS * 2;
  // LDS: fragment ring | state ring (R + 1 slots: a tile's state is read one iteration after its fragments) | accumulators
  // [2 buffers][4 matrix waves][3][64 lanes] f32x4 | transposition tiles [2 buffers][4 arrays] | dummy | poll ring | words
  constexpr int O_ST = R * SLOT_B, O_PART = O_ST + (R + 1) * 2048, O_TRANS = O_PART + 2 * 4 * 3 * 1024;
  constexpr int O_DUMMY = O_TRANS + 2 * 4 * TILE_B, O_POLL = O_DUMMY + 1024, O_FLAG = O_POLL + 3 * 256;
  static_assert(R == 3 || R == 4, "the poll ring has three slots");
  constexpr int OOB = (int)0x80000000u;            // a voffset beyond every descriptor: the access is dropped
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ArkGruSweep& p = pa.a;
  int* lflag = reinterpret_cast<int*>(smem + O_FLAG);   // words: [0] error / poisoned, [2], [3] command of iteration parity 0 / 1

  const int B = p.B, L = p.L, n = p.n_layers, NG = pa.NG, NSUB = pa.NSUB;
  const int RT = B >> 4, NSG = NSUB * NG, NP = n * NG;
  unsigned* sync = p.sync;
  unsigned* cnt = p.sync + kSweepSyncHdr;
  int pair, s;
  if (pa.xcd_pairs) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    pair = xcd + 8 * (j / NS);
    s = j % NS;
  } else {
    pair = blockIdx.x / NS;
    s = blockIdx.x % NS;
  }
  const unsigned need = sweep_enter(sync, lflag, (unsigned)NS);
  __syncthreads();
  if (pair >= NP || *lflag != 0) {
    sweep_leave(sync);
    return;
  }
  const int l = pair / NG, g = pair - l * NG;
  const ArkGruSweepLayer& Ly = p.layer[l];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kg = lane >> 4;

  // the subgroups of this row group: first tile, tile count
  int r0s[4], nts[4], NTS = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = NSUB * g + (i < NSUB ? i : NSUB - 1);
    r0s[i] = q * RT / NSG;
    nts[i] = (q + 1) * RT / NSG - r0s[i];
    if (i < NSUB) NTS += nts[i];
  }
  auto r0_of = [&](int sub) { return sub == 0 ? r0s[0] : sub == 1 ? r0s[1] : sub == 2 ? r0s[2] : r0s[3]; };
  auto nt_of = [&](int sub) { return sub == 0 ? nts[0] : sub == 1 ? nts[1] : sub == 2 ? nts[2] : nts[3]; };
  auto advance = [&](FatCur& c) {
    ++c.n;
    if (++c.k == nt_of(c.sub)) {
      c.k = 0;
      if (++c.sub == NSUB) { c.sub = 0; ++c.t; }
    }
  };
  const int NT = L * NTS;
  if (tid == 0) {   // command words: bit 0 = this iteration issues a real tile, bit 1 = leave the loop
    lflag[2] = (NT > 0 && l == 0) ? 1 : 0;   // iteration 0: step 0 of layer 0 waits for nobody
    lflag[3] = 0;
  }

  if (wave >= 4) {
    // ------------------------------------------------ matrix waves ------------------------------------------------
    const int m = wave - 4;
    const bool xw = m < 2;
    const int ut = m & 1;
    h8 w[3][16];
    {
      const h_t* ws = reinterpret_cast<const h_t*>(xw ? Ly.w_ih16 : Ly.w_hh16);
#pragma unroll
      for (int gt = 0; gt < 3; ++gt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          w[gt][ks] = *reinterpret_cast<const h8*>(ws + (long)(gt * D + s * US + ut * 16 + r) * D + ks * 32 + kg * 8);
    }
    __syncthreads();
    unsigned hist = 0;
    int nmul = 0;
    for (int e = 0;; ++e) {
      const int cmd = __builtin_amdgcn_readfirstlane(lds_word(smem + O_FLAG + 8 + (e & 1) * 4));
      if (cmd & 2) break;
      hist = (hist << 1) | (unsigned)(cmd & 1);
      if ((hist >> (R - 1)) & 1u) {
        const char* src = smem + (nmul % R) * SLOT_B + (xw ? 0 : OP_B) + lane * 16;
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        h8 af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const h8*>(src + j * 1024);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc's scheduler otherwise sinks every read next to its MFMAs: one buffer,
                                             //  a full LDS round trip per k-step)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const h8 a = af[ks & 3];
          if (ks + 4 < 16) {
            af[ks & 3] = *reinterpret_cast<const h8*>(src + (ks + 4) * 1024);
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[0] = PT::mfma(a, w[0][ks], acc[0]);
          acc[1] = PT::mfma(a, w[1][ks], acc[1]);
          acc[2] = PT::mfma(a, w[2][ks], acc[2]);
        }
        f32x4* pp = reinterpret_cast<f32x4*>(smem + O_PART) + ((e & 1) * 4 + m) * 3 * 64 + lane;
        pp[0] = acc[0];
        pp[64] = acc[1];
        pp[128] = acc[2];
        ++nmul;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    sweep_leave(sync);
    return;
  }

  // ---------------------------------------------------- helpers ----------------------------------------------------
  const bool xh = wave < 2;          // loads the x operand (waves 0, 1) / the h operand (2, 3); pieces (wave & 1) * 8 ..
  const int ut = wave & 1;           // gate math of unit tile ut (waves 0, 1)
  const int eu = s * US + ut * 16 + r;
  const float br = Ly.b_ih[eu] + Ly.b_hh[eu], bz = Ly.b_ih[D + eu] + Ly.b_hh[D + eu];
  const float bin = Ly.b_ih[2 * D + eu], bhn = Ly.b_hh[2 * D + eu];
  const bool drop = Ly.drop_p > 0.f;
  const bool below_drop = l > 0 && p.layer[l - 1].drop_p > 0.f;
  DropCtx dc{};
  if (drop) dc = drop_ctx(Ly.drop_seed, p.hyper, Ly.drop_p);
  const long slot = (long)B * D;
  const unsigned RG = (unsigned)(slot * 2);
  const __amdgpu_buffer_rsrc_t rex = sweep_rsrc(p.exch, RG * (unsigned)(2 * L * n));
  const __amdgpu_buffer_rsrc_t ry = sweep_rsrc(Ly.y_t, (unsigned)((L + 1) * slot * 4));
  const __amdgpu_buffer_rsrc_t rsr = sweep_rsrc(Ly.save_r, RG * (unsigned)L);
  const __amdgpu_buffer_rsrc_t rsz = sweep_rsrc(Ly.save_z, RG * (unsigned)L);
  const __amdgpu_buffer_rsrc_t rsn = sweep_rsrc(Ly.save_n, RG * (unsigned)L);
  const __amdgpu_buffer_rsrc_t rshn = sweep_rsrc(Ly.save_hn, RG * (unsigned)L);
  // the row-major 16-bit copies: wave 2 writes the forward-type pair (state -> y16a and exchange copy 0, state x mask ->
  // yd16a and exchange copy 1), wave 3 the backward-type pair (y16b, yd16b); a missing array gets an empty descriptor
  void* cp0 = wave == 2 ? Ly.y16a : Ly.y16b;
  void* cp1 = drop ? (wave == 2 ? Ly.yd16a : Ly.yd16b) : nullptr;
  const __amdgpu_buffer_rsrc_t rc0 = sweep_rsrc(cp0, cp0 ? RG * (unsigned)(L + 1) : 0u);
  const __amdgpu_buffer_rsrc_t rc1 = sweep_rsrc(cp1, cp1 ? RG * (unsigned)L : 0u);

  // per-lane source offset of this wave's first fragment piece (k-step (wave & 1) * 8) in the two layouts; piece i lies
  // i * 64 B (row-major) / i * 1024 B (exchange layout) further on -- a uniform stride, added to the scalar base
  const int voff_rm = (r * D + ut * 8 * 32 + kg * 8) * 2;
  const int voff_ex = ((ut * 8 * 2 + (kg >> 1)) * 16 + r) * 32 + (kg & 1) * 16;
  const int tile_bytes = 16 * D * 2;
  const char* x0b = reinterpret_cast<const char*>(p.x0_16);
  const char* exb = reinterpret_cast<const char*>(p.exch);
  const char* h0b = reinterpret_cast<const char*>(Ly.y16a);
  const char* ytb = reinterpret_cast<const char*>(Ly.y_t);

  unsigned hist = 0;                     // bit d: the issue slot d iterations ago carried a real tile
  unsigned long long tags = ~0ull;       // 16 bits per iteration: the group (t * NSUB + sub) that iteration's poll looked at
  FatCur ci{0, 0, 0, 0}, ce = ci, cs = ci;
  int p0 = -1, p1 = -1, p2 = -1;         // counters to bump, by age in iterations
  unsigned stalls = 0;
  unsigned long long stall_t0 = 0;
#ifdef ARK_FAT_STAMPS
  unsigned long long facc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long fstart_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long rst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rlast_ = fstart_;   // per-phase ticks of this wave
#endif
  // hipcc does not see the hand-written waits: every register loaded from global memory above gets a use HERE, so that the
  // wait for it stands in front of the loop and not (as `vmcnt(0)`, draining the DMA queue) at its first use inside
  asm volatile("" ::"v"(br), "v"(bz), "v"(bin), "v"(bhn), "v"(dc.hs), "v"(dc.p16), "v"(dc.ks), "v"(dc.s1));
  __syncthreads();

  for (int e = 0;; ++e) {
    RING_STAMP(6);
    // ---- 1. issue: the next tile of the stream if its inputs are published (wave 0's verdict of the last iteration), else
    //         a bubble ----
    const bool more = ci.n < NT;
    const bool ready = (__builtin_amdgcn_readfirstlane(lds_word(smem + O_FLAG + 8 + (e & 1) * 4)) & 1) != 0;
    int pt = ci.t, ps = ci.sub;          // the group the poll looks at: the cursor's own if it stands at its first tile
    if (ci.k > 0) {
      if (++ps == NSUB) { ps = 0; ++pt; }
    }
    const bool pvalid = more && pt < L;
    if (wave == 0) {
      const int q = NSUB * g + ps;
      const unsigned* cown = cnt + (pvalid && pt > 0 ? (((long)l * L + (pt - 1)) * NSG + q) * kSweepCntStride : 0);
      const unsigned* cbel = cnt + (pvalid && l > 0 ? (((long)(l - 1) * L + pt) * NSG + q) * kSweepCntStride : 0);
      const unsigned* src = lane == 0 ? cown : cbel;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + O_POLL + (e % 3) * 256), 4, 0, 16);
    }
    tags = (tags << 16) | (unsigned long long)(pvalid ? (unsigned)(pt * NSUB + ps) : 0xFFFFu);
    {
      const int t = ci.t, rt = r0_of(ci.sub) + ci.k;
      const int sl = ci.n % R;
      const char* base;
      bool rm;
      if (xh) {
        rm = l == 0;
        base = rm ? x0b + (size_t)t * RG : exb + (size_t)(((l - 1) * L + t) * 2 + (below_drop ? 1 : 0)) * RG;
      } else {
        rm = t == 0;
        base = rm ? h0b : exb + (size_t)((l * L + (t - 1)) * 2) * RG;
      }
      base += (size_t)rt * tile_bytes;
      if (!ready) { base = x0b; rm = true; }
#ifdef ARK_RING_DUMMY_SRC     // timing experiment: every piece from one cache-hot tile (results invalid)
      base = x0b; rm = true;
#endif
#ifdef ARK_RING_DUMMY_SRC_EX  // ... in the exchange layout (full lines)
      base = exb; rm = false;
#endif
      char* dst = ready ? smem + sl * SLOT_B + (xh ? 0 : OP_B) + ut * 8 * 1024 : smem + O_DUMMY;
      const int dstep = ready ? 1024 : 0;
      const int sstep = rm ? 64 : 1024;
      const int voff = rm ? voff_rm : voff_ex;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(i * sstep) + voff),
                                         (__attribute__((address_space(3))) void*)(dst + i * dstep), 16, 0, 0);
      if (xh) {   // the fp32 state of the tile's rows (unit tile ut): slot t of the tile-native state array
        const char* sb = ready ? ytb + ((size_t)t * slot + (size_t)(rt * NS16 + s * UT + ut) * 256) * 4 : ytb;
        char* sd = ready ? smem + O_ST + (ci.n % (R + 1)) * 2048 + ut * 1024 : smem + O_DUMMY;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + lane * 16),
                                         (__attribute__((address_space(3))) void*)sd, 16, 0, 0);
      }
    }
    hist = (hist << 1) | (ready ? 1u : 0u);
    if (ready) advance(ci);
    bool fail = false;
    if (wave == 0) {   // a real deadlock must end as an error, not as a hang
      if (more && !ready) {
        if ((++stalls & 1023u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          fail = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
          if (stall_t0 == 0) stall_t0 = now;
          else if (now - stall_t0 > kSweepTimeoutTicks) {
            __hip_atomic_store(sync + 1, (unsigned)(blockIdx.x << 12 | (ci.t & 2047) << 1 | (ci.sub & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fail = true;
          }
        }
      } else {
        stalls = 0;
        stall_t0 = 0;
      }
    }
#ifdef ARK_RING_NO_STORES   // timing experiment: every store dropped by its descriptor's range check (results invalid)
    const bool v_epi_m = (hist >> R) & 1u, v_st_m = (hist >> (R + 1)) & 1u;
    const bool v_epi = false, v_st = false;
#else
    const bool v_epi = (hist >> R) & 1u, v_st = (hist >> (R + 1)) & 1u;
    const bool v_epi_m = v_epi, v_st_m = v_st;
#endif
    RING_STAMP(0);

    if (xh) {
      // ---- 2. gate math of the tile multiplied in the last iteration (accumulator layout: lane = unit, 4 rows) ----
      const int t = ce.t, rt = r0_of(ce.sub) + ce.k;
      const int o = (rt * NS16 + s * UT + ut) * 256 + lane * 4;       // tile-native element offset of this lane's quad
      f32x4 rr, zz, nn, hn, h;
      if (v_epi) {
        const f32x4* px = reinterpret_cast<const f32x4*>(smem + O_PART) + (((e - 1) & 1) * 4 + ut) * 3 * 64 + lane;
        const f32x4* ph = px + 2 * 3 * 64;
        const f32x4 xr = px[0], xz = px[64], xn = px[128], gr = ph[0], gz = ph[64], gn = ph[128];
        const f32x4 hp = *reinterpret_cast<const f32x4*>(smem + O_ST + (ce.n % (R + 1)) * 2048 + ut * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rr[i] = fast_sigmoid(xr[i] + gr[i] + br);
          zz[i] = fast_sigmoid(xz[i] + gz[i] + bz);
          hn[i] = gn[i] + bhn;
          nn[i] = fast_tanh(xn[i] + bin + rr[i] * hn[i]);
          h[i] = nn[i] + zz[i] * (hp[i] - nn[i]);
        }
        char* tb = smem + O_TRANS + (e & 1) * 4 * TILE_B;
        h_t* ta = reinterpret_cast<h_t*>(tb);
        hb_t* tbw = reinterpret_cast<hb_t*>(tb + 2 * TILE_B);
        const int col = ut * 16 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ta[(4 * kg + i) * TS + col] = PT::cvt(h[i]);
          tbw[(4 * kg + i) * TS + col] = PB::cvt(h[i]);
        }
        if (drop) {
          const f32x4 m4 = dropout_quad(dc, (uint64_t)((long)(p.t0 + t) * slot + o) >> 2);
          h_t* tda = reinterpret_cast<h_t*>(tb + TILE_B);
          hb_t* tdb = reinterpret_cast<hb_t*>(tb + 3 * TILE_B);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            tda[(4 * kg + i) * TS + col] = PT::cvt(h[i] * m4[i]);
            tdb[(4 * kg + i) * TS + col] = PB::cvt(h[i] * m4[i]);
          }
        }
      } else {
        rr = zz = nn = hn = h = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int vo4 = v_epi ? o * 4 : OOB, vo2 = v_epi ? o * 2 : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), ry, vo4, (int)((unsigned)(t + 1) * (unsigned)(slot * 4)), 0);
      const int so = (int)((unsigned)t * RG);
      auto pk = [](const f32x4& v) {
        return __builtin_bit_cast(u32x2, half4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]});
      };
      __builtin_amdgcn_raw_buffer_store_b64(pk(rr), rsr, vo2, so, 0);
      __builtin_amdgcn_raw_buffer_store_b64(pk(zz), rsz, vo2, so, 0);
      __builtin_amdgcn_raw_buffer_store_b64(pk(nn), rsn, vo2, so, 0);
      __builtin_amdgcn_raw_buffer_store_b64(pk(hn), rshn, vo2, so, 0);
    } else {
      // ---- 3. the 16-bit copies of the tile whose gate math ran in the last iteration leave the transposition tiles ----
      const int t = cs.t, rt = r0_of(cs.sub) + cs.k;
      const int sut = lane >> 5, srow = (lane & 31) >> 1, half = lane & 1;
      const char* tb = smem + O_TRANS + (((e - 1) & 1) * 4 + (wave == 2 ? 0 : 2)) * TILE_B + (srow * TS + sut * 16 + half * 8) * 2;
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(tb);             // state
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(tb + TILE_B);    // state x dropout mask
      const int go = ((t * B + rt * 16 + srow) * D + s * US + sut * 16 + half * 8) * 2;   // row-major, bytes, slot t
      const int xo = ((rt * NS16 + s * UT + sut) * 16 + srow) * 32 + half * 16;
      if (wave == 2) {
        st_sc1(v0, rex, v_st ? xo : OOB, (int)((unsigned)((l * L + t) * 2) * RG));
        st_sc1(v1, rex, (v_st && drop) ? xo : OOB, (int)((unsigned)((l * L + t) * 2 + 1) * RG));
      }
      __builtin_amdgcn_raw_buffer_store_b128(v0, rc0, v_st ? go : OOB, (int)RG, 0);   // (y16a / y16b: slot t + 1)
      __builtin_amdgcn_raw_buffer_store_b128(v1, rc1, v_st ? go : OOB, 0, 0);
    }
    if (v_epi_m) advance(ce);
    if (v_st_m) {
      if (cs.k == nt_of(cs.sub) - 1) p0 = (int)(((long)l * L + cs.t) * NSG + NSUB * g + cs.sub);
      advance(cs);
    }
    RING_STAMP(1);

    // ---- 4. the pieces issued R-2 iterations ago have landed (and this wave's stores of two iterations ago) ----
    //         vector-memory instructions per iteration: wave 0: poll + 9 pieces + 5 stores; 1: 9 + 5; 2: 8 + 4; 3: 8 + 2
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 + (R - 2) * 15) : "memory");
    else if (wave == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 + (R - 2) * 14) : "memory");
    else if (wave == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + (R - 2) * 12) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + (R - 2) * 10) : "memory");
    RING_STAMP(2);
    const bool drained = !more && (hist & ((1u << (R + 2)) - 1u)) == 0u && p0 < 0 && p1 < 0 && p2 < 0;
    // a subgroup's last exchange copies were issued by wave 2 two iterations ago: 2 x 12 instructions lie behind them and
    // this wave has just waited for all but its youngest 16
    if (wave == 2 && p2 >= 0 && lane == 0)
      __hip_atomic_fetch_add(cnt + (long)p2 * kSweepCntStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      // the command for the next iteration: a tile inside a group follows its first tile; a group's first tile needs its
      // two counters, seen by the poll of R-2 iterations ago (older than the pieces this wave has just waited for, and this
      // wave issued it, so its bytes are in LDS now)
      int v = 0;
      if (ci.n < NT) {
        if (ci.k > 0 || (ci.t == 0 && l == 0)) v = 1;
        else {
          const unsigned tag = (unsigned)(tags >> (16 * (R - 2))) & 0xFFFFu;
          const char* pw = smem + O_POLL + ((e + 3 - (R - 2)) % 3) * 256;
          const unsigned own = (unsigned)lds_word(pw), bel = (unsigned)lds_word(pw + 4);
          const bool ok = tag == (unsigned)(ci.t * NSUB + ci.sub) && (ci.t == 0 || (int)(own - need) >= 0) && (l == 0 || (int)(bel - need) >= 0);
          v = __builtin_amdgcn_readfirstlane(ok ? 1 : 0);
        }
      }
      if (drained || fail) v = 2;
      if (lane == 0) {
        lds_set_word(smem + O_FLAG + 8 + ((e + 1) & 1) * 4, v);
        if (fail) lds_set_word(lflag, 1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    RING_STAMP(3);
    p2 = p1; p1 = p0; p0 = -1;
#ifdef ARK_FAT_STAMPS
    facc_[0] += 1;
    if (more && !ready) facc_[1] += 1;
#endif
    if ((__builtin_amdgcn_readfirstlane(lds_word(smem + O_FLAG + 8 + ((e + 1) & 1) * 4)) & 2) != 0) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ARK_FAT_STAMPS
  if (tid == 0 && blockIdx.x < 512) {
    facc_[2] = __builtin_amdgcn_s_memrealtime() - fstart_;
    facc_[4] = (unsigned long long)NT;
    for (int i_ = 0; i_ < 8; ++i_) ark_fat_stamp_buf[blockIdx.x * 8 + i_] = facc_[i_];
  }
  if (lane == 0 && blockIdx.x < 512)   // per-wave phase ticks behind the per-workgroup words: [512*8 + (wg*4 + wave)*8 + phase]
    for (int i_ = 0; i_ < 8; ++i_) ark_fat_stamp_buf[512 * 8 + (blockIdx.x * 4 + wave) * 8 + i_] = rst_[i_];
#endif
  sweep_leave(sync);
}

template <int PREC, int PRECB, int KSW, int UT>
static int launch_fat(const GruFatArgs& p, unsigned grid, unsigned valid, hipStream_t st) {
  constexpr int US = 16 * UT;
  constexpr int LDS = 2 * 4 * 4 * UT * 64 * 16 + kFatMaxTiles * UT * 64 * 16 + 2 * 4 * 16 * (US + 8) * 2 + 64;
  static_assert(LDS <= 160 * 1024, "LDS budget");   // (one workgroup per CU by registers: 4 waves of ~400 VGPRs)
  auto kern = gru_fat_fwd_kernel<PREC, PRECB, KSW, UT>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, LDS) != hipSuccess || per_cu < 1) return ARK_ERR_SHAPE;
  if ((long)valid > (long)cus * per_cu || (long)grid > (long)cus * per_cu) return ARK_ERR_SHAPE;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

template <int PREC, int PRECB>
static int launch_fat_ring(const GruFatArgs& p, unsigned grid, unsigned valid, hipStream_t st) {
  constexpr int R = kFatRing;
  constexpr int LDS = R * 32768 + (R + 1) * 2048 + 2 * 4 * 3 * 1024 + 2 * 4 * 16 * 40 * 2 + 1024 + 3 * 256 + 64;
  static_assert(LDS <= 160 * 1024 && LDS > 80 * 1024, "LDS budget; one workgroup per CU");
  auto kern = gru_fat_ring_kernel<PREC, PRECB>;
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return ARK_ERR_ARG;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, LDS) != hipSuccess || per_cu < 1) return ARK_ERR_SHAPE;
  if ((long)valid > (long)cus * per_cu || (long)grid > (long)cus * per_cu) return ARK_ERR_SHAPE;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

// Which kernel serves (D, wg_slices): the register-staged kernel, unless D = 512 and `wg_slices` == 3 asks for the LDS-ring
// kernel (measured slower: see its header; kept with its tests and tools/fat_time.py / fat_ring_stamps.py).
static bool fat_uses_ring(int D, int wg_slices) { return D == 512 && wg_slices == 3; }
// row groups the forward of (n_layers, B, D) would use on the current device; 0 = unsupported shape / does not fit
static int fat_row_groups(int n_layers, int B, int D, bool ring) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (n_layers <= 0 || n_layers > ARK_SWEEP_MAX_LAYERS || B <= 0 || B % 16 != 0 || (D != 512 && D != 1024)) return 0;
  const int ns = D == 512 ? 16 : 64, rt = B / 16;
  int ng = cus / (n_layers * ns);
  if (ng > rt / 2) ng = rt / 2;                       // every subgroup needs a tile
  if (ng <= 0) return 0;
  if (!ring)
    for (int g = 0; g < ng; ++g)                      // tiles per workgroup <= the LDS state array
      if ((2 * g + 2) * rt / (2 * ng) - (2 * g) * rt / (2 * ng) > ark::kFatMaxTiles) return 0;
  return ng;
}
// subgroups per row group: three where the row group has the tiles (two subgroups stream while the third's producers publish)
static int fat_subgroups(int B, int ng, bool ring) {
  if (!ring) return 2;
  const int per = (B / 16) / ng;
  return per >= 3 ? 3 : 2;
}
extern "C" int ark_gru_fat_row_groups(int n_layers, int B, int D) { return fat_row_groups(n_layers, B, D, false); }
extern "C" long ark_gru_fat_sync_words(int n_layers, int B, int D, int L) {
  const int ng = fat_row_groups(n_layers, B, D, false);   // (sized for three subgroups per row group: either kernel)
  return ark::kSweepSyncHdr + (long)n_layers * L * 3 * (ng > 0 ? ng : 1) * ark::kSweepCntStride;
}

extern "C" int ark_gru_fat_fwd(int prec, int prec_b, const ArkGruSweep* a, void* stream) {
  using namespace ark;
  if (!a || a->n_layers <= 0 || a->n_layers > ARK_SWEEP_MAX_LAYERS || a->B <= 0 || a->D <= 0 || a->L <= 0) return ARK_ERR_ARG;
  if (!a->x0_16 || !a->exch || !a->sync || a->t0 < 0) return ARK_ERR_ARG;
  const int D = a->D, B = a->B, L = a->L, n = a->n_layers;
  if (L > 2047 || 2.0 * n * L * B * D * 2 >= 2147483648.0 || 4.0 * (L + 1) * B * D >= 2147483648.0) return ARK_ERR_SHAPE;   // 32-bit buffer offsets
  const bool ring = fat_uses_ring(D, a->wg_slices);
  const int ng = fat_row_groups(n, B, D, ring);
  if (ng == 0) return ARK_ERR_SHAPE;
  for (int l = 0; l < n; ++l) {
    const ArkGruSweepLayer& y = a->layer[l];
    if (!y.w_ih16 || !y.w_hh16 || !y.b_ih || !y.b_hh || !y.y_t || !y.y16a) return ARK_ERR_ARG;
    if (y.drop_p < 0.f || y.drop_p >= 1.f || (y.drop_p > 0.f && (!y.yd16a || !a->hyper))) return ARK_ERR_ARG;
    if (!y.save_r || !y.save_z || !y.save_n || !y.save_hn) return ARK_ERR_ARG;   // (unconditional stores: see the kernels)
  }
  GruFatArgs p;
  p.a = *a;
  p.NG = ng;
  p.NSUB = fat_subgroups(B, ng, ring);
  const int ns = D == 512 ? 16 : 64, np = n * ng;
  const unsigned valid = (unsigned)(np * ns);
  // XCD map: pairs x, x + 8, ... on XCD x, each with its ns slices; needs ns * ceil(np / 8) <= 32 CUs of an XCD
  const int per_xcd = ns * ((np + 7) / 8);
  p.xcd_pairs = per_xcd <= 32 ? 1 : 0;
  const unsigned grid = p.xcd_pairs ? (unsigned)(8 * per_xcd) : valid;
  hipStream_t st = (hipStream_t)stream;
#define ARK_FAT_GO(PF, PBK)                                                          \
  do {                                                                               \
    if (ring) return launch_fat_ring<PF, PBK>(p, grid, valid, st);                   \
    if (D == 512) return launch_fat<PF, PBK, 4, 2>(p, grid, valid, st);              \
    return launch_fat<PF, PBK, 8, 1>(p, grid, valid, st);                            \
  } while (0)
  if (prec == PREC_F16 && prec_b == PREC_BF16) ARK_FAT_GO(PREC_F16, PREC_BF16);
  if (prec == PREC_F16 && prec_b == PREC_F16) ARK_FAT_GO(PREC_F16, PREC_F16);
  if (prec == PREC_BF16 && prec_b == PREC_BF16) ARK_FAT_GO(PREC_BF16, PREC_BF16);
#undef ARK_FAT_GO
  return ARK_ERR_ARG;
}

#ifdef ARK_FAT_STAMPS
extern "C" int ark_debug_fat_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ark::ark_fat_stamp_buf), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
