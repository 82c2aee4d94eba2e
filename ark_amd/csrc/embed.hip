// Embedding gathers and their scatter-add gradients (HBM/LDS-bound; no MFMA here).
// Replaces: nn.Embedding x3 + torch.cat + masked mean pool (reference kgvae/model/models.py:47-58),
// tok_emb / pos_emb lookup (models.py:138, 343) and autograd's embedding_backward.
#include "common.h"
#include "../../include/ark_amd.h"

namespace ark {

// ---- encoder: g[b, :] = (1/max(1,cnt_b)) * sum_t m[b,t] * [E[h] | R[r] | E[t]] -----------------
// One workgroup per graph; each thread owns float4 columns of the 3D-wide row, so every table row
// read is a coalesced 16 B/lane stream (tables of the syn-* sets are L2-resident).
__device__ __forceinline__ void put16x4(void* base, long idx4, f32x4 v, int prec) {
  if (prec == PREC_F16) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    auto c = [](float x) { return (_Float16)fminf(fmaxf(x, -65504.0f), 65504.0f); };
    reinterpret_cast<h4*>(base)[idx4] = h4{c(v[0]), c(v[1]), c(v[2]), c(v[3])};
  } else {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<b4*>(base)[idx4] = b4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
}

__device__ __forceinline__ void enc_pool_fwd_body(int b, const int64_t* __restrict__ triples, const float* __restrict__ E,
                                                           const float* __restrict__ R, float* __restrict__ g,
                                                           float* __restrict__ inv_cnt, int T, int D, long pad_rid,
                                                           void* g16a, int prec_a, void* g16b, int prec_b) {
  const int64_t* tr = triples + (long)b * T * 3;
  int cnt = 0;
  for (int t = 0; t < T; ++t) cnt += (pad_rid < 0 || tr[t * 3 + 1] != pad_rid) ? 1 : 0;
  // reference: masked sum / clamp(cnt,1) when pad_rid is set, plain mean over T otherwise
  const float w = 1.0f / (float)(pad_rid < 0 ? T : (cnt > 0 ? cnt : 1));
  if (threadIdx.x == 0 && inv_cnt) inv_cnt[b] = w;
  const int D4 = D >> 2;
  // One workgroup per graph walks its T triples alone: what it costs is load latency (the entity table of the wd-* sets
  // is HBM-resident, rows are random).  Every thread therefore owns TWO float4 columns at a time and keeps 2 x 8 row
  // loads in flight (a load-add chain of T = 212 was 307 us; 8 in flight 210 us).  Summation order per column: t ascending.
  for (int c0 = threadIdx.x; c0 < 3 * D4; c0 += 2 * blockDim.x) {
    const int c1 = c0 + blockDim.x;
    const bool two = c1 < 3 * D4;
    const int part0 = c0 / D4, d40 = c0 % D4;
    const int part1 = two ? c1 / D4 : part0, d41 = two ? c1 % D4 : d40;
    const float* tab0 = (part0 == 1) ? R : E;
    const float* tab1 = (part1 == 1) ? R : E;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += 8) {
      f32x4 v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u;
        const bool ok = t < T && !(pad_rid >= 0 && tr[t * 3 + 1] == pad_rid);
        const long id0 = ok ? tr[t * 3 + part0] : 0, id1 = ok ? tr[t * 3 + part1] : 0;
        v0[u] = ok ? *reinterpret_cast<const f32x4*>(tab0 + id0 * D + 4 * d40) : f32x4{0.f, 0.f, 0.f, 0.f};
        v1[u] = (ok && two) ? *reinterpret_cast<const f32x4*>(tab1 + id1 * D + 4 * d41) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += v0[u]; s1 += v1[u]; }
    }
    const f32x4 r0 = s0 * w, r1 = s1 * w;
    *reinterpret_cast<f32x4*>(g + (long)b * 3 * D + 4 * c0) = r0;
    if (g16a) put16x4(g16a, (long)b * 3 * D4 + c0, r0, prec_a);   // 16-bit operand copies for the MLP products
    if (g16b) put16x4(g16b, (long)b * 3 * D4 + c0, r0, prec_b);
    if (two) {
      *reinterpret_cast<f32x4*>(g + (long)b * 3 * D + 4 * c1) = r1;
      if (g16a) put16x4(g16a, (long)b * 3 * D4 + c1, r1, prec_a);
      if (g16b) put16x4(g16b, (long)b * 3 * D4 + c1, r1, prec_b);
    }
  }
}

// Long graphs (wd-articles: T = 212 triples, 16 graphs): one workgroup per graph is 16 workgroups walking 212 random table
// rows each, one after the other -- 210 us of load latency.  Here a workgroup owns 64 float4 columns of one graph and its
// four waves each sum a quarter of the triples (eight row loads in flight per lane, the indices staged in LDS); the
// quarters meet in LDS in a fixed order, so the result is deterministic (it differs from the t-ascending sum in rounding).
__device__ __forceinline__ void enc_pool_fwd_split(int b, int cb, const int64_t* __restrict__ triples, const float* __restrict__ E,
                                                   const float* __restrict__ R, float* __restrict__ g, float* __restrict__ inv_cnt,
                                                   int T, int D, long pad_rid, void* g16a, int prec_a, void* g16b, int prec_b) {
  extern __shared__ __attribute__((aligned(16))) char smem_pool[];
  long* ids = reinterpret_cast<long*>(smem_pool);                       // [T][3], relation id = -1 for padding triples
  f32x4* part = reinterpret_cast<f32x4*>(smem_pool + (size_t)((T * 3 * 8 + 15) / 16) * 16);   // [4 waves][64 columns]
  const int64_t* tr = triples + (long)b * T * 3;
  for (int i = threadIdx.x; i < 3 * T; i += 256) ids[i] = tr[i];
  __syncthreads();
  int cnt = 0;
  for (int t = 0; t < T; ++t) cnt += (pad_rid < 0 || ids[t * 3 + 1] != pad_rid) ? 1 : 0;
  const float w = 1.0f / (float)(pad_rid < 0 ? T : (cnt > 0 ? cnt : 1));
  if (threadIdx.x == 0 && cb == 0 && inv_cnt) inv_cnt[b] = w;
  const int D4 = D >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = cb * 64 + lane;
  const bool live = c < 3 * D4;
  const int part_i = live ? c / D4 : 0, d4 = live ? c % D4 : 0;
  const float* tab = (part_i == 1) ? R : E;
  const int tq = (T + 3) / 4, t_lo = wave * tq, t_hi = min(T, t_lo + tq);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = t_lo; t0 < t_hi; t0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u;
      const bool ok = live && t < t_hi && !(pad_rid >= 0 && ids[t * 3 + 1] == pad_rid);
      v[u] = ok ? *reinterpret_cast<const f32x4*>(tab + ids[t * 3 + part_i] * D + 4 * d4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  part[wave * 64 + lane] = s;
  __syncthreads();
  if (wave == 0 && live) {
    const f32x4 r = (((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane]) * w;
    *reinterpret_cast<f32x4*>(g + (long)b * 3 * D + 4 * c) = r;
    if (g16a) put16x4(g16a, (long)b * 3 * D4 + c, r, prec_a);
    if (g16b) put16x4(g16b, (long)b * 3 * D4 + c, r, prec_b);
  }
}

__global__ __launch_bounds__(256) void enc_pool_fwd_kernel(const int64_t* __restrict__ triples, const float* __restrict__ E,
                                                           const float* __restrict__ R, float* __restrict__ g,
                                                           float* __restrict__ inv_cnt, int T, int D, long pad_rid,
                                                           void* g16a, int prec_a, void* g16b, int prec_b) {
  enc_pool_fwd_body(blockIdx.x, triples, E, R, g, inv_cnt, T, D, pad_rid, g16a, prec_a, g16b, prec_b);
}

// The first launch of a SAIL step: the encoder pool (blocks [0, B)) and the decoder's token gather (the blocks after
// them) have nothing to do with each other except that both only need the batch indices, so they share one launch
// instead of two dependent ones.  The gather role also bumps the dropout draw counter (see tok_gather16_kernel).
struct PoolGatherArgs {
  const int64_t* triples; const float* E; const float* R; float* g; float* inv_cnt; void* g16a; void* g16b;
  const int64_t* seq; const float* Wt; void* xa; void* xb; float* hyper_tick; int* tok_tm;
  long pad_rid, ld_seq;
  int B, T, D, Dd, L, prec_a, prec_b;
  int col_blocks;   // 0: one pool workgroup per graph; > 0: this many 64-column workgroups per graph (long graphs)
};
__global__ __launch_bounds__(256) void pool_gather_fwd_kernel(PoolGatherArgs p) {
  ARK_CHAIN_PRIO();
  const int npool = p.col_blocks > 0 ? p.B * p.col_blocks : p.B;
  if ((int)blockIdx.x < npool) {
    if (p.col_blocks > 0)
      enc_pool_fwd_split(blockIdx.x / p.col_blocks, blockIdx.x % p.col_blocks, p.triples, p.E, p.R, p.g, p.inv_cnt, p.T, p.D,
                         p.pad_rid, p.g16a, p.prec_a, p.g16b, p.prec_b);
    else
      enc_pool_fwd_body(blockIdx.x, p.triples, p.E, p.R, p.g, p.inv_cnt, p.T, p.D, p.pad_rid, p.g16a, p.prec_a, p.g16b, p.prec_b);
    return;
  }
  const int blk = blockIdx.x - npool, nblk = gridDim.x - npool;
  if (p.hyper_tick && blk == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t*>(p.hyper_tick)[kHpDropStep] += 1u;
  if (!p.xa) {   // small vocabulary: the forward cells read rows of W_tok W_ih^T by token id, no embedding rows needed
    const long rows = (long)p.B * p.L;
    for (long row = (long)blk * 256 + threadIdx.x; row < rows; row += (long)nblk * 256) {
      const int t = (int)(row / p.B), b = (int)(row % p.B);
      p.tok_tm[row] = (int)p.seq[(long)b * p.ld_seq + t];
    }
    return;
  }
  const int D4 = p.Dd >> 2;
  const long total = (long)p.B * p.L * D4;
  for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
    const int d4 = (int)(i % D4);
    const long row = i / D4;
    const int t = (int)(row / p.B), b = (int)(row % p.B);
    const long tok = p.seq[(long)b * p.ld_seq + t];
    if (p.tok_tm && d4 == 0) p.tok_tm[row] = (int)tok;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.Wt + tok * p.Dd + 4 * d4);
    put16x4(p.xa, row * D4 + d4, v, p.prec_a);
    if (p.xb) put16x4(p.xb, row * D4 + d4, v, p.prec_b);
  }
}

__global__ __launch_bounds__(256) void tok_time_major_kernel(const int64_t* __restrict__ seq, long ld_seq, int* __restrict__ tok_tm,
                                                             int B, int L, float* hyper_tick) {
  if (hyper_tick && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t*>(hyper_tick)[kHpDropStep] += 1u;
  const long rows = (long)B * L;
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long)gridDim.x * 256) {
    const int t = (int)(row / B), b = (int)(row % B);
    tok_tm[row] = (int)seq[(long)b * ld_seq + t];
  }
}

// ---- decoder input: x[(t,b), :] = W_tok[seq[b, t]] (+ W_pos[t])   (time-major rows) -----------
__global__ __launch_bounds__(256) void tok_gather_kernel(const int64_t* __restrict__ seq, long ld_seq,
                                                         const float* __restrict__ Wt, const float* __restrict__ Wp,
                                                         float* __restrict__ x, int B, int L, int D, float* hyper_tick) {
  // training forward: bump the dropout draw counter (see common.h; the GRU cells run after this kernel)
  if (hyper_tick && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t*>(hyper_tick)[kHpDropStep] += 1u;
  const int D4 = D >> 2;
  const long total = (long)B * L * D4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const long row = i / D4;
    const int t = (int)(row / B), b = (int)(row % B);
    const long tok = seq[(long)b * ld_seq + t];
    f32x4 v = *reinterpret_cast<const f32x4*>(Wt + tok * D + 4 * d4);
    if (Wp) v += *reinterpret_cast<const f32x4*>(Wp + (long)t * D + 4 * d4);
    *reinterpret_cast<f32x4*>(x + row * D + 4 * d4) = v;
  }
}

// ---- scatter-add of gradient rows into a table --------------------------------------------
// Source item i (0..n_items-1) adds  scale * src[src_row(i), col_off + 0..D)  into dst[id(i), :].
// Small tables (syn-*: <= 192 rows) are privatised in LDS per workgroup: 64-column slices x item
// chunks, ds_add_f32 inside, then ONE global atomic per (row, column) per workgroup -- this keeps
// the float-atomic traffic at table-size x chunks instead of one contended add per item
// (MI355X: every workgroup adding into one row runs 14x below the atomic rate).
// Large tables (wd-*) use direct global atomics, one wave per item, 256 contiguous bytes per
// wave-instruction.
struct ScatterArgs {
  float* dst; const float* src; long ld_src; int col_off;
  const int64_t* ids; long id_stride_outer, id_stride_inner; int inner;  // id(i) = ids[(i/inner)*outer + (i%inner)*inner_stride + id_off]
  int id_off;
  const int64_t* mask_ids; int mask_off; long mask_val;   // skip item when mask_ids[... + mask_off] == mask_val
  const float* scale;     // per source row, nullable
  int src_div;            // src_row(i) = time_major ? (i%inner)*B_outer + i/inner : i / src_div
  int time_major; int n_outer;
  long skip_id;           // padding_idx row: never receives gradient (-1: none)
  int n_items, D, n_rows;
};

__device__ __forceinline__ bool scatter_item(const ScatterArgs& p, int i, long& id, long& srow, float& sc) {
  const int o = i / p.inner, j = i % p.inner;
  const long base = (long)o * p.id_stride_outer + (long)j * p.id_stride_inner;
  id = p.ids[base + p.id_off];
  if (id == p.skip_id) return false;
  if (p.mask_ids && p.mask_ids[base + p.mask_off] == p.mask_val) return false;
  srow = p.time_major ? ((long)j * p.n_outer + o) : (long)(i / p.src_div);
  sc = p.scale ? p.scale[srow] : 1.0f;
  return true;
}

__global__ __launch_bounds__(256) void scatter_lds_kernel(ScatterArgs p, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);  // [n_rows][64]
  const int slice = blockIdx.x / n_chunks, chunk = blockIdx.x % n_chunks;
  const int c0 = slice * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < p.n_rows * 64; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int per = (p.n_items + n_chunks - 1) / n_chunks;
  const int i0 = chunk * per, i1 = min(p.n_items, i0 + per);
  const bool col_ok = (c0 + lane) < p.D;
  // 4 items per wave per trip: the id / row / value loads of the four are independent, so their
  // latencies overlap (a single dependent chain per item made this kernel latency-bound)
  constexpr int U = 4;
  for (int i = i0 + wave * U; i < i1; i += 4 * U) {
    long id[U], srow[U]; float sc[U], v[U]; bool ok[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      id[k] = 0; srow[k] = 0; sc[k] = 0.f;
      ok[k] = (i + k < i1) && scatter_item(p, i + k, id[k], srow[k], sc[k]) && col_ok;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ok[k] ? p.src[srow[k] * p.ld_src + p.col_off + c0 + lane] : 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (ok[k]) atomicAdd(&tab[id[k] * 64 + lane], sc[k] * v[k]);
  }
  __syncthreads();
  if (col_ok)
    for (int r = wave; r < p.n_rows; r += 4) {
      const float v = tab[r * 64 + lane];
      if (v != 0.f) atomicAdd(&p.dst[(long)r * p.D + c0 + lane], v);
    }
}

__global__ __launch_bounds__(256) void scatter_global_kernel(ScatterArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  for (int i = wave_global; i < p.n_items; i += n_waves) {
    long id, srow; float sc;
    if (!scatter_item(p, i, id, srow, sc)) continue;
    const float* s = p.src + srow * p.ld_src + p.col_off;
    float* d = p.dst + id * p.D;
    // exact zeros are skipped: the rows behind PAD inputs (half of a padded batch) have an all-zero gradient and all
    // hit ONE table row -- same-address atomics serialise (wd-movies trace: 228 us for 2.3 M adds before, see DESIGN)
    for (int c = lane; c < p.D; c += 64) {
      const float v = sc * s[c];
      if (v != 0.f) atomicAdd(&d[c], v);
    }
  }
}

// all three encoder scatters (head -> E, relation -> R, tail -> E) of one column slice in one pass:
// both tables are privatised in LDS; item = (graph b, triple t)
__global__ __launch_bounds__(256) void enc_scatter_fused_kernel(const int64_t* __restrict__ triples, const float* __restrict__ dg,
                                                                const float* __restrict__ inv_cnt, float* __restrict__ dE,
                                                                float* __restrict__ dR, int B, int T, int D, int n_ent, int n_rel,
                                                                long pad_eid, long pad_rid, int n_chunks) {
  ARK_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* te = reinterpret_cast<float*>(smem);        // [n_ent][64]
  float* tr = te + (size_t)n_ent * 64;               // [n_rel][64]
  const int slice = blockIdx.x / n_chunks, chunk = blockIdx.x % n_chunks;
  const int c0 = slice * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (n_ent + n_rel) * 64; i += 256) te[i] = 0.f;
  __syncthreads();
  const int n_items = B * T;
  const int per = (n_items + n_chunks - 1) / n_chunks;
  const int i0 = chunk * per, i1 = min(n_items, i0 + per);
  const bool col_ok = (c0 + lane) < D;
  constexpr int U = 4;
  for (int i = i0 + wave * U; i < i1; i += 4 * U) {
    long h[U], r[U], t[U]; float sc[U], vh[U], vr[U], vt[U]; bool ok[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int it = min(i + k, i1 - 1);
      const int64_t* p3 = triples + (long)it * 3;
      h[k] = p3[0]; r[k] = p3[1]; t[k] = p3[2];
      ok[k] = (i + k < i1) && col_ok && !(pad_rid >= 0 && r[k] == pad_rid);
      sc[k] = inv_cnt[it / T];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int it = min(i + k, i1 - 1);
      const float* src = dg + (long)(it / T) * 3 * D + c0 + lane;
      vh[k] = ok[k] ? src[0] : 0.f; vr[k] = ok[k] ? src[D] : 0.f; vt[k] = ok[k] ? src[2 * D] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (ok[k]) {
        if (h[k] != pad_eid) atomicAdd(&te[h[k] * 64 + lane], sc[k] * vh[k]);
        if (r[k] != pad_rid) atomicAdd(&tr[r[k] * 64 + lane], sc[k] * vr[k]);
        if (t[k] != pad_eid) atomicAdd(&te[t[k] * 64 + lane], sc[k] * vt[k]);
      }
  }
  __syncthreads();
  if (col_ok) {
    for (int row = wave; row < n_ent; row += 4) {
      const float v = te[row * 64 + lane];
      if (v != 0.f) atomicAdd(&dE[(long)row * D + c0 + lane], v);
    }
    for (int row = wave; row < n_rel; row += 4) {
      const float v = tr[row * 64 + lane];
      if (v != 0.f) atomicAdd(&dR[(long)row * D + c0 + lane], v);
    }
  }
}

#ifndef ARK_SCATTER_CHUNK
#define ARK_SCATTER_CHUNK 48
#endif
constexpr int g_scatter_chunk_items = ARK_SCATTER_CHUNK;

static int launch_scatter(const ScatterArgs& p, hipStream_t st) {
  if (p.n_items <= 0) return 0;
  if ((long)p.n_rows * 64 * 4 <= 48 * 1024) {
    const int slices = (p.D + 63) / 64;
    // chunks trade item-loop length against the fixed per-workgroup cost (LDS table zero + flush of
    // n_rows x 64 global atomics): ~chunk_items items per workgroup, at most ~512 workgroups
    int n_chunks = (p.n_items + g_scatter_chunk_items - 1) / g_scatter_chunk_items;
    if (n_chunks > 2048 / slices) n_chunks = 2048 / slices;
    if (n_chunks < 1) n_chunks = 1;
    hipLaunchKernelGGL(scatter_lds_kernel, dim3(slices * n_chunks), dim3(256), (size_t)p.n_rows * 64 * 4, st, p, n_chunks);
  } else {
    int grid = (p.n_items + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scatter_global_kernel, dim3(grid), dim3(256), 0, st, p);
  }
  ARK_LAUNCH_CHECK();
  return 0;
}

}  // namespace ark

extern "C" int ark_enc_pool_fwd(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt,
                                int B, int T, int D, int64_t pad_rid, void* stream) {
  using namespace ark;
  if (!triples || !E || !R || !g || B <= 0 || T <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 4 != 0) return ARK_ERR_SHAPE;
  hipLaunchKernelGGL(enc_pool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, triples, E, R, g, inv_cnt, T, D,
                     (long)pad_rid, (void*)nullptr, 0, (void*)nullptr, 0);
  ARK_LAUNCH_CHECK();
  return 0;
}

// same, additionally writing 16-bit copies of g (g16a in prec_a, g16b nullable in prec_b)
extern "C" int ark_enc_pool_fwd16(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt,
                                  void* g16a, int prec_a, void* g16b, int prec_b, int B, int T, int D, int64_t pad_rid,
                                  void* stream) {
  using namespace ark;
  if (!triples || !E || !R || !g || !g16a || B <= 0 || T <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 4 != 0) return ARK_ERR_SHAPE;
  if ((prec_a != PREC_F16 && prec_a != PREC_BF16) || (g16b && prec_b != PREC_F16 && prec_b != PREC_BF16)) return ARK_ERR_ARG;
  hipLaunchKernelGGL(enc_pool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, triples, E, R, g, inv_cnt, T, D,
                     (long)pad_rid, g16a, prec_a, g16b, prec_b);
  ARK_LAUNCH_CHECK();
  return 0;
}

// encoder pool (as ark_enc_pool_fwd16) and decoder token gather (as ark_tok_gather16 without a position table) in ONE
// launch: x16a/x16b[(t, b), :] = cast(W_tok[seq[b, t]]), t < L, rows time-major, D_dec columns
extern "C" int ark_pool_gather_fwd16(const int64_t* triples, const float* E, const float* R, float* g, float* inv_cnt,
                                     void* g16a, int prec_a, void* g16b, int prec_b, int B, int T, int D, int64_t pad_rid,
                                     const int64_t* seq, int64_t ld_seq, const float* w_tok, void* x16a, void* x16b, int L,
                                     int D_dec, int* tok_tm, float* hyper_tick, void* stream) {
  using namespace ark;
  if (!triples || !E || !R || !g || !g16a || !seq || !w_tok || (!x16a && !tok_tm) || B <= 0 || T <= 0 || D <= 0 || L <= 0 || D_dec <= 0)
    return ARK_ERR_ARG;
  if (!x16a && x16b) return ARK_ERR_ARG;
  if (D % 4 != 0 || D_dec % 4 != 0) return ARK_ERR_SHAPE;
  if ((prec_a != PREC_F16 && prec_a != PREC_BF16) || (g16b && prec_b != PREC_F16 && prec_b != PREC_BF16)) return ARK_ERR_ARG;
  if (x16a && (g16b == nullptr) != (x16b == nullptr)) return ARK_ERR_ARG;   // one backward type for both (or none)
  PoolGatherArgs p{triples, E, R, g, inv_cnt, g16a, g16b, seq, w_tok, x16a, x16b, hyper_tick, tok_tm, (long)pad_rid, (long)ld_seq,
                   B, T, D, D_dec, L, prec_a, prec_b, 0};
  const long total = x16a ? (long)B * L * (D_dec / 4) : (long)B * L;
  long gb = (total + 255) / 256; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
  // long graphs in a small batch: the pool is all load latency -- spread each graph over column blocks and its triples over waves
  size_t lds = 0;
  if (T >= 32 && (long)B * ((3 * (D / 4) + 63) / 64) <= 2048) {
    p.col_blocks = (3 * (D / 4) + 63) / 64;
    lds = (size_t)((T * 3 * 8 + 15) / 16) * 16 + 4 * 64 * 16;
    if (lds > 60 * 1024) { p.col_blocks = 0; lds = 0; }
  }
  const long npool = p.col_blocks > 0 ? (long)B * p.col_blocks : B;
  hipLaunchKernelGGL(pool_gather_fwd_kernel, dim3((unsigned)(npool + gb)), dim3(256), lds, (hipStream_t)stream, p);
  ARK_LAUNCH_CHECK();
  return 0;
}

extern "C" int ark_tok_time_major(const int64_t* seq, int64_t ld_seq, int* tok_tm, int B, int L, float* hyper_tick, void* stream) {
  if (!seq || !tok_tm || B <= 0 || L <= 0) return ARK_ERR_ARG;
  const long rows = (long)B * L;
  long gb = (rows + 255) / 256; if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(ark::tok_time_major_kernel, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, seq, (long)ld_seq, tok_tm, B, L,
                     hyper_tick);
  ARK_LAUNCH_CHECK();
  return 0;
}

// dE / dR += scatter of dg (gradient of the pooled encoder input); accumulates into dE, dR.
extern "C" int ark_enc_pool_bwd(const int64_t* triples, const float* dg, const float* inv_cnt, float* dE, float* dR,
                                int B, int T, int D, int n_ent, int n_rel, int64_t pad_eid, int64_t pad_rid,
                                void* stream) {
  using namespace ark;
  if (!triples || !dg || !inv_cnt || !dE || !dR || B <= 0 || T <= 0 || D <= 0) return ARK_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if ((long)(n_ent + n_rel) * 64 * 4 <= 48 * 1024) {   // both tables fit LDS: one fused launch
    const int slices = (D + 63) / 64;
    int n_chunks = (B * T + g_scatter_chunk_items - 1) / g_scatter_chunk_items;
    if (n_chunks > 2048 / slices) n_chunks = 2048 / slices;
    if (n_chunks < 1) n_chunks = 1;
    hipLaunchKernelGGL(enc_scatter_fused_kernel, dim3(slices * n_chunks), dim3(256), (size_t)(n_ent + n_rel) * 64 * 4, st,
                       triples, dg, inv_cnt, dE, dR, B, T, D, n_ent, n_rel, (long)pad_eid, (long)pad_rid, n_chunks);
    ARK_LAUNCH_CHECK();
    return 0;
  }
  for (int part = 0; part < 3; ++part) {
    ScatterArgs p{};
    p.dst = (part == 1) ? dR : dE; p.src = dg; p.ld_src = 3L * D; p.col_off = part * D;
    p.ids = triples; p.id_stride_outer = 3L * T; p.id_stride_inner = 3; p.inner = T; p.id_off = part;
    p.mask_ids = pad_rid >= 0 ? triples : nullptr; p.mask_off = 1; p.mask_val = (long)pad_rid;
    p.scale = inv_cnt; p.src_div = T; p.time_major = 0; p.n_outer = B;
    p.skip_id = (part == 1) ? (long)pad_rid : (long)pad_eid;
    p.n_items = B * T; p.D = D; p.n_rows = (part == 1) ? n_rel : n_ent;
    int rc = launch_scatter(p, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int ark_tok_gather(const int64_t* seq, int64_t ld_seq, const float* w_tok, const float* w_pos, float* x,
                              int B, int L, int D, float* hyper_tick, void* stream) {
  using namespace ark;
  if (!seq || !w_tok || !x || B <= 0 || L <= 0 || D <= 0) return ARK_ERR_ARG;
  if (D % 4 != 0) return ARK_ERR_SHAPE;
  long total = (long)B * L * (D / 4);
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(tok_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, seq, (long)ld_seq, w_tok, w_pos,
                     x, B, L, D, hyper_tick);
  ARK_LAUNCH_CHECK();
  return 0;
}

// dW_tok[seq[b,t], :] += dx[(t,b), :]   (accumulates; the tied projection gradient is already in dW_tok)
extern "C" int ark_tok_scatter(const int64_t* seq, int64_t ld_seq, const float* dx, float* d_w_tok, int B, int L, int D,
                               int vocab, void* stream) {
  using namespace ark;
  if (!seq || !dx || !d_w_tok || B <= 0 || L <= 0 || D <= 0 || vocab <= 0) return ARK_ERR_ARG;
  ScatterArgs p{};
  p.dst = d_w_tok; p.src = dx; p.ld_src = D; p.col_off = 0;
  p.ids = seq; p.id_stride_outer = (long)ld_seq; p.id_stride_inner = 1; p.inner = L; p.id_off = 0;
  p.mask_ids = nullptr; p.scale = nullptr; p.src_div = 1; p.time_major = 1; p.n_outer = B; p.skip_id = -1;
  p.n_items = B * L; p.D = D; p.n_rows = vocab;
  return launch_scatter(p, (hipStream_t)stream);
}
