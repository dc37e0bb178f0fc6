// Pruned nearest-descriptor search: which (row block, column tile) products of the screening (nn_screen.hip) can be skipped.
//
// On large clouds the arg-min is most of the path (71 % of the flops at 16 k points, 89 % at 64 k), and
// screen_kernel multiplies every block of src descriptors with every tile of ref descriptors.  A tile t of 64 ref descriptors with
// centroid c_t and radius r_t = max_{b in t} |b - c_t| bounds all its columns from BELOW,
//       D(a, b) = |a - b|^2 >= (|a - c_t| - r_t)_+^2        (triangle inequality),
// and any actual distance of a row bounds its minimum from ABOVE.  Two such distances are at hand:
//   * T_j = D(a_j, b_{p_j}), p_j the previous iteration's match (exact fp32, row_prep_kernel), and
//   * the smallest screening upper bound U = L + 2 d >= D over the columns of the tile whose centroid is nearest to a_j
//     (tile_T_kernel; available in iteration 0 as well, and tight when the previous match is stale after a large pose update).
// A tile whose lower bound exceeds min of the two for EVERY row of a block cannot hold the arg-min of any of them - nor tie with
// it: the skipped columns are strictly farther - and the block may skip it.  Nothing approximate enters a decision: the bounds
// only remove work, the surviving products go through the screening and the exact fp32 pick unchanged.
// The bounds bite when tiles are compact and a block's rows agree on which tiles matter.  Descriptors are continuous
// functions of position (model.py:209-235: per-point MLPs of xyz, score and local features), so
//   * ref columns are taken in MORTON ORDER of their points (once per registration: the ref side is loop invariant) - tile
//     radius 0.82 -> 0.39 / 0.13 on 16 k / 64 k-point clouds - and
//   * src rows in the order of their nearest-centroid tile (per iteration; centroid_argmin_kernel).
// Measured on the engine's descriptors (tools/prune_stats.py, "tileT"): 48 - 61 % of the products of the five iterations are
// visited at 16 384 points, 35 - 61 % at 65 536, ~all at 5 000 (79 tiles: the search is not worth pruning there and is not).
// All arithmetic of the bounds is fp32 with explicit safety margins (the exact distances they are compared with are fp32
// evaluations, too): margins of 2e-5 (1 + |a|^2 + |b|^2) against evaluation errors below 4e-6 (1 + ...) - see the kernels.
#include <hipcub/hipcub.hpp>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int TILE = 64;            // = the screening's column tile (DSIR_SCREEN_BC)

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
inline int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

// bounding box of every cloud's points: one 1024-thread block per cloud
__global__ __launch_bounds__(1024) void bbox_kernel(const float* __restrict__ xyz, int64_t cs, int n, float* __restrict__ box) {
  const int cloud = blockIdx.x;
  const float* p = xyz + cloud * cs;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float v = p[(int64_t)i * 3 + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
  __shared__ float sh[6][16];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = lo[k], b = hi[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
    if ((threadIdx.x & 63) == 0) { sh[k][threadIdx.x >> 6] = a; sh[3 + k][threadIdx.x >> 6] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sh[threadIdx.x][0];
    for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
    box[cloud * 6 + threadIdx.x] = v;
  }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Morton code (up to 30 bits) of every point inside its cloud's box (non-finite coordinates: key 0 - any order is a valid order)
__global__ void morton_kernel(const float* __restrict__ xyz, int64_t cs, int n, const float* __restrict__ box, int64_t total, int mbits,
                              uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cloud = (int)(i / n), j = (int)(i % n);
    const float* p = xyz + cloud * cs + (int64_t)j * 3;
    const float* b = box + cloud * 6;
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float ext = b[3 + k] - b[k];
      float t = ext > 0.f ? (p[k] - b[k]) / ext * 1023.f : 0.f;
      t = t == t ? fminf(fmaxf(t, 0.f), 1023.f) : 0.f;
      q[k] = (uint32_t)t;
    }
    // (cloud, code) in one 32-bit key - the code's low bits go when the cloud index needs them -: one device-wide sort
    const uint32_t m = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    key[i] = ((uint32_t)cloud << mbits) | (m >> (30 - mbits));
    val[i] = (uint32_t)j;
  }
}

// order (position -> element) as int32, and its inverse (element -> position)
__global__ void order_kernel(const uint32_t* __restrict__ sorted_val, int n, int64_t total, bool identity, int32_t* __restrict__ order,
                             int32_t* __restrict__ inverse) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = identity ? (int)(i % n) : (int)sorted_val[i];
    order[i] = v;
    if (inverse) inverse[(i / n) * n + v] = (int)(i % n);
  }
}

// one wave per tile of 64 positions of the column order, lane = channel: centroid, |centroid|^2, radius (rounded UP)
__global__ __launch_bounds__(256) void tile_stats_kernel(const float* __restrict__ desc, const int32_t* __restrict__ cols, int K, int nt,
                                                         int64_t tiles_total, float* __restrict__ cen, float* __restrict__ cn2,
                                                         float* __restrict__ rad) {
  const int64_t tg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tg >= tiles_total) return;
  const int lane = threadIdx.x & 63;
  const int pair = (int)(tg / nt), t = (int)(tg % nt);
  const float* D = desc + (int64_t)pair * K * 64;
  const int32_t* C = cols + (int64_t)pair * K;
  const int n = min(TILE, K - t * TILE);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += D[(int64_t)C[t * TILE + i] * 64 + lane];
  const float c = s / (float)n;
  float r2 = 0.f;
  for (int i = 0; i < n; ++i) {
    const float d = D[(int64_t)C[t * TILE + i] * 64 + lane] - c;
    r2 = fmaxf(r2, wave_sum(d * d));
  }
  const float c2 = wave_sum(c * c);
  cen[tg * 64 + lane] = c;
  if (lane == 0) {
    cn2[tg] = c2;
    rad[tg] = sqrtf(r2) * 1.00001f + 1e-6f;          // upper bound of the radius: fp32 sum of 64 squares + sqrt, errors < 1e-6 relative
  }
}

// DOMAIN of the margins below (2e-5 (1 + |a|^2 + |b_prev|^2) on T, 1e-5 (1 + |a|^2 + |c|^2) on L): they cover the fp32 evaluation error
// of a SKIPPED column only while that column's own |b|^2 is of the order of the norms they are built from.  The pruned search is
// reachable from dsir_register alone (engine.hip: `prune` requires the registration's own descriptors), and those are L2-normalised by
// the aggregation chain (model.py:232-233: |a| = |b| = 1 up to rounding); it is not offered to caller-supplied descriptors
// (dsir_nn_match searches exhaustively or through the unpruned screening, whose bound carries every column's own norm).
// per src row: the sort key of the row order (its nearest-centroid tile) and the upper bound from the previous iteration's match,
// T = D(a, b_prev) + margin, D evaluated as nn_match.hip evaluates it (fmaf chain over the channels, the reference's roundings);
// iteration 0 (idx_prev == nullptr): +inf (tile_T_kernel supplies the bound)
__global__ __launch_bounds__(256) void row_prep_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ sa, const float* __restrict__ sb,
                                                       const int32_t* __restrict__ idx_prev, const int32_t* __restrict__ tstar, int J, int K,
                                                       int tbits, int64_t total, bool keep_all, uint32_t* __restrict__ key,
                                                       uint32_t* __restrict__ val, float* __restrict__ T) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pair = (int)(i / J);
    float bound = INFINITY;
    if (idx_prev && !keep_all) {
      int k = idx_prev[i];
      k = k < 0 ? 0 : (k >= K ? K - 1 : k);
      const float4* ap = reinterpret_cast<const float4*>(a + i * 64);
      const float4* bp = reinterpret_cast<const float4*>(b + ((int64_t)pair * K + k) * 64);
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 u = ap[q], v = bp[q];
        acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
      }
      const float san = sa[i], sbn = sb[(int64_t)pair * K + k];
      const float d = __fadd_rn(__fmaf_rn(acc, -2.f, san), sbn);
      // non-finite rows get an infinite bound (they visit every tile)
      if (d == d) bound = d + 2e-5f * (1.f + san + sbn);
    }
    T[i] = bound;
    key[i] = ((uint32_t)pair << tbits) | (uint32_t)tstar[i];
    val[i] = (uint32_t)(i % J);
  }
}

// the ref side's screening operands in column order: 16 bytes per thread (a row of 64 halves = 8 pieces, hi and lo), + the seeds
__global__ __launch_bounds__(256) void permute_ref_kernel(const uint4* __restrict__ bh, const uint4* __restrict__ bl, const float* __restrict__ sb,
                                                          const int32_t* __restrict__ cols, int K, int64_t total, uint4* __restrict__ ph,
                                                          uint4* __restrict__ pl, float* __restrict__ psb) {
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total * 8; f += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = f >> 3;
    const int piece = (int)(f & 7);
    const int64_t src = (i / K) * K + cols[i];
    ph[f] = bh[src * 8 + piece];
    pl[f] = bl[src * 8 + piece];
    if (piece == 0) psb[i] = sb[src];
  }
}

__global__ void prune_account_kernel(const int32_t* __restrict__ tcount, int n, int nt, unsigned long long* __restrict__ acc) {
  unsigned long long kept = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) kept += (unsigned long long)tcount[i];
  kept = (unsigned long long)wave_sum((double)kept);
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(acc, kept);
  if (threadIdx.x == 0) atomicAdd(acc + 1, (unsigned long long)n * (unsigned long long)nt);
}

struct Layout {
  int32_t *cols, *rows, *tlist, *tcount, *queue, *rborder, *tstar;
  float *cen, *cn2, *rad, *T, *box;
  void *ch, *cl;                       // the centroids as the screening's fp16 pairs
  void *pbh, *pbl;                     // the ref side's fp16 pairs in column order
  float* psb;
  uint32_t *k0, *k1, *v0, *v1;
  void* cub;
  size_t cub_bytes, total;
};

Layout carve(void* scratch, int pairs, int J, int K) {
  const int nt = (K + TILE - 1) / TILE;
  const int rpb = nn_screen_rows_per_block(J);
  const int nrb = (J + rpb - 1) / rpb;
  const size_t nmax = (size_t)pairs * (size_t)(J > K ? J : K);
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
  Layout L{};
  L.cols = reinterpret_cast<int32_t*>(take((size_t)pairs * K * 4));
  L.rows = reinterpret_cast<int32_t*>(take((size_t)pairs * J * 4));
  L.tstar = reinterpret_cast<int32_t*>(take((size_t)pairs * J * 4));
  L.tlist = reinterpret_cast<int32_t*>(take((size_t)pairs * nrb * nt * 4));
  L.tcount = reinterpret_cast<int32_t*>(take((size_t)pairs * nrb * 4));
  L.queue = reinterpret_cast<int32_t*>(take(8 * 4));
  L.rborder = reinterpret_cast<int32_t*>(take((size_t)pairs * nrb * 4));
  L.cen = reinterpret_cast<float*>(take((size_t)pairs * nt * 64 * 4));
  L.ch = take((size_t)pairs * nt * 64 * 2);
  L.cl = take((size_t)pairs * nt * 64 * 2);
  L.cn2 = reinterpret_cast<float*>(take((size_t)pairs * nt * 4));
  L.rad = reinterpret_cast<float*>(take((size_t)pairs * nt * 4));
  L.T = reinterpret_cast<float*>(take((size_t)pairs * J * 4));
  L.box = reinterpret_cast<float*>(take((size_t)pairs * 6 * 4));
  L.k0 = reinterpret_cast<uint32_t*>(take(nmax * 4));
  L.k1 = reinterpret_cast<uint32_t*>(take(nmax * 4));
  L.v0 = reinterpret_cast<uint32_t*>(take(nmax * 4));
  L.v1 = reinterpret_cast<uint32_t*>(take(nmax * 4));
  L.pbh = take((size_t)pairs * K * 64 * 2);
  L.pbl = take((size_t)pairs * K * 64 * 2);
  L.psb = reinterpret_cast<float*>(take((size_t)pairs * K * 4));
  // temporary storage of the two radix sorts: sized for the largest sort over all 32 key bits; every sort asks again for its own
  // size and bit range and refuses to run if the answer exceeds this (sort_pairs)
  size_t tmp = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                         (int)nmax, 0, 32) != hipSuccess)
    tmp = 0;                       // no storage: sort_pairs fails, the caller reports the error
  L.cub = take(tmp);
  L.cub_bytes = tmp;
  L.total = (size_t)(p - reinterpret_cast<char*>(scratch));
  return L;
}

// (key, value) radix sort k0 / v0 -> k1 / v1 over key bits [0, end_bit): the library is asked for the temporary size of THIS sort first
int sort_pairs(const Layout& L, int64_t total, int end_bit, hipStream_t st) {
  size_t need = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, need, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                         (int)total, 0, end_bit, st) != hipSuccess || need > L.cub_bytes || L.cub_bytes == 0)
    return 1;
  size_t tmp = L.cub_bytes;
  return hipcub::DeviceRadixSort::SortPairs(L.cub, tmp, L.k0, L.k1, L.v0, L.v1, (int)total, 0, end_bit, st) != hipSuccess;
}

inline int bits_for(int n) { int b = 0; while ((1ll << b) < n) ++b; return b; }   // smallest b with 2^b >= n

}  // namespace

bool nn_prune_supported(int pairs, int J, int K) {
  if (bits_for(pairs) + bits_for(K) > 32 || bits_for(pairs) > 12) return false;     // the sorts' composite 32-bit keys
  return (K + TILE - 1) / TILE <= nn_screen_max_bound_tiles() && (int64_t)pairs * (J > K ? J : K) <= 0x7fffffffll && nn_screen_rows_per_block(J) <= 512;
}

size_t nn_prune_scratch_bytes(int pairs, int J, int K) { return carve(nullptr, pairs, J, K).total; }

// once per registration: the column order (Morton order of the ref points) and the tiles' centroids / radii
int launch_prune_ref(const float* ref_xyz, int64_t xyz_cs, const float* desc_r, const void* bh, const void* bl, const float* sb, int pairs, int J,
                     int K, void* scratch, hipStream_t st) {
  const Layout L = carve(scratch, pairs, J, K);
  const int nt = (K + TILE - 1) / TILE;
  const int64_t total = (int64_t)pairs * K;
  const int pbits = bits_for(pairs);
  const int mbits = 32 - pbits < 30 ? 32 - pbits : 30;
  hipLaunchKernelGGL(bbox_kernel, dim3(pairs), dim3(1024), 0, st, ref_xyz, xyz_cs, K, L.box);
  hipLaunchKernelGGL(morton_kernel, dim3(grid_for(total)), dim3(256), 0, st, ref_xyz, xyz_cs, K, L.box, total, mbits, L.k0, L.v0);
  if (sort_pairs(L, total, mbits + pbits, st)) return 1;
  hipLaunchKernelGGL(order_kernel, dim3(grid_for(total)), dim3(256), 0, st, L.v1, K, total, false, L.cols, (int32_t*)nullptr);
  const int64_t tiles_total = (int64_t)pairs * nt;
  hipLaunchKernelGGL(tile_stats_kernel, dim3((unsigned)((tiles_total + 3) / 4)), dim3(256), 0, st, desc_r, L.cols, K, nt, tiles_total, L.cen, L.cn2,
                     L.rad);
  // the centroids in the screening's operand format (and |c|^2 in its summation order): the bound pass runs its MFMA chain
  launch_split16_norm(L.cen, tiles_total, L.ch, L.cl, L.cn2, st);
  hipLaunchKernelGGL(permute_ref_kernel, dim3(grid_for(total * 8)), dim3(256), 0, st, reinterpret_cast<const uint4*>(bh),
                     reinterpret_cast<const uint4*>(bl), sb, L.cols, K, total, reinterpret_cast<uint4*>(L.pbh), reinterpret_cast<uint4*>(L.pbl), L.psb);
  return 0;
}

// per iteration (>= 1): the row order, the rows' upper bounds and the tile lists of every row block -> ord
int launch_prune_rows(const float* desc_s, const float* desc_r, const void* ah, const void* al, const float* sa, const float* sb,
                      const int32_t* idx_prev, int pairs, int J, int K, void* scratch, hipStream_t st, ScreenOrder* ord, unsigned long long* acc) {
  const Layout L = carve(scratch, pairs, J, K);
  const int nt = (K + TILE - 1) / TILE;
  const int rpb = nn_screen_rows_per_block(J);
  const int nrb = (J + rpb - 1) / rpb;
  const int64_t total = (int64_t)pairs * J;
  const int tbits = bits_for(nt);
  static const bool id_rows = tuning_flag("DSIR_PRUNE_ID_ROWS");     // measurement hook: rows in their natural order
  static const bool no_lpt = tuning_flag("DSIR_PRUNE_NO_LPT");       // A/B hook: items in row-block order
  static const bool keep_all = tuning_flag("DSIR_PRUNE_KEEP_ALL");   // measurement hook: every tile on every list (the mechanism's own cost)
  static const bool no_tile_T = tuning_flag("DSIR_PRUNE_NO_TILE_T"); // A/B hook: upper bounds from the previous match only
  launch_centroid_argmin(ah, al, L.ch, L.cl, L.cn2, pairs, J, nt, L.tstar, st);
  hipLaunchKernelGGL(row_prep_kernel, dim3(grid_for(total)), dim3(256), 0, st, desc_s, desc_r, sa, sb, idx_prev, L.tstar, J, K, tbits, total,
                     keep_all, L.k0, L.v0, L.T);
  if (sort_pairs(L, total, tbits + bits_for(pairs), st)) return 1;
  hipLaunchKernelGGL(order_kernel, dim3(grid_for(total)), dim3(256), 0, st, L.v1, J, total, id_rows, L.rows, (int32_t*)nullptr);
  if (!keep_all && !no_tile_T) launch_tile_T(ah, al, sa, L.rows, L.tstar, L.pbh, L.pbl, L.psb, pairs, J, K, nt, L.T, st);
  launch_tile_bound(ah, al, sa, L.rows, L.T, L.ch, L.cl, L.cn2, L.rad, pairs, J, nt, L.tlist, L.tcount, nt, no_lpt ? nullptr : L.rborder, st);
  if (acc) hipLaunchKernelGGL(prune_account_kernel, dim3(1), dim3(256), 0, st, L.tcount, pairs * nrb, nt, acc);
  ord->rows = L.rows;
  ord->cols = L.cols;
  ord->bh = L.pbh;
  ord->bl = L.pbl;
  ord->sbp = L.psb;
  ord->tlist = L.tlist;
  ord->tcount = L.tcount;
  ord->tl_stride = nt;
  (void)hipMemsetAsync(L.queue, 0, 8 * 4, st);
  ord->queue = L.queue;
  ord->rborder = no_lpt ? nullptr : L.rborder;
  return 0;
}

}  // namespace dsir
