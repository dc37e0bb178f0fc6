// Exact 16-NN through a uniform grid, for the large pyramid levels (n >= 1024; DSIR_GRID_MIN).
// Same result, bit for bit, as the brute force of knn.hip / oracle/knn.py: fp32
// d = (dx*dx + dy*dy) + dz*dz without FMA contraction, neighbours ordered by
// (d, then lower index).  The grid only prunes: a query visits the cells of growing
// Chebyshev shells around its own cell and stops once its 16th best distance is
// strictly (with a safety margin against fp32 rounding) inside the cube already
// visited, so every point that could enter or tie the top 16 has been seen.
//
// Per level and cloud batch: bounding box + cell size, cell histogram, exclusive scan, scatter into cell order
// (xyz + original index packed in a float4) - one launch with the cell table in LDS (grid_build_kernel) up to 16384
// points, five launches beyond -, query (one lane per point, points
// taken in cell order so that a wave's lanes walk neighbouring cells).
// Candidates arrive in arbitrary index order, hence the lexicographic insertion.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

struct GridParams {
  float ox, oy, oz, h, inv_h;
  int gx, gy, gz;
};

constexpr int kPointsPerCell = 6;

__device__ __forceinline__ float sqdist3(float qx, float qy, float qz, float sx, float sy, float sz) {
  const float dx = __fsub_rn(sx, qx), dy = __fsub_rn(sy, qy), dz = __fsub_rn(sz, qz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int g) {
  int c = (int)floorf((p - o) * inv_h);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// one block per cloud: bounding box, then cell size / grid dimensions
__global__ __launch_bounds__(1024) void grid_setup_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n,
                                                          int max_cells, GridParams* __restrict__ gp) {
  __shared__ float red[6][16];
  const int cloud = blockIdx.x;
  const float* P = pts + cloud * cs;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = P[(int64_t)i * stride + k];
      lo[k] = fminf(lo[k], v);
      hi[k] = fmaxf(hi[k], v);
    }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = lo[k], b = hi[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
    if (lane == 0) { red[k][w] = a; red[3 + k][w] = b; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
      mn[k] = red[k][0]; mx[k] = red[3 + k][0];
      for (int ww = 1; ww < (int)(blockDim.x >> 6); ++ww) { mn[k] = fminf(mn[k], red[k][ww]); mx[k] = fmaxf(mx[k], red[3 + k][ww]); }
    }
    float ext[3], emax = 0.f;
    for (int k = 0; k < 3; ++k) { ext[k] = mx[k] - mn[k]; emax = fmaxf(emax, ext[k]); }
    if (!(emax > 0.f) || !isfinite(emax)) emax = 1.f;     // all points coincide (or garbage): one cell
    // cell size for ~kPointsPerCell points per occupied cell; flat / thin clouds are handled by
    // flooring each extent at a fraction of the largest one before taking the volume
    float vol = 1.f;
    for (int k = 0; k < 3; ++k) vol *= fmaxf(ext[k], emax * 1e-3f);
    const float target = fmaxf((float)n / (float)kPointsPerCell, 1.f);
    float h = cbrtf(vol / target);
    if (!(h > 0.f) || !isfinite(h)) h = emax;
    int g[3];
    for (int it = 0; it < 64; ++it) {
      long long prod = 1;
      for (int k = 0; k < 3; ++k) {
        g[k] = (int)fminf(floorf(ext[k] / h) + 1.f, 1.0e6f);
        if (g[k] < 1) g[k] = 1;
        prod *= g[k];
      }
      if (prod <= max_cells) break;
      h *= 1.26f;
      if (it == 63) g[0] = g[1] = g[2] = 1;   // infinite extent: one cell (the search degenerates to the brute force)
    }
    GridParams q;
    q.ox = mn[0]; q.oy = mn[1]; q.oz = mn[2]; q.h = h; q.inv_h = 1.f / h; q.gx = g[0]; q.gy = g[1]; q.gz = g[2];
    gp[cloud] = q;
  }
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n,
                                                         const GridParams* __restrict__ gp, int max_cells,
                                                         int* __restrict__ cell_of, int* __restrict__ counts) {
  const int cloud = blockIdx.y;
  const GridParams g = gp[cloud];
  const float* P = pts + cloud * cs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int cx = cell_coord(P[(int64_t)i * stride], g.ox, g.inv_h, g.gx);
    const int cy = cell_coord(P[(int64_t)i * stride + 1], g.oy, g.inv_h, g.gy);
    const int cz = cell_coord(P[(int64_t)i * stride + 2], g.oz, g.inv_h, g.gz);
    const int c = (cz * g.gy + cy) * g.gx + cx;
    cell_of[(int64_t)cloud * n + i] = c;
    atomicAdd(&counts[(int64_t)cloud * max_cells + c], 1);
  }
}

// one block per cloud: starts = exclusive scan(counts); cursor = starts (consumed by the scatter)
__global__ __launch_bounds__(1024) void grid_scan_kernel(const int* __restrict__ counts, int max_cells,
                                                         const GridParams* __restrict__ gp, int* __restrict__ starts,
                                                         int* __restrict__ cursor) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int cloud = blockIdx.x;
  const GridParams g = gp[cloud];
  const int ncell = g.gx * g.gy * g.gz;
  const int* C = counts + (int64_t)cloud * max_cells;
  int* S = starts + (int64_t)cloud * (max_cells + 1);
  int* U = cursor + (int64_t)cloud * max_cells;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ncell; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ncell ? C[i] : 0;
    int x = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0;
    for (int ww = 0; ww < w; ++ww) woff += wsum[ww];
    const int excl = carry_s + woff + x - v;
    if (i < ncell) { S[i] = excl; U[i] = excl; }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) S[ncell] = carry_s;
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n,
                                                           const int* __restrict__ cell_of, int max_cells,
                                                           int* __restrict__ cursor, float4* __restrict__ sorted) {
  const int cloud = blockIdx.y;
  const float* P = pts + cloud * cs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = cell_of[(int64_t)cloud * n + i];
    const int pos = atomicAdd(&cursor[(int64_t)cloud * max_cells + c], 1);
    sorted[(int64_t)cloud * n + pos] =
        make_float4(P[(int64_t)i * stride], P[(int64_t)i * stride + 1], P[(int64_t)i * stride + 2], __int_as_float(i));
  }
}

// One launch instead of five (memset, setup, count, scan, scatter) for clouds whose cell table fits LDS: one 1024-thread workgroup per
// cloud computes the bounding box and the grid, counts the points per cell in LDS, scans the counts, and scatters the points into cell
// order through LDS cursors.  The order of the points inside a cell follows the arrival of the atomics - as in the five-launch form;
// the searches do not depend on it (lexicographic top-16 / top-1).  kBuildMaxCells ints of LDS twice.
constexpr int kBuildMaxCells = 8256;      // n <= 16384
// body: one 1024-thread workgroup builds the grid of one cloud
__device__ __forceinline__ void grid_build_body(const float* __restrict__ pts, int64_t cs, int stride, int n, int max_cells,
                                                GridParams* __restrict__ gp, int* __restrict__ starts, float4* __restrict__ sorted, const int cloud) {
  __shared__ float red[6][16];
  __shared__ GridParams s_g;
  __shared__ int s_cnt[kBuildMaxCells];
  __shared__ int s_cur[kBuildMaxCells];
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const float* P = pts + cloud * cs;
  // ---- bounding box and grid (grid_setup_kernel, verbatim)
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = P[(int64_t)i * stride + k];
      lo[k] = fminf(lo[k], v);
      hi[k] = fmaxf(hi[k], v);
    }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = lo[k], b = hi[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
    if (lane == 0) { red[k][w] = a; red[3 + k][w] = b; }
  }
  for (int c = threadIdx.x; c < max_cells; c += blockDim.x) s_cnt[c] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
      mn[k] = red[k][0]; mx[k] = red[3 + k][0];
      for (int ww = 1; ww < (int)(blockDim.x >> 6); ++ww) { mn[k] = fminf(mn[k], red[k][ww]); mx[k] = fmaxf(mx[k], red[3 + k][ww]); }
    }
    float ext[3], emax = 0.f;
    for (int k = 0; k < 3; ++k) { ext[k] = mx[k] - mn[k]; emax = fmaxf(emax, ext[k]); }
    if (!(emax > 0.f) || !isfinite(emax)) emax = 1.f;
    float vol = 1.f;
    for (int k = 0; k < 3; ++k) vol *= fmaxf(ext[k], emax * 1e-3f);
    const float target = fmaxf((float)n / (float)kPointsPerCell, 1.f);
    float h = cbrtf(vol / target);
    if (!(h > 0.f) || !isfinite(h)) h = emax;
    int g[3];
    for (int it = 0; it < 64; ++it) {
      long long prod = 1;
      for (int k = 0; k < 3; ++k) {
        g[k] = (int)fminf(floorf(ext[k] / h) + 1.f, 1.0e6f);
        if (g[k] < 1) g[k] = 1;
        prod *= g[k];
      }
      if (prod <= max_cells) break;
      h *= 1.26f;
      if (it == 63) g[0] = g[1] = g[2] = 1;
    }
    GridParams q;
    q.ox = mn[0]; q.oy = mn[1]; q.oz = mn[2]; q.h = h; q.inv_h = 1.f / h; q.gx = g[0]; q.gy = g[1]; q.gz = g[2];
    s_g = q;
    gp[cloud] = q;
    carry_s = 0;
  }
  __syncthreads();
  const GridParams g = s_g;
  const int ncell = g.gx * g.gy * g.gz;
  auto cell_of_pt = [&](int i) {
    const int cx = cell_coord(P[(int64_t)i * stride], g.ox, g.inv_h, g.gx);
    const int cy = cell_coord(P[(int64_t)i * stride + 1], g.oy, g.inv_h, g.gy);
    const int cz = cell_coord(P[(int64_t)i * stride + 2], g.oz, g.inv_h, g.gz);
    return (cz * g.gy + cy) * g.gx + cx;
  };
  // ---- count
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_cnt[cell_of_pt(i)], 1);
  __syncthreads();
  // ---- exclusive scan (grid_scan_kernel): starts -> global, cursors -> LDS
  int* S = starts + (int64_t)cloud * (max_cells + 1);
  for (int base = 0; base < ncell; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ncell ? s_cnt[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0;
    for (int ww = 0; ww < w; ++ww) woff += wsum[ww];
    const int excl = carry_s + woff + x - v;
    if (i < ncell) { S[i] = excl; s_cur[i] = excl; }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) S[ncell] = carry_s;
  // ---- scatter into cell order
  float4* SO = sorted + (int64_t)cloud * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int pos = atomicAdd(&s_cur[cell_of_pt(i)], 1);
    SO[pos] = make_float4(P[(int64_t)i * stride], P[(int64_t)i * stride + 1], P[(int64_t)i * stride + 2], __int_as_float(i));
  }
}
__global__ __launch_bounds__(1024) void grid_build_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n, int max_cells,
                                                          GridParams* __restrict__ gp, int* __restrict__ starts,
                                                          float4* __restrict__ sorted) {
  grid_build_body(pts, cs, stride, n, max_cells, gp, starts, sorted, blockIdx.x);
}
// the grids of SEVERAL levels of the same clouds in one launch (few clouds in flight: launch_knn16_grid_levels); blockIdx.y = level
struct GridLevelsArgs {
  static constexpr int kMax = 4;
  struct Lev { int n, max_cells; GridParams* gp; int* starts; float4* sorted; int32_t* out; int b0; } lev[kMax];
  int nlev;
};
__global__ __launch_bounds__(1024) void grid_build_levels_kernel(const float* __restrict__ pts, int64_t cs, int stride, const GridLevelsArgs A) {
  const GridLevelsArgs::Lev& L = A.lev[blockIdx.y];
  grid_build_body(pts, cs, stride, L.n, L.max_cells, L.gp, L.starts, L.sorted, blockIdx.x);
}

// Sorted top-16 under the lexicographic order (distance, index).  A squared distance is a non-negative float, whose
// bit pattern is monotone as an unsigned integer, so (bits(d) << 32 | index) orders exactly like (d, index): one
// 64-bit compare and four selects per compare-exchange step, no branches.
struct TopLex {
  unsigned long long k[kKnn];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < kKnn; ++t) k[t] = 0x7f8000007fffffffull;   // (+inf, INT_MAX)
  }
  __device__ __forceinline__ float worst() const { return __uint_as_float((unsigned)(k[kKnn - 1] >> 32)); }
  // an empty slot (fewer than 16 finite distances: non-finite coordinates) points at `self`, never out of range
  __device__ __forceinline__ int index(int t, int self) const {
    const int i = (int)(unsigned)(k[t] & 0xffffffffull);
    return i == 0x7fffffff ? self : i;
  }
  static __device__ __forceinline__ void ce(unsigned long long& a, unsigned long long& b) {
    const bool sw = b < a;
    const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
  }
  // The 16 smallest of (this list) U (16 more keys in any order; unused slots hold the sentinel): Batcher's odd-even merge sort of
  // the newcomers (63 compare-exchanges, checked over all 2^16 0/1 inputs when the list was generated), then min(k[i], q[15 - i]) -
  // the lower half of the union as a bitonic sequence - and a bitonic merge (32 compare-exchanges).  ~500 VALU instructions whatever
  // the number of live newcomers, against ~130 per newcomer for insert(): the burst form of 16 insertions.  Keys are unique (the
  // sentinel aside), so the result is the insertions', bit for bit.
  __device__ __forceinline__ void merge16(unsigned long long (&q)[kKnn]) {
#define CE(a, b) ce(q[a], q[b]);
    CE(0,1) CE(2,3) CE(4,5) CE(6,7) CE(8,9) CE(10,11) CE(12,13) CE(14,15) CE(0,2) CE(1,3) CE(4,6) CE(5,7) CE(8,10) CE(9,11) CE(12,14) CE(13,15)
    CE(1,2) CE(5,6) CE(9,10) CE(13,14) CE(0,4) CE(1,5) CE(2,6) CE(3,7) CE(8,12) CE(9,13) CE(10,14) CE(11,15) CE(2,4) CE(3,5) CE(10,12) CE(11,13)
    CE(1,2) CE(3,4) CE(5,6) CE(9,10) CE(11,12) CE(13,14) CE(0,8) CE(1,9) CE(2,10) CE(3,11) CE(4,12) CE(5,13) CE(6,14) CE(7,15) CE(4,8) CE(5,9)
    CE(6,10) CE(7,11) CE(2,4) CE(3,5) CE(6,8) CE(7,9) CE(10,12) CE(11,13) CE(1,2) CE(3,4) CE(5,6) CE(7,8) CE(9,10) CE(11,12) CE(13,14)
#undef CE
#pragma unroll
    for (int t = 0; t < kKnn; ++t) k[t] = q[kKnn - 1 - t] < k[t] ? q[kKnn - 1 - t] : k[t];
#pragma unroll
    for (int j = kKnn / 2; j > 0; j >>= 1)
#pragma unroll
      for (int t = 0; t < kKnn; ++t)
        if ((t ^ j) > t) ce(k[t], k[t ^ j]);
  }
  __device__ __forceinline__ void insert(float dist, int idx) {
    const unsigned long long x = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)idx;
    if (x < k[kKnn - 1]) {
      k[kKnn - 1] = x;
#pragma unroll
      for (int t = kKnn - 1; t > 0; --t) {
        const bool sw = k[t] < k[t - 1];
        const unsigned long long lo = sw ? k[t] : k[t - 1], hi = sw ? k[t - 1] : k[t];
        k[t - 1] = lo; k[t] = hi;
      }
    }
  }
};

// Shell 1 (the 26 cells around the query's own) is walked NEAR FIRST: its nine (z, y) rows in the order centre, the four rows that
// share a face with it, the four diagonal ones - (dz + 1, dy + 1) of row j in two bits each.  Far-first orders fill the top 16 with
// candidates that are evicted again: every eviction is a queue slot, and queue slots are what the merge network is paid for.
constexpr unsigned kNearDz = 0x28215u, kNearDy = 0x22161u;
constexpr int QCAP = 16;   // per-lane candidate queue depth (= kKnn: a full queue is one TopLex::merge16)
constexpr int KB = 64;     // threads per query block (queue = QCAP * KB * 8 bytes of LDS)

// One lane per query (queries taken in cell order, so a wave's lanes walk neighbouring cells).  The sorted
// insertion (15 lexicographic compare-exchange steps, ~130 VALU instructions) is far more expensive than a
// candidate's distance, and with 64 independent lanes SOME lane wants to insert at almost every candidate, so an
// immediate insert() would run its body ~once per candidate for the whole wave.  Candidates that pass the lane's
// threshold (its 16th best distance at the last flush) are therefore parked in a per-lane LDS queue and inserted in
// bursts — all lanes together — when some lane's queue is full and at the end of every shell.  insert() re-checks
// exactly, and TopLex is order independent, so the result is unchanged.
// body: query block bx (KB queries) of one cloud
__device__ __forceinline__ void grid_knn_body(const float4* __restrict__ sorted, const int* __restrict__ starts,
                                              const GridParams* __restrict__ gp, int max_cells, int n,
                                              int32_t* __restrict__ out, int64_t ocs, const int bx, const int cloud) {
  __shared__ float qd[QCAP][KB];
  __shared__ int qi[QCAP][KB];
  const int tid = threadIdx.x;
  const int t = bx * KB + tid;
  const bool live = t < n;
  const GridParams g = gp[cloud];
  const float4* S = sorted + (int64_t)cloud * n;
  const int* ST = starts + (int64_t)cloud * (max_cells + 1);
  const float4 q = S[live ? t : n - 1];
  const int qidx = __float_as_int(q.w);
  const int cx = cell_coord(q.x, g.ox, g.inv_h, g.gx);
  const int cy = cell_coord(q.y, g.oy, g.inv_h, g.gy);
  const int cz = cell_coord(q.z, g.oz, g.inv_h, g.gz);
  TopLex top;
  top.init();
  float thr = INFINITY;
  int nq = 0;
  auto flush = [&]() {
    if (__any(nq > 3)) {                                  // a burst: all 16 queue slots at once (TopLex::merge16)
      unsigned long long x[QCAP];
#pragma unroll
      for (int c = 0; c < QCAP; ++c)
        x[c] = c < nq ? (((unsigned long long)__float_as_uint(qd[c][tid]) << 32) | (unsigned)qi[c][tid]) : 0x7f8000007fffffffull;
      top.merge16(x);
    } else {
#pragma unroll 1
      for (int c = 0; c < 3; ++c)
        if (c < nq) top.insert(qd[c][tid], qi[c][tid]);
    }
    nq = 0;
    thr = top.worst();
  };
  // Four candidates per step: their (clamped) loads are independent, so their latencies overlap instead of adding up.
  auto scan = [&](int b, int e) {
    for (int k = b; k < e; k += 4) {
      if (__any(nq > QCAP - 4)) flush();                  // before the loads: nothing of this step is live across the merge network
      // 32-bit byte offsets on the cloud's (wave-uniform) base: one add + one min per address (n * 16 < 2^32, launch_knn16_grid)
      const uint32_t kb = (uint32_t)k * 16u, eb = (uint32_t)(e - 1) * 16u;
      float4 sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S) + min(kb + 16u * u, eb));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = sqdist3(q.x, q.y, q.z, sv[u].x, sv[u].y, sv[u].z);
        if (k + u < e && d <= thr) { qd[nq][tid] = d; qi[nq][tid] = __float_as_int(sv[u].w); ++nq; }
      }
    }
  };
  const int rmax = live ? max(max(max(cx, g.gx - 1 - cx), max(cy, g.gy - 1 - cy)), max(cz, g.gz - 1 - cz)) : -1;
  // the walk starts with the whole 3 x 3 x 3 cube around the query's cell (r = 1: a cell holds ~6 points, the own cell alone never
  // ends a 16-NN search), then Chebyshev shells r = 2, 3, ...
  for (int r = 1; r <= max(rmax, 1) && live; ++r) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, g.gz - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, g.gy - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1);
    const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
    // the shell's rows (z, y) one after the other; the cell-range loads of row j+1 are issued before row j is scanned
    // row -> up to two candidate ranges [b0,e0), [b1,e1): the whole x-run when the row lies on a face of the shell,
    // else its two end cells
    auto bounds = [&](int j, int& b0, int& e0, int& b1, int& e1) {
      int z = z0 + j / ny, y = y0 + j % ny;
      if (r == 1) { z = cz + (int)((kNearDz >> (2 * j)) & 3u) - 1; y = cy + (int)((kNearDy >> (2 * j)) & 3u) - 1; }
      const bool face = r == 1 || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);   // r = 1: the full 3 x 3 x 3 cube
      const int rowbase = (z * g.gy + y) * g.gx;
      b0 = e0 = b1 = e1 = 0;
      if (r == 1 && ((unsigned)z >= (unsigned)g.gz || (unsigned)y >= (unsigned)g.gy)) return;      // row outside the grid
      if (face) { b0 = ST[rowbase + x0]; e0 = ST[rowbase + x1 + 1]; }
      else {
        if (cx - r >= 0) { b0 = ST[rowbase + cx - r]; e0 = ST[rowbase + cx - r + 1]; }
        if (cx + r <= g.gx - 1) { b1 = ST[rowbase + cx + r]; e1 = ST[rowbase + cx + r + 1]; }
      }
    };
    const int nr = r == 1 ? 9 : nrows;             // shell 1: all nine rows in near-first order, those outside the grid empty
    int b0 = 0, e0 = 0, b1 = 0, e1 = 0;
    if (nr > 0) bounds(0, b0, e0, b1, e1);
    for (int j = 0; j < nr; ++j) {
      int nb0 = 0, ne0 = 0, nb1 = 0, ne1 = 0;
      if (j + 1 < nr) bounds(j + 1, nb0, ne0, nb1, ne1);
      scan(b0, e0);
      scan(b1, e1);
      b0 = nb0; e0 = ne0; b1 = nb1; e1 = ne1;
    }
    flush();
    // distance from the query to the nearest face of the visited cube that still has cells behind it
    float bound = INFINITY;
    if (cx - r > 0) bound = fminf(bound, q.x - (g.ox + (float)(cx - r) * g.h));
    if (cx + r < g.gx - 1) bound = fminf(bound, (g.ox + (float)(cx + r + 1) * g.h) - q.x);
    if (cy - r > 0) bound = fminf(bound, q.y - (g.oy + (float)(cy - r) * g.h));
    if (cy + r < g.gy - 1) bound = fminf(bound, (g.oy + (float)(cy + r + 1) * g.h) - q.y);
    if (cz - r > 0) bound = fminf(bound, q.z - (g.oz + (float)(cz - r) * g.h));
    if (cz + r < g.gz - 1) bound = fminf(bound, (g.oz + (float)(cz + r + 1) * g.h) - q.z);
    if (bound == INFINITY) break;                   // the cube covers the whole grid
    // Points binned by clamped/rounded coordinates can sit a few ulps outside their cell: keep a margin.
    bound -= 1e-4f * g.h;
    if (bound > 0.f && top.worst() < bound * bound * 0.9999f) break;
  }
  if (live) {
    int32_t* o = out + cloud * ocs + (int64_t)qidx * kKnn;
#pragma unroll
    for (int k = 0; k < kKnn; k += 4) *reinterpret_cast<int4*>(o + k) = make_int4(top.index(k, qidx), top.index(k + 1, qidx), top.index(k + 2, qidx), top.index(k + 3, qidx));
  }
}

__global__ __launch_bounds__(KB) __attribute__((amdgpu_waves_per_eu(3, 3))) void grid_knn_kernel(const float4* __restrict__ sorted, const int* __restrict__ starts,
                                                       const GridParams* __restrict__ gp, int max_cells, int n,
                                                       int32_t* __restrict__ out, int64_t ocs) {
  grid_knn_body(sorted, starts, gp, max_cells, n, out, ocs, blockIdx.x, blockIdx.y);
}
// the searches of several levels in one launch: workgroup -> (level, query block) by the levels' first workgroups
__global__ __launch_bounds__(KB) __attribute__((amdgpu_waves_per_eu(3, 3))) void grid_knn_levels_kernel(const GridLevelsArgs A, int64_t ocs) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < GridLevelsArgs::kMax; ++k) l += (k < A.nlev && (int)blockIdx.x >= A.lev[k].b0) ? 1 : 0;
  const GridLevelsArgs::Lev& L = A.lev[l];
  grid_knn_body(L.sorted, L.starts, L.gp, L.max_cells, L.n, L.out, ocs, (int)blockIdx.x - L.b0, blockIdx.y);
}

// Four lanes per query, for launches too small to fill the chip with one lane per query (one or two clouds in flight:
// 5000 queries are 79 waves on 1024 SIMDs, and a query's walk over ~27 cells x 8 points is a chain of dependent loads).
// Lane s of a quad takes the candidates k = b + s, b + s + 4, ... of every cell run, keeps its own sorted top 16 and
// queue; the filter / termination threshold is the minimum over the quad of the lanes' 16th best - an upper bound of the
// query's true 16th best (the k-th smallest of a union is at most that of any part), so the walk stays exact, at worst a
// little longer.  At the end the four sorted lists meet in a 4-way merge by lane 0 of the quad.  Keys are unique
// (distance bits << 32 | index), so the 16 smallest keys - hence the output - are the one-lane kernel's, bit for bit.
constexpr int KB4 = 64;      // threads per block = 16 queries
// body: query block bx (16 queries) of one cloud
__device__ __forceinline__ void grid_knn4_body(const float4* __restrict__ sorted, const int* __restrict__ starts,
                                               const GridParams* __restrict__ gp, int max_cells, int n,
                                               int32_t* __restrict__ out, int64_t ocs, const int bx, const int cloud) {
  __shared__ float qd[QCAP][KB4];
  __shared__ int qi[QCAP][KB4];
  __shared__ unsigned long long mk[KB4][kKnn + 1];      // the quads' lists for the merge (+1: bank spread)
  const int tid = threadIdx.x, sub = tid & 3;
  const int t = bx * (KB4 / 4) + (tid >> 2);
  const bool live = t < n;
  const GridParams g = gp[cloud];
  const float4* S = sorted + (int64_t)cloud * n;
  const int* ST = starts + (int64_t)cloud * (max_cells + 1);
  const float4 q = S[live ? t : n - 1];
  const int qidx = __float_as_int(q.w);
  const int cx = cell_coord(q.x, g.ox, g.inv_h, g.gx);
  const int cy = cell_coord(q.y, g.oy, g.inv_h, g.gy);
  const int cz = cell_coord(q.z, g.oz, g.inv_h, g.gz);
  TopLex top;
  top.init();
  float thr = INFINITY;
  int nq = 0;
  auto quad_min = [&](float v) { v = fminf(v, __shfl_xor(v, 1)); return fminf(v, __shfl_xor(v, 2)); };
  auto flush = [&]() {
    if (__any(nq > 3)) {                                  // a burst: all 16 queue slots at once (TopLex::merge16)
      unsigned long long x[QCAP];
#pragma unroll
      for (int c = 0; c < QCAP; ++c)
        x[c] = c < nq ? (((unsigned long long)__float_as_uint(qd[c][tid]) << 32) | (unsigned)qi[c][tid]) : 0x7f8000007fffffffull;
      top.merge16(x);
    } else {
#pragma unroll 1
      for (int c = 0; c < 3; ++c)
        if (c < nq) top.insert(qd[c][tid], qi[c][tid]);
    }
    nq = 0;
    thr = quad_min(top.worst());
  };
  auto scan = [&](int b, int e) {
    for (int kb = b; kb < e; kb += 16) {                  // trip count uniform over the quad (its lanes shuffle in flush())
      const int k = kb + sub;
      if (__any(nq > QCAP - 4)) flush();
      float4 sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = S[min(k + 4 * u, n - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = sqdist3(q.x, q.y, q.z, sv[u].x, sv[u].y, sv[u].z);
        if (k + 4 * u < e && d <= thr) { qd[nq][tid] = d; qi[nq][tid] = __float_as_int(sv[u].w); ++nq; }
      }
    }
  };
  const int rmax = live ? max(max(max(cx, g.gx - 1 - cx), max(cy, g.gy - 1 - cy)), max(cz, g.gz - 1 - cz)) : -1;
  for (int r = 1; r <= max(rmax, 1) && live; ++r) {       // from the full 3 x 3 x 3 cube, as grid_knn_kernel
    const int z0 = max(cz - r, 0), z1 = min(cz + r, g.gz - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, g.gy - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1);
    const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
    auto bounds = [&](int j, int& b0, int& e0, int& b1, int& e1) {
      int z = z0 + j / ny, y = y0 + j % ny;
      if (r == 1) { z = cz + (int)((kNearDz >> (2 * j)) & 3u) - 1; y = cy + (int)((kNearDy >> (2 * j)) & 3u) - 1; }
      const bool face = r == 1 || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);   // r = 1: the full 3 x 3 x 3 cube
      const int rowbase = (z * g.gy + y) * g.gx;
      b0 = e0 = b1 = e1 = 0;
      if (r == 1 && ((unsigned)z >= (unsigned)g.gz || (unsigned)y >= (unsigned)g.gy)) return;      // row outside the grid
      if (face) { b0 = ST[rowbase + x0]; e0 = ST[rowbase + x1 + 1]; }
      else {
        if (cx - r >= 0) { b0 = ST[rowbase + cx - r]; e0 = ST[rowbase + cx - r + 1]; }
        if (cx + r <= g.gx - 1) { b1 = ST[rowbase + cx + r]; e1 = ST[rowbase + cx + r + 1]; }
      }
    };
    const int nr = r == 1 ? 9 : nrows;             // shell 1: all nine rows in near-first order, those outside the grid empty
    int b0 = 0, e0 = 0, b1 = 0, e1 = 0;
    if (nr > 0) bounds(0, b0, e0, b1, e1);
    for (int j = 0; j < nr; ++j) {
      int nb0 = 0, ne0 = 0, nb1 = 0, ne1 = 0;
      if (j + 1 < nr) bounds(j + 1, nb0, ne0, nb1, ne1);
      scan(b0, e0);
      scan(b1, e1);
      b0 = nb0; e0 = ne0; b1 = nb1; e1 = ne1;
    }
    flush();
    float bound = INFINITY;
    if (cx - r > 0) bound = fminf(bound, q.x - (g.ox + (float)(cx - r) * g.h));
    if (cx + r < g.gx - 1) bound = fminf(bound, (g.ox + (float)(cx + r + 1) * g.h) - q.x);
    if (cy - r > 0) bound = fminf(bound, q.y - (g.oy + (float)(cy - r) * g.h));
    if (cy + r < g.gy - 1) bound = fminf(bound, (g.oy + (float)(cy + r + 1) * g.h) - q.y);
    if (cz - r > 0) bound = fminf(bound, q.z - (g.oz + (float)(cz - r) * g.h));
    if (cz + r < g.gz - 1) bound = fminf(bound, (g.oz + (float)(cz + r + 1) * g.h) - q.z);
    if (bound == INFINITY) break;
    bound -= 1e-4f * g.h;
    if (bound > 0.f && thr < bound * bound * 0.9999f) break;      // thr = quad minimum of the 16th bests (set by flush)
  }
  // 4-way merge of the quad's sorted lists
#pragma unroll
  for (int k = 0; k < kKnn; ++k) mk[tid][k] = top.k[k];
  __builtin_amdgcn_wave_barrier();
  if (live && sub == 0) {
    int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    int32_t* o = out + cloud * ocs + (int64_t)qidx * kKnn;
    const unsigned long long sentinel = 0x7f8000007fffffffull;
    for (int k = 0; k < kKnn; ++k) {
      const unsigned long long a0 = h0 < kKnn ? mk[tid][h0] : ~0ull, a1 = h1 < kKnn ? mk[tid + 1][h1] : ~0ull;
      const unsigned long long a2 = h2 < kKnn ? mk[tid + 2][h2] : ~0ull, a3 = h3 < kKnn ? mk[tid + 3][h3] : ~0ull;
      const unsigned long long m01 = a0 < a1 ? a0 : a1, m23 = a2 < a3 ? a2 : a3, m = m01 < m23 ? m01 : m23;
      if (m == a0) ++h0; else if (m == a1) ++h1; else if (m == a2) ++h2; else ++h3;
      const int i = (int)(unsigned)(m & 0xffffffffull);
      o[k] = (m == sentinel || i == 0x7fffffff) ? qidx : i;
    }
  }
}
__global__ __launch_bounds__(KB4) void grid_knn4_kernel(const float4* __restrict__ sorted, const int* __restrict__ starts,
                                                        const GridParams* __restrict__ gp, int max_cells, int n,
                                                        int32_t* __restrict__ out, int64_t ocs) {
  grid_knn4_body(sorted, starts, gp, max_cells, n, out, ocs, blockIdx.x, blockIdx.y);
}
// the searches of several levels in one launch: workgroup -> (level, query block) by the levels' first workgroups
__global__ __launch_bounds__(KB4) void grid_knn4_levels_kernel(const GridLevelsArgs A, int64_t ocs) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < GridLevelsArgs::kMax; ++k) l += (k < A.nlev && (int)blockIdx.x >= A.lev[k].b0) ? 1 : 0;
  const GridLevelsArgs::Lev& L = A.lev[l];
  grid_knn4_body(L.sorted, L.starts, L.gp, L.max_cells, L.n, L.out, ocs, (int)blockIdx.x - L.b0, blockIdx.y);
}

}  // namespace

// Nearest SUPPORT point (support = the first n_support points of the cloud = the next pyramid level) of every query point,
// through the grid that level's own 16-NN search has just built: the same shell walk and termination as grid_knn_kernel with a
// top-1 instead of a top-16 - bit for bit the brute force of knn.hip::nn1_kernel (smallest (distance, index); a query
// without any finite distance gets 0).  One lane per query, queries in their natural order.  Brute force costs
// n_query x n_support distances per cloud (65536 x 16384 at the top level of a 64 k-point cloud: 296 us per launch of two
// clouds), the walk ~27 cells x 8 points per query.
__global__ __launch_bounds__(256) void grid_nn1_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n_query,
                                                       const float4* __restrict__ sorted, const int* __restrict__ starts,
                                                       const GridParams* __restrict__ gp, int max_cells, int n_support,
                                                       int32_t* __restrict__ out, int64_t ocs, const float4* __restrict__ qsorted) {
  const int cloud = blockIdx.y;
  int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= n_query) return;
  const float* P = pts + cloud * cs;
  float qx, qy, qz;
  if (qsorted) {           // the query level's points in ITS cell order (round 4): a wave's lanes are neighbours in space
    const float4 q4 = qsorted[(int64_t)cloud * n_query + qi];
    qx = q4.x; qy = q4.y; qz = q4.z; qi = __float_as_int(q4.w);
  } else {
    qx = P[(int64_t)qi * stride]; qy = P[(int64_t)qi * stride + 1]; qz = P[(int64_t)qi * stride + 2];
  }
  const GridParams g = gp[cloud];
  const float4* S = sorted + (int64_t)cloud * n_support;
  const int* ST = starts + (int64_t)cloud * (max_cells + 1);
  const int cx = cell_coord(qx, g.ox, g.inv_h, g.gx);
  const int cy = cell_coord(qy, g.oy, g.inv_h, g.gy);
  const int cz = cell_coord(qz, g.oz, g.inv_h, g.gz);
  unsigned long long best = ~0ull;
  auto scan = [&](int b, int e) {
    for (int k = b; k < e; k += 4) {
      float4 sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = S[min(k + u, e - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = sqdist3(qx, qy, qz, sv[u].x, sv[u].y, sv[u].z);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(sv[u].w);
        if (k + u < e && d < INFINITY && key < best) best = key;      // NaN and inf distances never win (nn1_kernel: d < bd)
      }
    }
  };
  const int rmax = max(max(max(cx, g.gx - 1 - cx), max(cy, g.gy - 1 - cy)), max(cz, g.gz - 1 - cz));
  for (int r = 0; r <= rmax; ++r) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, g.gz - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, g.gy - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        // a row of the shell: the whole x-run when it lies on a face, else its two end cells
        const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
        const int rowbase = (z * g.gy + y) * g.gx;
        if (face) scan(ST[rowbase + x0], ST[rowbase + x1 + 1]);
        else {
          if (cx - r >= 0) scan(ST[rowbase + cx - r], ST[rowbase + cx - r + 1]);
          if (cx + r <= g.gx - 1) scan(ST[rowbase + cx + r], ST[rowbase + cx + r + 1]);
        }
      }
    // distance from the query to the nearest face of the visited cube that still has cells behind it (grid_knn_kernel)
    float bound = INFINITY;
    if (cx - r > 0) bound = fminf(bound, qx - (g.ox + (float)(cx - r) * g.h));
    if (cx + r < g.gx - 1) bound = fminf(bound, (g.ox + (float)(cx + r + 1) * g.h) - qx);
    if (cy - r > 0) bound = fminf(bound, qy - (g.oy + (float)(cy - r) * g.h));
    if (cy + r < g.gy - 1) bound = fminf(bound, (g.oy + (float)(cy + r + 1) * g.h) - qy);
    if (cz - r > 0) bound = fminf(bound, qz - (g.oz + (float)(cz - r) * g.h));
    if (cz + r < g.gz - 1) bound = fminf(bound, (g.oz + (float)(cz + r + 1) * g.h) - qz);
    if (bound == INFINITY) break;                   // the cube covers the whole grid
    bound -= 1e-4f * g.h;                           // points binned by rounded coordinates can sit a few ulps outside their cell
    if (bound > 0.f && best != ~0ull && __uint_as_float((unsigned)(best >> 32)) < bound * bound * 0.9999f) break;
  }
  out[cloud * ocs + qi] = best == ~0ull ? 0 : (int32_t)(unsigned)(best & 0xffffffffull);
}

size_t knn_grid_scratch_bytes(int clouds, int n) {
  const size_t max_cells = (size_t)n / 2 + 64;
  size_t b = 0;
  b += ((size_t)clouds * sizeof(GridParams) + 255) & ~(size_t)255;
  b += ((size_t)clouds * n * sizeof(int) + 255) & ~(size_t)255;                 // cell_of
  b += ((size_t)clouds * max_cells * sizeof(int) + 255) & ~(size_t)255;        // counts
  b += ((size_t)clouds * (max_cells + 1) * sizeof(int) + 255) & ~(size_t)255;  // starts
  b += ((size_t)clouds * max_cells * sizeof(int) + 255) & ~(size_t)255;        // cursor
  b += ((size_t)clouds * n * sizeof(float4) + 255) & ~(size_t)255;             // sorted
  return b;
}

void launch_knn16_grid(const float* pts, int64_t cs, int stride, int n, int clouds, int32_t* out, int64_t ocs,
                       void* scratch, hipStream_t st) {
  const int max_cells = n / 2 + 64;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  GridParams* gp = reinterpret_cast<GridParams*>(take((size_t)clouds * sizeof(GridParams)));
  int* cell_of = reinterpret_cast<int*>(take((size_t)clouds * n * sizeof(int)));
  int* counts = reinterpret_cast<int*>(take((size_t)clouds * max_cells * sizeof(int)));
  int* starts = reinterpret_cast<int*>(take((size_t)clouds * (max_cells + 1) * sizeof(int)));
  int* cursor = reinterpret_cast<int*>(take((size_t)clouds * max_cells * sizeof(int)));
  float4* sorted = reinterpret_cast<float4*>(take((size_t)clouds * n * sizeof(float4)));
  static const bool no_build = tuning_flag("DSIR_GRID_NO_BUILD");     // A/B switch: the five-launch construction throughout
  if (!no_build && max_cells <= kBuildMaxCells) {
    // cell table in LDS: bounding box, grid, count, scan and scatter in ONE launch, one workgroup per cloud
    hipLaunchKernelGGL(grid_build_kernel, dim3(clouds), dim3(1024), 0, st, pts, cs, stride, n, max_cells, gp, starts, sorted);
  } else {
    hipMemsetAsync(counts, 0, (size_t)clouds * max_cells * sizeof(int), st);
    hipLaunchKernelGGL(grid_setup_kernel, dim3(clouds), dim3(1024), 0, st, pts, cs, stride, n, max_cells, gp);
    const int gx = (n + 255) / 256;
    hipLaunchKernelGGL(grid_count_kernel, dim3(gx, clouds), dim3(256), 0, st, pts, cs, stride, n, gp, max_cells, cell_of, counts);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(clouds), dim3(1024), 0, st, counts, max_cells, gp, starts, cursor);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(gx, clouds), dim3(256), 0, st, pts, cs, stride, n, cell_of, max_cells, cursor, sorted);
  }
  // one lane per query when that fills the chip (1024 SIMDs), four lanes per query for a few clouds in flight; same bits
  if ((int64_t)((n + KB - 1) / KB) * clouds >= 1024)
    hipLaunchKernelGGL(grid_knn_kernel, dim3((n + KB - 1) / KB, clouds), dim3(KB), 0, st, sorted, starts, gp, max_cells, n, out, ocs);
  else
    hipLaunchKernelGGL(grid_knn4_kernel, dim3((n + KB4 / 4 - 1) / (KB4 / 4), clouds), dim3(KB4), 0, st, sorted, starts, gp, max_cells, n, out, ocs);
}

// does launch_knn16_grid build this level's grid in one launch (the form the several-levels launch below takes)?
bool knn16_grid_can_merge(int n) {
  static const bool no_build = tuning_flag("DSIR_GRID_NO_BUILD");
  return !no_build && n / 2 + 64 <= kBuildMaxCells;
}

// The grid-pruned searches of several levels of the same clouds (each knn16_grid_can_merge) in TWO launches - all the grids, then all
// the searches - instead of two per level in a registration's dependent chain.  One lane per query when the launch as a whole fills the
// chip that way, else four (the rule of launch_knn16_grid on the sum of the levels).  The same kernels' bodies on the same operands,
// and the two search forms give the same bits: the result equals the separate launches'.  `scratch[l]` as for launch_knn16_grid(n[l]).
void launch_knn16_grid_levels(const float* pts, int64_t cs, int stride, int nlev, const int* n, int clouds, int32_t* const* out, int64_t ocs,
                              void* const* scratch, hipStream_t st) {
  GridLevelsArgs A{};
  A.nlev = nlev;
  int64_t b1 = 0;
  for (int l = 0; l < nlev; ++l) b1 += (n[l] + KB - 1) / KB;
  const bool one_lane = b1 * clouds >= 1024;
  const int per = one_lane ? KB : KB4 / 4;          // queries per workgroup
  int b = 0;
  for (int l = 0; l < nlev; ++l) {
    const int max_cells = n[l] / 2 + 64;
    char* p = reinterpret_cast<char*>(scratch[l]);
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    GridLevelsArgs::Lev& L = A.lev[l];
    L.n = n[l]; L.max_cells = max_cells; L.out = out[l]; L.b0 = b;
    L.gp = reinterpret_cast<GridParams*>(take((size_t)clouds * sizeof(GridParams)));
    take((size_t)clouds * n[l] * sizeof(int));                   // cell_of   (the carving of launch_knn16_grid)
    take((size_t)clouds * max_cells * sizeof(int));              // counts
    L.starts = reinterpret_cast<int*>(take((size_t)clouds * (max_cells + 1) * sizeof(int)));
    take((size_t)clouds * max_cells * sizeof(int));              // cursor
    L.sorted = reinterpret_cast<float4*>(take((size_t)clouds * n[l] * sizeof(float4)));
    b += (n[l] + per - 1) / per;
  }
  hipLaunchKernelGGL(grid_build_levels_kernel, dim3(clouds, nlev), dim3(1024), 0, st, pts, cs, stride, A);
  if (one_lane) hipLaunchKernelGGL(grid_knn_levels_kernel, dim3(b, clouds), dim3(KB), 0, st, A, ocs);
  else hipLaunchKernelGGL(grid_knn4_levels_kernel, dim3(b, clouds), dim3(KB4), 0, st, A, ocs);
}

// nn1 through the grid launch_knn16_grid(.., n = n_support, .., scratch) has left in `scratch` (same carving)
// the `sorted` array inside a scratch of launch_knn16_grid(.., n, ..) (same carving as below)
static const float4* sorted_of(const void* scratch, int clouds, int n) {
  const int max_cells = n / 2 + 64;
  const char* p = reinterpret_cast<const char*>(scratch);
  auto take = [&](size_t bytes) { const char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  take((size_t)clouds * sizeof(GridParams));
  take((size_t)clouds * n * sizeof(int));
  take((size_t)clouds * max_cells * sizeof(int));
  take((size_t)clouds * (max_cells + 1) * sizeof(int));
  take((size_t)clouds * max_cells * sizeof(int));
  return reinterpret_cast<const float4*>(take((size_t)clouds * n * sizeof(float4)));
}

void launch_nn1_grid(const float* pts, int64_t cs, int stride, int n_query, int n_support, int clouds, int32_t* out, int64_t ocs,
                     const void* scratch, hipStream_t st, const void* query_scratch) {
  const int n = n_support;
  const int max_cells = n / 2 + 64;
  const char* p = reinterpret_cast<const char*>(scratch);
  auto take = [&](size_t bytes) { const char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  const GridParams* gp = reinterpret_cast<const GridParams*>(take((size_t)clouds * sizeof(GridParams)));
  take((size_t)clouds * n * sizeof(int));                    // cell_of
  take((size_t)clouds * max_cells * sizeof(int));            // counts
  const int* starts = reinterpret_cast<const int*>(take((size_t)clouds * (max_cells + 1) * sizeof(int)));
  take((size_t)clouds * max_cells * sizeof(int));            // cursor
  const float4* sorted = reinterpret_cast<const float4*>(take((size_t)clouds * n * sizeof(float4)));
  hipLaunchKernelGGL(grid_nn1_kernel, dim3((n_query + 255) / 256, clouds), dim3(256), 0, st, pts, cs, stride, n_query, sorted, starts, gp,
                     max_cells, n_support, out, ocs, query_scratch ? sorted_of(query_scratch, clouds, n_query) : nullptr);
}

}  // namespace dsir
