// Brute-force KNN for the pyramid (reference dataloader/data_base.py:153-183;
// the reference delegates to torch_points_kernels.knn on the CPU).
//
// One block = 64 queries x 4 waves.  Wave w scans the w-th quarter of the
// support set, staged through a wave-private LDS tile (xyz padded to float4,
// broadcast ds_read_b128), and keeps a sorted top-16 (distance, index) list in
// registers (fully unrolled compare-exchange chain, static indexing).  The four
// partial lists are merged through LDS by wave 0.  Tie rule = oracle/knn.py:
// fp32 d = (dx*dx + dy*dy) + dz*dz without FMA contraction, ties to the lower
// support index (scan order is ascending and every comparison is strict).
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int QB = 64;      // queries per block (one per lane)
constexpr int NW = 4;       // waves per block = support slices
constexpr int TILE = 256;   // support points staged per wave per step
constexpr int QCAP = 16;    // per-lane candidate queue depth (flushed when fewer than 8 free slots remain)

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, const float4& s) {
  const float dx = __fsub_rn(s.x, qx), dy = __fsub_rn(s.y, qy), dz = __fsub_rn(s.z, qz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

struct Top16 {
  float d[kKnn];
  int i[kKnn];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < kKnn; ++t) { d[t] = INFINITY; i[t] = -1; }
  }
  __device__ __forceinline__ void insert(float dist, int idx) {
    if (dist < d[kKnn - 1]) {
      d[kKnn - 1] = dist; i[kKnn - 1] = idx;
#pragma unroll
      for (int t = kKnn - 1; t > 0; --t) {
        const bool sw = d[t] < d[t - 1];
        const float dl = sw ? d[t] : d[t - 1], dh = sw ? d[t - 1] : d[t];
        const int il = sw ? i[t] : i[t - 1], ih = sw ? i[t - 1] : i[t];
        d[t - 1] = dl; d[t] = dh; i[t - 1] = il; i[t] = ih;
      }
    }
  }
  // The 16 smallest of this list and another SORTED list under the order (distance, then index) - what inserting the other list's
  // entries one by one gives when its indices all exceed this list's at equal distance (the slices of the support set are merged in
  // ascending order).  min(k[t], q[15 - t]) is a bitonic sequence holding the lower half of the union; a bitonic merge (32
  // compare-exchanges) sorts it: ~350 instructions instead of 16 x 75.
  __device__ __forceinline__ void merge_sorted(const float (&qd)[kKnn], const int (&qi)[kKnn]) {
#pragma unroll
    for (int t = 0; t < kKnn; ++t) {
      const float e = qd[kKnn - 1 - t];
      const int ei = qi[kKnn - 1 - t];
      const bool sw = e < d[t] || (e == d[t] && ei < i[t]);
      d[t] = sw ? e : d[t]; i[t] = sw ? ei : i[t];
    }
#pragma unroll
    for (int j = kKnn / 2; j > 0; j >>= 1)
#pragma unroll
      for (int t = 0; t < kKnn; ++t)
        if ((t ^ j) > t) {
          const int u = t ^ j;
          const bool sw = d[u] < d[t] || (d[u] == d[t] && i[u] < i[t]);
          const float dl = sw ? d[u] : d[t], dh = sw ? d[t] : d[u];
          const int il = sw ? i[u] : i[t], ih = sw ? i[t] : i[u];
          d[t] = dl; d[u] = dh; i[t] = il; i[u] = ih;
        }
  }
};

struct Knn16Smem {
  float4 tile[NW][TILE];
  float md[NW - 1][kKnn][QB];
  int mi[NW - 1][kKnn][QB];
  float qd[NW][QCAP][QB];
  int qi[NW][QCAP][QB];
};
// body: query block bx of one cloud (P its points, out its lists); all QB * NW threads of the workgroup
__device__ __forceinline__ void knn16_body(const float* __restrict__ P, int stride, int n, int32_t* __restrict__ out, int bx, Knn16Smem& sm) {
  auto& tile = sm.tile;
  auto& md = sm.md;
  auto& mi = sm.mi;
  auto& qd = sm.qd;
  auto& qi = sm.qi;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = bx * QB + lane;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < n) { qx = P[(int64_t)q * stride]; qy = P[(int64_t)q * stride + 1]; qz = P[(int64_t)q * stride + 2]; }
  const int slice = (n + NW - 1) / NW;
  const int s_begin = w * slice;
  const int s_end = min(n, s_begin + slice);
  Top16 top;
  top.init();
  // Candidates that beat the lane's current 16th distance are parked in a small
  // per-lane LDS queue; the (75-instruction, divergent) sorted insertion runs only
  // when some lane's queue is full, for all lanes at once.  The threshold is only
  // refreshed at a flush, so the queue may hold a few candidates insert() rejects:
  // harmless, insert() re-checks.  Order of arrival (ascending index) is kept.
  float thr = INFINITY;
  int nq = 0;
  auto flush = [&]() {
#pragma unroll 1
    for (int c = 0; c < QCAP; ++c)
      if (c < nq) top.insert(qd[w][c][lane], qi[w][c][lane]);
    nq = 0;
    thr = top.d[kKnn - 1];
  };
  for (int t0 = 0; t0 < slice; t0 += TILE) {
    // stage: each lane loads TILE/64 support points of this wave's slice
#pragma unroll
    for (int r = 0; r < TILE / 64; ++r) {
      const int j = s_begin + t0 + r * 64 + lane;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < s_end) { v.x = P[(int64_t)j * stride]; v.y = P[(int64_t)j * stride + 1]; v.z = P[(int64_t)j * stride + 2]; }
      tile[w][r * 64 + lane] = v;
    }
    __syncthreads();
    const int cnt = max(0, min(TILE, s_end - (s_begin + t0)));
    // 8 support points per step: the 8 broadcast ds_read_b128 are independent, so their latency overlaps
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      float dd[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 s = tile[w][j0 + u];
        dd[u] = (j0 + u < cnt) ? sqdist(qx, qy, qz, s) : INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dd[u] < thr) { qd[w][nq][lane] = dd[u]; qi[w][nq][lane] = s_begin + t0 + j0 + u; ++nq; }
      if (__any(nq > QCAP - 8)) flush();
    }
    __syncthreads();
  }
  flush();
  if (w > 0) {
#pragma unroll
    for (int t = 0; t < kKnn; ++t) { md[w - 1][t][lane] = top.d[t]; mi[w - 1][t][lane] = top.i[t]; }
  }
  __syncthreads();
  if (w == 0 && q < n) {
#pragma unroll
    for (int s = 0; s < NW - 1; ++s) {
      float ld[kKnn];
      int li[kKnn];
#pragma unroll
      for (int t = 0; t < kKnn; ++t) { ld[t] = md[s][t][lane]; li[t] = mi[s][t][lane]; }
      top.merge_sorted(ld, li);
    }
    int32_t* o = out + (int64_t)q * kKnn;
    // fewer than 16 finite distances (non-finite coordinates): the empty slots point at the query itself, never out of range
#pragma unroll
    for (int t = 0; t < kKnn; ++t) top.i[t] = top.i[t] < 0 ? q : top.i[t];
#pragma unroll
    for (int t = 0; t < kKnn; t += 4) *reinterpret_cast<int4*>(o + t) = make_int4(top.i[t], top.i[t + 1], top.i[t + 2], top.i[t + 3]);
  }
}
__global__ __launch_bounds__(QB * NW) void knn16_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n,
                                                        int32_t* __restrict__ out, int64_t ocs) {
  __shared__ Knn16Smem sm;
  knn16_body(pts + blockIdx.y * cs, stride, n, out + blockIdx.y * ocs, blockIdx.x, sm);
}

// One WAVE per query, for the small pyramid levels when few clouds are in flight (312 and 78 points per cloud at N = 5000:
// with one lane per query that is 5 and 2 workgroups per cloud, each lane walking the whole support through a sorted
// insertion).  Lane l holds the keys (distance bits << 32 | index) of support points l, l + 64, ...; the 16 smallest keys
// of the wave are extracted by 16 rounds of a wave-wide minimum (keys are unique, so each round removes exactly one).
// Same keys, same order as Top16 / oracle/knn.py: distance, then lower index - bit-identical output.
constexpr int WQ_MAX = 16;    // support points per lane: n <= 1024
// body: query block bx of one cloud (P its points, out its lists); no LDS, no barrier
__device__ __forceinline__ void knn16_wave_body(const float* __restrict__ P, int stride, int n, int32_t* __restrict__ out, int bx) {
  const int lane = threadIdx.x & 63;
  const int q = bx * 4 + (threadIdx.x >> 6);
  if (q >= n) return;                                  // wave-uniform
  const float qx = P[(int64_t)q * stride], qy = P[(int64_t)q * stride + 1], qz = P[(int64_t)q * stride + 2];
  unsigned long long key[WQ_MAX];
#pragma unroll
  for (int i = 0; i < WQ_MAX; ++i) {
    const int j = lane + 64 * i;
    key[i] = ~0ull;
    if (j < n) {
      float4 s;
      s.x = P[(int64_t)j * stride]; s.y = P[(int64_t)j * stride + 1]; s.z = P[(int64_t)j * stride + 2]; s.w = 0.f;
      const float d = sqdist(qx, qy, qz, s);
      if (d == d && d < INFINITY) key[i] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;   // finite, >= 0: bits are monotone
    }
  }
  int mine = -1;                                        // lane t keeps the t-th neighbour
  for (int t = 0; t < kKnn; ++t) {
    unsigned long long m = key[0];
#pragma unroll
    for (int i = 1; i < WQ_MAX; ++i) m = key[i] < m ? key[i] : m;
    unsigned long long w = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long v = __shfl_xor(w, o);
      w = v < w ? v : w;
    }
    if (lane == t) mine = w == ~0ull ? q : (int)(unsigned)(w & 0xffffffffull);   // fewer than 16 finite distances: the query itself
#pragma unroll
    for (int i = 0; i < WQ_MAX; ++i) key[i] = key[i] == w ? ~0ull : key[i];      // unique keys: removes exactly the winner
  }
  if (lane < kKnn) out[(int64_t)q * kKnn + lane] = mine;
}
__global__ __launch_bounds__(256) void knn16_wave_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n,
                                                         int32_t* __restrict__ out, int64_t ocs) {
  knn16_wave_body(pts + blockIdx.y * cs, stride, n, out + blockIdx.y * ocs, blockIdx.x);
}

// nearest support point (support = first n_support points) of every query point
struct Nn1Smem {
  float4 tile[NW][TILE];
  float md[NW][QB];
  int mi[NW][QB];
};
// body: query block bx of one cloud (P its points, out its list); all QB * NW threads of the workgroup
__device__ __forceinline__ void nn1_body(const float* __restrict__ P, int stride, int n_query, int n_support, int32_t* __restrict__ out, int bx,
                                         Nn1Smem& sm) {
  auto& tile = sm.tile;
  auto& md = sm.md;
  auto& mi = sm.mi;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = bx * QB + lane;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < n_query) { qx = P[(int64_t)q * stride]; qy = P[(int64_t)q * stride + 1]; qz = P[(int64_t)q * stride + 2]; }
  const int slice = (n_support + NW - 1) / NW;
  const int s_begin = w * slice;
  const int s_end = min(n_support, s_begin + slice);
  float bd = INFINITY;
  int bi = -1;
  for (int t0 = 0; t0 < slice; t0 += TILE) {
#pragma unroll
    for (int r = 0; r < TILE / 64; ++r) {
      const int j = s_begin + t0 + r * 64 + lane;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < s_end) { v.x = P[(int64_t)j * stride]; v.y = P[(int64_t)j * stride + 1]; v.z = P[(int64_t)j * stride + 2]; }
      tile[w][r * 64 + lane] = v;
    }
    __syncthreads();
    const int cnt = max(0, min(TILE, s_end - (s_begin + t0)));
    for (int j = 0; j < cnt; ++j) {
      const float d = sqdist(qx, qy, qz, tile[w][j]);
      if (d < bd) { bd = d; bi = s_begin + t0 + j; }
    }
    __syncthreads();
  }
  md[w][lane] = bd; mi[w][lane] = bi;
  __syncthreads();
  if (w == 0 && q < n_query) {
#pragma unroll
    for (int s = 1; s < NW; ++s) {
      const float d = md[s][lane];
      if (d < bd) { bd = d; bi = mi[s][lane]; }
    }
    out[q] = bi < 0 ? 0 : bi;   // no finite distance (non-finite coordinates): stay in range
  }
}
__global__ __launch_bounds__(QB * NW) void nn1_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n_query,
                                                      int n_support, int32_t* __restrict__ out, int64_t ocs) {
  __shared__ Nn1Smem sm;
  nn1_body(pts + blockIdx.y * cs, stride, n_query, n_support, out + blockIdx.y * ocs, blockIdx.x, sm);
}

// Every level's points are a prefix of the input cloud (data_base.py:166-172), so the interpolation searches of ALL levels and the
// 16-NN searches of the levels without a grid read the input alone and are independent of one another - ONE launch for what were up
// to six dependent ones in a registration's chain (one pair in flight: 62 -> 29 us; eight: the 55 us lane-per-query search of level 2
// runs beside the others).  Job j of the table = (kind, level sizes, output, first workgroup); a workgroup finds its job by its
// index and runs the unchanged body: same bits.  K16: the table holds a lane-per-query search (its 72 KB of LDS).
template <bool K16>
struct SmallSmem { union { Nn1Smem nn1; Knn16Smem k16; }; };
template <>
struct SmallSmem<false> { Nn1Smem nn1; };
template <bool K16>
__global__ __launch_bounds__(256) void knn_small_levels_kernel(const float* __restrict__ pts, int64_t cs, int stride, const KnnSmallJobs J) {
  static_assert(QB * NW == 256, "the bodies run 256 threads");
  __shared__ SmallSmem<K16> sm;
  int j = 0;
#pragma unroll
  for (int k = 1; k < KnnSmallJobs::kMax; ++k) j += (k < J.njobs && (int)blockIdx.x >= J.job[k].b0) ? 1 : 0;
  const KnnSmallJobs::Job& jb = J.job[j];
  const int bx = (int)blockIdx.x - jb.b0;
  const float* P = pts + blockIdx.y * cs;
  if (jb.kind == 0) nn1_body(P, stride, jb.n, jb.n_support, jb.out + blockIdx.y * jb.ocs, bx, sm.nn1);
  else if (jb.kind == 1) knn16_wave_body(P, stride, jb.n, jb.out + blockIdx.y * jb.ocs, bx);
  else if constexpr (K16) knn16_body(P, stride, jb.n, jb.out + blockIdx.y * jb.ocs, bx, sm.k16);
}

}  // namespace

void launch_knn16(const float* pts, int64_t cs, int stride, int n, int clouds, int32_t* out, int64_t ocs, hipStream_t st) {
  // small level, few clouds: one wave per query (same bits); otherwise one lane per query
  if (knn16_takes_wave_kernel(n, clouds)) {
    hipLaunchKernelGGL(knn16_wave_kernel, dim3((n + 3) / 4, clouds), dim3(256), 0, st, pts, cs, stride, n, out, ocs);
    return;
  }
  dim3 grid((n + QB - 1) / QB, clouds);
  hipLaunchKernelGGL(knn16_kernel, grid, dim3(QB * NW), 0, st, pts, cs, stride, n, out, ocs);
}

bool knn16_takes_wave_kernel(int n, int clouds) { return n <= 64 * WQ_MAX && (int64_t)clouds * n <= 4096; }

void launch_knn_small_levels(const float* pts, int64_t cs, int stride, int clouds, KnnSmallJobs& J, hipStream_t st) {
  // the longest bodies first (the lane-per-query searches, then one wave per query, then the interpolation searches)
  KnnSmallJobs S{};
  for (int kind = 2; kind >= 0; --kind)
    for (int k = 0; k < J.njobs; ++k)
      if (J.job[k].kind == kind) S.job[S.njobs++] = J.job[k];
  int b = 0;
  bool k16 = false;
  for (int k = 0; k < S.njobs; ++k) {
    S.job[k].b0 = b;
    b += S.job[k].kind == 1 ? (S.job[k].n + 3) / 4 : (S.job[k].n + QB - 1) / QB;
    k16 = k16 || S.job[k].kind == 2;
  }
  J = S;
  if (b == 0 || clouds <= 0) return;
  if (k16) hipLaunchKernelGGL(knn_small_levels_kernel<true>, dim3(b, clouds), dim3(256), 0, st, pts, cs, stride, S);
  else hipLaunchKernelGGL(knn_small_levels_kernel<false>, dim3(b, clouds), dim3(256), 0, st, pts, cs, stride, S);
}

void launch_nn1(const float* pts, int64_t cs, int stride, int n_query, int n_support, int clouds, int32_t* out,
                int64_t ocs, hipStream_t st) {
  dim3 grid((n_query + QB - 1) / QB, clouds);
  hipLaunchKernelGGL(nn1_kernel, grid, dim3(QB * NW), 0, st, pts, cs, stride, n_query, n_support, out, ocs);
}

}  // namespace dsir
