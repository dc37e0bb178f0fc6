// Point-to-point ICP refinement of a registration result: the `use_icp` branch of the reference's
// pose_optimization (test.py:241-258), which hands the predicted pose to open3d's
//   registration_icp(src, tgt, max_correspondence_distance, T_init, TransformationEstimationPointToPoint())
// with the default convergence criteria (relative_fitness = relative_rmse = 1e-6, max_iteration = 30).  The branch is
// switched off in the reference (`use_icp = False`, test.py:216) and open3d is not installable here, so parity at this
// boundary is unpinned: the algorithm restated (and mirrored in oracle/icp.py) is open3d's RegistrationICP loop —
//   result = correspondences(T·src, tgt)                       nearest target within the radius, fitness, inlier RMSE
//   repeat: update = Kabsch(correspondences); T = update·T; src = update·src; result' = correspondences(...)
//           stop when |fitness' - fitness| < relative_fitness and |rmse' - rmse| < relative_rmse
// run for all pairs of a batch at once, entirely on device (no host round trip per iteration): a per-pair `done` flag
// turns the remaining iterations into no-ops.  Nearest neighbours: exact brute force in fp32 (squared distance
// (dx*dx + dy*dy) + dz*dz without FMA contraction, ties to the lower index), support staged through LDS.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int QB = 64;      // queries per block (one per lane)
constexpr int NW = 4;       // waves per block = support slices
constexpr int TILE = 256;   // support points staged per wave per step

// cur[pair][j] = T[pair] * src[pair][j]
__global__ void icp_apply_kernel(const float* __restrict__ src, int stride, int J, const float* __restrict__ T,
                                 float* __restrict__ cur) {
  const int pair = blockIdx.y;
  const float* t = T + (int64_t)pair * 12;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < J; j += gridDim.x * blockDim.x) {
    const float* p = src + ((int64_t)pair * J + j) * stride;
    const float x = p[0], y = p[1], z = p[2];
    float* o = cur + ((int64_t)pair * J + j) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      o[r] = __fadd_rn(fmaf(z, t[r * 4 + 2], fmaf(y, t[r * 4 + 1], __fmul_rn(x, t[r * 4 + 0]))), t[r * 4 + 3]);
  }
}

__global__ __launch_bounds__(QB * NW) void icp_nn_kernel(const float* __restrict__ cur, const float* __restrict__ ref,
                                                         int ref_stride, int J, int K, float r2,
                                                         int32_t* __restrict__ idx, float* __restrict__ d2,
                                                         const int32_t* __restrict__ skip) {
  // a converged pair keeps the correspondences of its last search: its points no longer move (block-uniform exit)
  if (skip && skip[blockIdx.y]) return;
  __shared__ float4 tile[NW][TILE];
  __shared__ float md[NW][QB];
  __shared__ int mi[NW][QB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int pair = blockIdx.y;
  const float* Q = cur + (int64_t)pair * J * 3;
  const float* S = ref + (int64_t)pair * K * ref_stride;
  const int q = blockIdx.x * QB + lane;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < J) { qx = Q[(int64_t)q * 3]; qy = Q[(int64_t)q * 3 + 1]; qz = Q[(int64_t)q * 3 + 2]; }
  const int slice = (K + NW - 1) / NW;
  const int s_begin = w * slice, s_end = min(K, s_begin + slice);
  float bd = INFINITY;
  int bi = -1;
  for (int t0 = 0; t0 < slice; t0 += TILE) {
#pragma unroll
    for (int r = 0; r < TILE / 64; ++r) {
      const int j = s_begin + t0 + r * 64 + lane;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < s_end) { v.x = S[(int64_t)j * ref_stride]; v.y = S[(int64_t)j * ref_stride + 1]; v.z = S[(int64_t)j * ref_stride + 2]; }
      tile[w][r * 64 + lane] = v;
    }
    __syncthreads();
    const int cnt = max(0, min(TILE, s_end - (s_begin + t0)));
    for (int j = 0; j < cnt; ++j) {
      const float4 s = tile[w][j];
      const float dx = __fsub_rn(s.x, qx), dy = __fsub_rn(s.y, qy), dz = __fsub_rn(s.z, qz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d < bd) { bd = d; bi = s_begin + t0 + j; }
    }
    __syncthreads();
  }
  md[w][lane] = bd; mi[w][lane] = bi;
  __syncthreads();
  if (w == 0 && q < J) {
#pragma unroll
    for (int s = 1; s < NW; ++s) {
      const float d = md[s][lane];
      if (d < bd) { bd = d; bi = mi[s][lane]; }   // slices are ascending in index: strict < keeps the lower index on a tie
    }
    const bool in = bi >= 0 && bd <= r2;
    idx[(int64_t)pair * J + q] = in ? bi : -1;
    d2[(int64_t)pair * J + q] = in ? bd : 0.f;
  }
}

// per pair: fitness = |corr| / J, inlier RMSE = sqrt(sum d2 / |corr|); convergence test against the previous values;
// correspondences turned into (clamped index, 0/1 weight) for the Kabsch kernel.  state = {fitness, rmse, done, iterations}
__global__ __launch_bounds__(256) void icp_stats_kernel(int32_t* __restrict__ idx, const float* __restrict__ d2,
                                                        float* __restrict__ w, int J, int check, float rel_fitness,
                                                        float rel_rmse, double* __restrict__ state) {
  __shared__ double s_cnt[4], s_sse[4];
  const int pair = blockIdx.x;
  double* st = state + (int64_t)pair * 4;
  if (st[2] != 0.0) return;   // converged earlier: frozen (block-uniform)
  double cnt = 0.0, sse = 0.0;
  for (int j = threadIdx.x; j < J; j += 256) {
    const int64_t o = (int64_t)pair * J + j;
    const int i = idx[o];
    const bool in = i >= 0;
    w[o] = in ? 1.f : 0.f;
    if (!in) idx[o] = 0;
    cnt += in ? 1.0 : 0.0;
    sse += in ? (double)d2[o] : 0.0;
  }
  cnt = wave_sum(cnt); sse = wave_sum(sse);
  if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = cnt; s_sse[threadIdx.x >> 6] = sse; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    sse = s_sse[0] + s_sse[1] + s_sse[2] + s_sse[3];
    const double fitness = cnt / (double)J;
    const double rmse = cnt > 0.0 ? sqrt(sse / cnt) : 0.0;
    if (check) {
      st[3] += 1.0;
      if (fabs(st[0] - fitness) < (double)rel_fitness && fabs(st[1] - rmse) < (double)rel_rmse) st[2] = 1.0;
    }
    st[0] = fitness; st[1] = rmse;
  }
}

__global__ void icp_done_flags_kernel(const double* __restrict__ state, int pairs, int32_t* __restrict__ skip) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < pairs) skip[p] = state[(int64_t)p * 4 + 2] != 0.0 ? 1 : 0;
}

}  // namespace

size_t icp_scratch_bytes(int pairs, int J) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return al((size_t)pairs * J * 12) + 3 * al((size_t)pairs * J * 4) + 2 * al((size_t)pairs * 48) + al((size_t)pairs * 32) +
         al((size_t)pairs * 4) + al((size_t)pairs * 48);
}

// T_init / T_out [pairs][3][4]; stats_out [pairs][4] doubles {fitness, inlier_rmse, converged, iterations} or nullptr
void launch_icp_refine(const float* src, const float* ref, int pairs, int J, int K, int stride, float max_corr_dist,
                       int max_iter, float rel_fitness, float rel_rmse, const float* T_init, float* T_out,
                       double* stats_out, void* scratch, hipStream_t st) {
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  float* cur = reinterpret_cast<float*>(take((size_t)pairs * J * 12));
  int32_t* idx = reinterpret_cast<int32_t*>(take((size_t)pairs * J * 4));
  float* d2 = reinterpret_cast<float*>(take((size_t)pairs * J * 4));
  float* w = reinterpret_cast<float*>(take((size_t)pairs * J * 4));
  float* Ta = reinterpret_cast<float*>(take((size_t)pairs * 48));
  float* Tb = reinterpret_cast<float*>(take((size_t)pairs * 48));
  double* state = reinterpret_cast<double*>(take((size_t)pairs * 32));
  int32_t* skip = reinterpret_cast<int32_t*>(take((size_t)pairs * 4));
  float* Tstep = reinterpret_cast<float*>(take((size_t)pairs * 48));
  const float r2 = max_corr_dist * max_corr_dist;
  hipMemsetAsync(state, 0, (size_t)pairs * 32, st);
  hipMemcpyAsync(Ta, T_init, (size_t)pairs * 48, hipMemcpyDeviceToDevice, st);
  const dim3 gj((J + 255) / 256, pairs), gq((J + QB - 1) / QB, pairs);
  hipLaunchKernelGGL(icp_apply_kernel, gj, dim3(256), 0, st, src, stride, J, Ta, cur);
  hipLaunchKernelGGL(icp_nn_kernel, gq, dim3(QB * NW), 0, st, cur, ref, stride, J, K, r2, idx, d2, (const int32_t*)nullptr);
  hipLaunchKernelGGL(icp_stats_kernel, dim3(pairs), dim3(256), 0, st, idx, d2, w, J, 0, rel_fitness, rel_rmse, state);
  float *Tp = Ta, *Tn = Tb;
  for (int it = 0; it < max_iter; ++it) {
    hipLaunchKernelGGL(icp_done_flags_kernel, dim3((pairs + 255) / 256), dim3(256), 0, st, state, pairs, skip);
    KabschArgs a{};
    a.src = cur; a.ref = ref; a.idx = idx; a.w = w; a.src_stride = (int64_t)J * 3; a.ref_stride = (int64_t)K * stride;
    a.ref_ld = stride; a.sigmoid = 0; a.pairs = pairs; a.m = J; a.T = Tstep; a.invalid = nullptr;
    a.src_out = cur; a.src_out_stride = (int64_t)J * 3; a.T_prev = Tp; a.T_cum = Tn; a.T_stride = 12; a.skip = skip;
    launch_kabsch(a, st);
    hipLaunchKernelGGL(icp_nn_kernel, gq, dim3(QB * NW), 0, st, cur, ref, stride, J, K, r2, idx, d2, (const int32_t*)skip);
    hipLaunchKernelGGL(icp_stats_kernel, dim3(pairs), dim3(256), 0, st, idx, d2, w, J, 1, rel_fitness, rel_rmse, state);
    float* t = Tp; Tp = Tn; Tn = t;
  }
  hipMemcpyAsync(T_out, Tp, (size_t)pairs * 48, hipMemcpyDeviceToDevice, st);
  if (stats_out) hipMemcpyAsync(stats_out, state, (size_t)pairs * 32, hipMemcpyDeviceToDevice, st);
}

}  // namespace dsir
