// Evaluation metrics of predicted poses on device (reference common/metrics_util.py:27-85
// compute_metrics, called per iteration by test.py:308-355 evaluate_align): DCP-style Euler / translation
// errors, isotropic residual rotation (deg) / translation, success flag and the modified Chamfer distance
// on the first M (<= 2048) points of each cloud.
//
// One 256-thread block per pair.  Thread 0 does the 3x4 pose algebra (fp32 where the reference uses
// torch fp32, fp64 for the Euler angles as scipy does); all threads then share the two nearest-neighbour
// sweeps of the Chamfer term: targets are staged through LDS tiles, each thread owns query points.
// Squared distances are fp32 (dx*dx + dy*dy) + dz*dz like torch.sum((a-b)**2, -1).
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int NT = 256;
constexpr int TILE = 512;

struct Pose { float R[3][3]; float t[3]; };

__device__ __forceinline__ void load_pose(const float* T, Pose& p) {
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) p.R[r][c] = T[r * 4 + c]; p.t[r] = T[r * 4 + 3]; }
}
// (R1 R2, R1 t2 + t1)   se3_torch.py:34-57
__device__ __forceinline__ Pose compose(const Pose& a, const Pose& b) {
  Pose o;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.R[r][c] = fmaf(a.R[r][2], b.R[2][c], fmaf(a.R[r][1], b.R[1][c], a.R[r][0] * b.R[0][c]));
    o.t[r] = fmaf(a.R[r][2], b.t[2], fmaf(a.R[r][1], b.t[1], a.R[r][0] * b.t[0])) + a.t[r];
  }
  return o;
}
// (R^T, -R^T t)   se3_torch.py:14-31
__device__ __forceinline__ Pose inverse(const Pose& a) {
  Pose o;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.R[r][c] = a.R[c][r];
    o.t[r] = fmaf(a.R[2][r], -a.t[2], fmaf(a.R[1][r], -a.t[1], a.R[0][r] * -a.t[0]));
  }
  return o;
}
__device__ __forceinline__ float3 apply(const Pose& p, float x, float y, float z) {   // p R^T + t
  float3 o;
  o.x = fmaf(z, p.R[0][2], fmaf(y, p.R[0][1], x * p.R[0][0])) + p.t[0];
  o.y = fmaf(z, p.R[1][2], fmaf(y, p.R[1][1], x * p.R[1][0])) + p.t[1];
  o.z = fmaf(z, p.R[2][2], fmaf(y, p.R[2][1], x * p.R[2][0])) + p.t[2];
  return o;
}
__device__ __forceinline__ void euler_xyz_deg(const Pose& p, double (&e)[3]) {   // extrinsic x-y-z, R = Rz Ry Rx
  const double k = 57.29577951308232;
  e[0] = atan2((double)p.R[2][1], (double)p.R[2][2]) * k;
  e[1] = atan2(-(double)p.R[2][0], sqrt((double)p.R[0][0] * p.R[0][0] + (double)p.R[1][0] * p.R[1][0])) * k;
  e[2] = atan2((double)p.R[1][0], (double)p.R[0][0]) * k;
}
__device__ __forceinline__ float sqd(const float3& a, const float3& b) {
  const float dx = __fsub_rn(a.x, b.x), dy = __fsub_rn(a.y, b.y), dz = __fsub_rn(a.z, b.z);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// out[pair][8] = r_mse, r_mae, t_mse, t_mae, err_r_deg, err_t, succ, chamfer
__global__ __launch_bounds__(NT) void eval_metrics_kernel(const float* __restrict__ pred, int64_t pred_stride,
                                                          const float* __restrict__ gt, const float* __restrict__ src,
                                                          const float* __restrict__ ref, int n, int stride, int m,
                                                          float rte_thresh, float rre_thresh, double* __restrict__ out) {
  __shared__ float3 tile[TILE];
  __shared__ double red[NT / 64];
  __shared__ Pose s_pred, s_gt, s_inter;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const float* S = src + (int64_t)pair * n * stride;
  const float* Rf = ref + (int64_t)pair * n * stride;
  double* o = out + (int64_t)pair * 8;
  if (tid == 0) {
    Pose pp, pg;
    load_pose(pred + pair * pred_stride, pp);
    load_pose(gt + (int64_t)pair * 12, pg);
    double eg[3], ep[3], rm = 0, ra = 0;
    euler_xyz_deg(pg, eg); euler_xyz_deg(pp, ep);
    for (int k = 0; k < 3; ++k) { const double d = eg[k] - ep[k]; rm += d * d; ra += fabs(d); }
    float tm = 0.f, ta = 0.f;
    for (int k = 0; k < 3; ++k) { const float d = pg.t[k] - pp.t[k]; tm += d * d; ta += fabsf(d); }
    const Pose cat = compose(inverse(pg), pp);
    const float tr = cat.R[0][0] + cat.R[1][1] + cat.R[2][2];
    float cs = 0.5f * (tr - 1.f);
    cs = fminf(fmaxf(cs, -1.f + 1e-16f), 1.f - 1e-16f);          // the reference's clamp (a no-op in fp32, kept for fidelity)
    const float rot = acosf(cs) * 180.0f / 3.14159265358979323846f;
    const float trn = sqrtf(cat.t[0] * cat.t[0] + cat.t[1] * cat.t[1] + cat.t[2] * cat.t[2]);
    o[0] = rm / 3.0; o[1] = ra / 3.0; o[2] = (double)(tm / 3.f); o[3] = (double)(ta / 3.f);
    o[4] = (double)rot; o[5] = (double)trn; o[6] = (trn < rte_thresh && rot < rre_thresh) ? 1.0 : 0.0;
    s_pred = pp; s_gt = pg; s_inter = compose(pp, inverse(pg));
  }
  __syncthreads();
  const Pose pp = s_pred, pg = s_gt, pi = s_inter;
  // points_raw = [gt(src) ; ref] (2m points).  Sweep 1: queries pred(src) against raw.
  // Sweep 2: queries ref against inter(raw).   Both sweeps share the tile loop; a thread owns queries tid, tid+NT, ...
  double sums[2] = {0.0, 0.0};
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int q0 = 0; q0 < m; q0 += NT) {
      const int q = q0 + tid;
      float3 qp = make_float3(0.f, 0.f, 0.f);
      if (q < m) {
        if (sweep == 0) qp = apply(pp, S[(int64_t)q * stride], S[(int64_t)q * stride + 1], S[(int64_t)q * stride + 2]);
        else qp = make_float3(Rf[(int64_t)q * stride], Rf[(int64_t)q * stride + 1], Rf[(int64_t)q * stride + 2]);
      }
      float best = INFINITY;
      for (int t0 = 0; t0 < 2 * m; t0 += TILE) {
        __syncthreads();
        for (int j = tid; j < TILE; j += NT) {
          const int k = t0 + j;
          float3 v = make_float3(0.f, 0.f, 0.f);
          if (k < 2 * m) {
            if (k < m) v = apply(pg, S[(int64_t)k * stride], S[(int64_t)k * stride + 1], S[(int64_t)k * stride + 2]);
            else v = make_float3(Rf[(int64_t)(k - m) * stride], Rf[(int64_t)(k - m) * stride + 1], Rf[(int64_t)(k - m) * stride + 2]);
            if (sweep == 1) v = apply(pi, v.x, v.y, v.z);
          }
          tile[j] = v;
        }
        __syncthreads();
        const int cnt = min(TILE, 2 * m - t0);
        for (int j = 0; j < cnt; ++j) best = fminf(best, sqd(qp, tile[j]));
      }
      if (q < m) sums[sweep] += (double)best;
    }
  }
  for (int sweep = 0; sweep < 2; ++sweep) {
    double v = wave_sum(sums[sweep]);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < NT / 64; ++w) t += red[w];
      sums[sweep] = t;
    }
  }
  if (tid == 0) o[7] = (double)((float)(sums[0] / m) + (float)(sums[1] / m));
}

}  // namespace

void launch_eval_metrics(const float* pred, int64_t pred_stride, const float* gt, const float* src, const float* ref,
                         int pairs, int n, int stride, float rte_thresh, float rre_thresh, double* out, hipStream_t st) {
  if (pairs <= 0) return;
  const int m = n < 2048 ? n : 2048;   // compute_metrics slices [:2048] (metrics_util.py:36-37)
  hipLaunchKernelGGL(eval_metrics_kernel, dim3(pairs), dim3(NT), 0, st, pred, pred_stride, gt, src, ref, n, stride, m,
                     rte_thresh, rre_thresh, out);
}

}  // namespace dsir
