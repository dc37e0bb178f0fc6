// The training loss of the `align` pipeline and its gradient down to the inlier logits (the
// backward half): ScanAlignmentLoss (reference network/loss.py:705-851, called at train.py:401) through
// se3_torch.concatenate (common/math/se3_torch.py:34-57, model.py:595) and compute_rigid_transform_2 (model.py:22-66).
//
// In forward_align_4 the matching runs under no_grad (model.py:556) and the src cloud is moved by R_t.detach()
// (model.py:590): the loss reaches the network's parameters only through the inlier logits x_ij,
//     w = sigmoid(x)  ->  wn = w / (sum |w| + 1e-16)  ->  c_s, c_t, H = sum wn (s - c_s)(t - c_t)^T  ->  H = U S V^T,
//     R = V diag(1, 1, d) U^T,  t = -R c_s + c_t  ->  Tc_i = T_i o Tc_{i-1}  ->  mean |Tc_i p - T_gt p|  (or squared),
// and through the BCE-with-logits term on the same logits.  d total / d x is what the inlier RandLA's backward consumes.
//
// One 1024-thread workgroup per pair replays the chain forward (as kabsch.hip: fp32 products, fp64 sums, fp64 3x3 SVD) and
// walks it back:
//   dL/dTc_i   = discount_i / (P J 3) sum_j sign(Tc_i p_j - g_j) [p_j; 1]^T        (mse: 2 (Tc_i p_j - g_j))
//   concat     dL/dR_i = G^R Rc_{i-1}^T + G^t tc_{i-1}^T,  dL/dt_i = G^t,  dL/dTc_{i-1} += R_i^T G
//   Kabsch     R H = V S~ V^T is symmetric (S~ = diag(s1, s2, d s3)), so dR = V W V^T R with W skew and
//              W_ab = -K_ab / (s~_a + s~_b),  K = V^T (R dH - dH^T R^T) V   =>   dL/dH = -2 (U D) Z V^T,
//              Z_ab = skew(V^T (dL/dR) R^T V)_ab / (s~_a + s~_b)
//   weights    dL/dwn_j = (s_j - c_s)^T (dL/dH) (t_j - c_t) + dL/dc_s . s_j + dL/dc_t . t_j,  then the normalisation and
//              sigmoid' = w (1 - w);  BCE: discount_i wt (w - y) / (P J)
// Pinned by tests/golden/align_loss_cases.npz (the imported reference's autograd, oracle/gen_golden_align_loss.py).
#include "kernels.h"
#include "device_utils.h"
#include "svd3.h"

namespace dsir {

namespace {

constexpr int AL_THREADS = 1024;
constexpr int AL_WAVES = AL_THREADS / 64;
constexpr int AL_MAX_ITER = 8;

struct AlignLossArgs {
  const float* src; const float* ref;        // [P][J][3], [P][K][3]
  const int32_t* idx;                         // [n_iter][P][J]
  const float* logits; const float* labels;   // [n_iter][P][J]; labels may be nullptr (no confidence term)
  const float* T_gt;                          // [P][3][4]
  int P, J, K, n_iter, mse;
  float wt_pt, wt_in, discount;
  float* T_out;                               // [P][n_iter][3][4] or nullptr
  double* losses;                             // [n_iter][2] (point-distance term, confidence term), summed over the pairs by align_loss_reduce_kernel
  double* loss_part;                          // [P][n_iter][2] every pair's terms (nullptr: no losses wanted)
  float* grad;                                // [n_iter][P][J]
};

template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* sh /* [AL_WAVES][NV] + [NV] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[w * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int ww = 0; ww < AL_WAVES; ++ww) s += sh[ww * NV + threadIdx.x];
    sh[AL_WAVES * NV + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = sh[AL_WAVES * NV + i];
}

struct IterState {
  double R[3][3], t[3];        // this iteration's transform
  double Rc[3][3], tc[3];      // cumulative transform after it
  double cs[3], ct[3];         // weighted centroids
  double U[3][3], V[3][3], St[3];   // H = U S V^T, St = (s1, s2, d s3); U already multiplied by D
  double den, sw;              // sum |w| + 1e-16, sum wn
  double G[12];                // sum_j dl/dpred_j [p_j; 1]^T (row major 3x4), unscaled
};

__global__ __launch_bounds__(AL_THREADS) void align_loss_kernel(AlignLossArgs a) {
  __shared__ double sh[AL_WAVES * 14 + 14];
  __shared__ IterState st[AL_MAX_ITER];
  __shared__ double s_GH[9], s_gcs[3], s_gct[3];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int J = a.J;
  const float* S = a.src + (int64_t)pair * J * 3;
  const float* Rf = a.ref + (int64_t)pair * a.K * 3;
  const float* Tg = a.T_gt + (int64_t)pair * 12;
  auto src_at = [&](int it, int j, double (&s)[3]) {      // current src point of iteration it = Tc_{it-1} p_j
    const double x = S[3 * j], y = S[3 * j + 1], z = S[3 * j + 2];
    if (it == 0) { s[0] = x; s[1] = y; s[2] = z; return; }
    const IterState& q = st[it - 1];
#pragma unroll
    for (int r = 0; r < 3; ++r) s[r] = (double)(float)(q.Rc[r][0] * x + q.Rc[r][1] * y + q.Rc[r][2] * z + q.tc[r]);
  };
  auto tgt_at = [&](int it, int j, double (&t)[3]) {
    const int k = a.idx[((int64_t)it * a.P + pair) * J + j];
    t[0] = Rf[3 * k]; t[1] = Rf[3 * k + 1]; t[2] = Rf[3 * k + 2];
  };
  auto weight = [&](int it, int j) -> double {
    const float x = a.logits[((int64_t)it * a.P + pair) * J + j];
    return (double)(1.f / (1.f + expf(-x)));
  };
  const double inv_pts = 1.0 / ((double)a.P * J * 3.0), inv_rows = 1.0 / ((double)a.P * J);

  // ------------------------------------------------------------------ forward replay
  for (int it = 0; it < a.n_iter; ++it) {
    double v1[1] = {0.0};
    for (int j = tid; j < J; j += AL_THREADS) v1[0] += fabs(weight(it, j));
    block_sum_d<1>(v1, sh);
    const double den = (double)((float)v1[0] + 1e-16f);
    double v7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int j = tid; j < J; j += AL_THREADS) {
      const double wn = (double)((float)weight(it, j) / (float)den);
      double s[3], t[3];
      src_at(it, j, s); tgt_at(it, j, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) { v7[k] += (double)((float)s[k] * (float)wn); v7[3 + k] += (double)((float)t[k] * (float)wn); }
      v7[6] += wn;
    }
    block_sum_d<7>(v7, sh);
    const double cs[3] = {(double)(float)v7[0], (double)(float)v7[1], (double)(float)v7[2]};
    const double ct[3] = {(double)(float)v7[3], (double)(float)v7[4], (double)(float)v7[5]};
    double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = tid; j < J; j += AL_THREADS) {
      const float wn = (float)weight(it, j) / (float)den;
      double s[3], t[3];
      src_at(it, j, s); tgt_at(it, j, t);
      float sc[3], tw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { sc[k] = (float)s[k] - (float)cs[k]; tw[k] = ((float)t[k] - (float)ct[k]) * wn; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v9[r * 3 + c] += (double)(sc[r] * tw[c]);
    }
    block_sum_d<9>(v9, sh);
    if (tid == 0) {
      IterState& q = st[it];
      double H[3][3], U[3][3], Sv[3], V[3][3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] = (double)(float)v9[r * 3 + c];
      svd3(H, U, Sv, V);
      double Rp[3][3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rp[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
      const double d = det3(Rp) > 0 ? 1.0 : -1.0;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          q.R[r][c] = (double)(float)(V[r][0] * U[c][0] + V[r][1] * U[c][1] + d * V[r][2] * U[c][2]);
          q.V[r][c] = V[r][c];
          q.U[r][c] = c == 2 ? d * U[r][c] : U[r][c];      // U D
        }
      }
      q.St[0] = Sv[0]; q.St[1] = Sv[1]; q.St[2] = d * Sv[2];
      for (int r = 0; r < 3; ++r) {
        q.t[r] = (double)((float)(-(q.R[r][0] * cs[0] + q.R[r][1] * cs[1] + q.R[r][2] * cs[2])) + (float)ct[r]);
        q.cs[r] = cs[r]; q.ct[r] = ct[r];
      }
      q.den = den; q.sw = v7[6];
      if (it == 0) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) q.Rc[r][c] = q.R[r][c]; q.tc[r] = q.t[r]; }
      } else {
        const IterState& p = st[it - 1];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) q.Rc[r][c] = (double)(float)(q.R[r][0] * p.Rc[0][c] + q.R[r][1] * p.Rc[1][c] + q.R[r][2] * p.Rc[2][c]);
          q.tc[r] = (double)(float)(q.R[r][0] * p.tc[0] + q.R[r][1] * p.tc[1] + q.R[r][2] * p.tc[2] + q.t[r]);
        }
      }
      if (a.T_out) {
        float* o = a.T_out + ((int64_t)pair * a.n_iter + it) * 12;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) o[r * 4 + c] = (float)q.Rc[r][c]; o[r * 4 + 3] = (float)q.tc[r]; }
      }
    }
    __syncthreads();
    // loss terms of this iteration and dL/dTc (unscaled sums)
    double v14[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) v14[k] = 0.0;
    {
      const IterState& q = st[it];
      for (int j = tid; j < J; j += AL_THREADS) {
        const double x = S[3 * j], y = S[3 * j + 1], z = S[3 * j + 2];
        double g[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float pred = (float)(q.Rc[r][0] * x + q.Rc[r][1] * y + q.Rc[r][2] * z + q.tc[r]);
          const float gt = (float)((double)Tg[r * 4] * x + (double)Tg[r * 4 + 1] * y + (double)Tg[r * 4 + 2] * z + (double)Tg[r * 4 + 3]);
          const double df = (double)(pred - gt);
          if (a.mse) { v14[12] += df * df; g[r] = 2.0 * df; }
          else { v14[12] += fabs(df); g[r] = df > 0.0 ? 1.0 : (df < 0.0 ? -1.0 : 0.0); }
          v14[r * 4 + 0] += g[r] * x; v14[r * 4 + 1] += g[r] * y; v14[r * 4 + 2] += g[r] * z; v14[r * 4 + 3] += g[r];
        }
        if (a.labels) {
          const int64_t o = ((int64_t)it * a.P + pair) * J + j;
          const double xl = a.logits[o], yl = a.labels[o];
          v14[13] += fmax(xl, 0.0) - xl * yl + log1p(exp(-fabs(xl)));       // BCEWithLogits
        }
      }
    }
    block_sum_d<14>(v14, sh);
    if (tid == 0) {
      for (int k = 0; k < 12; ++k) st[it].G[k] = v14[k];
      if (a.loss_part) {      // this pair's terms; align_loss_reduce_kernel adds the pairs in pair order (no atomics: same bits every run)
        a.loss_part[((int64_t)pair * a.n_iter + it) * 2] = a.wt_pt > 0.f ? v14[12] * inv_pts : 0.0;
        a.loss_part[((int64_t)pair * a.n_iter + it) * 2 + 1] = (a.labels && a.wt_in > 0.f) ? v14[13] * inv_rows * (double)a.wt_in : 0.0;
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ backward
  double carry[12];      // dL/dTc_it arriving from the later iterations (thread 0)
  for (int k = 0; k < 12; ++k) carry[k] = 0.0;
  for (int it = a.n_iter - 1; it >= 0; --it) {
    const double disc = pow((double)a.discount, (double)(a.n_iter - it - 1));
    if (tid == 0) {
      const IterState& q = st[it];
      double G[3][4];
      const double sc = a.wt_pt > 0.f ? disc * inv_pts : 0.0;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) G[r][c] = sc * q.G[r * 4 + c] + carry[r * 4 + c];
      double gR[3][3], gt[3];
      if (it == 0) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) gR[r][c] = G[r][c]; gt[r] = G[r][3]; }
      } else {
        const IterState& p = st[it - 1];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            gR[r][c] = G[r][0] * p.Rc[c][0] + G[r][1] * p.Rc[c][1] + G[r][2] * p.Rc[c][2] + G[r][3] * p.tc[c];
          gt[r] = G[r][3];
        }
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 4; ++c) carry[r * 4 + c] = q.R[0][r] * G[0][c] + q.R[1][r] * G[1][c] + q.R[2][r] * G[2][c];
      }
      // t = -R c_s + c_t
      double gcs[3], gct[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) gR[r][c] -= gt[r] * q.cs[c];
        gct[r] = gt[r];
      }
      for (int c = 0; c < 3; ++c) gcs[c] = -(q.R[0][c] * gt[0] + q.R[1][c] * gt[1] + q.R[2][c] * gt[2]);
      // R = V D U^T  <-  H = U S V^T
      double A1[3][3], Q[3][3];   // A1 = gR R^T, Q = V^T A1 V
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A1[r][c] = gR[r][0] * q.R[c][0] + gR[r][1] * q.R[c][1] + gR[r][2] * q.R[c][2];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double acc = 0.0;
          for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) acc += q.V[i][r] * A1[i][k] * q.V[k][c];
          Q[r][c] = acc;
        }
      double Z[3][3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          const double dn = q.St[r] + q.St[c];
          Z[r][c] = (r == c || fabs(dn) < 1e-300) ? 0.0 : 0.5 * (Q[r][c] - Q[c][r]) / dn;
        }
      double GH[3][3];            // -2 (U D) Z V^T
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double acc = 0.0;
          for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) acc += q.U[r][i] * Z[i][k] * q.V[c][k];
          GH[r][c] = -2.0 * acc;
        }
      // H also depends on the centroids: sum wn (t - c_t) = c_t (1 - sw), sum wn (s - c_s) = c_s (1 - sw)
      for (int r = 0; r < 3; ++r) {
        double a1 = 0.0, a2 = 0.0;
        for (int c = 0; c < 3; ++c) { a1 += GH[r][c] * q.ct[c]; a2 += GH[c][r] * q.cs[c]; }
        gcs[r] -= a1 * (1.0 - q.sw);
        gct[r] -= a2 * (1.0 - q.sw);
      }
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) s_GH[r * 3 + c] = GH[r][c]; s_gcs[r] = gcs[r]; s_gct[r] = gct[r]; }
    }
    __syncthreads();
    const IterState& q = st[it];
    auto g_wn = [&](int j) -> double {
      double s[3], t[3];
      src_at(it, j, s); tgt_at(it, j, t);
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) row += s_GH[r * 3 + c] * (t[c] - q.ct[c]);
        acc += (s[r] - q.cs[r]) * row + s_gcs[r] * s[r] + s_gct[r] * t[r];
      }
      return acc;
    };
    double v1[1] = {0.0};
    for (int j = tid; j < J; j += AL_THREADS) v1[0] += g_wn(j) * weight(it, j);
    block_sum_d<1>(v1, sh);
    const double dotw = v1[0], bce = (a.labels && a.wt_in > 0.f) ? disc * (double)a.wt_in * inv_rows : 0.0;
    for (int j = tid; j < J; j += AL_THREADS) {
      const double w = weight(it, j);
      const double gw = g_wn(j) / q.den - dotw / (q.den * q.den);      // w > 0: d|w|/dw = 1
      const int64_t o = ((int64_t)it * a.P + pair) * J + j;
      double g = gw * w * (1.0 - w);
      if (bce != 0.0) g += bce * (w - (double)a.labels[o]);
      a.grad[o] = (float)g;
    }
    __syncthreads();
  }
}

// losses[it][term] = sum over the pairs, in pair order
__global__ void align_loss_reduce_kernel(const double* __restrict__ part, int P, int n2, double* __restrict__ losses) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n2) return;
  double s = 0.0;
  for (int p = 0; p < P; ++p) s += part[(int64_t)p * n2 + k];
  losses[k] = s;
}

}  // namespace

int launch_align_loss(const float* src, const float* ref, const int32_t* idx, const float* logits, const float* labels,
                      const float* T_gt, int P, int J, int K, int n_iter, int mse, float wt_pt, float wt_in, float discount,
                      float* T_out, double* losses, float* grad, hipStream_t st, double* loss_part) {
  if (n_iter < 1 || n_iter > AL_MAX_ITER) return 1;
  AlignLossArgs a;
  a.src = src; a.ref = ref; a.idx = idx; a.logits = logits; a.labels = labels; a.T_gt = T_gt; a.P = P; a.J = J; a.K = K;
  a.n_iter = n_iter; a.mse = mse; a.wt_pt = wt_pt; a.wt_in = wt_in; a.discount = discount; a.T_out = T_out; a.losses = losses;
  a.grad = grad;
  if (losses && !loss_part) return 2;
  a.loss_part = losses ? loss_part : nullptr;
  hipLaunchKernelGGL(align_loss_kernel, dim3(P), dim3(AL_THREADS), 0, st, a);
  if (losses) hipLaunchKernelGGL(align_loss_reduce_kernel, dim3(1), dim3(64), 0, st, loss_part, P, 2 * n_iter, losses);
  return 0;
}

}  // namespace dsir
