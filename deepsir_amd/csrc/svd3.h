// 3x3 float64 linear algebra shared by the pose kernels (kabsch.hip, align_loss.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dsir {

// One-sided (Hestenes) Jacobi SVD of a 3x3 matrix in fp64: A = U diag(s) V^T, s sorted descending.
__device__ inline void svd3(const double A[3][3], double U[3][3], double S[3], double V[3][3]) {
  double G[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { G[i][j] = A[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += G[i][p] * G[i][p]; be += G[i][q] * G[i][q]; ga += G[i][p] * G[i][q]; }
        const double lim = 1e-300 + 1e-32 * al * be;
        if (ga * ga <= lim) continue;
        off = fmax(off, ga * ga / (al * be));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double gp = G[i][p], gq = G[i][q];
          G[i][p] = c * gp - s * gq; G[i][q] = s * gp + c * gq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-30) break;
  }
  double nrm[3];
  for (int j = 0; j < 3; ++j) nrm[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
  int o[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2 - a; ++b)
      if (nrm[o[b]] < nrm[o[b + 1]]) { int t = o[b]; o[b] = o[b + 1]; o[b + 1] = t; }
  double Vs[3][3];
  const double tiny = 1e-280;
  for (int j = 0; j < 3; ++j) {
    S[j] = nrm[o[j]];
    for (int i = 0; i < 3; ++i) { Vs[i][j] = V[i][o[j]]; U[i][j] = S[j] > tiny ? G[i][o[j]] / S[j] : 0.0; }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
  // complete U for (numerically) rank-deficient input; R is unique iff rank >= 2
  const double thr = S[0] * 1e-14;
  if (!(S[0] > tiny)) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
  } else {
    if (!(S[1] > thr)) {  // pick any unit vector orthogonal to u0
      int m = 0;
      if (fabs(U[1][0]) < fabs(U[m][0])) m = 1;
      if (fabs(U[2][0]) < fabs(U[m][0])) m = 2;
      double e[3] = {0, 0, 0};
      e[m] = 1.0;
      const double dot = U[m][0];
      double n2 = 0;
      for (int i = 0; i < 3; ++i) { U[i][1] = e[i] - dot * U[i][0]; n2 += U[i][1] * U[i][1]; }
      n2 = sqrt(n2);
      for (int i = 0; i < 3; ++i) U[i][1] /= n2;
    }
    if (!(S[2] > thr)) {
      U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
      U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
      U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
  }
}

__device__ __forceinline__ double det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}


}  // namespace dsir
