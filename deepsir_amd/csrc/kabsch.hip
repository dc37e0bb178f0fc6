// Weighted Kabsch pose solve, entirely on device (reference network/model.py:22-66
// compute_rigid_transform_2 — which round-trips to the CPU for a float64 LAPACK
// SVD every iteration — plus the SE(3) bookkeeping of forward_align_4,
// model.py:586-595, common/math/se3_torch.py:28-77).
//
// One 1024-thread block per pair.  Three passes over the (L2-resident) points:
//   S = sum |w|;  c_s = sum s*wn, c_t = sum t*wn (wn = w/(S+1e-16));
//   H = sum (s-c_s) ((t-c_t)*wn)^T.
// Every product is rounded to fp32 exactly as the reference's element-wise ops
// produce it; the sums are accumulated in fp64 (wave shuffles + LDS), i.e. the
// exact value the reference's fp32 reductions approximate in some order.
// Thread 0 then runs a one-sided Jacobi SVD of H in fp64, R = V diag(1,1,d) U^T
// with d = sign(det(V U^T)), casts R to fp32 and forms t = -R c_s + c_t in fp32
// (model.py:53,57).  Non-finite H => identity + invalid flag (model.py:61-64).
// The same launch applies the transform to the src points, gathers the matched
// ref points and composes the cumulative transform.
#include <cstdlib>

#include "kernels.h"
#include "device_utils.h"
#include "svd3.h"

namespace dsir {

namespace {

// the passes' loops are unrolled x4: a thread's iterations are independent up to the fp64 adds (kept in order: same bits), and
// on large clouds - 64 points per thread and pass at 65536 - the index -> ref gather chains of consecutive iterations overlap
#define DSIR_KABSCH_UNROLL _Pragma("unroll 4")
constexpr int NTHR = 1024;
constexpr int NWAVE = NTHR / 64;

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh /* [NWAVE][NV] + [NV] */) {
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
    for (int ww = 0; ww < NWAVE; ++ww) s += sh[ww * NV + threadIdx.x];
    sh[NWAVE * NV + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = sh[NWAVE * NV + i];
}

// thread 0 of a pair: H (fp32, as the reference forms it) -> SVD in fp64 -> R, t; the pair's transform, flag and cumulative
// transform to global memory, the transform to sT (12 floats) for the apply step
__device__ void kabsch_solve(const KabschArgs& a, int pair, const double (&v9)[9], const float (&cs)[3], const float (&ct)[3], float* sT) {
  bool finite = true;
  double H[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const float h = (float)v9[r * 3 + c];   // the reference's H is fp32, then .double()
      H[r][c] = (double)h;
      finite = finite && isfinite(h);
    }
  float T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  int bad = 1;
  if (finite) {
    double U[3][3], S[3], V[3][3];
    svd3(H, U, S, V);
    double Rp[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rp[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
    const double d = det3(Rp) > 0 ? 1.0 : -1.0;
    float R[3][3];
    bool ok = true;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        R[r][c] = (float)(V[r][0] * U[c][0] + V[r][1] * U[c][1] + d * V[r][2] * U[c][2]);
        ok = ok && isfinite(R[r][c]);
      }
    if (ok) {
      bad = 0;
      for (int r = 0; r < 3; ++r) {
        T[r * 4 + 0] = R[r][0]; T[r * 4 + 1] = R[r][1]; T[r * 4 + 2] = R[r][2];
        float acc = __fmul_rn(-R[r][0], cs[0]);
        acc = fmaf(-R[r][1], cs[1], acc);
        acc = fmaf(-R[r][2], cs[2], acc);
        T[r * 4 + 3] = __fadd_rn(acc, ct[r]);
      }
    }
  }
  for (int k = 0; k < 12; ++k) { sT[k] = T[k]; a.T[(int64_t)pair * 12 + k] = T[k]; }
  if (a.invalid && bad) a.invalid[pair] |= 1;   // one thread per pair; bit 1 (clamped caller index, misc.hip) stays
  if (a.T_cum) {   // concatenate(R_t, T_prev): (R1 R2, R1 t2 + t1)   se3_torch.py:34-57
    float* out = a.T_cum + pair * a.T_stride;
    if (!a.T_prev) {
      for (int k = 0; k < 12; ++k) out[k] = T[k];
    } else {
      const float* P = a.T_prev + pair * a.T_stride;
      float C[12];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          C[r * 4 + c] = fmaf(T[r * 4 + 2], P[2 * 4 + c], fmaf(T[r * 4 + 1], P[1 * 4 + c], __fmul_rn(T[r * 4 + 0], P[c])));
        const float rt = fmaf(T[r * 4 + 2], P[2 * 4 + 3], fmaf(T[r * 4 + 1], P[1 * 4 + 3], __fmul_rn(T[r * 4 + 0], P[3])));
        C[r * 4 + 3] = __fadd_rn(rt, T[r * 4 + 3]);
      }
      for (int k = 0; k < 12; ++k) out[k] = C[k];
    }
  }
}

__global__ __launch_bounds__(NTHR) void kabsch_kernel(const KabschArgs a) {
  __shared__ double sh[NWAVE * 9 + 9];
  __shared__ float sT[12];
  const int pair = blockIdx.x;
  const int m = a.m;
  if (a.skip && a.skip[pair]) {   // block-uniform: frozen pair (ICP converged): identity step, cumulative transform carried over
    if (threadIdx.x == 0) {
      const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      for (int k = 0; k < 12; ++k) a.T[(int64_t)pair * 12 + k] = I[k];
      if (a.T_cum) {
        float* out = a.T_cum + pair * a.T_stride;
        const float* P = a.T_prev ? a.T_prev + pair * a.T_stride : I;
        for (int k = 0; k < 12; ++k) out[k] = P[k];
      }
    }
    return;
  }
  const int ld = a.ref_ld ? a.ref_ld : 3;
  const float* src = a.src + pair * a.src_stride;
  const float* ref = a.ref + pair * a.ref_stride;
  const int32_t* idx = a.idx ? a.idx + (int64_t)pair * m : nullptr;
  const float* wl = a.w + (int64_t)pair * m;
  auto weight = [&](int i) -> float {
    const float x = wl[i];
    return a.sigmoid ? 1.f / (1.f + expf(-x)) : x;
  };
  auto target = [&](int i, float& x, float& y, float& z) {
    const int64_t j = idx ? idx[i] : i;
    x = ref[j * ld]; y = ref[j * ld + 1]; z = ref[j * ld + 2];
  };

  // pass 1: S = sum |w|
  double v1[1] = {0.0};
  DSIR_KABSCH_UNROLL
  for (int i = threadIdx.x; i < m; i += NTHR) v1[0] += (double)fabsf(weight(i));
  block_sum<1>(v1, sh);
  const float den = (float)v1[0] + 1e-16f;   // model.py:35 (fp32 sum + _EPS)

  // pass 2: weighted centroids
  double v6[6] = {0, 0, 0, 0, 0, 0};
  DSIR_KABSCH_UNROLL
  for (int i = threadIdx.x; i < m; i += NTHR) {
    const float wn = weight(i) / den;
    float tx, ty, tz;
    target(i, tx, ty, tz);
    v6[0] += (double)__fmul_rn(src[(int64_t)i * 3], wn);
    v6[1] += (double)__fmul_rn(src[(int64_t)i * 3 + 1], wn);
    v6[2] += (double)__fmul_rn(src[(int64_t)i * 3 + 2], wn);
    v6[3] += (double)__fmul_rn(tx, wn);
    v6[4] += (double)__fmul_rn(ty, wn);
    v6[5] += (double)__fmul_rn(tz, wn);
  }
  block_sum<6>(v6, sh);
  const float cs[3] = {(float)v6[0], (float)v6[1], (float)v6[2]};
  const float ct[3] = {(float)v6[3], (float)v6[4], (float)v6[5]};

  // pass 3: covariance H[a][b] = sum (s_a - cs_a) * ((t_b - ct_b) * wn)
  double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  DSIR_KABSCH_UNROLL
  for (int i = threadIdx.x; i < m; i += NTHR) {
    const float wn = weight(i) / den;
    float t[3];
    target(i, t[0], t[1], t[2]);
    float sc[3], tw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sc[k] = __fsub_rn(src[(int64_t)i * 3 + k], cs[k]);
      tw[k] = __fmul_rn(__fsub_rn(t[k], ct[k]), wn);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) v9[r * 3 + c] += (double)__fmul_rn(sc[r], tw[c]);
  }
  block_sum<9>(v9, sh);

  if (threadIdx.x == 0) kabsch_solve(a, pair, v9, cs, ct, sT);
  __syncthreads();
  // apply: p' = p R^T + t (se3_torch.py:60-77); gather the matched ref points
  if (a.src_out || a.matched_out) {
    float* so = a.src_out ? a.src_out + pair * a.src_out_stride : nullptr;
    float* mo = a.matched_out ? a.matched_out + (int64_t)pair * m * 3 : nullptr;
    for (int i = threadIdx.x; i < m; i += NTHR) {
      if (mo) {
        float tx, ty, tz;
        target(i, tx, ty, tz);
        mo[(int64_t)i * 3] = tx; mo[(int64_t)i * 3 + 1] = ty; mo[(int64_t)i * 3 + 2] = tz;
      }
      if (so) {
        const float x = src[(int64_t)i * 3], y = src[(int64_t)i * 3 + 1], z = src[(int64_t)i * 3 + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float d = fmaf(z, sT[r * 4 + 2], fmaf(y, sT[r * 4 + 1], __fmul_rn(x, sT[r * 4 + 0])));
          so[(int64_t)i * 3 + r] = __fadd_rn(d, sT[r * 4 + 3]);
        }
      }
    }
  }
}

// ---- clouds of up to TR x 1024 points: the same kernel with the points HELD IN REGISTERS between the passes.  With one pair in
// flight (the reference's evaluation mode) the solve sits in the dependent chain of every iteration, and each pass of kabsch_kernel
// opens with its own weight -> index -> ref-point load chain (three round trips to L2 plus the apply step's fourth); here the chain
// is paid once.  A thread owns the same points i = tid, tid + 1024, .. and adds them in the same order: same bits as kabsch_kernel.
template <int TR>
__global__ __launch_bounds__(NTHR) void kabsch_reg_kernel(const KabschArgs a) {
  __shared__ double sh[NWAVE * 9 + 9];
  __shared__ float sT[12];
  const int pair = blockIdx.x;
  const int m = a.m;
  if (a.skip && a.skip[pair]) {   // block-uniform: frozen pair (ICP converged): identity step, cumulative transform carried over
    if (threadIdx.x == 0) {
      const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      for (int k = 0; k < 12; ++k) a.T[(int64_t)pair * 12 + k] = I[k];
      if (a.T_cum) {
        float* out = a.T_cum + pair * a.T_stride;
        const float* P = a.T_prev ? a.T_prev + pair * a.T_stride : I;
        for (int k = 0; k < 12; ++k) out[k] = P[k];
      }
    }
    return;
  }
  const int ld = a.ref_ld ? a.ref_ld : 3;
  const float* src = a.src + pair * a.src_stride;
  const float* ref = a.ref + pair * a.ref_stride;
  const int32_t* idx = a.idx ? a.idx + (int64_t)pair * m : nullptr;
  const float* wl = a.w + (int64_t)pair * m;

  float w[TR], s[TR][3], t[TR][3];
#pragma unroll
  for (int k = 0; k < TR; ++k) {
    const int i = threadIdx.x + k * NTHR;
    w[k] = 0.f;
    s[k][0] = s[k][1] = s[k][2] = t[k][0] = t[k][1] = t[k][2] = 0.f;
    if (i < m) {
      const int64_t j = idx ? idx[i] : i;
      w[k] = wl[i];
      t[k][0] = ref[j * ld]; t[k][1] = ref[j * ld + 1]; t[k][2] = ref[j * ld + 2];
      s[k][0] = src[(int64_t)i * 3]; s[k][1] = src[(int64_t)i * 3 + 1]; s[k][2] = src[(int64_t)i * 3 + 2];
    }
  }
  if (a.sigmoid) {
#pragma unroll
    for (int k = 0; k < TR; ++k) w[k] = 1.f / (1.f + expf(-w[k]));
  }

  // pass 1: S = sum |w|
  double v1[1] = {0.0};
#pragma unroll
  for (int k = 0; k < TR; ++k)
    if (threadIdx.x + k * NTHR < m) v1[0] += (double)fabsf(w[k]);
  block_sum<1>(v1, sh);
  const float den = (float)v1[0] + 1e-16f;   // model.py:35 (fp32 sum + _EPS)

  // pass 2: weighted centroids
  double v6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < TR; ++k)
    if (threadIdx.x + k * NTHR < m) {
      w[k] = w[k] / den;          // wn: the same quotient both passes of kabsch_kernel form
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v6[c] += (double)__fmul_rn(s[k][c], w[k]);
        v6[3 + c] += (double)__fmul_rn(t[k][c], w[k]);
      }
    }
  block_sum<6>(v6, sh);
  const float cs[3] = {(float)v6[0], (float)v6[1], (float)v6[2]};
  const float ct[3] = {(float)v6[3], (float)v6[4], (float)v6[5]};

  // pass 3: covariance H[a][b] = sum (s_a - cs_a) * ((t_b - ct_b) * wn)
  double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < TR; ++k)
    if (threadIdx.x + k * NTHR < m) {
      float sc[3], tw[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sc[c] = __fsub_rn(s[k][c], cs[c]);
        tw[c] = __fmul_rn(__fsub_rn(t[k][c], ct[c]), w[k]);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v9[r * 3 + c] += (double)__fmul_rn(sc[r], tw[c]);
    }
  block_sum<9>(v9, sh);

  if (threadIdx.x == 0) kabsch_solve(a, pair, v9, cs, ct, sT);
  __syncthreads();
  // apply: p' = p R^T + t (se3_torch.py:60-77); the matched ref points
  if (a.src_out || a.matched_out) {
    float* so = a.src_out ? a.src_out + pair * a.src_out_stride : nullptr;
    float* mo = a.matched_out ? a.matched_out + (int64_t)pair * m * 3 : nullptr;
#pragma unroll
    for (int k = 0; k < TR; ++k) {
      const int i = threadIdx.x + k * NTHR;
      if (i >= m) continue;
      if (mo) { mo[(int64_t)i * 3] = t[k][0]; mo[(int64_t)i * 3 + 1] = t[k][1]; mo[(int64_t)i * 3 + 2] = t[k][2]; }
      if (so) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float d = fmaf(s[k][2], sT[r * 4 + 2], fmaf(s[k][1], sT[r * 4 + 1], __fmul_rn(s[k][0], sT[r * 4 + 0])));
          so[(int64_t)i * 3 + r] = __fadd_rn(d, sT[r * 4 + 3]);
        }
      }
    }
  }
}

// ---- large clouds: the same three passes over CHUNKS of 4096 points, one workgroup per (chunk, pair) and pass, partial sums
// in fp64 in a.part [pairs][chunks][16] = {S, c_s, c_t, H}; every pass adds the chunks' partials in chunk order (every
// workgroup for itself: a few dozen doubles), so the result does not depend on the grid.  One workgroup per pair walks
// 65536 points in 64 dependent trips per pass (218 us per solve with 2 pairs in flight); in chunks: five short launches.
constexpr int CHUNK = 4096;

template <int PHASE>
__global__ __launch_bounds__(NTHR) void kabsch_part_kernel(const KabschArgs a, int nch) {
  __shared__ double sh[NWAVE * 9 + 9];
  const int pair = blockIdx.y, ch = blockIdx.x;
  if (a.skip && a.skip[pair]) return;                 // block-uniform (frozen pair): kabsch_final_kernel writes the identity
  const int m = a.m;
  double* part = a.part + (int64_t)pair * nch * 16;
  const int ld = a.ref_ld ? a.ref_ld : 3;
  const float* src = a.src + pair * a.src_stride;
  const float* ref = a.ref + pair * a.ref_stride;
  const int32_t* idx = a.idx ? a.idx + (int64_t)pair * m : nullptr;
  const float* wl = a.w + (int64_t)pair * m;
  auto weight = [&](int i) -> float {
    const float x = wl[i];
    return a.sigmoid ? 1.f / (1.f + expf(-x)) : x;
  };
  auto target = [&](int i, float& x, float& y, float& z) {
    const int64_t j = idx ? idx[i] : i;
    x = ref[j * ld]; y = ref[j * ld + 1]; z = ref[j * ld + 2];
  };
  const int i1 = min(m, (ch + 1) * CHUNK);
  float den = 0.f, cs[3] = {0.f, 0.f, 0.f}, ct[3] = {0.f, 0.f, 0.f};
  if (PHASE >= 1) {
    double S = 0.0;
    for (int c = 0; c < nch; ++c) S += part[c * 16];
    den = (float)S + 1e-16f;                          // model.py:35 (fp32 sum + _EPS)
  }
  if (PHASE >= 2) {
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < nch; ++c)
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] += part[c * 16 + 1 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { cs[k] = (float)v[k]; ct[k] = (float)v[3 + k]; }
  }
  if (PHASE == 0) {
    double v1[1] = {0.0};
    DSIR_KABSCH_UNROLL
    for (int i = ch * CHUNK + threadIdx.x; i < i1; i += NTHR) v1[0] += (double)fabsf(weight(i));
    block_sum<1>(v1, sh);
    if (threadIdx.x == 0) part[ch * 16] = v1[0];
  } else if (PHASE == 1) {
    double v6[6] = {0, 0, 0, 0, 0, 0};
    DSIR_KABSCH_UNROLL
    for (int i = ch * CHUNK + threadIdx.x; i < i1; i += NTHR) {
      const float wn = weight(i) / den;
      float tx, ty, tz;
      target(i, tx, ty, tz);
      v6[0] += (double)__fmul_rn(src[(int64_t)i * 3], wn);
      v6[1] += (double)__fmul_rn(src[(int64_t)i * 3 + 1], wn);
      v6[2] += (double)__fmul_rn(src[(int64_t)i * 3 + 2], wn);
      v6[3] += (double)__fmul_rn(tx, wn);
      v6[4] += (double)__fmul_rn(ty, wn);
      v6[5] += (double)__fmul_rn(tz, wn);
    }
    block_sum<6>(v6, sh);
    if (threadIdx.x < 6) part[ch * 16 + 1 + threadIdx.x] = v6[threadIdx.x];
  } else {
    double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    DSIR_KABSCH_UNROLL
    for (int i = ch * CHUNK + threadIdx.x; i < i1; i += NTHR) {
      const float wn = weight(i) / den;
      float t[3];
      target(i, t[0], t[1], t[2]);
      float sc[3], tw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sc[k] = __fsub_rn(src[(int64_t)i * 3 + k], cs[k]);
        tw[k] = __fmul_rn(__fsub_rn(t[k], ct[k]), wn);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v9[r * 3 + c] += (double)__fmul_rn(sc[r], tw[c]);
    }
    block_sum<9>(v9, sh);
    if (threadIdx.x < 9) part[ch * 16 + 7 + threadIdx.x] = v9[threadIdx.x];
  }
}

// one thread per pair: the chunks' partials in order -> the solve of kabsch_kernel
__global__ __launch_bounds__(64) void kabsch_final_kernel(const KabschArgs a, int nch) {
  const int pair = blockIdx.x * 64 + threadIdx.x;
  if (pair >= a.pairs) return;
  if (a.skip && a.skip[pair]) {
    const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int k = 0; k < 12; ++k) a.T[(int64_t)pair * 12 + k] = I[k];
    if (a.T_cum) {
      float* out = a.T_cum + pair * a.T_stride;
      const float* P = a.T_prev ? a.T_prev + pair * a.T_stride : I;
      for (int k = 0; k < 12; ++k) out[k] = P[k];
    }
    return;
  }
  const double* part = a.part + (int64_t)pair * nch * 16;
  double v6[6] = {0, 0, 0, 0, 0, 0}, v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < nch; ++c) {
    for (int k = 0; k < 6; ++k) v6[k] += part[c * 16 + 1 + k];
    for (int k = 0; k < 9; ++k) v9[k] += part[c * 16 + 7 + k];
  }
  const float cs[3] = {(float)v6[0], (float)v6[1], (float)v6[2]};
  const float ct[3] = {(float)v6[3], (float)v6[4], (float)v6[5]};
  float sT[12];
  kabsch_solve(a, pair, v9, cs, ct, sT);
}

// apply: p' = p R^T + t (se3_torch.py:60-77); gather the matched ref points; one thread per point
__global__ __launch_bounds__(256) void kabsch_apply_kernel(const KabschArgs a) {
  const int pair = blockIdx.y;
  if (a.skip && a.skip[pair]) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m) return;
  const int ld = a.ref_ld ? a.ref_ld : 3;
  const float* src = a.src + pair * a.src_stride;
  if (a.matched_out) {
    const float* ref = a.ref + pair * a.ref_stride;
    const int64_t j = a.idx ? a.idx[(int64_t)pair * a.m + i] : i;
    float* mo = a.matched_out + ((int64_t)pair * a.m + i) * 3;
    mo[0] = ref[j * ld]; mo[1] = ref[j * ld + 1]; mo[2] = ref[j * ld + 2];
  }
  if (a.src_out) {
    const float* T = a.T + (int64_t)pair * 12;
    float* so = a.src_out + pair * a.src_out_stride;
    const float x = src[(int64_t)i * 3], y = src[(int64_t)i * 3 + 1], z = src[(int64_t)i * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float d = fmaf(z, T[r * 4 + 2], fmaf(y, T[r * 4 + 1], __fmul_rn(x, T[r * 4 + 0])));
      so[(int64_t)i * 3 + r] = __fadd_rn(d, T[r * 4 + 3]);
    }
  }
}

}  // namespace

// clouds of that many points and more take the chunked path (dsir_set_kabsch_chunked_min moves the threshold of a context)
static int chunked_min(int chunk_min) { return chunk_min > 0 ? chunk_min : kKabschChunkedMin; }

size_t kabsch_part_bytes(int pairs, int m, int chunk_min) {
  return m >= chunked_min(chunk_min) ? (size_t)pairs * ((m + CHUNK - 1) / CHUNK) * 16 * sizeof(double) : 0;
}

void launch_kabsch(const KabschArgs& a, hipStream_t st) {
  if (a.pairs <= 0) return;
  if (a.part && a.m >= chunked_min(a.chunk_min)) {
    // the choice depends on the cloud size alone: a pair's pose does not depend on what else is in the batch
    const int nch = (a.m + CHUNK - 1) / CHUNK;
    const dim3 grid(nch, a.pairs);
    hipLaunchKernelGGL(kabsch_part_kernel<0>, grid, dim3(NTHR), 0, st, a, nch);
    hipLaunchKernelGGL(kabsch_part_kernel<1>, grid, dim3(NTHR), 0, st, a, nch);
    hipLaunchKernelGGL(kabsch_part_kernel<2>, grid, dim3(NTHR), 0, st, a, nch);
    hipLaunchKernelGGL(kabsch_final_kernel, dim3((a.pairs + 63) / 64), dim3(64), 0, st, a, nch);
    if (a.src_out || a.matched_out) hipLaunchKernelGGL(kabsch_apply_kernel, dim3((a.m + 255) / 256, a.pairs), dim3(256), 0, st, a);
    return;
  }
  // (the choice depends on the cloud size alone, and the two kernels give the same bits)
  static const bool no_reg = tuning_flag("DSIR_KABSCH_STREAM");      // A/B switch: the streaming kernel for every size
  if (a.m <= 5 * NTHR && !no_reg) hipLaunchKernelGGL(kabsch_reg_kernel<5>, dim3(a.pairs), dim3(NTHR), 0, st, a);
  else if (a.m <= 8 * NTHR && !no_reg) hipLaunchKernelGGL(kabsch_reg_kernel<8>, dim3(a.pairs), dim3(NTHR), 0, st, a);
  else hipLaunchKernelGGL(kabsch_kernel, dim3(a.pairs), dim3(NTHR), 0, st, a);
}

}  // namespace dsir
