// Training operators (include/dsir_train.h): forward-with-saved-activations and backward of the ATen calls RandLA.forward
// makes (reference network/RandLANet.py:140-230, :311-408), point-major fp32.  The inference engine (engine.hip) fuses and
// never materialises most of these tensors; a training step needs them all, so this path is layer by layer: one
// exact-fp32 MFMA GEMM for every 1x1 convolution (forward, d input, d weight), and row-streaming kernels around it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdlib>

#include "device_utils.h"
#include "dsir_train.h"
#include "kernels.h"

namespace dsir {
namespace {

constexpr int TB = 64;        // GEMM tile: 64 rows x 64 columns
constexpr int TK = 16;        // K chunk
constexpr int TLD = TK + 2;   // LDS row (floats): fragment reads (row r, k = 4 s + q) hit banks 2 r + q

// Y[r][n] = beta Y[r][n] + bias[n] + sum_k X[r][k] W[n wn + k wk]; block = 4 waves, wave w owns rows 16 w .. 16 w + 15
__global__ __launch_bounds__(256) void t_gemm_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int wn, int wk,
                                                     const float* __restrict__ bias, float* __restrict__ Y, int ldy, int64_t rows,
                                                     int K, int N, float beta) {
  __shared__ float Xs[2][TB * TLD];
  __shared__ float Ws[2][TB * TLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * TB;
  const int n0 = blockIdx.y * TB;
  const int sr = tid >> 2, sk = (tid & 3) * 4;          // staging: row / column sr, four consecutive k
  const int64_t xr = m0 + sr;
  const int wc = n0 + sr;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // rows of X (and of W when it is k-contiguous) are read as one 16-byte load per thread where alignment allows
  const bool xv = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  const bool wv = wk == 1 && (wn & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  // the next K chunk is fetched into registers while the MFMAs of the current one run, then written to the other LDS buffer:
  // one barrier per chunk and no exposed load latency (round 3; the deep layers - 40 workgroups, K = 512 - were a chain of 32
  // fetch -> barrier -> 4 MFMAs -> barrier steps, 43 us)
  float xq[4], wq[4];
  auto gload = [&](int k0) {
    const int kb = k0 + sk;
    if (xv && xr < rows && kb + 3 < K) {
      const float4 v = *reinterpret_cast<const float4*>(X + xr * ldx + kb);
      xq[0] = v.x; xq[1] = v.y; xq[2] = v.z; xq[3] = v.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) xq[u] = (xr < rows && kb + u < K) ? X[xr * ldx + kb + u] : 0.f;
    }
    if (wv && wc < N && kb + 3 < K) {
      const float4 v = *reinterpret_cast<const float4*>(W + (int64_t)wc * wn + kb);
      wq[0] = v.x; wq[1] = v.y; wq[2] = v.z; wq[3] = v.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) wq[u] = (wc < N && kb + u < K) ? W[(int64_t)wc * wn + (int64_t)(kb + u) * wk] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { Xs[buf][sr * TLD + sk + u] = xq[u]; Ws[buf][sr * TLD + sk + u] = wq[u]; }
  };
  gload(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += TK) {
    const bool more = k0 + TK < K;
    if (more) gload(k0 + TK);
#pragma unroll
    for (int s = 0; s < TK / 4; ++s) {
      const float a = Xs[buf][(16 * w + fr) * TLD + 4 * s + fq];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Ws[buf][(16 * t + fr) * TLD + 4 * s + fq], acc[t], 0, 0, 0);
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + fr;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + 16 * w + 4 * fq + r;
      if (row < rows) {
        float v = acc[t][r] + bv;
        if (beta != 0.f) v += beta * Y[row * ldy + n];
        Y[row * ldy + n] = v;
      }
    }
  }
}

// partial[split][n][k'] = sum over the split's rows of dY[r][n] X'[r][k'], X' = [X, 1] when the bias column is wanted
constexpr int DLD = TB + 16;   // LDS row of the transposed-use tiles: (k-row q, column fr) -> banks 16 q + fr
__global__ __launch_bounds__(256) void t_gemm_dw_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx,
                                                        int64_t rows, int64_t rows_per_split, int N, int K, int Kp,
                                                        float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float As[2][TK * DLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][TK * DLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * TB, k0 = blockIdx.y * TB;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = min(rows, r_begin + rows_per_split);
  const int sr = tid >> 4, sc = (tid & 15) * 4;         // staging: row sr of the chunk, four consecutive columns
  const bool av = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
  const bool bv = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // as in t_gemm_kernel: the next 16-row chunk travels to registers during the MFMAs of the current one; one barrier per chunk
  float4 aq, bq;
  auto gload = [&](int64_t r0) {
    const int64_t r = r0 + sr;
    const bool rin = r < r_end;
    if (av && rin && n0 + sc + 3 < N) {
      aq = *reinterpret_cast<const float4*>(dY + r * ldy + n0 + sc);
    } else {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = (rin && n0 + sc + u < N) ? dY[r * ldy + n0 + sc + u] : 0.f;
      aq = make_float4(t[0], t[1], t[2], t[3]);
    }
    if (bv && rin && k0 + sc + 3 < K) {
      bq = *reinterpret_cast<const float4*>(X + r * ldx + k0 + sc);
    } else {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + sc + u;
        t[u] = !rin ? 0.f : (k < K ? X[r * ldx + k] : (k == K && Kp > K ? 1.f : 0.f));
      }
      bq = make_float4(t[0], t[1], t[2], t[3]);
    }
  };
  auto sstore = [&](int buf) {
    *reinterpret_cast<float4*>(&As[buf][sr * DLD + sc]) = aq;
    *reinterpret_cast<float4*>(&Bs[buf][sr * DLD + sc]) = bq;
  };
  int buf = 0;
  if (r_begin < r_end) {
    gload(r_begin);
    sstore(0);
  }
  __syncthreads();
  for (int64_t r0 = r_begin; r0 < r_end; r0 += TK) {
    const bool more = r0 + TK < r_end;
    if (more) gload(r0 + TK);
#pragma unroll
    for (int s = 0; s < TK / 4; ++s) {
      const float a = As[buf][(4 * s + fq) * DLD + 16 * w + fr];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[buf][(4 * s + fq) * DLD + 16 * t + fr], acc[t], 0, 0, 0);
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  float* P = partial + (int64_t)blockIdx.z * N * Kp;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = k0 + 16 * t + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * w + 4 * fq + r;
      if (n < N && k < Kp) P[(int64_t)n * Kp + k] = acc[t][r];
    }
  }
}

// block = 32 consecutive elements x 8 split lanes: lane (el, sl) adds the splits sl, sl + 8, ... of its element in order - a wave reads
// two 128-byte runs per trip instead of 64 scattered floats -, then the 8 lanes of an element meet in LDS in a fixed order
// (deterministic)
__global__ __launch_bounds__(256) void t_gemm_dw_reduce_kernel(const float* __restrict__ partial, int splits, int N, int K, int Kp,
                                                               float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sh[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + el;
  const int64_t total = (int64_t)N * Kp;
  float s = 0.f;
  if (e < total) {
#pragma unroll 4
    for (int i = sl; i < splits; i += 8) s += partial[(int64_t)i * total + e];
  }
  sh[sl][el] = s;
  __syncthreads();
  if (sl != 0 || e >= total) return;
  float a = sh[0][el];
#pragma unroll
  for (int q = 1; q < 8; ++q) a += sh[q][el];
  const int n = (int)(e / Kp), k = (int)(e % Kp);
  if (k < K) dW[(int64_t)n * K + k] += a;
  else if (db) db[n] += a;
}

// ---- GroupNorm / BatchNorm(train) -------------------------------------------------------------------------------------
// row chunks of the GroupNorm reductions: 128 rows and up to 256 chunks per cloud (round 3; 1024 / 64 before: a per-point layer of
// 8 x 5000 rows ran as 40 workgroups whose threads walked 128 rows each, 60 us - the trace showed 24 - 80 workgroups per launch
// on most of the 148 layers of a step)
constexpr int GN_CHUNK_ROWS = 128, GN_MAX_CHUNKS = 256;
inline int gn_chunks(int M) { const int c = (M + GN_CHUNK_ROWS - 1) / GN_CHUNK_ROWS; return c < 1 ? 1 : (c > GN_MAX_CHUNKS ? GN_MAX_CHUNKS : c); }

// stage 1: one block per (row chunk, group, cloud): sum and sum of squares over its rows x gw channels, fp64
__global__ __launch_bounds__(256) void t_gn_stats_kernel(const float* __restrict__ Y, int M, int C, int groups, int rows_per_chunk,
                                                         double* __restrict__ partial) {
  const int chunk = blockIdx.x, g = blockIdx.y, cloud = blockIdx.z, gw = C / groups, nchunks = gridDim.x;
  const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  const float* base = Y + ((int64_t)cloud * M + r0) * C + g * gw;
  double s1 = 0.0, s2 = 0.0;
  const int64_t total = (int64_t)max(r1 - r0, 0) * gw;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const float v = base[(e / gw) * C + (e % gw)];
    s1 += v; s2 += (double)v * v;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  __shared__ double sh[2][4];
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + (((int64_t)cloud * groups + g) * nchunks + chunk) * 2;
    o[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    o[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// stage 2: mean, 1 / sqrt(var + eps).  Eight lanes per (cloud, group): lane j adds the chunks j, j + 8, ... in order, then a fixed
// xor tree - deterministic, and 32 dependent loads instead of 256 on the large layers
__global__ __launch_bounds__(256) void t_gn_stats_final_kernel(const double* __restrict__ partial, int nchunks, int count, double inv_total,
                                                               float* __restrict__ stats) {
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3);      // cloud * groups + g
  const int j = threadIdx.x & 7;
  double a = 0.0, b = 0.0;
  if (i < count)
    for (int c = j; c < nchunks; c += 8) { a += partial[((int64_t)i * nchunks + c) * 2]; b += partial[((int64_t)i * nchunks + c) * 2 + 1]; }
  a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
  a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
  a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
  if (i >= count || j != 0) return;
  const double mean = a * inv_total;
  double var = b * inv_total - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

__global__ void t_gn_apply_kernel(const float* __restrict__ Y, const float* __restrict__ stats, int M, int C, int groups,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* __restrict__ out,
                                  int64_t total) {
  const int gw = C / groups;
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    // four consecutive channels of one row per trip (C % 4 == 0: a float4 never straddles rows), same arithmetic per element
    for (int64_t e = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); e < total; e += 4 * (int64_t)gridDim.x * blockDim.x) {
      const int c = (int)(e % C);
      const int64_t cloud = e / ((int64_t)M * C);
      const float4 y4 = *reinterpret_cast<const float4*>(Y + e);
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* st = stats + (cloud * groups + (c + u) / gw) * 2;
        float v = (yv[u] - st[0]) * st[1] * gamma[c + u] + beta[c + u];
        if (act && !(v > 0.f)) v *= 0.2f;
        o[u] = v;
      }
      *reinterpret_cast<float4*>(out + e) = make_float4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t cloud = e / ((int64_t)M * C);
    const float* st = stats + (cloud * groups + c / gw) * 2;
    float v = (Y[e] - st[0]) * st[1] * gamma[c] + beta[c];
    if (act && !(v > 0.f)) v *= 0.2f;
    out[e] = v;
  }
}

// stage 1, per (cloud, row chunk, channel): s1 = sum_r g, s2 = sum_r g xhat, g = dOut * slope(v); block = 32 channels x 8 row lanes
__global__ __launch_bounds__(256) void t_gn_bwd_sums_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                            const float* __restrict__ stats, int M, int C, int groups,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                            int rows_per_chunk, float* __restrict__ partial) {
  const int cloud = blockIdx.z, chunk = blockIdx.y, nchunks = gridDim.y;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), ry = threadIdx.x >> 5;
  const int gw = C / groups;
  const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float* st = stats + ((int64_t)cloud * groups + c / gw) * 2;
    const float mean = st[0], rstd = st[1], ga = gamma[c], be = beta[c];
    for (int r = r0 + ry; r < r1; r += 8) {
      const int64_t e = ((int64_t)cloud * M + r) * C + c;
      const float xh = (Y[e] - mean) * rstd;
      float g = dOut[e];
      if (act && !(xh * ga + be > 0.f)) g *= 0.2f;
      s1 += g; s2 += (double)g * xh;
    }
  }
  __shared__ double sh[2][8][32];
  sh[0][ry][threadIdx.x & 31] = s1; sh[1][ry][threadIdx.x & 31] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < 8; ++i) { a += sh[0][i][threadIdx.x]; b += sh[1][i][threadIdx.x]; }
    float* o = partial + (((int64_t)cloud * nchunks + chunk) * C + c) * 2;
    o[0] = (float)a; o[1] = (float)b;
  }
}

// stage 2: sums[cloud][C][2]; eight lanes per output as in t_gn_stats_final_kernel
__global__ __launch_bounds__(256) void t_gn_bwd_sums_final_kernel(const float* __restrict__ partial, int nchunks, int C, int64_t count,
                                                                  float* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);     // (cloud * C + c) * 2 + which
  const int j = threadIdx.x & 7;
  double a = 0.0;
  if (i < count) {
    const int64_t cloud = i / (2 * C), rest = i % (2 * C);
    for (int k = j; k < nchunks; k += 8) a += partial[(cloud * nchunks + k) * 2 * C + rest];
  }
  a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
  if (i < count && j == 0) sums[i] = (float)a;
}

// dY = rstd (gamma g - mean_group(gamma g) - xhat mean_group(gamma g xhat)); one block per (row range, cloud)
__global__ __launch_bounds__(256) void t_gn_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                             const float* __restrict__ stats, const float* __restrict__ sums, int M,
                                                             int C, int groups, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act, float* __restrict__ dY,
                                                             int rows_per_block, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float gm[];   // [groups][2]: mean(gamma g), mean(gamma g xhat)
  const int cloud = blockIdx.y, gw = C / groups;
  // the parameter gradients (sums over the clouds, in cloud order) ride along in the first workgroup: one launch less per layer
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int c = threadIdx.x; c < C; c += 256) {
      float a = 0.f, b = 0.f;
      for (int i = 0; i < (int)gridDim.y; ++i) { a += sums[((int64_t)i * C + c) * 2]; b += sums[((int64_t)i * C + c) * 2 + 1]; }
      dbeta[c] += a;
      dgamma[c] += b;
    }
  for (int g = threadIdx.x; g < groups; g += 256) {
    double a = 0.0, b = 0.0;
    for (int c = g * gw; c < (g + 1) * gw; ++c) {
      a += (double)gamma[c] * sums[((int64_t)cloud * C + c) * 2];
      b += (double)gamma[c] * sums[((int64_t)cloud * C + c) * 2 + 1];
    }
    const double inv = 1.0 / ((double)M * gw);
    gm[2 * g] = (float)(a * inv); gm[2 * g + 1] = (float)(b * inv);
  }
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int64_t n = (int64_t)min(rows_per_block, M - r0) * C;
  const int64_t base = ((int64_t)cloud * M + r0) * C;
  // C a power of two <= 1024 (every layer of the network): a thread's four consecutive channels are the same in every trip
  // (4 tid + 1024 k mod C), so their parameters are loaded once and the tensors travel as float4 - same arithmetic per element
  if ((C & 3) == 0 && (1024 % C) == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dOut) | reinterpret_cast<uintptr_t>(dY)) & 15) == 0) {
    const int c0 = (4 * threadIdx.x) % C;
    float mean[4], rstd[4], ga[4], be[4], g0[4], g1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u, g = c / gw;
      const float* st = stats + ((int64_t)cloud * groups + g) * 2;
      mean[u] = st[0]; rstd[u] = st[1]; ga[u] = gamma[c]; be[u] = beta[c]; g0[u] = gm[2 * g]; g1[u] = gm[2 * g + 1];
    }
    for (int64_t i = 4 * (int64_t)threadIdx.x; i < n; i += 1024) {
      const float4 y4 = *reinterpret_cast<const float4*>(Y + base + i);
      const float4 d4 = *reinterpret_cast<const float4*>(dOut + base + i);
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
      float dv[4] = {d4.x, d4.y, d4.z, d4.w};
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xh = (yv[u] - mean[u]) * rstd[u];
        if (act && !(xh * ga[u] + be[u] > 0.f)) dv[u] *= 0.2f;
        o[u] = rstd[u] * (ga[u] * dv[u] - g0[u] - xh * g1[u]);
      }
      *reinterpret_cast<float4*>(dY + base + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t e = base + i;
    const int c = (int)(e % C), g = c / gw;
    const float* st = stats + ((int64_t)cloud * groups + g) * 2;
    const float xh = (Y[e] - st[0]) * st[1];
    float d = dOut[e];
    if (act && !(xh * gamma[c] + beta[c] > 0.f)) d *= 0.2f;
    dY[e] = st[1] * (gamma[c] * d - gm[2 * g] - xh * gm[2 * g + 1]);
  }
}

// ---- gathers ----------------------------------------------------------------------------------------------------------
__global__ void t_gather_kernel(const float* __restrict__ X, int n, int C, const int32_t* __restrict__ idx, int m, float* __restrict__ Y,
                                int ldy, int col_off, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t j = e / C;                 // cloud * m + j
    const int64_t cloud = j / m;
    int i = idx[j];
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    Y[j * ldy + col_off + c] = X[(cloud * n + i) * C + c];
  }
}

// backward of a gather: dX[d][c] = sum of dY over the sources of destination row d, in ascending source order (the plan of
// launch_scatter_plan): every sum has one order - no float atomics, same bits every run
__global__ void t_scatter_sum_kernel(const float* __restrict__ dY, int ldy, int col_off, const int32_t* __restrict__ order,
                                     const int32_t* __restrict__ offsets, float* __restrict__ dX, int C, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t d = e / C;                 // cloud * n + i
    float s = 0.f;
    for (int q = offsets[d]; q < offsets[d + 1]; ++q) s += dY[(int64_t)order[q] * ldy + col_off + c];
    dX[e] = s;
  }
}

__global__ void t_relpos_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int n, int k, float* __restrict__ out,
                                int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = e / k;                 // cloud * n + i
    const int64_t cloud = p / n;
    int j = idx[e];
    j = j < 0 ? 0 : (j >= n ? n - 1 : j);
    const float* pi = xyz + p * 3;
    const float* pj = xyz + (cloud * n + j) * 3;
    const float dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];
    float* o = out + e * 10;
    o[0] = sqrtf(dx * dx + dy * dy + dz * dz);
    o[1] = dx; o[2] = dy; o[3] = dz;
    o[4] = pi[0]; o[5] = pi[1]; o[6] = pi[2];
    o[7] = pj[0]; o[8] = pj[1]; o[9] = pj[2];
  }
}

// the inlier model's input of one registration iteration (model.py:571-573 with :587): [T x_src ; x_ref[idx]]
__global__ void t_inlier_input_kernel(const float* __restrict__ xs, const float* __restrict__ xr, const int32_t* __restrict__ idx,
                                      const float* __restrict__ T, int t_stride, int J, int K, float* __restrict__ out, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pair = e / J;
    const float* p = xs + e * 3;
    float x = p[0], y = p[1], z = p[2];
    if (T) {
      const float* t = T + pair * t_stride;
      const float a = t[0] * x + t[1] * y + t[2] * z + t[3];
      const float b = t[4] * x + t[5] * y + t[6] * z + t[7];
      const float c = t[8] * x + t[9] * y + t[10] * z + t[11];
      x = a; y = b; z = c;
    }
    int i = idx[e];
    i = i < 0 ? 0 : (i >= K ? K - 1 : i);
    const float* q = xr + (pair * K + i) * 3;
    float* o = out + e * 6;
    o[0] = x; o[1] = y; o[2] = z; o[3] = q[0]; o[4] = q[1]; o[5] = q[2];
  }
}

// ---- attentive pooling ------------------------------------------------------------------------------------------------
// k = 16 neighbours (the network's only value): a point's column of scores and features lives in registers - one read of S and
// cat and one exp per element instead of three reads and two exps; same operations in the same order (same bits)
__global__ void t_attpool_fwd_kernel(const float* __restrict__ cat, float* __restrict__ S, int k, int C, float* __restrict__ out,
                                     int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t p = e / C;
    const int64_t b = p * k * C + c;
    if (k == 16) {
      float sv[16], cv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { sv[t] = S[b + (int64_t)t * C]; cv[t] = cat[b + (int64_t)t * C]; }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 16; ++t) mx = fmaxf(mx, sv[t]);
      float se = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) { sv[t] = expf(sv[t] - mx); se += sv[t]; }
      const float inv = 1.f / se;
      float o = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float a = sv[t] * inv;
        S[b + (int64_t)t * C] = a;
        o += a * cv[t];
      }
      out[e] = o;
      continue;
    }
    float mx = -INFINITY;
    for (int t = 0; t < k; ++t) mx = fmaxf(mx, S[b + (int64_t)t * C]);
    float se = 0.f;
    for (int t = 0; t < k; ++t) se += expf(S[b + (int64_t)t * C] - mx);
    const float inv = 1.f / se;
    float o = 0.f;
    for (int t = 0; t < k; ++t) {
      const float a = expf(S[b + (int64_t)t * C] - mx) * inv;
      S[b + (int64_t)t * C] = a;
      o += a * cat[b + (int64_t)t * C];
    }
    out[e] = o;
  }
}

__global__ void t_attpool_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ cat, const float* __restrict__ A, int k,
                                     int C, float* __restrict__ dCat, float* __restrict__ dS, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t p = e / C;
    const int64_t b = p * k * C + c;
    const float g = dOut[e];
    if (k == 16) {
      float av[16], cv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { av[t] = A[b + (int64_t)t * C]; cv[t] = cat[b + (int64_t)t * C]; }
      float dot = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) dot += av[t] * cv[t];
      dot *= g;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        dCat[b + (int64_t)t * C] = av[t] * g;
        dS[b + (int64_t)t * C] = av[t] * (cv[t] * g - dot);
      }
      continue;
    }
    float dot = 0.f;
    for (int t = 0; t < k; ++t) dot += A[b + (int64_t)t * C] * cat[b + (int64_t)t * C];
    dot *= g;                                // sum_t a_t (cat_t g)
    for (int t = 0; t < k; ++t) {
      const float a = A[b + (int64_t)t * C];
      dCat[b + (int64_t)t * C] = a * g;
      dS[b + (int64_t)t * C] = a * (cat[b + (int64_t)t * C] * g - dot);
    }
  }
}

// ---- pooling / element-wise ---------------------------------------------------------------------------------------------
__global__ void t_maxpool_fwd_kernel(const float* __restrict__ X, int n, int C, const int32_t* __restrict__ pool, int m, int k,
                                     float* __restrict__ out, int32_t* __restrict__ arg, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t j = e / C;                 // cloud * m + j
    const int64_t cloud = j / m;
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < k; ++t) {
      int i = pool[j * k + t];
      i = i < 0 ? 0 : (i >= n ? n - 1 : i);
      const float v = X[(cloud * n + i) * C + c];
      if (t == 0 || v > best) { best = v; bi = i; }
    }
    out[e] = best;
    arg[e] = bi;
  }
}

// backward of the max over k pooled rows: dX[i][c] = sum of dOut[j][c] over the pooled outputs j whose winner for channel c was row i,
// walked through the inverse of the pool index in ascending (j, t) order; a row listed twice by one output counts once
__global__ void t_maxpool_bwd_kernel(const float* __restrict__ dOut, const int32_t* __restrict__ arg, const int32_t* __restrict__ order,
                                     const int32_t* __restrict__ offsets, int m, int k, int C, float* __restrict__ dX, int n, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t d = e / C;                 // cloud * n + i
    const int i = (int)(d % n);
    float s = 0.f;
    int64_t prev = -1;
    for (int q = offsets[d]; q < offsets[d + 1]; ++q) {
      const int64_t j = (int64_t)order[q] / k;           // cloud * m + output row
      if (j != prev && arg[j * C + c] == i) s += dOut[j * C + c];
      prev = j;
    }
    dX[e] = s;
  }
}

__global__ void t_add_leaky_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float v = a[e] + b[e];
    if (!(v > 0.f)) v *= 0.2f;
    out[e] = v;
  }
}
__global__ void t_add_leaky_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ out, int64_t n, float* __restrict__ d) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    d[e] = out[e] > 0.f ? dOut[e] : 0.2f * dOut[e];
}
__global__ void t_mul_mask_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float scale, int64_t n,
                                  float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    y[e] = mask[e] ? x[e] * scale : 0.f;
}
__global__ void t_sigmoid_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] = 1.f / (1.f + expf(-x[e]));
}
__global__ void t_axpy_kernel(float a, const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] += a * x[e];
}
__global__ void t_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                              int64_t n, float b1, float b2, float eps, float step_size, float inv_sqrt_bc2) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[e];
    const float mi = b1 * m[e] + (1.f - b1) * gi;
    const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi; v[e] = vi;
    p[e] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

// running statistics of nn.BatchNorm1d in training mode: running = (1 - momentum) running + momentum batch, the variance
// unbiased (M / (M - 1)); stats [C][2] = {mean, rstd} of dsir_t_gn_fwd with groups = C
__global__ void t_bn_running_kernel(const float* __restrict__ stats, int C, double M, float momentum, float* __restrict__ rmean,
                                    float* __restrict__ rvar) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = stats[2 * c], rstd = stats[2 * c + 1];
  const double var = fmax(1.0 / (rstd * rstd) - 1e-5, 0.0) * (M > 1.0 ? M / (M - 1.0) : 1.0);
  rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
  rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var);
}

// ---- L2 normalisation over channels (F.normalize, p = 2, eps 1e-12) -------------------------------------------------
__global__ void t_l2norm_fwd_kernel(const float* __restrict__ x, int64_t rows, int C, float* __restrict__ y, float* __restrict__ nrm) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += x[r * C + c] * x[r * C + c];
    const float n = fmaxf(sqrtf(s), 1e-12f);
    nrm[r] = n;
    for (int c = 0; c < C; ++c) y[r * C + c] = x[r * C + c] / n;
  }
}
__global__ void t_l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ nrm, int64_t rows,
                                    int C, float* __restrict__ dx) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += y[r * C + c] * dy[r * C + c];
    const float inv = 1.f / nrm[r];
    for (int c = 0; c < C; ++c) dx[r * C + c] = (dy[r * C + c] - y[r * C + c] * dot) * inv;
  }
}

// ---- DetDesLoss = CircleLoss + detection term (reference network/loss.py:500-571, :667-702) -----------------------------
// Restated INCLUDING what its arithmetic does (oracle/train.py::det_des_loss): dist_min = min_j(dist_pc * false_negative),
// pos_mask = (dist_pc == dist_min), masked entries enter the log-sum-exps with exponent 0.
constexpr float kCEps = 1e5f, kCScale = 10.f, kCPosM = 0.1f, kCNegM = 1.4f;
struct CircleRow { float dist_min, lse_pos, lse_neg, far, close; int arg_far, arg_close, far_masked; };

// dist_pc, dist_feat from the dot products (dot[i][j] = anc_i . pos_j) and the squared norms
__global__ void t_circle_dist_kernel(const float* __restrict__ dot, const float* __restrict__ anc, const float* __restrict__ pos,
                                     const float* __restrict__ anc_pc, const float* __restrict__ src_pc, const float* __restrict__ T, int M,
                                     int C, float* __restrict__ dist_pc, float* __restrict__ dist_feat, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e % M);
    const int64_t pi = e / M;                 // pair * M + i
    const int64_t pair = pi / M;
    const int64_t pj = pair * M + j;
    const float* t = T + pair * 12;
    const float* q = src_pc + pj * 3;
    const float px = t[0] * q[0] + t[1] * q[1] + t[2] * q[2] + t[3], py = t[4] * q[0] + t[5] * q[1] + t[6] * q[2] + t[7],
                pz = t[8] * q[0] + t[9] * q[1] + t[10] * q[2] + t[11];
    const float* a = anc_pc + pi * 3;
    const float dx = a[0] - px, dy = a[1] - py, dz = a[2] - pz;
    dist_pc[e] = sqrtf(dx * dx + dy * dy + dz * dz);
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; ++c) { sa += anc[pi * C + c] * anc[pi * C + c]; sb += pos[pj * C + c] * pos[pj * C + c]; }
    dist_feat[e] = sqrtf(((-2.f * dot[e]) + sa) + sb + 1e-16f);
  }
}

__device__ __forceinline__ void circle_terms(float dpc, float dft, float dist_min, float thres, float& posw, float& negw, float& pw,
                                             float& nw, bool& pmask) {
  const bool fn = dpc < thres;
  pmask = dpc == dist_min;
  const bool negm = !(pmask || fn);
  const float p = dft - (negm ? kCEps : 0.f);
  pw = fmaxf(p - kCPosM, 0.f);
  posw = kCScale * (p - kCPosM) * pw;
  const float n = dft + (negm ? 0.f : kCEps);
  nw = fmaxf(kCNegM - n, 0.f);
  negw = kCScale * (kCNegM - n) * nw;
}

template <typename F>
__device__ __forceinline__ float block_reduce(float v, float* sh, F op) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = op(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

// one block per (row i, pair): dist_min, the two row log-sum-exps, furthest positive / closest negative with their columns
__global__ __launch_bounds__(256) void t_circle_rows_kernel(const float* __restrict__ dist_pc, const float* __restrict__ dist_feat, int M,
                                                            float thres, CircleRow* __restrict__ rows) {
  __shared__ float sh[256];
  __shared__ int shi[256];
  const int64_t pi = (int64_t)blockIdx.y * M + blockIdx.x;
  const float* dp = dist_pc + pi * M;
  const float* df = dist_feat + pi * M;
  float v = INFINITY;
  for (int j = threadIdx.x; j < M; j += 256) v = fminf(v, dp[j] < thres ? dp[j] : 0.f);
  const float dist_min = block_reduce(v, sh, [](float a, float b) { return fminf(a, b); });
  float mp = -INFINITY, mn = -INFINITY, far = -INFINITY, close = INFINITY;
  int afar = 0x7fffffff, aclose = 0x7fffffff;
  for (int j = threadIdx.x; j < M; j += 256) {
    float posw, negw, pw, nw; bool pm;
    circle_terms(dp[j], df[j], dist_min, thres, posw, negw, pw, nw, pm);
    mp = fmaxf(mp, posw); mn = fmaxf(mn, negw);
    const float f = pm ? df[j] : 0.f, c = df[j] + (pm ? kCEps : 0.f);
    if (f > far) { far = f; afar = j; }
    if (c < close) { close = c; aclose = j; }
  }
  const float gmp = block_reduce(mp, sh, [](float a, float b) { return fmaxf(a, b); });
  const float gmn = block_reduce(mn, sh, [](float a, float b) { return fmaxf(a, b); });
  float sp = 0.f, sn = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    float posw, negw, pw, nw; bool pm;
    circle_terms(dp[j], df[j], dist_min, thres, posw, negw, pw, nw, pm);
    sp += expf(posw - gmp); sn += expf(negw - gmn);
  }
  sp = block_reduce(sp, sh, [](float a, float b) { return a + b; });
  sn = block_reduce(sn, sh, [](float a, float b) { return a + b; });
  // arg max / arg min: the first column attaining the extreme (torch.max / torch.min return the first on the CPU)
  const float gfar = block_reduce(far, sh, [](float a, float b) { return fmaxf(a, b); });
  const float gclose = block_reduce(close, sh, [](float a, float b) { return fminf(a, b); });
  shi[threadIdx.x] = far == gfar ? afar : 0x7fffffff;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) shi[threadIdx.x] = min(shi[threadIdx.x], shi[threadIdx.x + o]); __syncthreads(); }
  const int jfar = shi[0];
  __syncthreads();
  shi[threadIdx.x] = close == gclose ? aclose : 0x7fffffff;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) shi[threadIdx.x] = min(shi[threadIdx.x], shi[threadIdx.x + o]); __syncthreads(); }
  const int jclose = shi[0];
  if (threadIdx.x == 0) {
    CircleRow r;
    r.dist_min = dist_min; r.lse_pos = gmp + logf(sp); r.lse_neg = gmn + logf(sn); r.far = gfar; r.close = gclose;
    r.arg_far = jfar; r.arg_close = jclose; r.far_masked = (jfar < M && dp[jfar] == dist_min) ? 1 : 0;
    rows[pi] = r;
  }
}

// one block per (column j, pair): the column log-sum-exp of the negative terms
__global__ __launch_bounds__(256) void t_circle_cols_kernel(const float* __restrict__ dist_pc, const float* __restrict__ dist_feat, int M,
                                                            float thres, const CircleRow* __restrict__ rows, float* __restrict__ lse_col) {
  __shared__ float sh[256];
  const int64_t pair = blockIdx.y;
  const int j = blockIdx.x;
  float mn = -INFINITY;
  for (int i = threadIdx.x; i < M; i += 256) {
    const int64_t e = (pair * M + i) * M + j;
    float posw, negw, pw, nw; bool pm;
    circle_terms(dist_pc[e], dist_feat[e], rows[pair * M + i].dist_min, thres, posw, negw, pw, nw, pm);
    mn = fmaxf(mn, negw);
  }
  const float gmn = block_reduce(mn, sh, [](float a, float b) { return fmaxf(a, b); });
  float sn = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) {
    const int64_t e = (pair * M + i) * M + j;
    float posw, negw, pw, nw; bool pm;
    circle_terms(dist_pc[e], dist_feat[e], rows[pair * M + i].dist_min, thres, posw, negw, pw, nw, pm);
    sn += expf(negw - gmn);
  }
  sn = block_reduce(sn, sh, [](float a, float b) { return a + b; });
  if (threadIdx.x == 0) lse_col[pair * M + j] = gmn + logf(sn);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus (threshold 20)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// one block: the loss values and the per-row / per-column coefficients of the backward pass
// coef[pi] = {A = sigmoid(lse_pos + lse_neg_row) / (10 P M), B = sigmoid(lse_pos + lse_neg_col) / (10 P M), c = det_w score / sum / (P M)}
__global__ __launch_bounds__(256) void t_circle_final_kernel(const CircleRow* __restrict__ rows, const float* __restrict__ lse_col,
                                                             const float* __restrict__ score, int P, int M, float det_w,
                                                             float* __restrict__ coef, double* __restrict__ out) {
  __shared__ double sh[3][256];
  __shared__ float ssum;
  double lf = 0.0, ld = 0.0, acc = 0.0;
  const float inv = 1.f / ((float)P * (float)M);
  for (int p = 0; p < P; ++p) {
    float s = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) s += score[(int64_t)p * M + i];
    sh[0][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < 256; ++k) t += sh[0][k]; ssum = (float)t; }
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += 256) {
      const int64_t pi = (int64_t)p * M + i;
      const CircleRow r = rows[pi];
      const float xr = r.lse_pos + r.lse_neg, xc = r.lse_pos + lse_col[pi];
      lf += (double)(softplus_f(xr) / kCScale + softplus_f(xc) / kCScale);
      const float diff = r.far - r.close, sc = score[pi] / ssum;
      ld += (double)(diff * sc);
      acc += diff < 0.f ? 1.0 : 0.0;
      coef[pi * 3] = sigmoid_f(xr) / kCScale * inv;
      coef[pi * 3 + 1] = sigmoid_f(xc) / kCScale * inv;
      coef[pi * 3 + 2] = det_w * sc * inv;
    }
    __syncthreads();
  }
  sh[0][threadIdx.x] = lf; sh[1][threadIdx.x] = ld; sh[2][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = 0; k < 256; ++k) { a += sh[0][k]; b += sh[1][k]; c += sh[2][k]; }
    out[1] = a * inv; out[2] = b * inv; out[0] = out[1] + out[2] * det_w; out[3] = c * 100.0 / (double)M;
  }
}

// one block per (row i, pair): Wn[i][j] = -2 d loss / d sq[i][j] (and its transpose), rowsum[i] = sum_j 2 d loss / d sq[i][j]
__global__ __launch_bounds__(256) void t_circle_grad_kernel(const float* __restrict__ dist_pc, const float* __restrict__ dist_feat, int M,
                                                            float thres, const CircleRow* __restrict__ rows, const float* __restrict__ lse_col,
                                                            const float* __restrict__ coef, float* __restrict__ Wn, float* __restrict__ Wnt,
                                                            float* __restrict__ rowsum) {
  __shared__ float sh[256];
  const int64_t pair = blockIdx.y, pi = pair * M + blockIdx.x;
  const CircleRow r = rows[pi];
  const float A = coef[pi * 3], B = coef[pi * 3 + 1], cd = coef[pi * 3 + 2];
  float rs = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    const int64_t e = pi * M + j;
    float posw, negw, pw, nw; bool pm;
    circle_terms(dist_pc[e], dist_feat[e], r.dist_min, thres, posw, negw, pw, nw, pm);
    const float Bj = coef[(pair * M + j) * 3 + 1];
    float g = (A + B) * expf(posw - r.lse_pos) * kCScale * pw - A * expf(negw - r.lse_neg) * kCScale * nw -
              Bj * expf(negw - lse_col[pair * M + j]) * kCScale * nw;
    if (j == r.arg_far && r.far_masked) g += cd;
    if (j == r.arg_close) g -= cd;
    const float w2 = g / dist_feat[e];                    // 2 d loss / d sq (d sqrt(u) = du / (2 sqrt(u)))
    Wn[e] = -w2;
    Wnt[(pair * M + j) * M + blockIdx.x] = -w2;
    rs += w2;
  }
  rs = block_reduce(rs, sh, [](float a, float b) { return a + b; });
  if (threadIdx.x == 0) rowsum[pi] = rs;
}

__global__ __launch_bounds__(256) void t_rowsum_neg_kernel(const float* __restrict__ Wnt, int M, float* __restrict__ colsum) {
  __shared__ float sh[256];
  const int64_t pj = (int64_t)blockIdx.y * M + blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) s -= Wnt[pj * M + i];
  s = block_reduce(s, sh, [](float a, float b) { return a + b; });
  if (threadIdx.x == 0) colsum[pj] = s;
}

__global__ void t_scale_rows_kernel(const float* __restrict__ s, const float* __restrict__ x, int C, float* __restrict__ y, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) y[e] = s[e / C] * x[e];
}

// weighted cross entropy, stage 1: one thread per row; labels 0 = ignored, class = label - 1 (SemanticLoss.compute_loss);
// dlogits = w_y (softmax - onehot) (scaled by 1 / sum w in stage 2); per-block partial {sum w nll, sum w, correct, valid}
__global__ __launch_bounds__(256) void t_wce_rows_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                         const float* __restrict__ cw, int64_t rows, int C, float* __restrict__ dlogits,
                                                         double* __restrict__ partial) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double nll = 0.0, w = 0.0, correct = 0.0, valid = 0.0;
  if (r < rows) {
    const float* x = logits + r * C;
    float* d = dlogits + r * C;
    const int lab = labels[r];
    if (lab < 1 || lab > C) {
      for (int c = 0; c < C; ++c) d[c] = 0.f;
    } else {
      const int y = lab - 1;
      float mx = x[0];
      int am = 0;
      for (int c = 1; c < C; ++c) if (x[c] > mx) { mx = x[c]; am = c; }
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
      const float lse = mx + logf(se), wy = cw[y];
      for (int c = 0; c < C; ++c) d[c] = wy * (expf(x[c] - lse) - (c == y ? 1.f : 0.f));
      nll = (double)wy * (double)(lse - x[y]); w = wy; correct = am == y ? 1.0 : 0.0; valid = 1.0;
    }
  }
  nll = wave_sum(nll); w = wave_sum(w); correct = wave_sum(correct); valid = wave_sum(valid);
  __shared__ double sh[4][4];
  if ((threadIdx.x & 63) == 0) { const int i = threadIdx.x >> 6; sh[0][i] = nll; sh[1][i] = w; sh[2][i] = correct; sh[3][i] = valid; }
  __syncthreads();
  if (threadIdx.x < 4) partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// stage 2 (one block): blocks in order -> out {loss = sum w nll / sum w, sum w, correct, valid}
__global__ __launch_bounds__(256) void t_wce_final_kernel(const double* __restrict__ partial, int blocks, double* __restrict__ out) {
  __shared__ double sh[4][256];
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < blocks; b += 256)
    for (int k = 0; k < 4; ++k) a[k] += partial[(int64_t)b * 4 + k];
  for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = a[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += sh[threadIdx.x][i];
    sh[threadIdx.x][0] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = sh[1][0] > 0.0 ? sh[0][0] / sh[1][0] : 0.0;
    out[1] = sh[1][0]; out[2] = sh[2][0]; out[3] = sh[3][0];
  }
}

__global__ void t_wce_scale_kernel(float* __restrict__ d, int64_t n, const double* __restrict__ out, float scale) {
  const float f = out[1] > 0.0 ? (float)((double)scale / out[1]) : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) d[e] *= f;
}

// flag[0] = 1 when any element is NaN (train.py:437-441: the step is skipped then)
__global__ void t_any_nan_kernel(const float* __restrict__ x, int64_t n, int32_t* __restrict__ flag) {
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) bad |= x[e] != x[e];
  if (__any(bad) && (threadIdx.x & 63) == 0) flag[0] = 1;
}

inline unsigned grid1(int64_t n) { const int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
inline int done() { return (int)hipGetLastError(); }

struct DwPlan { int splits; int64_t rows_per_split; int Kp; };
inline DwPlan dw_plan(int64_t rows, int N, int K, bool bias) {
  DwPlan p;
  p.Kp = K + (bias ? 1 : 0);
  const int64_t tiles = (int64_t)((N + TB - 1) / TB) * ((p.Kp + TB - 1) / TB);
  int64_t sp = (rows + 127) / 128;                       // >= 128 rows per split (512 before: 32 barrier-separated K steps, 40 us, for 79 workgroups)
  const int64_t cap = tiles >= 2048 ? 1 : 2048 / tiles;   // ~ 2048 workgroups in all
  sp = sp > cap ? cap : sp;
  sp = sp > 1024 ? 1024 : sp;
  sp = sp < 1 ? 1 : sp;
  p.rows_per_split = (((rows + sp - 1) / sp) + TK - 1) / TK * TK;
  p.splits = (int)((rows + p.rows_per_split - 1) / p.rows_per_split);
  if (p.splits < 1) p.splits = 1;
  return p;
}

}  // namespace
}  // namespace dsir

using namespace dsir;

extern "C" {

int dsir_t_gemm(void* stream, const float* X, int ldx, const float* W, int wn, int wk, const float* bias, float* Y, int ldy,
                int64_t rows, int K, int N, float beta) {
  if (!X || !W || !Y || rows < 1 || K < 1 || N < 1) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((rows + TB - 1) / TB), (unsigned)((N + TB - 1) / TB));
  hipLaunchKernelGGL(t_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, wn, wk, bias, Y, ldy, rows, K, N, beta);
  return done();
}

size_t dsir_t_gemm_dw_scratch(int64_t rows, int N, int K) {
  // dsir_t_gemm_dw plans with or without the bias column (db == NULL): one tile column fewer can mean a larger split cap,
  // i.e. MORE partial matrices than the other plan - size for the larger of the two
  const DwPlan pb = dw_plan(rows, N, K, true), pn = dw_plan(rows, N, K, false);
  const size_t a = (size_t)pb.splits * N * pb.Kp, b = (size_t)pn.splits * N * pn.Kp;
  return (a > b ? a : b) * sizeof(float);
}

int dsir_t_gemm_dw(void* stream, const float* dY, int ldy, const float* X, int ldx, int64_t rows, int N, int K, float* dW, float* db,
                   void* scratch) {
  if (!dY || !X || !dW || !scratch || rows < 1 || K < 1 || N < 1) return (int)hipErrorInvalidValue;
  const DwPlan p = dw_plan(rows, N, K, db != nullptr);
  float* partial = reinterpret_cast<float*>(scratch);
  const dim3 grid((unsigned)((N + TB - 1) / TB), (unsigned)((p.Kp + TB - 1) / TB), (unsigned)p.splits);
  hipLaunchKernelGGL(t_gemm_dw_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, ldy, X, ldx, rows, p.rows_per_split, N, K, p.Kp,
                     partial);
  hipLaunchKernelGGL(t_gemm_dw_reduce_kernel, dim3((unsigned)(((int64_t)N * p.Kp + 31) / 32)), dim3(256), 0, (hipStream_t)stream, partial, p.splits, N, K,
                     p.Kp, dW, db);
  return done();
}

size_t dsir_t_gn_scratch(int clouds, int M, int C) {
  const size_t nch = (size_t)gn_chunks(M);
  return ((size_t)clouds * nch * C * 2 + (size_t)clouds * C * 2) * sizeof(double);
}

int dsir_t_gn_fwd(void* stream, const float* Y, int clouds, int M, int C, int groups, const float* gamma, const float* beta, int act,
                  float* out, float* stats, void* scratch) {
  if (!Y || !gamma || !beta || !out || !stats || !scratch || clouds < 1 || M < 1 || C < 1 || groups < 1 || C % groups)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int nch = gn_chunks(M), rpc = (M + nch - 1) / nch;
  double* partial = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(t_gn_stats_kernel, dim3(nch, groups, clouds), dim3(256), 0, st, Y, M, C, groups, rpc, partial);
  hipLaunchKernelGGL(t_gn_stats_final_kernel, dim3((clouds * groups + 31) / 32), dim3(256), 0, st, partial, nch, clouds * groups,
                     1.0 / ((double)M * (C / groups)), stats);
  const int64_t total = (int64_t)clouds * M * C;
  hipLaunchKernelGGL(t_gn_apply_kernel, dim3(grid1((C & 3) == 0 ? total / 4 : total)), dim3(256), 0, st, Y, stats, M, C, groups, gamma, beta, act, out, total);
  return done();
}

int dsir_t_gn_bwd(void* stream, const float* dOut, const float* Y, const float* stats, int clouds, int M, int C, int groups,
                  const float* gamma, const float* beta, int act, float* dY, float* dgamma, float* dbeta, void* scratch) {
  if (!dOut || !Y || !stats || !gamma || !beta || !dY || !dgamma || !dbeta || !scratch || clouds < 1 || M < 1 || C < 1 || groups < 1 ||
      C % groups || groups > 4096)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int nch = gn_chunks(M), rpc = (M + nch - 1) / nch;
  float* partial = reinterpret_cast<float*>(scratch);
  float* sums = partial + (size_t)clouds * nch * C * 2;
  hipLaunchKernelGGL(t_gn_bwd_sums_kernel, dim3((C + 31) / 32, nch, clouds), dim3(256), 0, st, dOut, Y, stats, M, C, groups, gamma, beta, act,
                     rpc, partial);
  const int64_t cnt = (int64_t)clouds * C * 2;
  hipLaunchKernelGGL(t_gn_bwd_sums_final_kernel, dim3((unsigned)((cnt + 31) / 32)), dim3(256), 0, st, partial, nch, C, cnt, sums);
  // elements per block: 16 k on the large layers, down to 4 k where that is needed for ~1000 workgroups (a block first forms the
  // group means from `sums`: C loads - not less than that many elements per thread-block pass)
  const int64_t total = (int64_t)clouds * M * C;
  int64_t per = total / 1024;
  per = per < 4096 ? 4096 : (per > 16384 ? 16384 : per);
  int rpb = (int)((per + C - 1) / C);
  rpb = rpb < 1 ? 1 : rpb;
  hipLaunchKernelGGL(t_gn_bwd_apply_kernel, dim3((M + rpb - 1) / rpb, clouds), dim3(256), (size_t)groups * 2 * sizeof(float), st, dOut, Y,
                     stats, sums, M, C, groups, gamma, beta, act, dY, rpb, dgamma, dbeta);
  return done();
}

int dsir_t_gather(void* stream, const float* X, int n, int C, const int32_t* idx, int m, int clouds, float* Y, int ldy, int col_off) {
  if (!X || !idx || !Y || n < 1 || C < 1 || m < 1 || clouds < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)clouds * m * C;
  hipLaunchKernelGGL(t_gather_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, X, n, C, idx, m, Y, ldy, col_off, total);
  return done();
}

size_t dsir_t_scatter_plan_scratch(int64_t total) { return total > 0 && total <= 0x7fffffffll ? scatter_plan_scratch_bytes(total) : 0; }

int dsir_t_scatter_plan(void* stream, const int32_t* idx, int m, int clouds, int n, int32_t* order, int32_t* offsets, void* scratch) {
  if (!idx || !order || !offsets || !scratch || m < 1 || clouds < 1 || n < 1) return (int)hipErrorInvalidValue;
  if (int r = launch_scatter_plan(idx, m, clouds, n, order, offsets, scratch, (hipStream_t)stream)) return r;
  return done();
}

int dsir_t_scatter_add(void* stream, const float* dY, int ldy, int col_off, const int32_t* order, const int32_t* offsets, int clouds,
                       float* dX, int n, int C) {
  if (!dY || !order || !offsets || !dX || n < 1 || C < 1 || clouds < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)clouds * n * C;
  hipLaunchKernelGGL(t_scatter_sum_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, dY, ldy, col_off, order, offsets, dX, C, total);
  return done();
}

int dsir_t_relpos(void* stream, const float* xyz, const int32_t* idx, int n, int k, int clouds, float* out) {
  if (!xyz || !idx || !out || n < 1 || k < 1 || clouds < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)clouds * n * k;
  hipLaunchKernelGGL(t_relpos_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, xyz, idx, n, k, out, total);
  return done();
}

int dsir_t_inlier_input(void* stream, const float* xyz_src, const float* xyz_ref, const int32_t* idx, const float* T, int t_stride,
                        int pairs, int J, int K, float* out) {
  if (!xyz_src || !xyz_ref || !idx || !out || pairs < 1 || J < 1 || K < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)pairs * J;
  hipLaunchKernelGGL(t_inlier_input_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, xyz_src, xyz_ref, idx, T, t_stride, J, K,
                     out, total);
  return done();
}

int dsir_t_attpool_fwd(void* stream, const float* cat, float* S, int64_t points, int k, int C, float* out) {
  if (!cat || !S || !out || points < 1 || k < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_attpool_fwd_kernel, dim3(grid1(points * C)), dim3(256), 0, (hipStream_t)stream, cat, S, k, C, out, points * C);
  return done();
}

int dsir_t_attpool_bwd(void* stream, const float* dOut, const float* cat, const float* A, int64_t points, int k, int C, float* dCat,
                       float* dS) {
  if (!dOut || !cat || !A || !dCat || !dS || points < 1 || k < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_attpool_bwd_kernel, dim3(grid1(points * C)), dim3(256), 0, (hipStream_t)stream, dOut, cat, A, k, C, dCat, dS,
                     points * C);
  return done();
}

int dsir_t_maxpool_fwd(void* stream, const float* X, int n, int C, const int32_t* pool, int m, int k, int clouds, float* out,
                       int32_t* arg) {
  if (!X || !pool || !out || !arg || n < 1 || C < 1 || m < 1 || k < 1 || clouds < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)clouds * m * C;
  hipLaunchKernelGGL(t_maxpool_fwd_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, X, n, C, pool, m, k, out, arg, total);
  return done();
}

int dsir_t_maxpool_bwd(void* stream, const float* dOut, const int32_t* arg, const int32_t* order, const int32_t* offsets, int m, int k, int C,
                       int clouds, float* dX, int n) {
  if (!dOut || !arg || !order || !offsets || !dX || n < 1 || C < 1 || m < 1 || k < 1 || clouds < 1) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)clouds * n * C;
  hipLaunchKernelGGL(t_maxpool_bwd_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, dOut, arg, order, offsets, m, k, C, dX, n, total);
  return done();
}

int dsir_t_bn_running(void* stream, const float* stats, int C, int64_t M, float momentum, float* running_mean, float* running_var) {
  if (!stats || !running_mean || !running_var || C < 1 || M < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, C, (double)M, momentum,
                     running_mean, running_var);
  return done();
}

int dsir_t_l2norm_fwd(void* stream, const float* x, int64_t rows, int C, float* y, float* norms) {
  if (!x || !y || !norms || rows < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_l2norm_fwd_kernel, dim3(grid1(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, C, y, norms);
  return done();
}

int dsir_t_l2norm_bwd(void* stream, const float* dy, const float* y, const float* norms, int64_t rows, int C, float* dx) {
  if (!dy || !y || !norms || !dx || rows < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_l2norm_bwd_kernel, dim3(grid1(rows)), dim3(256), 0, (hipStream_t)stream, dy, y, norms, rows, C, dx);
  return done();
}

size_t dsir_t_det_des_loss_scratch(int pairs, int M) {
  const size_t mm = (size_t)pairs * M * M, pm = (size_t)pairs * M;
  return (5 * mm + pm * 8 + pm * 8) * sizeof(float) + pm * sizeof(CircleRow) + 256;
}

int dsir_t_det_des_loss(void* stream, const float* feat_ref, const float* feat_src, const float* pt_ref, const float* pt_src,
                        const float* score_ref, const float* transform_gt, int pairs, int M, int C, float thres_radius, float det_loss_weight,
                        double* out, float* d_feat_ref, float* d_feat_src, void* scratch) {
  if (!feat_ref || !feat_src || !pt_ref || !pt_src || !score_ref || !transform_gt || !out || !d_feat_ref || !d_feat_src || !scratch ||
      pairs < 1 || M < 1 || C < 1 || !(thres_radius > 0.f))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const size_t mm = (size_t)pairs * M * M, pm = (size_t)pairs * M;
  float* dot = reinterpret_cast<float*>(scratch);
  float* dist_pc = dot + mm; float* dist_feat = dist_pc + mm; float* Wn = dist_feat + mm; float* Wnt = Wn + mm;
  float* lse_col = Wnt + mm; float* coef = lse_col + pm; float* rowsum = coef + 3 * pm; float* colsum = rowsum + pm;
  CircleRow* rows = reinterpret_cast<CircleRow*>(colsum + pm + (((uintptr_t)(colsum + pm) & 7) ? 1 : 0));
  for (int p = 0; p < pairs; ++p)       // dot[i][j] = anc_i . pos_j: the GEMM of square_distance_V2 (matchnet.py:110)
    hipLaunchKernelGGL(t_gemm_kernel, dim3((M + TB - 1) / TB, (M + TB - 1) / TB), dim3(256), 0, st, feat_ref + (size_t)p * M * C, C,
                       feat_src + (size_t)p * M * C, C, 1, (const float*)nullptr, dot + (size_t)p * M * M, M, (int64_t)M, C, M, 0.f);
  hipLaunchKernelGGL(t_circle_dist_kernel, dim3(grid1((int64_t)mm)), dim3(256), 0, st, dot, feat_ref, feat_src, pt_ref, pt_src, transform_gt, M,
                     C, dist_pc, dist_feat, (int64_t)mm);
  hipLaunchKernelGGL(t_circle_rows_kernel, dim3(M, pairs), dim3(256), 0, st, dist_pc, dist_feat, M, thres_radius, rows);
  hipLaunchKernelGGL(t_circle_cols_kernel, dim3(M, pairs), dim3(256), 0, st, dist_pc, dist_feat, M, thres_radius, rows, lse_col);
  hipLaunchKernelGGL(t_circle_final_kernel, dim3(1), dim3(256), 0, st, rows, lse_col, score_ref, pairs, M, det_loss_weight, coef, out);
  hipLaunchKernelGGL(t_circle_grad_kernel, dim3(M, pairs), dim3(256), 0, st, dist_pc, dist_feat, M, thres_radius, rows, lse_col, coef, Wn, Wnt,
                     rowsum);
  hipLaunchKernelGGL(t_rowsum_neg_kernel, dim3(M, pairs), dim3(256), 0, st, Wnt, M, colsum);
  // d anc_i = rowsum_i anc_i + sum_j Wn_ij pos_j;  d pos_j = colsum_j pos_j + sum_i Wn_ij anc_i
  hipLaunchKernelGGL(t_scale_rows_kernel, dim3(grid1((int64_t)pm * C)), dim3(256), 0, st, rowsum, feat_ref, C, d_feat_ref, (int64_t)pm * C);
  hipLaunchKernelGGL(t_scale_rows_kernel, dim3(grid1((int64_t)pm * C)), dim3(256), 0, st, colsum, feat_src, C, d_feat_src, (int64_t)pm * C);
  for (int p = 0; p < pairs; ++p) {
    hipLaunchKernelGGL(t_gemm_kernel, dim3((M + TB - 1) / TB, (C + TB - 1) / TB), dim3(256), 0, st, Wn + (size_t)p * M * M, M,
                       feat_src + (size_t)p * M * C, 1, C, (const float*)nullptr, d_feat_ref + (size_t)p * M * C, C, (int64_t)M, M, C, 1.f);
    hipLaunchKernelGGL(t_gemm_kernel, dim3((M + TB - 1) / TB, (C + TB - 1) / TB), dim3(256), 0, st, Wnt + (size_t)p * M * M, M,
                       feat_ref + (size_t)p * M * C, 1, C, (const float*)nullptr, d_feat_src + (size_t)p * M * C, C, (int64_t)M, M, C, 1.f);
  }
  return done();
}

size_t dsir_t_weighted_ce_scratch(int64_t rows) { return (size_t)((rows + 255) / 256) * 4 * sizeof(double); }

int dsir_t_weighted_ce(void* stream, const float* logits, const int32_t* labels, const float* class_weights, int64_t rows, int C,
                       float grad_scale, float* dlogits, double* out, void* scratch) {
  if (!logits || !labels || !class_weights || !dlogits || !out || !scratch || rows < 1 || C < 1) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((rows + 255) / 256);
  double* partial = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(t_wce_rows_kernel, dim3(blocks), dim3(256), 0, st, logits, labels, class_weights, rows, C, dlogits, partial);
  hipLaunchKernelGGL(t_wce_final_kernel, dim3(1), dim3(256), 0, st, partial, blocks, out);
  hipLaunchKernelGGL(t_wce_scale_kernel, dim3(grid1(rows * C)), dim3(256), 0, st, dlogits, rows * C, out, grad_scale);
  return done();
}

int dsir_t_add_leaky_fwd(void* stream, const float* a, const float* b, int64_t n, float* out) {
  if (!a || !b || !out || n < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_add_leaky_fwd_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
  return done();
}

int dsir_t_add_leaky_bwd(void* stream, const float* dOut, const float* out, int64_t n, float* d) {
  if (!dOut || !out || !d || n < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_add_leaky_bwd_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, dOut, out, n, d);
  return done();
}

int dsir_t_mul_mask(void* stream, const float* x, const uint8_t* mask, float scale, int64_t n, float* y) {
  if (!x || !mask || !y || n < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_mul_mask_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, n, y);
  return done();
}

size_t dsir_t_topk_scratch(int clouds, int n) { return topk_scratch_bytes(clouds, n); }

int dsir_t_topk(void* stream, const float* score, int clouds, int n, int k, int32_t* idx, float* score_out, void* scratch) {
  if (!score || !idx || !score_out || !scratch || clouds < 1 || n < 1 || k < 1 || k > n) return (int)hipErrorInvalidValue;
  if (int r = launch_topk(score, clouds, n, k, idx, score_out, scratch, (hipStream_t)stream)) return r;
  return done();
}

int dsir_t_sigmoid(void* stream, const float* x, int64_t n, float* y) {
  if (!x || !y || n < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_sigmoid_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
  return done();
}

int dsir_t_axpy(void* stream, float a, const float* x, int64_t n, float* y) {
  if (!x || !y || n < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(t_axpy_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, a, x, n, y);
  return done();
}

int dsir_t_any_nan(void* stream, const float* x, int64_t n, int32_t* flag) {
  if (!x || !flag || n < 1) return (int)hipErrorInvalidValue;
  (void)hipMemsetAsync(flag, 0, sizeof(int32_t), (hipStream_t)stream);
  hipLaunchKernelGGL(t_any_nan_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, x, n, flag);
  return done();
}

int dsir_t_adam(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int step) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return (int)hipErrorInvalidValue;
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  hipLaunchKernelGGL(t_adam_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, b1, b2, eps, (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)));
  return done();
}

}  // extern "C"
