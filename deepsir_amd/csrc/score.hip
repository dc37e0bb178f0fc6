// Saliency score of every point (reference network/model.py:638-639 torch.max(logits),
// :701-757 score_fun).  Two launches:
//   1. per-point (prob,label) = max/argmax of the semantic logits + three
//      per-cloud maxima (feature max, label-weight max, prob max) via wave
//      reductions and order-preserving integer atomics;
//   2. 16 lanes per point, four feature channels each: neighbour mean of the
//      normalised feature (16-byte pieces of 16 gathered 256-B rows), softplus saliency,
//      density gate, channel-max ratio, semantic weight, max over channels.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

__constant__ float c_label_weights[32] = {3, 1, 1, 3, 2, 0, 0, 0, 6, 5, 6, 4, 7, 7, 6, 8, 4, 9, 9,
                                          0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // model.py:146-149

constexpr float kEps = 1e-16f;  // model.py:18

__global__ __launch_bounds__(256) void score_reduce_kernel(const float* __restrict__ feat, const float* __restrict__ logits,
                                                           int ncls, int n, ScoreScratch s) {
  const int cloud = blockIdx.y;
  const float* F = feat + (int64_t)cloud * n * 64;
  const float* L = logits + (int64_t)cloud * n * ncls;
  float fmx = -INFINITY, lmx = -INFINITY, pmx = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float best = L[(int64_t)i * ncls];
    int arg = 0;
    for (int c = 1; c < ncls; ++c) {
      const float v = L[(int64_t)i * ncls + c];
      if (v > best) { best = v; arg = c; }   // first maximum wins, as torch.max on CPU
    }
    s.prob[(int64_t)cloud * n + i] = best;
    s.label[(int64_t)cloud * n + i] = arg;
    pmx = fmaxf(pmx, best);
    lmx = fmaxf(lmx, c_label_weights[arg]);
  }
  const int64_t total = (int64_t)n * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    fmx = fmaxf(fmx, F[e]);
  fmx = wave_max(fmx); lmx = wave_max(lmx); pmx = wave_max(pmx);
  if ((threadIdx.x & 63) == 0) {
    float* r = s.red + cloud * 4;
    atomic_max_float(r + 0, fmx);
    atomic_max_float(r + 1, lmx);
    atomic_max_float(r + 2, pmx);
  }
}

__device__ __forceinline__ float softplus(float x) {  // F.softplus, beta = 1, threshold = 20
  return x > 20.f ? x : log1pf(expf(x));
}

// 16 lanes per point, four channels each (round 4; one wave per point before: 64 lanes x one channel, every reduction over
// the channels a 6-step wave butterfly and every gather a dword).  A lane gathers 16-byte pieces of the 17 feature rows, the
// reductions over channels are an in-lane step plus four shuffles inside the 16-lane group; four points per wave.
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
  return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void score_point_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                          int64_t xyz_cs, const int32_t* __restrict__ neigh,
                                                          int64_t neigh_cs, int n, ScoreScratch s,
                                                          float* __restrict__ score, int32_t* __restrict__ label_out, int bpc) {
  // 1-D grid of bpc workgroups per cloud, XCD-aware: every XCD takes whole clouds in turn, so the 17 feature rows a point gathers
  // (256 B each, 1.28 MB per cloud) stay in ONE L2.  Round 3 dealt a cloud's workgroups over all eight XCDs: 2.5 GB of HBM-side
  // fetches per 256-cloud launch for 0.33 GB of features.
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  const int cloud = wi / bpc;
  const int l = threadIdx.x & 15;
  const int i = (wi % bpc) * 16 + (threadIdx.x >> 4);
  const bool ok = i < n;
  const int ic = ok ? i : n - 1;                       // clamped: all lanes of a wave take part in the group shuffles
  const float* F = feat + (int64_t)cloud * n * 64;
  const float* X = xyz + cloud * xyz_cs;
  const int32_t* nb = neigh + cloud * neigh_cs + (int64_t)ic * kKnn;
  const float* red = s.red + cloud * 4;
  const float fden = red[0] + kEps;
  // 1. saliency.  The reference divides every gathered feature by the cloud's maximum and then averages; the division is
  // linear, so the 16 neighbour rows are summed first and divided once (17 correctly rounded divisions per channel made the kernel
  // instruction-bound; the mean differs from the divide-then-sum order by ~1e-7 of its value)
  const int4* ip = reinterpret_cast<const int4*>(nb);
  int nbk[kKnn];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int4 v = ip[q]; nbk[4 * q] = v.x; nbk[4 * q + 1] = v.y; nbk[4 * q + 2] = v.z; nbk[4 * q + 3] = v.w; }
  const float4 own = *reinterpret_cast<const float4*>(F + (uint32_t)ic * 64u + 4u * (uint32_t)l);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < kKnn; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(F + (uint32_t)nbk[k] * 64u + 4u * (uint32_t)l);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float fn[4] = {own.x / fden, own.y / fden, own.z / fden, own.w / fden};
  const float mean[4] = {(acc.x * 0.0625f) / fden, (acc.y * 0.0625f) / fden, (acc.z * 0.0625f) / fden, (acc.w * 0.0625f) / fden};
  // 2. density gate: mean neighbour distance < 2.0 (lane l of the group takes neighbour l)
  float dist;
  {
    int jl = nbk[0];
#pragma unroll
    for (int k = 1; k < kKnn; ++k) jl = l == k ? nbk[k] : jl;
    const float dx = __fsub_rn(X[(int64_t)jl * 3], X[(int64_t)ic * 3]);
    const float dy = __fsub_rn(X[(int64_t)jl * 3 + 1], X[(int64_t)ic * 3 + 1]);
    const float dz = __fsub_rn(X[(int64_t)jl * 3 + 2], X[(int64_t)ic * 3 + 2]);
    dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  }
  const float gate = (group16_sum(dist) / 16.f < 2.0f) ? 1.f : 0.f;
  // 3. channel-wise max ratio
  const float fmx = group16_max(fmaxf(fmaxf(fn[0], fn[1]), fmaxf(fn[2], fn[3]))) + kEps;
  // 4. semantic weight
  const int lab = s.label[(int64_t)cloud * n + ic];
  float ls = c_label_weights[lab] / (red[1] + kEps);
  const float pr = s.prob[(int64_t)cloud * n + ic] / (red[2] + kEps);
  ls = ls * (pr > 0.2f ? 1.f : 0.f);
  // 5. total, max over channels
  float v = -INFINITY;
#pragma unroll
  for (int c = 0; c < 4; ++c) v = fmaxf(v, ((softplus(fn[c] - mean[c]) * gate) * (fn[c] / fmx)) * ls);
  v = group16_max(v);
  if (ok && l == 0) {
    score[(int64_t)cloud * n + i] = v;
    if (label_out) label_out[(int64_t)cloud * n + i] = lab;
  }
}

}  // namespace

void launch_score(const float* feat, const float* logits, int ncls, const float* xyz, int64_t xyz_cs,
                  const int32_t* neigh, int64_t neigh_cs, int clouds, int n, ScoreScratch s, float* score,
                  int32_t* label_out, hipStream_t st, bool red_preset) {
  if (!red_preset) hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s.red), 0xff800000u, (size_t)clouds * 4, st);  // -inf
  int gx = (n + 255) / 256;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(score_reduce_kernel, dim3(gx, clouds), dim3(256), 0, st, feat, logits, ncls, n, s);
  const int bpc = (n + 15) / 16;
  hipLaunchKernelGGL(score_point_kernel, dim3((unsigned)((int64_t)bpc * clouds)), dim3(256), 0, st, feat, xyz, xyz_cs, neigh, neigh_cs,
                     n, s, score, label_out, bpc);
}

}  // namespace dsir
