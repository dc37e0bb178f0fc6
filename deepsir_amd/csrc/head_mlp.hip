// Fused per-point head of RandLA.forward (reference network/RandLANet.py:363-367):
//   feat   = mlp_out(x)              32 -> 64, no bias, no norm        (x = last decoder block, GroupNorm+LeakyReLU lazy)
//   logits = fc_label(feat)          64 -> 64 -> 32 -> ncls, eval-BatchNorm folded, LeakyReLU(0.2) between
// Four row-wise GEMMs whose 64/64/32-wide intermediates never leave the CU: a wave owns 16-row tiles; every
// layer is a burst of exact-fp32 MFMAs with the layer's weight fragments held in registers for the whole kernel;
// between layers the 16 x C accumulator tile is transposed through a wave-private LDS tile (C layout ->
// row-per-lane A fragments).  The k order of every layer is the one pw_stream.hip uses for the same shapes
// (lane (r,q) holds channels [q*C/4,(q+1)*C/4) of row r), so the results are bit-identical to the four
// separate launches this kernel replaces, at ~1/4 of the HBM traffic and 1/4 of the launches.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int LDT = 64 + 4;   // transposition tile row (floats): rows shift by 4 banks

template <int K>
__device__ __forceinline__ void load_wrow(const float* __restrict__ W, int col, int ncols, int ld, int c_lo, float (&w)[K]) {
  if (col < ncols) {
#pragma unroll
    for (int i = 0; i < K / 4; ++i) {
      const float4 t = *reinterpret_cast<const float4*>(W + (int64_t)col * ld + c_lo + 4 * i);
      w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) w[i] = 0.f;
  }
}

// C-layout accumulators (col = lane & 15, row = 4*(lane>>4)+reg) -> T[row][col]; then lane (r,q) reads K
// contiguous channels of row r starting at q*K
template <int NTL, int K>
__device__ __forceinline__ void transpose_tile(float* T, const f32x4 (&acc)[NTL], int fr, int fq, float (&a)[K]) {
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(4 * fq + r) * LDT + 16 * t + fr] = acc[t][r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < K / 4; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(&T[fr * LDT + K * fq + 4 * i]);
    a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
  }
}

template <int NT4>
__global__ __launch_bounds__(256) void head_mlp_kernel(const HeadArgs p) {
  __shared__ float s_sc[32], s_sh[32];
  __shared__ float s_T[4][16 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cloud = blockIdx.y;

  if (tid < 32) {
    float scale = 1.f, shift = 0.f;
    const Seg& s = p.in;
    if (s.gn.stats) {
      const int g = tid / (32 / s.gn.groups);
      const double* st = s.gn.stats + ((int64_t)cloud * s.gn.groups + g) * kGnWords;
      const double mean = gn_stat_get(st) * s.gn.inv_count;
      double var = gn_stat_get(st + 2) * s.gn.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = gn_rstd(var);
      const double scd = (double)s.gn.gamma[tid] * rstd;
      scale = (float)scd;
      shift = (float)((double)s.gn.beta[tid] - mean * scd);
    }
    s_sc[tid] = scale;
    s_sh[tid] = shift;
  }
  __syncthreads();

  // weight fragments of the four layers (k order: lane q holds the contiguous slice q*K .. q*K+K-1)
  float w1[4][8], w2[4][16], w3[2][16], w4[NT4][8];
  float b2[4], b3[2], b4[NT4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    load_wrow<8>(p.W1, 16 * t + fr, 64, 32, 8 * fq, w1[t]);
    load_wrow<16>(p.W2, 16 * t + fr, 64, 64, 16 * fq, w2[t]);
    b2[t] = p.b2[16 * t + fr];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    load_wrow<16>(p.W3, 16 * t + fr, 32, 64, 16 * fq, w3[t]);
    b3[t] = p.b3[16 * t + fr];
  }
#pragma unroll
  for (int t = 0; t < NT4; ++t) {
    load_wrow<8>(p.W4, 16 * t + fr, p.ncls, 32, 8 * fq, w4[t]);
    b4[t] = (16 * t + fr) < p.ncls ? p.b4[16 * t + fr] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = s_sc[8 * fq + j]; sh[j] = s_sh[8 * fq + j]; }
  const int act = p.in.act;
  float* T = s_T[w];
  const float* X = p.in.x + cloud * p.in.cloud_stride;
  float* feat = p.feat_out ? p.feat_out + (int64_t)cloud * p.M * 64 : nullptr;
  float* logit = p.logits_out + (int64_t)cloud * p.M * p.ncls;

  const int ntiles = (p.M + 15) >> 4;
  const int nwaves = gridDim.x * 4;
  float a0[8], a0n[8];
  auto load_a0 = [&](int tile, float (&a)[8]) {
    const int row = min(tile * 16 + fr, p.M - 1);        // clamped, not predicated (results of padded rows are dropped)
    const float* src = X + (int64_t)row * p.in.ld + 8 * fq;
    const float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
    a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w; a[4] = v.x; a[5] = v.y; a[6] = v.z; a[7] = v.w;
  };
  int tile = blockIdx.x * 4 + w;
  if (tile < ntiles) load_a0(tile, a0);
  for (; tile < ntiles; tile += nwaves) {
    if (tile + nwaves < ntiles) load_a0(tile + nwaves, a0n);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = fmaf(a0[j], sc[j], sh[j]);
      a0[j] = (act && v < 0.f) ? 0.2f * v : v;
    }
    const int rbase = tile * 16 + 4 * fq;
    // ---- mlp_out: 32 -> 64 (no bias)
    f32x4 c1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) c1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], w1[t][s], c1[t], 0, 0, 0);
    if (feat) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rbase + r < p.M) feat[(int64_t)(rbase + r) * 64 + 16 * t + fr] = c1[t][r];
    }
    float a1[16];
    transpose_tile<4, 16>(T, c1, fr, fq, a1);
    // ---- fc_label.0 (+ folded BN) : 64 -> 64, LeakyReLU
    f32x4 c2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) c2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) c2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], w2[t][s], c2[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float v = c2[t][r] + b2[t]; c2[t][r] = v < 0.f ? v * 0.2f : v; }
    float a2[16];
    transpose_tile<4, 16>(T, c2, fr, fq, a2);
    // ---- fc_label.3 (+ folded BN) : 64 -> 32, LeakyReLU
    f32x4 c3[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) c3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) c3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], w3[t][s], c3[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float v = c3[t][r] + b3[t]; c3[t][r] = v < 0.f ? v * 0.2f : v; }
    float a3[8];
    transpose_tile<2, 8>(T, c3, fr, fq, a3);
    // ---- fc_label.6 : 32 -> ncls
    f32x4 c4[NT4];
#pragma unroll
    for (int t = 0; t < NT4; ++t) c4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < NT4; ++t) c4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[s], w4[t][s], c4[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      const int col = 16 * t + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rbase + r < p.M && col < p.ncls) logit[(int64_t)(rbase + r) * p.ncls + col] = c4[t][r] + b4[t];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a0n[j];
  }
}

}  // namespace

bool launch_head_mlp(const HeadArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.clouds <= 0) return true;
  if (a.in.C != 32 || (a.in.ld % 4) != 0 || (a.in.cloud_stride % 4) != 0 || a.in.idx) return false;
  if (a.ncls < 1 || a.ncls > 32 || !a.logits_out) return false;
  if ((reinterpret_cast<uintptr_t>(a.in.x) | reinterpret_cast<uintptr_t>(a.W1) | reinterpret_cast<uintptr_t>(a.W2) |
       reinterpret_cast<uintptr_t>(a.W3) | reinterpret_cast<uintptr_t>(a.W4)) % 16) return false;
  const int ntiles = (a.M + 15) / 16;
  int blocks = (ntiles + 31) / 32;              // ~8 tiles per wave: the 144 weight registers are loaded once per wave
  if (blocks < 16) blocks = (ntiles + 3) / 4 < 16 ? (ntiles + 3) / 4 : 16;
  if (blocks < 1) blocks = 1;
  dim3 grid(blocks, a.clouds);
  if (a.ncls <= 16) hipLaunchKernelGGL((head_mlp_kernel<1>), grid, dim3(256), 0, st, a);
  else              hipLaunchKernelGGL((head_mlp_kernel<2>), grid, dim3(256), 0, st, a);
  return true;
}

}  // namespace dsir
