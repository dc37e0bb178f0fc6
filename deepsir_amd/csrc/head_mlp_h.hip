// head_mlp_kernel (head_mlp.hip: mlp_out + fc_label, 32 -> 64 -> 64 -> 32 -> ncls per point, RandLANet.py:363-367) on the
// fp16 matrix pipe at fp32 accuracy, by the operand split of agg_chain_h.hip:
//       x = xh + xl + r,  xh = fp16(x),  xl = fp16(x - xh);      a.b = ah.bh + ah.bl + al.bh  (three v_mfma_f32_16x16x32_f16)
// The fp32 kernel issues 136 - 144 v_mfma_f32_16x16x4_f32 per 16 points (4.4 - 4.6 k cycles of the matrix pipe: the kernel
// is bound by it); this one 51 - 54 fp16 MFMAs (0.8 k cycles) plus ~6 VALU instructions per activation for the split.
// Weights are split once at load (dsir_finalize_weights) and held as B fragments in registers for the whole kernel, as in
// the fp32 kernel; between layers the accumulator tile is transposed through a wave-private LDS tile and split when it is
// read back as the next layer's A fragments (lane (fr, fq): row fr, channels 8 fq .. + 7 of each 32-channel k-step).
// Results are within ~1e-6 (relative to the layer's scale) of head_mlp_kernel's; that kernel (exact fp32, bit-identical to
// the four separate launches) stays as the reference: dsir_enable_agg_split(0) / DSIR_AGG_F32=1 select it.
#include <hip/hip_fp16.h>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int LDT = 64 + 4;   // transposition tile row (floats): 16-byte aligned rows, conflict-free both ways

__device__ __forceinline__ void split8(const float4 u, const float4 v, h8& h, h8& l) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 t = (_Float16)f[k];
    h[k] = t;
    l[k] = (_Float16)(f[k] - (float)t);
  }
}

// B fragment of weight row `col` (zero past ncols): channels 32 s + 8 fq .. + 7
__device__ __forceinline__ void load_w(const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl, int col, int ncols, int ld, int k0,
                                       h8& h, h8& l) {
  if (col < ncols) {
    h = *reinterpret_cast<const h8*>(Wh + (int64_t)col * ld + k0);
    l = *reinterpret_cast<const h8*>(Wl + (int64_t)col * ld + k0);
  } else {
    h = h8{0, 0, 0, 0, 0, 0, 0, 0};
    l = h8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

#define DSIR_MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)
#define DSIR_MMA3(acc, ah, al, bh, bl) \
  do { DSIR_MFMA16(acc, al, bh); DSIR_MFMA16(acc, ah, bl); DSIR_MFMA16(acc, ah, bh); } while (0)

template <int NT4>
__global__ __launch_bounds__(256) void head_mlp_h_kernel(const HeadArgs p) {
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

  const _Float16* const* Wh = reinterpret_cast<const _Float16* const*>(p.Wh);
  const _Float16* const* Wl = reinterpret_cast<const _Float16* const*>(p.Wl);
  h8 w1h[4], w1l[4], w2h[4][2], w2l[4][2], w3h[2][2], w3l[2][2], w4h[NT4], w4l[NT4];
  float b2[4], b3[2], b4[NT4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    load_w(Wh[0], Wl[0], 16 * t + fr, 64, 32, 8 * fq, w1h[t], w1l[t]);
#pragma unroll
    for (int s = 0; s < 2; ++s) load_w(Wh[1], Wl[1], 16 * t + fr, 64, 64, 32 * s + 8 * fq, w2h[t][s], w2l[t][s]);
    b2[t] = p.b2[16 * t + fr];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int s = 0; s < 2; ++s) load_w(Wh[2], Wl[2], 16 * t + fr, 32, 64, 32 * s + 8 * fq, w3h[t][s], w3l[t][s]);
    b3[t] = p.b3[16 * t + fr];
  }
#pragma unroll
  for (int t = 0; t < NT4; ++t) {
    load_w(Wh[3], Wl[3], 16 * t + fr, p.ncls, 32, 8 * fq, w4h[t], w4l[t]);
    b4[t] = (16 * t + fr) < p.ncls ? p.b4[16 * t + fr] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = s_sc[8 * fq + j]; sh[j] = s_sh[8 * fq + j]; }
  const float slope = p.in.act ? 0.2f : 1.f;
  float* T = s_T[w];
  const float* X = p.in.x + cloud * p.in.cloud_stride;
  float* feat = p.feat_out ? p.feat_out + (int64_t)cloud * p.M * 64 : nullptr;
  float* logit = p.logits_out + (int64_t)cloud * p.M * p.ncls;

  // C-layout accumulators -> T[row][col]; then lane (fr, fq) reads channels 8 fq .. + 7 (and 32 + 8 fq .. + 7) of row fr, split
  auto spill4 = [&](const f32x4 (&acc)[4]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * fq + r) * LDT + 16 * t + fr] = acc[t][r];
    __builtin_amdgcn_wave_barrier();
  };
  auto spill2 = [&](const f32x4 (&acc)[2]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * fq + r) * LDT + 16 * t + fr] = acc[t][r];
    __builtin_amdgcn_wave_barrier();
  };
  auto frag = [&](int s, h8& h, h8& l) {
    const float4* q = reinterpret_cast<const float4*>(T + fr * LDT + 32 * s + 8 * fq);
    split8(q[0], q[1], h, l);
  };

  const int ntiles = (p.M + 15) >> 4;
  const int nwaves = gridDim.x * 4;
  float4 x0, x1, x0n, x1n;
  auto load_a0 = [&](int tile, float4& u, float4& v) {
    const int row = min(tile * 16 + fr, p.M - 1);        // clamped, not predicated (results of padded rows are dropped)
    const float* src = X + (int64_t)row * p.in.ld + 8 * fq;
    u = *reinterpret_cast<const float4*>(src);
    v = *reinterpret_cast<const float4*>(src + 4);
  };
  int tile = blockIdx.x * 4 + w;
  if (tile < ntiles) load_a0(tile, x0, x1);
  for (; tile < ntiles; tile += nwaves) {
    if (tile + nwaves < ntiles) load_a0(tile + nwaves, x0n, x1n);
    h8 ah, al;
    {
      float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = fmaf(v[j], sc[j], sh[j]);
        v[j] = fmaxf(u, slope * u);
      }
      split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), ah, al);
    }
    const int rbase = tile * 16 + 4 * fq;
    // ---- mlp_out: 32 -> 64 (no bias)
    f32x4 c1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { c1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; DSIR_MMA3(c1[t], ah, al, w1h[t], w1l[t]); }
    if (feat) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rbase + r < p.M) feat[(int64_t)(rbase + r) * 64 + 16 * t + fr] = c1[t][r];
    }
    spill4(c1);
    h8 a1h[2], a1l[2];
    frag(0, a1h[0], a1l[0]); frag(1, a1h[1], a1l[1]);
    // ---- fc_label.0 (+ folded BN): 64 -> 64, LeakyReLU
    f32x4 c2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      c2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) DSIR_MMA3(c2[t], a1h[s], a1l[s], w2h[t][s], w2l[t][s]);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = c2[t][r] + b2[t]; c2[t][r] = v < 0.f ? v * 0.2f : v; }
    }
    spill4(c2);
    h8 a2h[2], a2l[2];
    frag(0, a2h[0], a2l[0]); frag(1, a2h[1], a2l[1]);
    // ---- fc_label.3 (+ folded BN): 64 -> 32, LeakyReLU
    f32x4 c3[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      c3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) DSIR_MMA3(c3[t], a2h[s], a2l[s], w3h[t][s], w3l[t][s]);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = c3[t][r] + b3[t]; c3[t][r] = v < 0.f ? v * 0.2f : v; }
    }
    spill2(c3);
    h8 a3h, a3l;
    frag(0, a3h, a3l);
    // ---- fc_label.6: 32 -> ncls
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
      DSIR_MMA3(c4, a3h, a3l, w4h[t], w4l[t]);
      const int col = 16 * t + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rbase + r < p.M && col < p.ncls) logit[(int64_t)(rbase + r) * p.ncls + col] = c4[r] + b4[t];
    }
    x0 = x0n; x1 = x1n;
  }
}

}  // namespace

bool launch_head_mlp_h(const HeadArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.clouds <= 0) return true;
  for (int k = 0; k < 4; ++k)
    if (!a.Wh[k] || !a.Wl[k]) return false;
  if (a.in.C != 32 || (a.in.ld % 4) != 0 || (a.in.cloud_stride % 4) != 0 || a.in.idx) return false;
  if (a.ncls < 1 || a.ncls > 32 || !a.logits_out) return false;
  if (reinterpret_cast<uintptr_t>(a.in.x) % 16) return false;
  const int ntiles = (a.M + 15) / 16;
  int blocks = (ntiles + 31) / 32;              // ~8 tiles per wave: the weight fragments are loaded once per wave
  if (blocks < 16) blocks = (ntiles + 3) / 4 < 16 ? (ntiles + 3) / 4 : 16;
  if (blocks < 1) blocks = 1;
  dim3 grid(blocks, a.clouds);
  if (a.ncls <= 16) hipLaunchKernelGGL((head_mlp_h_kernel<1>), grid, dim3(256), 0, st, a);
  else              hipLaunchKernelGGL((head_mlp_h_kernel<2>), grid, dim3(256), 0, st, a);
  return true;
}

}  // namespace dsir
