// The per-iteration half of Network.aggregation (reference network/model.py:223-233) in ONE launch:
//   g    = mlp_att([xyz ; score])        4 -> 32 -> 64 -> 128 -> 256 -> 64   (Conv1d + eval-BatchNorm folded + LeakyReLU 0.2,
//                                                                             last layer bare; RandLANet.py:34-55)
//   desc = normalize(mlp_proj(F + g))    64 -> 64, L2 over channels           (F = mlp_feat(feat0): loop invariant, hoisted)
// Unfused these are six launches whose 32..256-wide intermediates make a round trip through HBM (4.3 kB per point per
// iteration) and whose K loops are only 1..8 chunks long, so prologue / epilogue dominate.  Here a block owns 128
// points (4 waves x 2 row tiles of 16; 64 points with one row tile per wave when few clouds are in flight) and walks the whole chain:
//   * activations never leave the CU: a layer's accumulators (C layout) are transposed through a wave-private LDS tile
//     into the A fragments (registers) of the next layer; the 256-wide layer is produced 64 columns at a time and
//     consumed immediately as one K chunk of the following 256 -> 64 layer;
//   * weights stream through a double-buffered LDS tile of 64 columns x 64 channels shared by the four waves
//     (16 chunks per block: 224 kB from L2 per 128 points), the next chunk's global loads in flight during the MFMAs;
//   * exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), every layer in the SAME k order as the kernel it replaces
//     (pw_stream.hip for 4->32, 32->64 and 64->64, pw_tile.hip for the wide ones), so descriptors are bit-identical
//     to the unfused path.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int LDW = 66;   // weight tile row (floats): fragment reads (row r, k = 4 s + q) hit banks 2 r + q
constexpr int LDT = 66;   // transposition tile row

template <int RT>
__global__ __launch_bounds__(256, 2) void agg_chain_kernel(const AggArgs p) {
  __shared__ float Ws[2][64 * LDW];
  __shared__ float Ts[4][2][16 * LDT];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cloud = blockIdx.y;
  const int r0 = blockIdx.x * (64 * RT) + 16 * RT * w;      // first row of this wave

  // ---- weight chunk schedule
  const int srow = tid >> 4, sk = (tid & 15) * 4;
  float4 rw[4];
  auto chunk_src = [&](int i, const float*& W, int& ld, int& col0, int& k0, int& nk) {
    if (i == 0) { W = p.W2; ld = 32; col0 = 0; k0 = 0; nk = 32; }
    else if (i <= 2) { W = p.W3; ld = 64; col0 = 64 * (i - 1); k0 = 0; nk = 64; }
    else if (i == 15) { W = p.W6; ld = 64; col0 = 0; k0 = 0; nk = 64; }
    else {
      const int j = (i - 3) / 3, u = (i - 3) % 3;
      if (u < 2) { W = p.W4; ld = 128; col0 = 64 * j; k0 = 64 * u; nk = 64; }
      else { W = p.W5; ld = 256; col0 = 0; k0 = 64 * j; nk = 64; }
    }
  };
  auto gload = [&](int i) {
    const float* W; int ld, col0, k0, nk;
    chunk_src(i, W, ld, col0, k0, nk);
    const int k = min(sk, nk - 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) rw[u] = *reinterpret_cast<const float4*>(W + (int64_t)(col0 + srow + 16 * u) * ld + k0 + k);
  };
  auto lstore = [&](int i, int buf) {
    const int nk = i == 0 ? 32 : 64;
    const int k = min(sk, nk - 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float2* d = reinterpret_cast<float2*>(&Ws[buf][(srow + 16 * u) * LDW + k]);
      d[0] = make_float2(rw[u].x, rw[u].y);
      d[1] = make_float2(rw[u].z, rw[u].w);
    }
  };
  int buf = 0, ci = 0;
  // finish chunk ci: stage chunk ci+1 (already in registers) into the other buffer, barrier, flip
  auto next_chunk = [&]() {
    if (ci + 1 < 16) lstore(ci + 1, buf ^ 1);
    __syncthreads();
    buf ^= 1;
    ++ci;
    if (ci + 1 < 16) gload(ci + 1);
  };

  gload(0);
  lstore(0, 0);
  gload(1);

  // ---- layer 1: [xyz ; score] (4) -> 32, weights in registers (one MFMA k-step), k = q
  float* T0 = Ts[w][0];
  float* T1 = Ts[w][1];
  {
    float w1[2], b1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { w1[t] = p.W1[(16 * t + fr) * 4 + fq]; b1[t] = p.b1[16 * t + fr]; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = min(r0 + 16 * rt + fr, p.n - 1);
      const float x = fq < 3 ? p.xyz[cloud * p.xyz_cs + (int64_t)row * 3 + fq] : p.score[(int64_t)cloud * p.n + row];
      const float a = fmaf(x, 1.f, 0.f);
      float* T = rt ? T1 : T0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[t], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = c[r] + b1[t];
          v = v < 0.f ? v * 0.2f : v;
          T[(4 * fq + r) * LDT + 16 * t + fr] = v;
        }
      }
    }
  }
  __syncthreads();   // chunk 0 staged (and T written)

  // C layout accumulators -> LDS tile, with bias (+ LeakyReLU)
  auto spill_act = [&](float* T, const f32x4 (&acc)[4], const float* bias, bool act) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = bias[16 * t + fr];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[t][r] + b;
        if (act) v = v < 0.f ? v * 0.2f : v;
        T[(4 * fq + r) * LDT + 16 * t + fr] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto zero = [&](f32x4 (&acc)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- layer 2: 32 -> 64 (chunk 0), k order of pw_stream<8,...>: lane (r,q) holds channels [8q, 8q+8)
  float a3[RT][16];
  {
    float a2[RT][8];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float* T = rt ? T1 : T0;
#pragma unroll
      for (int s = 0; s < 8; ++s) a2[rt][s] = T[fr * LDT + 8 * fq + s];
    }
    f32x4 acc[RT][4];
    zero(acc);
    const float* Wt = Ws[buf];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = Wt[(16 * t + fr) * LDW + 8 * fq + s];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt][s], b[t], acc[rt][t], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float* T = rt ? T1 : T0;
      spill_act(T, acc[rt], p.b2, true);
#pragma unroll
      for (int s = 0; s < 16; ++s) a3[rt][s] = T[fr * LDT + 4 * s + fq];     // natural k order from here on
    }
    next_chunk();
  }

  // one 64-column x 64-channel weight chunk against A fragments in natural order (k = k0 + 4 s + q)
  auto mma64 = [&](f32x4 (&acc)[RT][4], const float (&a0)[16], const float (&a1)[16]) {
    const float* Wt = Ws[buf];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = Wt[(16 * t + fr) * LDW + 4 * s + fq];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[t], acc[0][t], 0, 0, 0);
      if (RT > 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[RT - 1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[t], acc[RT - 1][t], 0, 0, 0);
      }
    }
  };
  auto read_nat = [&](const float* T, float (&a)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = T[fr * LDT + 4 * s + fq];
  };

  // ---- layer 3: 64 -> 128 (chunks 1, 2 = column halves); its output is layer 4's A operand (2 x 16 regs per row tile)
  float a4[RT][2][16];   // [row tile][k half][s]
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    f32x4 acc[RT][4];
    zero(acc);
    mma64(acc, a3[0], a3[RT - 1]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float* T = rt ? T1 : T0;
      spill_act(T, acc[rt], p.b3 + 64 * c, true);
      read_nat(T, a4[rt][c]);
    }
    next_chunk();
  }

  // ---- layers 4 + 5 interleaved: column chunk j of 128 -> 256 (two K chunks), activated, becomes K chunk j of 256 -> 64
  f32x4 acc5[RT][4];
  zero(acc5);
  float fres[RT][4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 acc[RT][4];
    zero(acc);
    mma64(acc, a4[0][0], a4[RT - 1][0]);
    next_chunk();
    mma64(acc, a4[0][1], a4[RT - 1][1]);
    float a5[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float* T = rt ? T1 : T0;
      spill_act(T, acc[rt], p.b4 + 64 * j, true);
      read_nat(T, a5[rt]);
    }
    next_chunk();
    if (j == 3) {
      // the residual F rows (C layout) land while the last K chunk of layer 5 runs
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(r0 + 16 * rt + 4 * fq + r, p.n - 1);
          const float* f = p.F + ((int64_t)cloud * p.n + row) * 64 + fr;
#pragma unroll
          for (int t = 0; t < 4; ++t) fres[rt][t][r] = f[16 * t];
        }
    }
    mma64(acc5, a5[0], a5[RT - 1]);
    next_chunk();
  }

  // ---- layer 5 epilogue (bias, + F) -> layer 6 (mlp_proj, k order of pw_stream<16,...>) -> L2 normalise -> store
  {
    float a6[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float* T = rt ? T1 : T0;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = p.b5[16 * t + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc5[rt][t][r] + b;
          v += fres[rt][t][r];
          T[(4 * fq + r) * LDT + 16 * t + fr] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 16; ++s) a6[rt][s] = T[fr * LDT + 16 * fq + s];
    }
    f32x4 acc[RT][4];
    zero(acc);
    const float* Wt = Ws[buf];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = Wt[(16 * t + fr) * LDW + 16 * fq + s];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a6[rt][s], b[t], acc[rt][t], 0, 0, 0);
    }
    float bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = p.b6[16 * t + fr];
    float* Y = p.desc + (int64_t)cloud * p.n * 64;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float ss[4] = {0.f, 0.f, 0.f, 0.f};
      float v[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[t][r] = acc[rt][t][r] + bv[t];
          ss[r] += v[t][r] * v[t][r];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ss[r] += __shfl_xor(ss[r], 1); ss[r] += __shfl_xor(ss[r], 2);
        ss[r] += __shfl_xor(ss[r], 4); ss[r] += __shfl_xor(ss[r], 8);
        const float den = fmaxf(__fsqrt_rn(ss[r]), 1e-12f);
        const int row = r0 + 16 * rt + 4 * fq + r;
        if (row < p.n) {
#pragma unroll
          for (int t = 0; t < 4; ++t) Y[(int64_t)row * 64 + 16 * t + fr] = v[t][r] / den;
        }
      }
    }
  }
}

}  // namespace

bool launch_agg_chain(const AggArgs& a, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  // every point's chain is independent of the blocking, so the rows per workgroup follow the launch size: 128 (two row
  // tiles per wave, weights re-read once per 128 points) when that already fills the chip, 64 for a few clouds in flight
  // (batch-1 latency: twice the workgroups, half the chain per wave) - same bits either way
  if ((int64_t)((a.n + 127) / 128) * a.clouds >= 256) {
    dim3 grid((a.n + 127) / 128, a.clouds);
    hipLaunchKernelGGL(agg_chain_kernel<2>, grid, dim3(256), 0, st, a);
  } else {
    dim3 grid((a.n + 63) / 64, a.clouds);
    hipLaunchKernelGGL(agg_chain_kernel<1>, grid, dim3(256), 0, st, a);
  }
  return true;
}

}  // namespace dsir
