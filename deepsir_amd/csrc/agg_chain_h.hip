// agg_chain_kernel (agg_chain.hip) with the five wide layers of the chain on the fp16 matrix pipe, at fp32 accuracy.
//
// The per-iteration half of Network.aggregation (model.py:223-233) is the one MFMA-bound stage of the path besides the
// descriptor search: 127 kFLOP per point, 77 % of it in 128 -> 256 -> 64, on v_mfma_f32_16x16x4_f32 (157 TFLOP/s, the
// slowest matrix rate of the chip; the fp32 kernel keeps that pipe 75 % busy).  gfx950 has no TF32, but its fp16 MFMA runs
// 16 x faster and accumulates in fp32, so each fp32 operand is split into two fp16 numbers,
//       x = xh + xl + r,   xh = fp16(x),  xl = fp16(x - xh),   |r| <= max(2^-22 |x|, 2^-25)
// (x - xh is exact in fp32; fp16 subnormals are honoured by the matrix core - tools/ubench/mfma_denorm.hip), and
//       a.b = ah.bh + ah.bl + al.bh  (+ al.bl + r-terms <= 4 2^-22 |a||b| + 2^-25 (|a|_1 + |b|_1): dropped)
// is three v_mfma_f32_16x16x32_f16 (every product of two fp16 numbers is exact in fp32; the sum is accumulated in fp32 by
// the matrix core, whose accumulation error was measured at <= 12 fp32 roundings per 192 products,
// tests/test_gpu_screen_bound.py).  Per dot product that is the error fp32 arithmetic itself makes in a 64..256-term sum
// (K 2^-24 |a||b|), at 3/16 of the fp32 MFMA time.  Descriptors differ from the fp32 kernel's by ~1e-7 (asserted <= 2e-6
// by tests/test_gpu_parity.py::test_agg_chain_split_matches_fp32_chain; the reference's own thread-count noise on these
// descriptors is of the same size) - NOT bit-identical to the unfused path, which stays available as the
// reference (DSIR_AGG_F32=1 -> agg_chain_kernel; DSIR_NO_AGG=1 -> six launches).
// Domain: |activation| <= 65504 (fp16 range).  Larger values (or non-finite ones) make that point's descriptor
// non-finite, which the pose solve reports as the pair's `invalid` bit 0 - the same outcome as a non-finite input point.
//
// Structure as agg_chain_kernel: a block owns 64 RT points (4 waves x RT row tiles of 16), activations stay on the CU
// (accumulators -> wave-private LDS tile -> split -> A fragments in registers), weights (pre-split at load) stream through
// a double-buffered LDS tile of 64 columns x 64 channels x {high, low} shared by the four waves.
#include <hip/hip_fp16.h>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int SRS = 80;   // halfs per LDS weight row: 64 + 16 pad (the fragment layout of nn_screen.hip's ref tile)
constexpr int LDT = 68;   // floats per row of the transposition tile: 16-byte aligned rows, conflict-free both ways

__device__ __forceinline__ void split8(const float4 u, const float4 v, h8& h, h8& l) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 t = (_Float16)f[k];
    h[k] = t;
    l[k] = (_Float16)(f[k] - (float)t);
  }
}

#define DSIR_MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)

template <int RT>
__global__ __launch_bounds__(256, 2) void agg_chain_h_kernel(const AggArgs p) {
  __shared__ _Float16 Bs[2][2][64 * SRS];          // [buffer][high | low]
  __shared__ float Ts[4][RT][16 * LDT];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cloud = blockIdx.y;
  const int r0 = blockIdx.x * (64 * RT) + 16 * RT * w;      // first row of this wave

  // ---- weight chunk schedule (the same 16 chunks as agg_chain_kernel): layer index, first column, first channel, channels
  auto chunk_src = [&](int i, int& layer, int& ld, int& col0, int& k0, int& nk) {
    if (i == 0) { layer = 0; ld = 32; col0 = 0; k0 = 0; nk = 32; }
    else if (i <= 2) { layer = 1; ld = 64; col0 = 64 * (i - 1); k0 = 0; nk = 64; }
    else if (i == 15) { layer = 4; ld = 64; col0 = 0; k0 = 0; nk = 64; }
    else {
      const int j = (i - 3) / 3, u = (i - 3) % 3;
      if (u < 2) { layer = 2; ld = 128; col0 = 64 * j; k0 = 64 * u; nk = 64; }
      else { layer = 3; ld = 256; col0 = 0; k0 = 64 * j; nk = 64; }
    }
  };
  // Two register sets: chunk i travels in set i & 1 and is fetched TWO chunks before it is staged.  A chunk's 48 - 96 MFMAs (0.3 - 0.6 us
  // with two workgroups per CU) hide a fraction of one L2 round trip: with one chunk of lookahead (round 3) the sixteen chunks of a
  // workgroup each waited for their weights (1.8 us per chunk, matrix pipe 36 % busy).
  h8 rwh[2][2], rwl[2][2];
  auto gload = [&](int i) {
    int layer, ld, col0, k0, nk;
    chunk_src(i, layer, ld, col0, k0, nk);
    const _Float16* Wh = reinterpret_cast<const _Float16*>(p.Wh[layer]);
    const _Float16* Wl = reinterpret_cast<const _Float16*>(p.Wl[layer]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + 256 * u;
      const int piece = min(f & 7, nk / 8 - 1);
      const int64_t o = (int64_t)(col0 + (f >> 3)) * ld + k0 + 8 * piece;
      rwh[i & 1][u] = *reinterpret_cast<const h8*>(Wh + o);
      rwl[i & 1][u] = *reinterpret_cast<const h8*>(Wl + o);
    }
  };
  auto lstore = [&](int buf, int i) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + 256 * u;
      *reinterpret_cast<h8*>(&Bs[buf][0][(f >> 3) * SRS + 8 * (f & 7)]) = rwh[i & 1][u];
      *reinterpret_cast<h8*>(&Bs[buf][1][(f >> 3) * SRS + 8 * (f & 7)]) = rwl[i & 1][u];
    }
  };
  int buf = 0, ci = 0;
  // finish chunk ci: stage chunk ci + 1 (in registers since the chunk before) into the other buffer, barrier, flip, fetch chunk ci + 3
  // into the set that has just been staged
  auto next_chunk = [&]() {
    if (ci + 1 < 16) lstore(buf ^ 1, ci + 1);
    __syncthreads();
    buf ^= 1;
    ++ci;
    if (ci + 2 < 16) gload(ci + 2);
  };

  gload(0);
  lstore(0, 0);
  gload(1);
  gload(2);

  // ---- layer 1: [xyz ; score] (4) -> 32 in exact fp32 (one MFMA k-step, as agg_chain_kernel)
  {
    float w1[2], b1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { w1[t] = p.W1[(16 * t + fr) * 4 + fq]; b1[t] = p.b1[16 * t + fr]; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = min(r0 + 16 * rt + fr, p.n - 1);
      const float a = fq < 3 ? p.xyz[cloud * p.xyz_cs + (int64_t)row * 3 + fq] : p.score[(int64_t)cloud * p.n + row];
      float* T = Ts[w][rt];
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

  struct AFrag { h8 h[2], l[2]; };                 // one row tile x 64 channels: k-steps 0 / 1, high / low parts
  // C-layout accumulators (+ bias, LeakyReLU) -> LDS tile -> the next layer's A fragments (row fr, channels 8 fq.. and 32 + 8 fq..)
  auto spill = [&](float* T, const f32x4 (&acc)[4], const float* bias, bool act) {
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
  auto read_frag = [&](const float* T, AFrag& a, bool both) {
    const float4* q = reinterpret_cast<const float4*>(T + fr * LDT + 8 * fq);
    split8(q[0], q[1], a.h[0], a.l[0]);
    if (both) split8(q[8], q[9], a.h[1], a.l[1]);
  };
  auto zero = [&](f32x4 (&acc)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // one weight chunk (64 columns x 32 KS channels) against the A fragments of the RT row tiles: three products per k-step
  auto mma = [&](f32x4 (&acc)[RT][4], const AFrag (&a)[RT], const int KS) {
    const _Float16* bh = &Bs[buf][0][fr * SRS + 8 * fq];
    const _Float16* bl = &Bs[buf][1][fr * SRS + 8 * fq];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s < KS) {
          const h8 vh = *reinterpret_cast<const h8*>(bh + 16 * t * SRS + 32 * s);
          const h8 vl = *reinterpret_cast<const h8*>(bl + 16 * t * SRS + 32 * s);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            DSIR_MFMA16(acc[rt][t], a[rt].l[s], vh);
            DSIR_MFMA16(acc[rt][t], a[rt].h[s], vl);
            DSIR_MFMA16(acc[rt][t], a[rt].h[s], vh);
          }
        }
      }
    }
  };

  // ---- layer 2: 32 -> 64 (chunk 0: one k-step)
  AFrag a3[RT];
  {
    AFrag a2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) read_frag(Ts[w][rt], a2[rt], false);
    f32x4 acc[RT][4];
    zero(acc);
    mma(acc, a2, 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      spill(Ts[w][rt], acc[rt], p.b2, true);
      read_frag(Ts[w][rt], a3[rt], true);
    }
    next_chunk();
  }

  // ---- layer 3: 64 -> 128 (chunks 1, 2 = column halves); its output is layer 4's A operand
  AFrag a4[2][RT];   // [k half][row tile]
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    f32x4 acc[RT][4];
    zero(acc);
    mma(acc, a3, 2);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      spill(Ts[w][rt], acc[rt], p.b3 + 64 * c, true);
      read_frag(Ts[w][rt], a4[c][rt], true);
    }
    next_chunk();
  }

  // ---- layers 4 + 5 interleaved: column chunk j of 128 -> 256 (two K chunks), activated, becomes K chunk j of 256 -> 64
  f32x4 acc5[RT][4];
  zero(acc5);
  float fpre[RT][4][4];      // the mlp_feat rows the epilogue adds (model.py:226): fetched during the last column chunk, not in front of their use
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 acc[RT][4];
    zero(acc);
    if (j == 3) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(r0 + 16 * rt + 4 * fq + r, p.n - 1);
          const float* f = p.F + ((int64_t)cloud * p.n + row) * 64 + fr;
#pragma unroll
          for (int t = 0; t < 4; ++t) fpre[rt][r][t] = f[16 * t];
        }
    }
    mma(acc, a4[0], 2);
    next_chunk();
    mma(acc, a4[1], 2);
    AFrag a5[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      spill(Ts[w][rt], acc[rt], p.b4 + 64 * j, true);
      read_frag(Ts[w][rt], a5[rt], true);
    }
    next_chunk();
    mma(acc5, a5, 2);
    next_chunk();
  }

  // ---- layer 5 epilogue (bias, + F) -> layer 6 (mlp_proj) -> L2 normalise -> store
  {
    AFrag a6[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float* T = Ts[w][rt];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = 0; t < 4; ++t) T[(4 * fq + r) * LDT + 16 * t + fr] = (acc5[rt][t][r] + p.b5[16 * t + fr]) + fpre[rt][r][t];
      }
      __builtin_amdgcn_wave_barrier();
      read_frag(T, a6[rt], true);
    }
    f32x4 acc[RT][4];
    zero(acc);
    mma(acc, a6, 2);
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
      const bool extras = p.sq || p.hi || p.packed_init;      // block-uniform
      float* T = Ts[w][rt];
      if (extras) __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ss[r] += __shfl_xor(ss[r], 1); ss[r] += __shfl_xor(ss[r], 2);
        ss[r] += __shfl_xor(ss[r], 4); ss[r] += __shfl_xor(ss[r], 8);
        const float den = fmaxf(__fsqrt_rn(ss[r]), 1e-12f);
        const int row = r0 + 16 * rt + 4 * fq + r;
        if (extras) {
#pragma unroll
          for (int t = 0; t < 4; ++t) T[(4 * fq + r) * LDT + 16 * t + fr] = v[t][r] / den;
        } else if (row < p.n) {
#pragma unroll
          for (int t = 0; t < 4; ++t) Y[(int64_t)row * 64 + 16 * t + fr] = v[t][r] / den;
        }
      }
      if (extras) {
        // The descriptors go back through the wave's LDS tile into the row layout the search's preparation kernels read them in (16
        // lanes per row, four consecutive channels each): |desc|^2 in THEIR summation order (sqnorm_row16: the value enters the
        // distance every arg-min is decided on), the screening's fp16 operand pair by THEIR split - same bits, one pass less over
        // the descriptors (and 16-byte descriptor stores instead of 4-byte ones).
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int lr = 4 * q + fq, row = r0 + 16 * rt + lr;           // lane group fq takes row 4 q + fq, lane fr its channels 4 fr ..
          const float4 d4 = *reinterpret_cast<const float4*>(T + lr * LDT + 4 * fr);
          const float s2 = sqnorm_row16(d4);
          if (row < p.n) {
            const int64_t gr = (int64_t)cloud * p.n + row;
            *reinterpret_cast<float4*>(Y + (int64_t)row * 64 + 4 * fr) = d4;
            if (p.hi) {
              dsir_h4 hh, ll;
              screen_split4(d4, hh, ll, nullptr);
              *reinterpret_cast<dsir_h4*>(reinterpret_cast<_Float16*>(p.hi) + gr * 64 + 4 * fr) = hh;
              *reinterpret_cast<dsir_h4*>(reinterpret_cast<_Float16*>(p.lo) + gr * 64 + 4 * fr) = ll;
            }
            if (fr == 0) {
              if (p.sq) p.sq[gr] = s2;
              if (p.packed_init) p.packed_init[gr] = ~0ull;
            }
          }
        }
      }
    }
  }
}

}  // namespace

// fp32 weights -> the two fp16 parts of the split (host side, at weight load; the whole blob: 2.6 M values).  The same loop
// twice: compiled for the baseline x86-64 the fp32 <-> fp16 conversions are library calls (35 ms per load), with F16C they are
// one instruction (2 ms); both round to nearest even, so the parts are the same bits either way.
#define DSIR_SPLIT_LOOP                                 \
  for (size_t i = 0; i < n; ++i) {                      \
    const _Float16 t = (_Float16)w[i];                  \
    const _Float16 l = (_Float16)(w[i] - (float)t);     \
    __builtin_memcpy(hi + i, &t, 2);                    \
    __builtin_memcpy(lo + i, &l, 2);                    \
  }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("f16c"))) static void split_weights_f16c(const float* w, size_t n, uint16_t* hi, uint16_t* lo) { DSIR_SPLIT_LOOP }
#endif
void split_weights_f16(const float* w, size_t n, uint16_t* hi, uint16_t* lo) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
  if (__builtin_cpu_supports("f16c")) { split_weights_f16c(w, n, hi, lo); return; }
#endif
  DSIR_SPLIT_LOOP
}
#undef DSIR_SPLIT_LOOP

bool launch_agg_chain_h(const AggArgs& a, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  for (int l = 0; l < 5; ++l)
    if (!a.Wh[l] || !a.Wl[l]) return false;
  if ((int64_t)((a.n + 127) / 128) * a.clouds >= 256) {
    dim3 grid((a.n + 127) / 128, a.clouds);
    hipLaunchKernelGGL(agg_chain_h_kernel<2>, grid, dim3(256), 0, st, a);
  } else {
    dim3 grid((a.n + 63) / 64, a.clouds);
    hipLaunchKernelGGL(agg_chain_h_kernel<1>, grid, dim3(256), 0, st, a);
  }
  return true;
}

}  // namespace dsir
