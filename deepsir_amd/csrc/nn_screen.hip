// Nearest-descriptor search, screened in fp16 and DECIDED in exact fp32 (same result, bit for bit, as nn_match.hip).
//
// The reference distance D(j,k) = fl(fl(-2 dot32(a_j, b_k) + |a_j|^2) + |b_k|^2)   (matchnet.py:96-113, model.py:558-569)
// needs a J x K x 64 fp32 contraction per pair and iteration: 45 % of the path's FLOPs on the slowest MFMA rate of the
// chip (fp32: 1/16 of fp16).  Only the ARG-MIN matters, so the contraction is first done approximately where a rigorous
// error bound lets almost every column be discarded, and the exact fp32 formula is evaluated only for the survivors:
//
//   split   x = xh + 2^-11 xl + r,  xh = fp16(x) (flushed to 0 below 2^-14), xl = fp16((x - xh) 2^11), |r| <= 2^-22 |x|
//   S = |a|^2 + |b|^2 - 2 (ah.bh + 2^-11 (ah.bl + al.bh))                     |S - D| <= d = 2^-14 (|a|^2 + |b|^2)
//   pass 1  Umin[j] = min_k (S + d)              >= min_k D(j,k)
//   pass 2  candidates(j) = { k : S - d <= Umin[j] }   contains every k with D(j,k) = min_k D(j,k)
//           (a cheaper one-term pass 1 was measured: its 2^-8 bound admits ~100 candidates per row on real descriptors,
//            whose distances sit within ~1e-2 of each other; with the tight bound 1.2-1.3 candidates survive on average)
//   pass 3  exact D (the fmaf chain over channels 0..63 that v_mfma_f32_16x16x4_f32 evaluates, then the reference's two
//           roundings) for the candidates only; arg-min with ties to the lower index.  A row with more than CAP
//           candidates is scanned exhaustively in exact fp32 (nothing is ever decided by an approximate value).
//
// Error budget for |a|,|b| <= ~1 (descriptors are L2-normalised by the aggregation, model.py:232-233; the bounds scale
// with |a|^2 + |b|^2): representation 3 * 2^-22, fp32 accumulation of 64 (+128 scaled) exact fp16 products <= 64 * 2^-24
// relative to |a||b|, the reference's own fp32 chain <= 64 * 2^-24, final roundings 2^-22: |S2 - D| < 3e-5 for unit vectors
// against d = 1.2e-4 (measured: 7e-7).  Both passes are fp16 MFMAs (v_mfma_f32_16x16x32_f16, fp32 accumulate): 6/16 of the
// fp32 MFMA time.
#include <hip/hip_fp16.h>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#ifndef DSIR_SCREEN_BC
#define DSIR_SCREEN_BC 64
#endif
constexpr int SBC = DSIR_SCREEN_BC;   // ref columns per LDS tile
constexpr int SRS = 72;     // halfs per LDS row: 64 + 8 pad (144 B: the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads)
constexpr int CAP = 16;     // candidates kept per row; more => exhaustive exact scan of that row
constexpr float kC2 = 1.0f / 16384.0f;

__device__ __forceinline__ unsigned int order_bits(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned int b) {
  return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
}

// x [rows][64] fp32 -> hi, lo [rows][64] fp16 (see header); one thread per 4 channels
__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int64_t n4, _Float16* __restrict__ hi,
                                                      _Float16* __restrict__ lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float f[4] = {v.x, v.y, v.z, v.w};
    h4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      _Float16 t = (_Float16)f[k];
      if (fabsf((float)t) < 6.103515625e-05f) t = (_Float16)0.f;          // no fp16 subnormals in the high part
      h[k] = t;
      l[k] = (_Float16)((f[k] - (float)t) * 2048.0f);
    }
    reinterpret_cast<h4*>(hi)[i] = h;
    reinterpret_cast<h4*>(lo)[i] = l;
  }
}

// Common skeleton of the two screening passes.  Block = 4 waves, wave w owns RT row tiles of 16 src rows whose fp16
// fragments stay in registers; ref tiles of 64 columns stream through double-buffered LDS.  XCD-aware work mapping as
// in nn_match.hip.  PASS 1: running min of the upper bound.  PASS 2: candidate collection against Umin.
template <int RT, int PASS, int NWV>
__global__ __launch_bounds__(NWV * 64) void screen_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                     const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                                     const float* __restrict__ sa, const float* __restrict__ sb, int J, int K,
                                                     int cols_per_split, int rb_count, int splits,
                                                     unsigned int* __restrict__ umin, int32_t* __restrict__ cnt,
                                                     int32_t* __restrict__ cand) {
  constexpr int NB = 2;                             // B tiles per buffer: high and low parts
  __shared__ _Float16 Bs[2][NB][SBC * SRS];
  __shared__ float sbs[2][SBC];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int wi = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int rb = wi % rb_count;
  const int split = (wi / rb_count) % splits;
  const int pair = wi / (rb_count * splits);
  const int row0 = (rb * NWV + w) * (16 * RT);
  const int64_t arow = (int64_t)pair * J, brow = (int64_t)pair * K;

  // A fragments (lane: row fr, channels 32 c + 8 fq .. +7) and the per-row constants of this lane's C rows (4 fq + r)
  h8 ah[RT][2], al[RT][2];
  float srow[RT][4], thr[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = min(row0 + rt * 16 + fr, J - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ah[rt][c] = *reinterpret_cast<const h8*>(Ah + (arow + row) * 64 + 32 * c + 8 * fq);
      al[rt][c] = *reinterpret_cast<const h8*>(Al + (arow + row) * 64 + 32 * c + 8 * fq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = min(row0 + rt * 16 + 4 * fq + r, J - 1);
      const float s = sa[arow + rr];
      srow[rt][r] = PASS == 1 ? s + kC2 * s : s - kC2 * s;
      thr[rt][r] = PASS == 1 ? INFINITY : unorder_bits(umin[arow + rr]);
    }
  }

  const int c_begin = split * cols_per_split;
  const int c_end = min(K, c_begin + cols_per_split);
  // staging: thread -> 16-byte piece (8 channels) f of the tile: column f >> 3, piece f & 7; 512 pieces per part
  constexpr int PIECES = SBC * 8;   // 16-byte pieces per part of a tile
  constexpr int NP = (NWV * 64 >= PIECES) ? 1 : PIECES / (NWV * 64);
  const bool stager = NWV * 64 <= PIECES || tid < PIECES;
  h8 pre[NB][NP];
  float pre_sb = 0.f;
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = stager ? tid + NWV * 64 * i : 0;
      const int r = min(c0 + (f >> 3), K - 1);
      pre[0][i] = *reinterpret_cast<const h8*>(Bh + (brow + r) * 64 + 8 * (f & 7));
      pre[1][i] = *reinterpret_cast<const h8*>(Bl + (brow + r) * 64 + 8 * (f & 7));
    }
    if (tid < SBC) {
      const int col = c0 + tid;
      const float s = sb[brow + min(col, K - 1)];
      // -(|b|^2 +- d_b) / 2: the accumulators START from it, so that -2 acc already contains the column term
      pre_sb = col < c_end ? -0.5f * (PASS == 1 ? s + kC2 * s : s - kC2 * s) : -INFINITY;   // columns past the range never win
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = tid + NWV * 64 * i;
      if (stager) {
        *reinterpret_cast<h8*>(&Bs[buf][0][(f >> 3) * SRS + 8 * (f & 7)]) = pre[0][i];
        *reinterpret_cast<h8*>(&Bs[buf][1][(f >> 3) * SRS + 8 * (f & 7)]) = pre[1][i];
      }
    }
    if (tid < SBC) sbs[buf][tid] = pre_sb;
  };
  gload(c_begin);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += SBC) {
    const bool has_next = c0 + SBC < c_end;
    if (has_next) gload(c0 + SBC);
#pragma unroll
    for (int t = 0; t < SBC / 16; ++t) {
      f32x4 hh[RT], mx[RT];
      const float h0 = sbs[buf][16 * t + fr];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        hh[rt] = f32x4{h0, h0, h0, h0};
        mx[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const h8 bh = *reinterpret_cast<const h8*>(&Bs[buf][0][(16 * t + fr) * SRS + 32 * c + 8 * fq]);
        const h8 bl = *reinterpret_cast<const h8*>(&Bs[buf][1][(16 * t + fr) * SRS + 32 * c + 8 * fq]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          hh[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][c], bh, hh[rt], 0, 0, 0);
          mx[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][c], bl, mx[rt], 0, 0, 0);
          mx[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][c], bh, mx[rt], 0, 0, 0);
        }
      }
      const int col = c0 + 16 * t + fr;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // |a|^2 -+ d_a  -  2 (hh + 2^-11 mx),  hh started at -(|b|^2 -+ d_b) / 2
          const float x = fmaf(hh[rt][r], -2.f, fmaf(mx[rt][r], -9.765625e-4f, srow[rt][r]));
          if (PASS == 1) {
            thr[rt][r] = fminf(thr[rt][r], x);                                 // x = upper bound of D(row, col)
          } else {
            const float l = x;                                                 // x = lower bound of D(row, col)
            if (l <= thr[rt][r]) {
              const int row = row0 + rt * 16 + 4 * fq + r;
              if (row < J) {
                const int slot = atomicAdd(cnt + arow + row, 1);
                if (slot < CAP) cand[(arow + row) * CAP + slot] = col;
              }
            }
          }
        }
    }
    if (has_next) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (PASS == 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = thr[rt][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) u = fminf(u, __shfl_xor(u, o));
        const int row = row0 + rt * 16 + 4 * fq + r;
        if (fr == 0 && row < J) atomicMin(umin + arow + row, order_bits(u));
      }
  }
}

// exact D(row, k) exactly as nn_match.hip evaluates it: the k-ordered fmaf chain of v_mfma_f32_16x16x4_f32 from a zero
// accumulator, then fl(fl(-2 dot + |a|^2) + |b|^2)
__device__ __forceinline__ float exact_dist(const float4 (&a)[16], const float* __restrict__ b, float san, float sbn) {
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float4 v = reinterpret_cast<const float4*>(b)[q];
    acc = fmaf(a[q].x, v.x, acc); acc = fmaf(a[q].y, v.y, acc); acc = fmaf(a[q].z, v.z, acc); acc = fmaf(a[q].w, v.w, acc);
  }
  return __fadd_rn(__fmaf_rn(acc, -2.f, san), sbn);
}

// 16 lanes per src row.  A single survivor IS the arg-min (nothing to evaluate); several survivors: one lane each
// evaluates the exact distance; a row whose candidate list overflowed: its 16 lanes scan every ref column exactly.
__global__ __launch_bounds__(256) void exact_pick_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         const float* __restrict__ sa, const float* __restrict__ sb, int J,
                                                         int K, int64_t rows, const int32_t* __restrict__ cnt,
                                                         const int32_t* __restrict__ cand, int32_t* __restrict__ idx) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const bool live = row < rows;
  const int64_t rr = live ? row : rows - 1;
  const int n = cnt[rr];
  if (n == 1) {                                    // uniform over the row's 16 lanes
    if (live && l == 0) idx[row] = cand[rr * CAP];
    return;
  }
  const bool exhaustive = n > CAP || n <= 0;      // n <= 0 cannot happen for finite inputs (the minimiser always qualifies)
  const int64_t pair = rr / J;
  const float* Bp = B + pair * K * 64;
  const float* sbp = sb + pair * K;
  unsigned long long best = ~0ull;
  if (exhaustive || l < n) {
    float4 a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = reinterpret_cast<const float4*>(A + rr * 64)[q];
    const float san = sa[rr];
    if (!exhaustive) {
      const int k = cand[rr * CAP + l];
      best = ((unsigned long long)order_bits(exact_dist(a, Bp + (int64_t)k * 64, san, sbp[k])) << 32) | (unsigned int)k;
    } else {
      for (int k = l; k < K; k += 16) {
        const unsigned long long key =
            ((unsigned long long)order_bits(exact_dist(a, Bp + (int64_t)k * 64, san, sbp[k])) << 32) | (unsigned int)k;
        best = key < best ? key : best;
      }
    }
  }
  // the rows of a wave that reach this point may differ: shuffles only among the 16 lanes of one row (all of them are here)
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const unsigned int lo = __shfl_xor((unsigned int)(best & 0xffffffffull), o);
    const unsigned int hi = __shfl_xor((unsigned int)(best >> 32), o);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    best = other < best ? other : best;
  }
  if (live && l == 0) idx[row] = (int32_t)(best & 0xffffffffull);
}

// diagnostics: out[0] = total candidates, out[1] = rows scanned exhaustively (candidate list overflowed)
__global__ void screen_stats_kernel(const int32_t* __restrict__ cnt, int64_t rows, unsigned long long* __restrict__ out) {
  unsigned long long c = 0, o = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = cnt[i];
    c += (unsigned long long)(n > 0 ? n : 0);
    o += (n > CAP || n <= 0) ? 1ull : 0ull;
  }
  if (c) atomicAdd(out, c);
  if (o) atomicAdd(out + 1, o);
}

inline int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

// scratch: Umin u32 [rows] | cnt i32 [rows] | cand i32 [rows][CAP]
size_t nn_screen_scratch_bytes(int pairs, int J) {
  const size_t rows = (size_t)pairs * J;
  return ((rows * 4 + 255) & ~(size_t)255) * 2 + ((rows * CAP * 4 + 255) & ~(size_t)255);
}

void launch_split16(const float* x, int64_t rows, void* hi, void* lo, hipStream_t st) {
  const int64_t n4 = rows * 16;
  hipLaunchKernelGGL(split16_kernel, dim3(grid_for(n4)), dim3(256), 0, st, x, n4, reinterpret_cast<_Float16*>(hi),
                     reinterpret_cast<_Float16*>(lo));
}

// a, b: fp32 descriptors [pairs][J|K][64] with their fp16 splits (ah, al, bh, bl) and squared norms (sa, sb)
void launch_nn_screen(const float* a, const float* b, const void* ah, const void* al, const void* bh, const void* bl,
                      const float* sa, const float* sb, int pairs, int J, int K, int32_t* idx, void* scratch, hipStream_t st,
                      hipEvent_t ev0, hipEvent_t ev1, unsigned long long* stats) {
  const size_t rows = (size_t)pairs * J;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  unsigned int* umin = reinterpret_cast<unsigned int*>(take(rows * 4));
  int32_t* cnt = reinterpret_cast<int32_t*>(take(rows * 4));
  int32_t* cand = reinterpret_cast<int32_t*>(take(rows * CAP * 4));
  if (ev0) (void)hipEventRecord(ev0, st);
  (void)hipMemsetAsync(umin, 0xff, rows * 4, st);
  (void)hipMemsetAsync(cnt, 0, rows * 4, st);
  constexpr int RT = 2;
  constexpr int NWV = 8;   // waves per block: 8 x 32 rows share one staged ref tile (the L2 -> LDS fill is the scarce resource)
  const int rows_per_block = NWV * 16 * RT;
  const int rb_count = (J + rows_per_block - 1) / rows_per_block;
  const int64_t base = (int64_t)pairs * rb_count;
  const int tiles = (K + SBC - 1) / SBC;
  const int resident = 256 * (NWV == 16 ? 1 : (NWV == 8 ? 2 : 4));
  int splits = 1;
  double best_eff = -1.0;
  for (int sp = 1; sp <= 16 && sp <= tiles; ++sp) {
    const int tiles_per = (tiles + sp - 1) / sp;
    if (sp > 1 && tiles_per < 8) break;
    const int nsp = (tiles + tiles_per - 1) / tiles_per;
    const int64_t blocks = base * nsp;
    const int64_t rounds = (blocks + resident - 1) / resident;
    double eff = (double)blocks / (double)(rounds * resident);
    eff *= (double)tiles / (double)(tiles_per * nsp);
    if (eff > best_eff + 1e-9) { best_eff = eff; splits = sp; }
  }
  const int cols = ((tiles + splits - 1) / splits) * SBC;
  splits = (K + cols - 1) / cols;
  const dim3 grid((unsigned)((int64_t)rb_count * splits * pairs));
  const _Float16 *Ah = reinterpret_cast<const _Float16*>(ah), *Al = reinterpret_cast<const _Float16*>(al);
  const _Float16 *Bh = reinterpret_cast<const _Float16*>(bh), *Bl = reinterpret_cast<const _Float16*>(bl);
  hipLaunchKernelGGL((screen_kernel<RT, 1, NWV>), grid, dim3(NWV * 64), 0, st, Ah, Al, Bh, Bl, sa, sb, J, K, cols, rb_count, splits, umin,
                     cnt, cand);
  hipLaunchKernelGGL((screen_kernel<RT, 2, NWV>), grid, dim3(NWV * 64), 0, st, Ah, Al, Bh, Bl, sa, sb, J, K, cols, rb_count, splits, umin,
                     cnt, cand);
  hipLaunchKernelGGL(exact_pick_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, a, b, sa, sb, J, K,
                     (int64_t)rows, cnt, cand, idx);
  if (ev1) (void)hipEventRecord(ev1, st);
  if (stats) {
    (void)hipMemsetAsync(stats, 0, 16, st);
    hipLaunchKernelGGL(screen_stats_kernel, dim3(256), dim3(256), 0, st, cnt, (int64_t)rows, stats);
  }
}

}  // namespace dsir
