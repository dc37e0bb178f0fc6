// Nearest-descriptor search, screened in fp16 and DECIDED in exact fp32 (same result, bit for bit, as nn_match.hip).
//
// The reference distance D(j,k) = fl(fl(-2 dot32(a_j, b_k) + |a_j|^2) + |b_k|^2)   (matchnet.py:96-113, model.py:558-569)
// needs a J x K x 64 fp32 contraction per pair and iteration: 45 % of the path's FLOPs on the slowest MFMA rate of the
// chip (fp32: 1/16 of fp16).  Only the ARG-MIN matters, so the contraction is first done approximately where a rigorous
// error bound lets almost every column be discarded, and the exact fp32 formula is evaluated only for the survivors:
//
//   split   x = xh + 2^-11 xl + r,  xh = fp16(x) (flushed to 0 below 2^-14), xl = fp16((x - xh) 2^11); stored: 2^11 xh, xl
//   S = |a|^2 + |b|^2 - 2 (ah.bh + 2^-11 (ah.bl + al.bh))     |S - D| <= d = 2^-15 (|a|^2 + |b|^2) + 2^-20
//   screen  ONE pass of fp16 MFMAs (v_mfma_f32_16x16x32_f16, fp32 accumulate; 3/16 of the fp32 MFMA time).  Every lane
//           keeps, for each of its rows, the two smallest lower bounds L = S - d among the columns it sees (its class:
//           columns = fr mod 16 of the workgroup's column range) and the column of the smallest.  At the end of the
//           workgroup T = min over the lanes of (smallest L + 2 d) >= min_k D(row, k), and
//             - every lane whose smallest L <= T emits that column as a candidate entry,
//             - a lane whose runner-up is <= T as well emits its class (a second qualifying column exists, unknown which),
//             - the row's global threshold is lowered to T (atomicMin over the workgroups that share the row).
//           Every column with D = min D has L <= D <= T for every workgroup's T, so it is emitted, or covered by a class.
//   pick    entries above the row's final threshold are dropped.  One column left: it IS the arg-min.  Several: their
//           exact D (the fmaf chain over channels 0..63 that v_mfma_f32_16x16x4_f32 evaluates, then the reference's two
//           roundings) decides, ties to the lower index.
//   rest    rows with a surviving class entry or an overflowed entry list are searched by the exhaustive exact-fp32 MFMA
//           kernel (nn_match.hip) through a row list; a pair with a quarter of its rows in that state is searched by it
//           as a whole and skips the screening for the rest of the registration.  Nothing is ever decided by an
//           approximate value.
//
// Error budget, in units of M = |a|^2 + |b|^2 (|a||b| <= M/2): representation 2^-22 per element (3 2^-22 M with the
// dropped al.bl term); fp32 accumulation: both high parts are stored pre-scaled by 2^11 (exact) and the kernel forms
// z' = 2^22 (c + ah.bh) + 2^11 (ah.bl + al.bh) = 2^22 z in ONE chain of six MFMAs (192 exact fp16 products + the seed;
// a pure power-of-two scaling: the roundings are those of the unscaled sum), pessimistically one fp32 rounding per
// addition relative to the sum of the magnitudes (|a||b| + |b|^2 / 2) <= M: 198 2^-24 M on z, 2^-15.4 M after the factor 2; the reference's own
// fmaf chain 64 * 2^-24 * 2 |a||b| <= 2^-18 M; final roundings 2^-22 M: 2.8e-5 M against 2^-15 M = 3.05e-5 M (measured on
// unit descriptors: < 1e-6 against 6e-5).  Elements below 2^-25 lose their low part
// (fp16 underflow): <= 2^-25 per element, 2^-21 (|a| + |b|) <= 2^-21 (1 + M/2) on the distance: the constant term 2^-20.
// MEASURED on the matrix core (round 3; tests/test_gpu_screen_bound.py through dsir_screen_bounds, which runs this file's MFMA
// chain on this file's operands and returns L, U and the exact D of EVERY (row, column)): 19 input regimes chosen against the
// bound - same-sign components (no cancellation in the accumulator), constant vectors / 64 identical products, components on
// fp16 rounding ties, |x| = 16, norms 1e-3 .. 30, one-hot, sparse, below the fp16 normal range, near-duplicates, geometric
// decay - x three shapes, 4.4 M entries: no entry outside [L, U]; worst |D - (L + U) / 2| = 0.066 of the half width (a margin
// of 15 on d); the accumulation error of the six chained MFMAs against an fp64 sum of the same fp16 products never exceeded
// 11.9 fp32 roundings of the magnitude sum, against the 198 budgeted above: the v_mfma_f32_16x16x32_f16 adder of gfx950 rounds
// far less often than once per product (and not by truncation: same-sign inputs err LESS than signed ones).  The bound is
// kept at its pessimistic width; the test asserts a margin of 2 so that a different stepping would be noticed.
// Elements with |x| > 16 (the 2^11 pre-scaling of the high part must stay inside fp16: 2^15 < 65504) or not finite: split16_kernel
// raises a flag and every pair is searched exhaustively (the engine's descriptors are L2-normalised, model.py:232-233,
// and never take that path).
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
#ifndef DSIR_SCREEN_SRS
#define DSIR_SCREEN_SRS 80
#endif
#ifndef DSIR_SCREEN_RT
#define DSIR_SCREEN_RT 0      // row tiles per wave: 0 = chosen per launch (launch_nn_screen), 2 / 4 = forced
#endif
#ifndef DSIR_SCREEN_NWV
#define DSIR_SCREEN_NWV 8
#endif
constexpr int SRS = DSIR_SCREEN_SRS;     // halfs per LDS row: 64 + 16 pad (160 B; measured 1 % faster than the 144 B of a minimal pad)
constexpr int CAP = 16;     // entries kept per row; more => the row goes to the exhaustive kernel
constexpr float kC1 = 1.0f / 32768.0f;      // bound width: d = kC1 (|a|^2 + |b|^2) + kC0 (see the header)
constexpr float kC0 = 1.0f / 1048576.0f;
constexpr float kW = 2.0f * 1.015625f;       // upper - lower bound = 2 d, with slack for the rounding of its own evaluation

__device__ __forceinline__ unsigned int order_bits(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned int b) {
  return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
}

// four channels fp32 -> fp16 high / low parts (see the header): device_utils.h, screen_split4 (shared with agg_chain_h.hip's epilogue)
__device__ __forceinline__ void split4(const float4 v, h4& h, h4& l, int32_t* __restrict__ bad) { screen_split4(v, h, l, bad); }

// x [rows][64] fp32 -> hi, lo [rows][64] fp16; one thread per 4 channels
__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int64_t n4, _Float16* __restrict__ hi,
                                                      _Float16* __restrict__ lo, int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    h4 h, l;
    split4(reinterpret_cast<const float4*>(x)[i], h, l, bad);
    reinterpret_cast<h4*>(hi)[i] = h;
    reinterpret_cast<h4*>(lo)[i] = l;
  }
}

// the same split and the squared norm of each row (sqnorm_kernel's arithmetic) in one pass over the descriptors:
// 16 lanes per row, float4 each
__global__ __launch_bounds__(256) void split_norm_kernel(const float* __restrict__ x, int64_t rows, _Float16* __restrict__ hi,
                                                         _Float16* __restrict__ lo, float* __restrict__ sq,
                                                         int32_t* __restrict__ bad) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) v = *reinterpret_cast<const float4*>(x + row * 64 + l * 4);
  const float s = sqnorm_row16(v);
  if (row >= rows) return;
  h4 h, lw;
  split4(v, h, lw, bad);
  *reinterpret_cast<h4*>(hi + row * 64 + l * 4) = h;
  *reinterpret_cast<h4*>(lo + row * 64 + l * 4) = lw;
  if (l == 0) sq[row] = s;
}

// Block = NWV waves, wave w owns RT row tiles of 16 src rows whose fp16 fragments stay in registers (RT = 4: two operand
// sets of 8 registers per tile - 64 -, 224 VGPRs in all, two waves per SIMD); ref tiles of SBC columns stream through
// double-buffered LDS (fetched two tiles ahead through registers).  XCD-aware work mapping as in nn_match.hip.
//
// ONE accumulator chain per row tile: with BOTH high parts pre-scaled by 2^11 (exact in fp16 for |x| <= 16, which split4
// enforces) the three partial products carry the same factor, z' = 2^22 (c + ah.bh + 2^-11 (ah.bl + al.bh)), and come out
// of six chained MFMAs whose first C operand is 2^22 c, c = -(|b|^2 - d_b) / 2 from LDS: no VALU instruction joins the
// partial products, and the src side needs two operand sets (round 2 started with three: ah, 2^11 ah, al), which is
// what lets a wave hold four row tiles: every ref fragment read from LDS then feeds four MFMA chains instead of two and a
// staged tile serves 512 rows instead of 256 (1.243 -> 1.097 ms per 128-pair launch).
// Ranking, per accumulator element, exactly four VALU instructions (v_med3_f32, v_cmp_gt_f32, v_cndmask_b32, v_max_f32;
// inline asm, so no canonicalising v_max x, x and no re-association).  gfx950 overlaps a wave's VALU work with the matrix
// pipe only marginally (tools/ubench/rank_overlap.hip: clustered, interleaved and role-staggered schedules all land within
// 7 % of MFMA time + VALU time), so the levers are the instruction count and the exposed latencies: the ranking of step
// t-1 and the LDS fragment reads of step t+1 are issued between the MFMAs of step t (order pinned with sched_barrier),
// which takes the ds_read latency off the critical path.
#ifndef DSIR_SCREEN_WPE
#define DSIR_SCREEN_WPE 2
#endif
#define DSIR_FENCE() __builtin_amdgcn_sched_barrier(0)
// (z1 >= z2: running top two of z, k1: column of z1)  <-  z at column col
// DSIR_ABL_*: timing-only ablation builds (tools/ab_screen.sh; WRONG results, never part of libdsir.so): one VALU
// instruction per element instead of the ranking, no LDS fragment reads after a tile's first step, no tile staging / barrier.
#ifdef DSIR_ABL_NORANK
#define DSIR_RANK(z1, z2, k1, z, col) asm volatile("v_max_f32 %0, %0, %1" : "+v"(z1) : "v"(z))
#else
#define DSIR_RANK(z1, z2, k1, z, col)                                                                                         \
  asm volatile("v_med3_f32 %1, %0, %1, %3\n\tv_cmp_gt_f32 vcc, %3, %0\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_max_f32 %0, %0, %3" \
               : "+v"(z1), "+v"(z2), "+v"(k1) : "v"(z), "v"(col) : "vcc")
#endif
#ifdef DSIR_ABL_NOLDS
#define DSIR_MORE(x) false
#else
#define DSIR_MORE(x) (x)
#endif
constexpr int kMaxBoundTiles = 4096;   // tiles of a pruned search's column order (LDS: tile flags of the bound pass, an item's tile list)
// ORD = false: the dense search (every tile of the item's column range, natural row / column order) - the `ord` fields are not
// touched and the code is the round-2 kernel's; ORD = true: the pruned search of nn_prune.hip (row / column orders, tile lists)
template <int RT, int NWV, bool ORD>
__device__ __forceinline__ void screen_item(const int wi, const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                     const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                                     const float* __restrict__ sa, const float* __restrict__ sb, int J, int K,
                                                     int cols_per_split, int rb_count, int splits,
                                                     unsigned int* __restrict__ umin, int32_t* __restrict__ cnt,
                                                     int2* __restrict__ cand, int32_t* __restrict__ ovf,
                                                     int32_t* __restrict__ rowlist, int ovf_min, const ScreenOrder ord) {
  constexpr int NP = SBC * 8 / (NWV * 64);          // 16-byte pieces of each tile part per thread
  static_assert(NP >= 1 && NP * NWV * 64 == SBC * 8, "staging layout");
  constexpr int NSB = SBC * 4 / (NWV * 64) > 0 ? SBC * 4 / (NWV * 64) : 1;   // replicated accumulator seeds per thread
  static_assert(SBC * 4 <= NWV * 64 * NSB, "seed layout");
  __shared__ _Float16 Bs[2][2][SBC * SRS];          // [buffer][high | low part]
  __shared__ float4 sbs[2][SBC];                    // 2^11 c, c = -(|b|^2 - d_b) / 2, replicated x4: the first MFMA's C operand
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int rb = wi % rb_count;
  const int split = (wi / rb_count) % splits;
  const int pair = wi / (rb_count * splits);
  if (ORD && ord.rborder) rb = ord.rborder[pair * rb_count + rb];   // pruned search: row blocks in the order of their tile counts, longest first
  const int row0 = (rb * NWV + w) * (16 * RT);
  const int64_t arow = (int64_t)pair * J, brow = (int64_t)pair * K;
  // a pair with that many undecidable rows is searched exhaustively as a whole.  The counter grows while the kernel runs
  // (other workgroups list rows), so ONE thread reads it and the workgroup decides together: exiting on a value seen
  // mid-launch is safe, a workgroup whose threads disagree would hang at the first barrier
  __shared__ int s_skip;
  if (tid == 0) s_skip = ovf[pair] >= ovf_min ? 1 : 0;
  __syncthreads();
  const int skip = s_skip;
  __syncthreads();                                   // the next item of a persistent workgroup rewrites s_skip
  if (skip) return;

  // A fragments: lane holds row fr, channels 32 c + 8 fq .. +7; ah = 2^11 x (high part), al = 2^11 x (low part)
  h8 ah[RT][2], al[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = min(row0 + rt * 16 + fr, J - 1);
    if (ORD && ord.rows) row = ord.rows[arow + row];   // pruned search: the block's rows are rows [row0, ..) of the given ORDER
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ah[rt][c] = *reinterpret_cast<const h8*>(Ah + (arow + row) * 64 + 32 * c + 8 * fq);
      al[rt][c] = *reinterpret_cast<const h8*>(Al + (arow + row) * 64 + 32 * c + 8 * fq);
    }
  }
  // per C element (row 4 fq + r of tile rt, column class fr): the two largest z' and the column of the largest
  float z1[RT][4], z2[RT][4];
  int k1[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { z1[rt][r] = -INFINITY; z2[rt][r] = -INFINITY; k1[rt][r] = -1; }

  const int c_begin = split * cols_per_split;
  const int c_end = min(K, c_begin + cols_per_split);
  // the item's tile sequence: dense (every tile of its column range) or, pruned search, its share of the block's tile LIST;
  // columns are positions in the given column ORDER (ord.cols: position -> ref row; nullptr: identity)
  const int32_t* tl = nullptr;
  int n_i = (c_end - c_begin + SBC - 1) / SBC;
  if (ORD && ord.tlist) {
    const int n_all = ord.tcount[pair * rb_count + rb];
    const int li0 = (int)((int64_t)split * n_all / splits), li1 = (int)((int64_t)(split + 1) * n_all / splits);
    tl = ord.tlist + ((int64_t)pair * rb_count + rb) * ord.tl_stride + li0;
    n_i = li1 - li0;
  }
  n_i = __builtin_amdgcn_readfirstlane(n_i);
  if (n_i <= 0) return;                                // block-uniform; nothing of this row block falls to this split
  // the item's tile list in LDS: a tile's fetch must not wait for a global load of its own address first (the prefetch
  // distance of three tiles, ~4 us, does not cover two dependent trips to L2 / HBM under load)
  __shared__ int tls[ORD ? kMaxBoundTiles : 1];
  if (ORD && tl) {
    for (int i = tid; i < n_i; i += NWV * 64) tls[i] = tl[i];
    __syncthreads();                                   // (the item's first barrier pair above keeps the previous item's readers out)
  }
  auto tile_c0 = [&](int i) -> int {
    if (!ORD) return c_begin + i * SBC;                // dense: plain arithmetic, addresses clamped by the loads
    const int ic = min(i, n_i - 1);
    return tl ? tls[ic] * SBC : c_begin + ic * SBC;
  };
  const int c_lim = (ORD && tl) ? K : c_end;
  // staging: thread -> 16-byte piece (8 channels) of the tile: column f >> 3, piece f & 7
  // two register sets: a tile is fetched two iterations before it is needed (the L2 / MALL latency under load exceeds
  // the time of one tile) and written to the free LDS buffer at the end of the iteration before
  // sb: the accumulator seeds.  Pruned search: |b|^2 as loaded + column-in-range bits (`ok`), the seed formed when the tile is
  // stored to LDS: its gload branches, and with the use next to the load the compiler waited THERE - s_waitcnt vmcnt(0) behind
  // the tile fetches just issued, once per tile (ISA; 15 % of the kernel).  The dense search keeps the round-2 form (the same
  // restructuring measured 3 % slower there)
  struct Pre { h8 h[NP], l[NP]; float sb[NSB]; unsigned ok; };
  Pre preA, preB;
  auto gload = [&](Pre& pre, int it) {                 // tile `it` of the sequence (past its end: clamped, never used)
    const int c0 = tile_c0(it);
    const bool live = !ORD || it < n_i;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = tid + NWV * 64 * i;
      const int r = min(c0 + (f >> 3), K - 1);       // ORD: Bh / Bl are the copies in column order
      pre.h[i] = *reinterpret_cast<const h8*>(Bh + (brow + r) * 64 + 8 * (f & 7));
      pre.l[i] = *reinterpret_cast<const h8*>(Bl + (brow + r) * 64 + 8 * (f & 7));
    }
    pre.ok = 0u;
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      const int f = tid + NWV * 64 * i;
      const int col = c0 + (f >> 2);
      const int r = min(col, K - 1);
      const bool ok = live && col < c_lim && f < SBC * 4;                 // columns past the range never win
      if (ORD) { pre.sb[i] = ord.sbp[brow + r]; pre.ok |= ok ? 1u << i : 0u; }
      else { const float s = sb[brow + r]; pre.sb[i] = ok ? -2097152.f * (s - kC1 * s) : -INFINITY; }
    }
  };
  auto lstore = [&](const Pre& pre, int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = tid + NWV * 64 * i;
      *reinterpret_cast<h8*>(&Bs[buf][0][(f >> 3) * SRS + 8 * (f & 7)]) = pre.h[i];
      *reinterpret_cast<h8*>(&Bs[buf][1][(f >> 3) * SRS + 8 * (f & 7)]) = pre.l[i];
    }
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      const int f = tid + NWV * 64 * i;
      // the accumulator seed 2^22 c, c = -(|b|^2 - d_b) / 2; columns past the range never win
      if (f < SBC * 4)
        reinterpret_cast<float*>(sbs[buf])[f] = !ORD ? pre.sb[i] : ((pre.ok >> i) & 1u) ? -2097152.f * (pre.sb[i] - kC1 * pre.sb[i]) : -INFINITY;
    }
  };
  // fragments of one 16-column step: lane holds column fr of the step, channels 8 fq .. (+32)
  struct Frag { h8 bh0, bh1, bl0, bl1; f32x4 cin; };
  f32x4 zP[RT];
  int colP = 0;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) zP[rt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#define DSIR_MFMA(acc, a, b, c) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
  // one step: the six MFMAs per row tile on `cur`, between them the ranking of the previous step's accumulators (zP) and
  // the fragment reads of the next step (`nxt`; skipped when !more)
  auto step = [&](const Frag& cur, Frag& nxt, const _Float16* bhp, const _Float16* blp, const float4* cp, bool more, int col) {
    f32x4 zN[RT];
#define DSIR_RANK_ALL(r)                                                                          \
    DSIR_FENCE();                                                                                 \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) DSIR_RANK(z1[rt][r], z2[rt][r], k1[rt][r], zP[rt][r], colP); \
    DSIR_FENCE()
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], ah[rt][0], cur.bh0, cur.cin);
    if (more) { nxt.bh0 = *reinterpret_cast<const h8*>(bhp); const float4 v = *cp; nxt.cin = f32x4{v.x, v.y, v.z, v.w}; }
    DSIR_RANK_ALL(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], ah[rt][1], cur.bh1, zN[rt]);
    if (more) nxt.bl0 = *reinterpret_cast<const h8*>(blp);
    DSIR_RANK_ALL(1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], ah[rt][0], cur.bl0, zN[rt]);
    if (more) nxt.bh1 = *reinterpret_cast<const h8*>(bhp + 32);
    DSIR_RANK_ALL(2);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], al[rt][0], cur.bh0, zN[rt]);
    if (more) nxt.bl1 = *reinterpret_cast<const h8*>(blp + 32);
    DSIR_RANK_ALL(3);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], ah[rt][1], cur.bl1, zN[rt]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) DSIR_MFMA(zN[rt], al[rt][1], cur.bh1, zN[rt]);
#undef DSIR_RANK_ALL
    DSIR_FENCE();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) zP[rt] = zN[rt];
    colP = col;
  };
  // one tile: ranks into (z1, z2, k1); `pre` holds the tile after it and is refilled with the one two further on
  auto tile = [&](int it, int buf, Pre& pre) {
    const int c0 = tile_c0(it);
    const _Float16* bhp = &Bs[buf][0][fr * SRS + 8 * fq];
    const _Float16* blp = &Bs[buf][1][fr * SRS + 8 * fq];
    const float4* cp = &sbs[buf][fr];
    Frag fa, fb;
    fa.bh0 = *reinterpret_cast<const h8*>(bhp); fa.bh1 = *reinterpret_cast<const h8*>(bhp + 32);
    fa.bl0 = *reinterpret_cast<const h8*>(blp); fa.bl1 = *reinterpret_cast<const h8*>(blp + 32);
    { const float4 v = *cp; fa.cin = f32x4{v.x, v.y, v.z, v.w}; }
#pragma unroll
    for (int t = 0; t < SBC / 16; t += 2) {
      step(fa, fb, bhp + 16 * (t + 1) * SRS, blp + 16 * (t + 1) * SRS, cp + 16 * (t + 1), DSIR_MORE(true), c0 + 16 * t + fr);
      step(fb, fa, bhp + 16 * (t + 2) * SRS, blp + 16 * (t + 2) * SRS, cp + 16 * (t + 2), DSIR_MORE(t + 2 < SBC / 16), c0 + 16 * (t + 1) + fr);
    }
#ifndef DSIR_ABL_NOSTAGE
    lstore(pre, buf ^ 1);
    gload(pre, it + 3);                               // clamped addresses: harmless past the range
    __syncthreads();
#endif
  };
#undef DSIR_MFMA
  // all three fetches in flight before the first store waits for its own (vmcnt counts them in order)
  Pre pre0;
  gload(pre0, 0);
  gload(preA, 1);
  gload(preB, 2);
  lstore(pre0, 0);
  __syncthreads();
  if (ORD) {
    // two tiles per trip, the odd last one peeled: a straight-line loop body, for which the compiler's vmcnt bookkeeping of the
    // two fetch sets in flight is exact (vmcnt(5), (4), (3) in front of the three stores of a set)
    int it = 0;
    for (; it + 1 < n_i; it += 2) {
      tile(it, 0, preA);
      tile(it + 1, 1, preB);
    }
    if (it < n_i) tile(it, 0, preA);
  } else {
    for (int it = 0; it < n_i; it += 2) {
      tile(it, 0, preA);
      if (it + 1 < n_i) tile(it + 1, 1, preB);
    }
  }
  // the last step's accumulators
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) DSIR_RANK(z1[rt][r], z2[rt][r], k1[rt][r], zP[rt][r], colP);

  // T = min over the lanes of (their smallest lower bound + its bound width) >= min_k D(row, k) over this block's
  // columns, hence over all columns.  Every column of the block with lower bound <= T is either some lane's smallest
  // (emitted) or makes that lane's runner-up <= T (class emitted).
  // the epilogue's gathers (|a|^2 of the lane's rows, |b|^2 of their best columns) first, all independent: one memory
  // latency for the 4 RT elements instead of one each in front of the atomics
  float sanv[RT][4], sbkv[RT][4];
  int rowv[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ro = min(row0 + rt * 16 + 4 * fq + r, J - 1);
      if (ORD && ord.rows) ro = ord.rows[arow + ro];
      rowv[rt][r] = ro;
      int ko = max(k1[rt][r], 0);
      if (ORD && ord.cols) {                           // k1 is a position in the column order: back to the ref row
        ko = ord.cols[brow + ko];
        if (k1[rt][r] >= 0) k1[rt][r] = ko;
      }
      sanv[rt][r] = sa[arow + ro];
      sbkv[rt][r] = sb[brow + ko];
    }
  // three passes over the lane's 4 RT elements, so that the memory round trips of a pass overlap instead of queueing behind
  // each other (16 dependent trips per item before; rows in a given order scatter them over as many cache lines):
  // thresholds + atomicMin (no return) -> one counted atomicAdd per element (both of its entries) -> the entries' stores
  float l1v[RT][4], l2v[RT][4];
  int nent[RT][4], base[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool live_row = row0 + rt * 16 + 4 * fq + r < J;
      const float san = sanv[rt][r];
      const float slo = san - kC1 * san - kC0;       // |a|^2 - d_a
      // z' = 2^22 z: L = slo - 2 z = slo - 2^-21 z'
      const float l1 = fmaf(z1[rt][r], -4.76837158203125e-7f, slo), l2 = fmaf(z2[rt][r], -4.76837158203125e-7f, slo);
      const int k = k1[rt][r];
      float u = INFINITY;
      if (k >= 0) u = l1 + kW * (kC1 * (san + sbkv[rt][r]) + kC0);
      float T = u;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) T = fminf(T, __shfl_xor(T, o));
      if (live_row && fr == 0) atomicMin(umin + arow + rowv[rt][r], order_bits(T));
      l1v[rt][r] = l1; l2v[rt][r] = l2;
      // entries of this element: its best column (code k) and / or its class: a second column of this lane's class (columns
      // c_begin + fr + 16 m of this split) may qualify as well - which one is not tracked, so the class itself becomes an
      // entry and the row goes to the exhaustive kernel
      nent[rt][r] = live_row ? ((k >= 0 && l1 <= T) ? 1 : 0) + (l2 <= T ? 2 : 0) : 0;      // bit 0: column entry, bit 1: class entry
    }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (nent[rt][r] & 1) + (nent[rt][r] >> 1);
      base[rt][r] = n ? atomicAdd(cnt + arow + rowv[rt][r], n) : 0;
    }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rowv[rt][r];
      int slot = base[rt][r];
      auto emit = [&](int code, float lower) {
        if (slot < CAP) cand[(int64_t)slot * ord.cand_rs + arow + row] = make_int2(code, __float_as_int(lower));   // [slot][row]: see exact_pick_kernel
        else if (slot == CAP) rowlist[arow + atomicAdd(ovf + pair, 1)] = row;   // this row just overflowed (once per row)
        ++slot;
      };
      if (nent[rt][r] & 1) emit(k1[rt][r], l1v[rt][r]);
      if (nent[rt][r] & 2) emit(-(1 + split * 16 + fr), l2v[rt][r]);
    }
}

// The launch: one work item (pair, ref split, row block) per workgroup, or - DSIR_SCREEN_PERSIST workgroups per CU - a
// persistent grid whose workgroups walk the items with stride gridDim (no workgroup launch / drain between items).  XCD-aware
// in both forms: workgroups are dealt round-robin over the 8 XCDs, the remap hands every XCD a contiguous range of items.
template <int RT, int NWV, bool ORD>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(DSIR_SCREEN_WPE, DSIR_SCREEN_WPE))) void screen_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                     const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                                     const float* __restrict__ sa, const float* __restrict__ sb, int J, int K,
                                                     int cols_per_split, int rb_count, int splits,
                                                     unsigned int* __restrict__ umin, int32_t* __restrict__ cnt,
                                                     int2* __restrict__ cand, int32_t* __restrict__ ovf,
                                                     int32_t* __restrict__ rowlist, int ovf_min, int total, const ScreenOrder ord) {
  const int nwg = gridDim.x, id = blockIdx.x;
  if (ORD) {
    // pruned search: items differ in length (per pair and along the row order), so the static deal below would leave XCDs idle
    // while others still work.  A persistent grid instead: every XCD owns a queue - the contiguous item range the static deal
    // would give it, for the same L2 locality - and its workgroups pop items from it; a workgroup whose queue is empty helps
    // the next XCD's.  Every workgroup leaves after finding all 8 queues empty.
    __shared__ int s_wi;
    const int q8 = total >> 3, r8 = total & 7;
    for (int q = 0; q < 8; ++q) {
      const int x = ((id & 7) + q) & 7;
      const int start = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, len = q8 + (x < r8 ? 1 : 0);
      for (;;) {
        if (threadIdx.x == 0) s_wi = atomicAdd(ord.queue + x, 1);
        __syncthreads();
        const int i = s_wi;
        __syncthreads();
        if (i >= len) break;
        screen_item<RT, NWV, ORD>(start + i, Ah, Al, Bh, Bl, sa, sb, J, K, cols_per_split, rb_count, splits, umin, cnt, cand, ovf, rowlist, ovf_min, ord);
      }
    }
    return;
  }
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  for (int wi = first; wi < total; wi += nwg)
    screen_item<RT, NWV, ORD>(wi, Ah, Al, Bh, Bl, sa, sb, J, K, cols_per_split, rb_count, splits, umin, cnt, cand, ovf, rowlist, ovf_min, ord);
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

// One thread per src row.  Entries whose lower bound exceeds the row's final threshold (the min over all blocks) are
// dropped; a single surviving column IS the arg-min (nothing to evaluate); several surviving columns: their exact
// distances decide (ties to the lower index).  Left to the exhaustive fp32 MFMA kernel (nn_match.hip;
// unpack_listed_kernel copies its results): rows whose entry list overflowed (listed by screen_kernel), rows with a
// surviving class entry (code < 0: some column of that class, not tracked which) or without any entry (non-finite
// input) - listed here -, and every row of a pair with ovf_min or more listed rows.
__global__ __launch_bounds__(256) void exact_pick_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         const float* __restrict__ sa, const float* __restrict__ sb, int J,
                                                         int K, const unsigned int* __restrict__ umin,
                                                         const int32_t* __restrict__ cnt, const int2* __restrict__ cand,
                                                         int32_t* __restrict__ ovf, int ovf_min,
                                                         int32_t* __restrict__ rowlist, int32_t* __restrict__ idx) {
  const int pair = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  const int64_t row = (int64_t)pair * J + j;
  const int n = cnt[row];
  if (n > CAP || ovf[pair] >= ovf_min) return;
  const float T = unorder_bits(umin[row]);
  // entry lists slot-major, [slot][row] (round 4; [row][CAP] before): a row holds two to four entries (one per ref split and more), and
  // with one thread per row the slot-s entries of neighbouring rows are neighbours in memory - 8 bytes per entry read instead of the
  // 128-byte line of a row's list (the lists were ~80 of the 569 MB a 128-pair search moved)
  const int64_t crs = (int64_t)gridDim.y * J;
  const int2* ce = cand + row;
  int kept = 0, first = 0;
  bool cls = false;
  for (int e = 0; e < n; ++e) {
    const int2 c = ce[e * crs];
    if (__int_as_float(c.y) <= T) {
      if (c.x < 0) cls = true;
      else if (kept++ == 0) first = c.x;
    }
  }
  if (kept == 0) {                                   // no column entry at all (non-finite input): the exhaustive kernel
    rowlist[(int64_t)pair * J + atomicAdd(ovf + pair, 1)] = j;
    return;
  }
  if (kept == 1 && !cls) {
    idx[row] = first;
    return;
  }
  float4 a[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = reinterpret_cast<const float4*>(A + row * 64)[q];
  const float san = sa[row];
  const float* Bp = B + (int64_t)pair * K * 64;
  const float* sbp = sb + (int64_t)pair * K;
  unsigned long long best = ~0ull;
  for (int e = 0; e < n; ++e) {
    const int2 c = ce[e * crs];
    if (c.x >= 0 && __int_as_float(c.y) <= T) {
      const unsigned long long key =
          ((unsigned long long)order_bits(exact_dist(a, Bp + (int64_t)c.x * 64, san, sbp[c.x])) << 32) | (unsigned int)c.x;
      best = key < best ? key : best;
    }
  }
  if (cls) {
    // a class entry stands for columns whose lower bounds are >= its own: it matters only if that bound does not exceed the
    // best EXACT distance found among the column entries (an actual distance of the row, so D >= L > it rules the class out,
    // ties included) - a tighter test than the screening threshold T, which sits a bound width above the minimum
    const float dbest = unorder_bits((unsigned int)(best >> 32));
    bool open = !(dbest == dbest);
    for (int e = 0; e < n && !open; ++e) {
      const int2 c = ce[e * crs];
      open = c.x < 0 && __int_as_float(c.y) <= T && !(__int_as_float(c.y) > dbest);
    }
    if (open) {
      rowlist[(int64_t)pair * J + atomicAdd(ovf + pair, 1)] = j;
      return;
    }
  }
  idx[row] = (int32_t)(best & 0xffffffffull);
}

// running totals of a context (dsir_screen_stats): searches, rows searched, rows left to the exhaustive kernel, pairs
// searched exhaustively as a whole
__device__ __forceinline__ void screen_account(const int32_t* __restrict__ ovf, int pairs, int ovf_min, int J,
                                               unsigned long long* __restrict__ acc) {
  unsigned long long rows = 0, full = 0;
  for (int p = threadIdx.x; p < pairs; p += blockDim.x) {
    const int g = ovf[p];
    rows += (unsigned long long)(g >= ovf_min ? J : g);
    full += g >= ovf_min ? 1ull : 0ull;
  }
  if (rows) atomicAdd(acc + 2, rows);
  if (full) atomicAdd(acc + 3, full);
  if (threadIdx.x == 0) { atomicAdd(acc, 1ull); atomicAdd(acc + 1, (unsigned long long)pairs * (unsigned long long)J); }
}

// results of the exhaustive kernel -> idx: every row of a pair with ovf_min or more listed rows, else the listed rows
// (its first workgroup also adds the search to the context's running totals, acc - what screen_account_kernel did in a launch of
// its own: the listed-row counts are final once the exhaustive kernel has been queued)
__global__ __launch_bounds__(256) void unpack_listed_kernel(const unsigned long long* __restrict__ packed,
                                                            const int32_t* __restrict__ ovf, int ovf_min,
                                                            const int32_t* __restrict__ rowlist, int J,
                                                            int32_t* __restrict__ idx, unsigned long long* __restrict__ acc) {
  const int pair = blockIdx.y;
  const int g = ovf[pair];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int64_t base = (int64_t)pair * J;
  if (g >= ovf_min) {
    if (i < J) idx[base + i] = packed_index(packed[base + i]);
  } else if (i < g) {
    const int r = rowlist[base + i];
    idx[base + r] = packed_index(packed[base + r]);
  }
  if (acc && blockIdx.x == 0 && blockIdx.y == 0) screen_account(ovf, (int)gridDim.y, ovf_min, J, acc);
}

// diagnostics: out[0] = total entries, out[1] = rows left to the exhaustive kernel
__global__ void screen_stats_kernel(const int32_t* __restrict__ cnt, int64_t rows, int J, const int32_t* __restrict__ ovf,
                                    int ovf_min, unsigned long long* __restrict__ out) {
  unsigned long long c = 0, o = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = cnt[i];
    c += (unsigned long long)(n > 0 ? n : 0);
    if (i % J == 0) { const int g = ovf[i / J]; o += (unsigned long long)(g >= ovf_min ? J : g); }
  }
  if (c) atomicAdd(out, c);
  if (o) atomicAdd(out + 1, o);
}

// between the searches of one registration: a pair found not selective stays so (its descriptors barely change from one
// iteration to the next), every other counter restarts.  Input outside the bound's domain: every pair exhaustive.
// Also the per-row scratch of one search: threshold = +max, entry count = 0, packed result = all ones.
__global__ void screen_reset_kernel(int32_t* __restrict__ ovf, int pairs, int ovf_min, int keep, const int32_t* __restrict__ bad,
                                    unsigned int* __restrict__ umin, int32_t* __restrict__ cnt,
                                    unsigned long long* __restrict__ packed, int64_t rows) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 < pairs) ovf[i0] = (bad && *bad) ? ovf_min : ((keep && ovf[i0] >= ovf_min) ? ovf[i0] : 0);
  for (int64_t i = i0; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    umin[i] = 0xffffffffu;
    cnt[i] = 0;
    packed[i] = ~0ull;
  }
}

// ---- diagnostics (dsir_screen_bounds): the screening's arithmetic laid bare for EVERY (row, column) of one small pair.
// One wave per 16 x 16 tile runs the SAME chain of six v_mfma_f32_16x16x32_f16 as screen_item (same operands, same
// order, same seed 2^22 c) and the epilogue's arithmetic for the lower bound L and the upper bound U = L + 2 d, and evaluates the exact
// distance D beside them, so that a test can assert L <= D <= U entry by entry on adversarial inputs - the
// property the whole screened path rests on, checked on the matrix core itself rather than derived from an assumed
// rounding model.  tests/test_gpu_screen_bound.py also ties this kernel to the product: every candidate entry
// screen_kernel emits must carry the bits of this kernel's L at its (row, column).
__global__ __launch_bounds__(64) void screen_bounds_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                           const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                                           const float* __restrict__ A, const float* __restrict__ B,
                                                           const float* __restrict__ sa, const float* __restrict__ sb, int J, int K,
                                                           float* __restrict__ lower, float* __restrict__ upper,
                                                           float* __restrict__ exact, float* __restrict__ zacc) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
  const int arow = min(row0 + fr, J - 1), bcol = min(col0 + fr, K - 1);
  h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    ah[c] = *reinterpret_cast<const h8*>(Ah + (int64_t)arow * 64 + 32 * c + 8 * fq);
    al[c] = *reinterpret_cast<const h8*>(Al + (int64_t)arow * 64 + 32 * c + 8 * fq);
    bh[c] = *reinterpret_cast<const h8*>(Bh + (int64_t)bcol * 64 + 32 * c + 8 * fq);
    bl[c] = *reinterpret_cast<const h8*>(Bl + (int64_t)bcol * 64 + 32 * c + 8 * fq);
  }
  const float sbk = sb[bcol];
  const float seed = -2097152.f * (sbk - kC1 * sbk);          // screen_item's gload: 2^22 c, c = -(|b|^2 - d_b) / 2
  f32x4 z = f32x4{seed, seed, seed, seed};
  // screen_item's step(), one row tile: (ah0,bh0) (ah1,bh1) (ah0,bl0) (al0,bh0) (ah1,bl1) (al1,bh1)
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh[0], z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh[1], z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl[0], z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh[0], z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl[1], z, 0, 0, 0);
  z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh[1], z, 0, 0, 0);
  const int col = col0 + fr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * fq + r;
    if (row >= J || col >= K) continue;
    const float san = sa[row];
    const float slo = san - kC1 * san - kC0;                                  // the epilogue of screen_item, verbatim
    const float l1 = fmaf(z[r], -4.76837158203125e-7f, slo);
    const float u = l1 + kW * (kC1 * (san + sbk) + kC0);
    float4 a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = reinterpret_cast<const float4*>(A + (int64_t)row * 64)[q];
    const int64_t o = (int64_t)row * K + col;
    lower[o] = l1;
    upper[o] = u;
    if (zacc) zacc[o] = z[r];                                                 // the raw accumulator 2^22 (c + a.b)
    exact[o] = exact_dist(a, B + (int64_t)col * 64, san, sbk);
  }
}

// the product's candidate lists of one pair, out of launch_nn_screen's scratch
__global__ void screen_export_kernel(const unsigned int* __restrict__ umin, const int32_t* __restrict__ cnt,
                                     const int2* __restrict__ cand, int J, float* __restrict__ thresh,
                                     int32_t* __restrict__ count, int32_t* __restrict__ code, float* __restrict__ lower) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  thresh[j] = unorder_bits(umin[j]);
  const int n = cnt[j];
  count[j] = n;
  for (int e = 0; e < CAP; ++e) {
    const bool live = e < n;
    code[j * CAP + e] = live ? cand[(int64_t)e * J + j].x : 0;             // one pair: the launch had J rows
    lower[j * CAP + e] = live ? __int_as_float(cand[(int64_t)e * J + j].y) : 0.f;
  }
}

// ---- the bound pass of the pruned search (nn_prune.hip) on the matrix core: which column tiles must a row block visit?
// A tile t (64 ref columns, centroid c_t, radius r_t) can be skipped by a row whose minimum is known to be <= T iff
// (|a - c_t| - r_t)_+^2 > T.  |a - c_t|^2 is bounded from BELOW by this file's own screening bound: the centroids are split like
// descriptors and L(a, c_t) <= D(a, c_t) comes out of the same six-MFMA chain (rows x nt centroids: 1/64 of a search).
// Block = 8 waves x RT row tiles = one row block of the screening (rows in the given order); per 16-centroid step every lane
// tests its column against its 4 RT rows, the wave folds the answers into a 16-bit column mask, and the tiles some row needs
// are compacted into the block's list.
template <int RT>
__global__ __launch_bounds__(512) void tile_bound_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                         const float* __restrict__ sa, const int32_t* __restrict__ rows,
                                                         const float* __restrict__ T, const _Float16* __restrict__ Ch,
                                                         const _Float16* __restrict__ Cl, const float* __restrict__ cn2,
                                                         const float* __restrict__ rad, int J, int nt, int32_t* __restrict__ tlist,
                                                         int32_t* __restrict__ tcount, int tl_stride) {
  __shared__ unsigned int flags[kMaxBoundTiles / 16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int rb = blockIdx.x, pair = blockIdx.y, nrb = gridDim.x;
  const int64_t arow = (int64_t)pair * J;
  const int row0 = (rb * 8 + w) * (16 * RT);
  const int nsteps = (nt + 15) >> 4;
  for (int i = tid; i < nsteps; i += 512) flags[i] = 0u;
  h8 ah[RT][2], al[RT][2];
  // skip iff (sqrt(lo) 0.99999 - r)_+^2 > T, tested without a square root per element as lo > ((sqrt(T) + r) k)^2, k = 1.00002 (the
  // factor 1 / 0.99999 and the rounding of sqrtf): slo2 = |a|^2 - d_a - margin per row (+inf for rows past the end: they need
  // nothing), sT = sqrt(T) k per row, r k per column
  float slo2[RT][4], sT[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int ra = rows[arow + min(row0 + rt * 16 + fr, J - 1)];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ah[rt][c] = *reinterpret_cast<const h8*>(Ah + (arow + ra) * 64 + 32 * c + 8 * fq);
      al[rt][c] = *reinterpret_cast<const h8*>(Al + (arow + ra) * 64 + 32 * c + 8 * fq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pos = row0 + rt * 16 + 4 * fq + r;
      const int re = rows[arow + min(pos, J - 1)];
      const float san = sa[arow + re];
      slo2[rt][r] = pos < J ? (san - kC1 * san - kC0) - 1e-5f * (1.f + san) : INFINITY;
      sT[rt][r] = pos < J ? sqrtf(T[arow + re]) * 1.00002f : 0.f;      // T < 0 or NaN: NaN - the row visits every tile
    }
  }
  __syncthreads();
  const _Float16* ch = Ch + (int64_t)pair * nt * 64;
  const _Float16* cl = Cl + (int64_t)pair * nt * 64;
  // the next step's centroid fragments travel while the current step's MFMAs run (a step's own loads were exposed: 24 MFMAs
  // cannot start before four 16-byte loads and two scalars have come back)
  struct CFrag { h8 bh0, bh1, bl0, bl1; float c2, rad; };
  auto cload = [&](int s) {
    const int tc = min(16 * s + fr, nt - 1);
    CFrag f;
    f.bh0 = *reinterpret_cast<const h8*>(ch + (int64_t)tc * 64 + 8 * fq); f.bh1 = *reinterpret_cast<const h8*>(ch + (int64_t)tc * 64 + 32 + 8 * fq);
    f.bl0 = *reinterpret_cast<const h8*>(cl + (int64_t)tc * 64 + 8 * fq); f.bl1 = *reinterpret_cast<const h8*>(cl + (int64_t)tc * 64 + 32 + 8 * fq);
    f.c2 = cn2[(int64_t)pair * nt + tc];
    f.rad = rad[(int64_t)pair * nt + tc];
    return f;
  };
  CFrag nxt = cload(0);
  for (int s = 0; s < nsteps; ++s) {
    const CFrag cur = nxt;
    if (s + 1 < nsteps) nxt = cload(s + 1);
    const int t = 16 * s + fr;
    const h8 bh0 = cur.bh0, bh1 = cur.bh1, bl0 = cur.bl0, bl1 = cur.bl1;
    const float c2 = cur.c2;
    const float rk = cur.rad * 1.00002f;
    const float cterm = 1e-5f * c2;
    const float seed = t < nt ? -2097152.f * (c2 - kC1 * c2) : -INFINITY;      // tiles past the end: L = +inf (their flags are not read)
    bool need = false;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 z = f32x4{seed, seed, seed, seed};
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][0], bh0, z, 0, 0, 0);      // screen_item's chain
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][1], bh1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][0], bl0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][0], bh0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][1], bl1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][1], bh1, z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // L <= D(a, c) (the screening bound), D(a, c) within 1e-5 (1 + |a|^2 + |c|^2) of the true |a - c|^2: a lower bound of that
        const float lo = fmaf(z[r], -4.76837158203125e-7f, slo2[rt][r]) - cterm;
        const float thr = sT[rt][r] + rk;
        need = need || !(lo > thr * thr);                                        // NaN anywhere: visit
      }
    }
    const unsigned long long m = __ballot(need);
    const unsigned int m16 = (unsigned int)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffull);
    if (lane == 0 && m16) atomicOr(&flags[s], m16);
  }
  __syncthreads();
  if (tid < 64) {
    int32_t* out = tlist + ((int64_t)pair * nrb + rb) * tl_stride;
    int cnt = 0;
    for (int base = 0; base < nt; base += 64) {
      const int t = base + tid;
      const bool f = t < nt && ((flags[t >> 4] >> (t & 15)) & 1u);
      const unsigned long long m = __ballot(f);
      if (f) out[cnt + __popcll(m & ((1ull << tid) - 1ull))] = t;
      cnt += __popcll(m);
    }
    if (tid == 0) tcount[pair * nrb + rb] = cnt;
  }
}

// ---- where does a row look first?  The tile whose CENTROID is nearest (smallest screening lower bound L(a, c_t)), per src row, rows
// in their natural order.  That tile orders the rows (rows that start in the same tile sit in the same row block and agree on
// which tiles matter) and supplies an upper bound of the row's minimum that does not need a previous iteration (tile_T_kernel).
// Same MFMA chain and operands as tile_bound_kernel; L = slo_row - 2^-21 z', so the arg-min of L over t is the arg-max of z'.
template <int RT>
__global__ __launch_bounds__(512) void centroid_argmin_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                              const _Float16* __restrict__ Ch, const _Float16* __restrict__ Cl,
                                                              const float* __restrict__ cn2, int J, int nt, int32_t* __restrict__ tstar) {
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int rb = blockIdx.x, pair = blockIdx.y;
  const int64_t arow = (int64_t)pair * J;
  const int row0 = (rb * 8 + w) * (16 * RT);
  if (row0 >= J) return;                                 // wave-uniform; no barrier in this kernel
  h8 ah[RT][2], al[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int ra = min(row0 + rt * 16 + fr, J - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ah[rt][c] = *reinterpret_cast<const h8*>(Ah + (arow + ra) * 64 + 32 * c + 8 * fq);
      al[rt][c] = *reinterpret_cast<const h8*>(Al + (arow + ra) * 64 + 32 * c + 8 * fq);
    }
  }
  float bz[RT][4];
  int bt[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bz[rt][r] = -INFINITY; bt[rt][r] = 0; }
  const _Float16* ch = Ch + (int64_t)pair * nt * 64;
  const _Float16* cl = Cl + (int64_t)pair * nt * 64;
  const int nsteps = (nt + 15) >> 4;
  for (int s = 0; s < nsteps; ++s) {
    const int t = 16 * s + fr;
    const int tc = min(t, nt - 1);
    const h8 bh0 = *reinterpret_cast<const h8*>(ch + (int64_t)tc * 64 + 8 * fq), bh1 = *reinterpret_cast<const h8*>(ch + (int64_t)tc * 64 + 32 + 8 * fq);
    const h8 bl0 = *reinterpret_cast<const h8*>(cl + (int64_t)tc * 64 + 8 * fq), bl1 = *reinterpret_cast<const h8*>(cl + (int64_t)tc * 64 + 32 + 8 * fq);
    const float c2 = cn2[(int64_t)pair * nt + tc];
    const float seed = t < nt ? -2097152.f * (c2 - kC1 * c2) : -INFINITY;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 z = f32x4{seed, seed, seed, seed};
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][0], bh0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][1], bh1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][0], bl0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][0], bh0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][1], bl1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][1], bh1, z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (z[r] > bz[rt][r]) { bz[rt][r] = z[r]; bt[rt][r] = t; }        // NaN never wins; first (lowest) tile on ties within a lane
    }
  }
  // the 16 lanes of a row group hold its column classes: larger z' wins, the lower tile on ties (any tile is a valid choice)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z = bz[rt][r];
      int t = bt[rt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float zo = __shfl_xor(z, o);
        const int to = __shfl_xor(t, o);
        if (zo > z || (zo == z && to < t)) { z = zo; t = to; }
      }
      const int row = row0 + rt * 16 + 4 * fq + r;
      if (fr == 0 && row < J) tstar[arow + row] = t;
    }
}

// ---- an upper bound of every row's minimum from the tiles its 16-row group points at.  One wave per 16 consecutive rows of the row
// ORDER (rows sorted by their nearest-centroid tile: a group points at one or two tiles): for every distinct tile among the
// group's rows the screening chain on (16 rows x 64 columns of the ref operands in column order), U = L + 2 d >= D(row, column)
// for every (row, column) - the screening's proven upper bound of the exact fp32 distance -, so min U over ANY columns bounds the
// row minimum from above.  T[row] = min(T[row], min U + margin): T arrives holding the bound from the previous iteration's match
// (or +inf in iteration 0) and leaves as what tile_bound_kernel prunes against.  (64-row groups - a tile's operands read once per
// 64 rows - measured twice as slow: 4 x fewer waves with 3 x longer dependent chains; the kernel is latency-, not bandwidth-bound.)
__global__ __launch_bounds__(256) void tile_T_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                     const float* __restrict__ sa, const int32_t* __restrict__ rows,
                                                     const int32_t* __restrict__ tstar, const _Float16* __restrict__ Bh,
                                                     const _Float16* __restrict__ Bl, const float* __restrict__ sbp, int J, int K, int nt,
                                                     float* __restrict__ T) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int pair = blockIdx.y;
  const int g0 = (blockIdx.x * 4 + w) * 16;              // first position of the wave's group in the row order
  if (g0 >= J) return;                                   // wave-uniform; no barrier in this kernel
  const int64_t arow = (int64_t)pair * J, brow = (int64_t)pair * K;
  const int ra = rows[arow + min(g0 + fr, J - 1)];        // the row whose A fragment this lane holds
  h8 ah[2], al[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    ah[c] = *reinterpret_cast<const h8*>(Ah + (arow + ra) * 64 + 32 * c + 8 * fq);
    al[c] = *reinterpret_cast<const h8*>(Al + (arow + ra) * 64 + 32 * c + 8 * fq);
  }
  int rowe[4];
  float san[4], slo[4], best[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rowe[r] = rows[arow + min(g0 + 4 * fq + r, J - 1)];
    san[r] = sa[arow + rowe[r]];
    slo[r] = san[r] - kC1 * san[r] - kC0;
    best[r] = INFINITY;
  }
  int mine = tstar[arow + ra];                            // the tile this lane's row points at (lanes fr, all four fq copies)
  mine = mine < 0 ? 0 : (mine >= nt ? nt - 1 : mine);
  bool pending = g0 + fr < J;
  for (int guard = 0; guard < 16; ++guard) {              // at most 16 distinct tiles per group
    // the lowest tile some lane still waits for
    int t = pending ? mine : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) t = min(t, __shfl_xor(t, o));
    if (t == 0x7fffffff) break;                           // wave-uniform
    if (mine == t) pending = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int pos = min(64 * t + 16 * s + fr, K - 1);
      const bool live = 64 * t + 16 * s + fr < K;
      const _Float16* bh = Bh + (brow + pos) * 64;
      const _Float16* bl = Bl + (brow + pos) * 64;
      const h8 bh0 = *reinterpret_cast<const h8*>(bh + 8 * fq), bh1 = *reinterpret_cast<const h8*>(bh + 32 + 8 * fq);
      const h8 bl0 = *reinterpret_cast<const h8*>(bl + 8 * fq), bl1 = *reinterpret_cast<const h8*>(bl + 32 + 8 * fq);
      const float sbk = sbp[brow + pos];
      const float seed = -2097152.f * (sbk - kC1 * sbk);
      f32x4 z = f32x4{seed, seed, seed, seed};
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh0, z, 0, 0, 0);         // screen_item's chain
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh0, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl1, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh1, z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float l = fmaf(z[r], -4.76837158203125e-7f, slo[r]);
        // U = L + 2 d (the epilogue of screen_item) + the margin row_prep_kernel puts on an exact distance
        const float u = l + kW * (kC1 * (san[r] + sbk) + kC0) + 2e-5f * (1.f + san[r] + sbk);
        if (live) best[r] = fminf(best[r], u);           // NaN: ignored (a row without finite bound keeps +inf: it visits every tile)
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float b = best[r];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) b = fminf(b, __shfl_xor(b, o));
    if (fr == 0 && g0 + 4 * fq + r < J) {
      const float old = T[arow + rowe[r]];
      T[arow + rowe[r]] = fminf(old, b);                   // one writer per row
    }
  }
}

// the row blocks of every pair by descending tile count (ties: by index): the persistent search takes the long items first, so
// that its last round is made of short ones (longest-processing-time-first; the tail of a launch was up to one full-length item,
// 18 % of the kernel at 65536 points).  One workgroup per pair, rank by counting - a pair has at most a few hundred row blocks
__global__ __launch_bounds__(256) void rank_blocks_kernel(const int32_t* __restrict__ tcount, int nrb, int32_t* __restrict__ rborder) {
  const int32_t* c = tcount + (int64_t)blockIdx.x * nrb;
  int32_t* out = rborder + (int64_t)blockIdx.x * nrb;
  for (int i = threadIdx.x; i < nrb; i += 256) {
    const int ci = c[i];
    int rank = 0;
    for (int j = 0; j < nrb; ++j) { const int cj = c[j]; rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
    out[rank] = i;
  }
}

inline int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

int nn_screen_max_bound_tiles() { return kMaxBoundTiles; }

// nearest-centroid tile of every src row (natural row order)
void launch_centroid_argmin(const void* ah, const void* al, const void* ch, const void* cl, const float* cn2, int pairs, int J, int nt,
                            int32_t* tstar, hipStream_t st) {
  const int rpb = nn_screen_rows_per_block(J);
  const dim3 grid((J + rpb - 1) / rpb, pairs);
  const _Float16 *Ah = reinterpret_cast<const _Float16*>(ah), *Al = reinterpret_cast<const _Float16*>(al);
  const _Float16 *Ch = reinterpret_cast<const _Float16*>(ch), *Cl = reinterpret_cast<const _Float16*>(cl);
  if (rpb == 512) hipLaunchKernelGGL(centroid_argmin_kernel<4>, grid, dim3(512), 0, st, Ah, Al, Ch, Cl, cn2, J, nt, tstar);
  else            hipLaunchKernelGGL(centroid_argmin_kernel<2>, grid, dim3(512), 0, st, Ah, Al, Ch, Cl, cn2, J, nt, tstar);
}

// T[row] = min(T[row], upper bound from the tiles the row's 16-row group (in the row order) points at); bh / bl / sbp: the ref
// operands in column order
void launch_tile_T(const void* ah, const void* al, const float* sa, const int32_t* rows, const int32_t* tstar, const void* bh, const void* bl,
                   const float* sbp, int pairs, int J, int K, int nt, float* T, hipStream_t st) {
  hipLaunchKernelGGL(tile_T_kernel, dim3((J + 63) / 64, pairs), dim3(256), 0, st, reinterpret_cast<const _Float16*>(ah),
                     reinterpret_cast<const _Float16*>(al), sa, rows, tstar, reinterpret_cast<const _Float16*>(bh),
                     reinterpret_cast<const _Float16*>(bl), sbp, J, K, nt, T);
}

// tile lists of every (pair, row block) for the row order `rows` and the per-row upper bounds T (nn_prune.hip)
void launch_tile_bound(const void* ah, const void* al, const float* sa, const int32_t* rows, const float* T, const void* ch, const void* cl,
                       const float* cn2, const float* rad, int pairs, int J, int nt, int32_t* tlist, int32_t* tcount, int tl_stride,
                       int32_t* rborder, hipStream_t st) {
  const int rpb = nn_screen_rows_per_block(J);
  const dim3 grid((J + rpb - 1) / rpb, pairs);
  const _Float16 *Ah = reinterpret_cast<const _Float16*>(ah), *Al = reinterpret_cast<const _Float16*>(al);
  const _Float16 *Ch = reinterpret_cast<const _Float16*>(ch), *Cl = reinterpret_cast<const _Float16*>(cl);
  if (rpb == 512) hipLaunchKernelGGL(tile_bound_kernel<4>, grid, dim3(512), 0, st, Ah, Al, sa, rows, T, Ch, Cl, cn2, rad, J, nt, tlist, tcount, tl_stride);
  else            hipLaunchKernelGGL(tile_bound_kernel<2>, grid, dim3(512), 0, st, Ah, Al, sa, rows, T, Ch, Cl, cn2, rad, J, nt, tlist, tcount, tl_stride);
  if (rborder) hipLaunchKernelGGL(rank_blocks_kernel, dim3(pairs), dim3(256), 0, st, tcount, (int)grid.x, rborder);
}

int nn_screen_cap() { return CAP; }

// row tiles per wave: four (512-row blocks: every ref fragment read from LDS feeds four MFMA chains, -12 % kernel time at
// J = 5000) unless the padding of J to whole blocks costs more than that
int nn_screen_rows_per_block(int J) {
  constexpr int NWV = DSIR_SCREEN_NWV;
  static const int force_rt = (int)tuning_int("DSIR_SCREEN_RT", DSIR_SCREEN_RT);   // A/B hook
  auto padded = [&](int rt) { const int64_t r = NWV * 16 * rt; return ((J + r - 1) / r) * r; };
  const int RT = force_rt == 2 || force_rt == 4 ? force_rt : (0.88 * (double)padded(4) <= (double)padded(2) ? 4 : 2);
  return NWV * 16 * RT;
}

void launch_screen_bounds(const float* a, const float* b, const void* ah, const void* al, const void* bh, const void* bl,
                          const float* sa, const float* sb, int J, int K, float* lower, float* upper, float* exact,
                          float* zacc, hipStream_t st) {
  hipLaunchKernelGGL(screen_bounds_kernel, dim3((K + 15) / 16, (J + 15) / 16), dim3(64), 0, st,
                     reinterpret_cast<const _Float16*>(ah), reinterpret_cast<const _Float16*>(al),
                     reinterpret_cast<const _Float16*>(bh), reinterpret_cast<const _Float16*>(bl), a, b, sa, sb, J, K, lower, upper,
                     exact, zacc);
}

// scratch = what launch_nn_screen ran on with pairs = 1 (layout: see nn_screen_scratch_bytes)
void launch_screen_export(const void* scratch, int J, float* thresh, int32_t* count, int32_t* code, float* lower, hipStream_t st) {
  const size_t rows = (size_t)J;
  const char* p = reinterpret_cast<const char*>(scratch);
  auto take = [&](size_t bytes) { const char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  const unsigned int* umin = reinterpret_cast<const unsigned int*>(take(rows * 4));
  const int32_t* cnt = reinterpret_cast<const int32_t*>(take(rows * 4));
  const int2* cand = reinterpret_cast<const int2*>(take(rows * CAP * 8));
  hipLaunchKernelGGL(screen_export_kernel, dim3((J + 255) / 256), dim3(256), 0, st, umin, cnt, cand, J, thresh, count, code, lower);
}

// scratch: Umin u32 [rows] | cnt i32 [rows] | cand {col, lower bound} [rows][CAP] | undecidable rows per pair i32 [pairs] |
//          packed results of the exhaustive fallback u64 [rows] | list of the undecidable rows i32 [rows]
size_t nn_screen_scratch_bytes(int pairs, int J) {
  const size_t rows = (size_t)pairs * J;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return al(rows * 4) * 3 + al(rows * CAP * 8) + al((size_t)pairs * 4) + al(rows * 8);
}

void launch_split16(const float* x, int64_t rows, void* hi, void* lo, hipStream_t st, int32_t* bad) {
  const int64_t n4 = rows * 16;
  hipLaunchKernelGGL(split16_kernel, dim3(grid_for(n4)), dim3(256), 0, st, x, n4, reinterpret_cast<_Float16*>(hi),
                     reinterpret_cast<_Float16*>(lo), bad);
}

void launch_split16_norm(const float* x, int64_t rows, void* hi, void* lo, float* sq, hipStream_t st, int32_t* bad) {
  hipLaunchKernelGGL(split_norm_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, x, rows,
                     reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo), sq, bad);
}

// a, b: fp32 descriptors [pairs][J|K][64] with their fp16 splits (ah, al, bh, bl) and squared norms (sa, sb)
void launch_nn_screen(const float* a, const float* b, const void* ah, const void* al, const void* bh, const void* bl,
                      const float* sa, const float* sb, int pairs, int J, int K, int32_t* idx, void* scratch, hipStream_t st,
                      hipEvent_t ev0, hipEvent_t ev1, unsigned long long* stats, bool keep_gate, const int32_t* bad,
                      unsigned long long* acc, hipEvent_t evk0, hipEvent_t evk1, const ScreenOrder& ord_in) {
  ScreenOrder ord = ord_in;
  ord.cand_rs = (long long)pairs * J;
  const size_t rows = (size_t)pairs * J;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
  unsigned int* umin = reinterpret_cast<unsigned int*>(take(rows * 4));
  int32_t* cnt = reinterpret_cast<int32_t*>(take(rows * 4));
  int2* cand = reinterpret_cast<int2*>(take(rows * CAP * 8));
  int32_t* ovf = reinterpret_cast<int32_t*>(take((size_t)pairs * 4));
  unsigned long long* packed = reinterpret_cast<unsigned long long*>(take(rows * 8));
  int32_t* rowlist = reinterpret_cast<int32_t*>(take(rows * 4));
  // Screening that is not selective (descriptors closer to each other than the bound width) leaves rows undecided: they
  // are searched by the exhaustive fp32 MFMA kernel - row by row through a list, or the whole pair once a quarter of its
  // rows is affected (then the pair also skips the screening in the remaining iterations of the registration).
  static const int force_min = (int)tuning_int("DSIR_SCREEN_OVF_MIN", 0);   // tuning/test hook
  const int ovf_min = force_min > 0 ? force_min : J / 4 + 1;
  if (ev0) (void)hipEventRecord(ev0, st);
  {
    const int64_t need = ((int64_t)rows + 255) / 256, min_blocks = (pairs + 255) / 256;
    const int blocks = (int)(need > 2048 ? 2048 : (need < min_blocks ? min_blocks : need));
    hipLaunchKernelGGL(screen_reset_kernel, dim3(blocks), dim3(256), 0, st, ovf, pairs, ovf_min, keep_gate ? 1 : 0, bad, umin, cnt,
                       packed, (int64_t)rows);
  }
  constexpr int NWV = DSIR_SCREEN_NWV;   // waves per block: they share one staged ref tile (the L2 -> LDS fill is the scarce resource)
  // row tiles per wave: four (512-row blocks: every ref fragment read from LDS feeds four MFMA chains, -12 % kernel time
  // at J = 5000) unless the padding of J to whole blocks costs more than that
  const int rows_per_block = nn_screen_rows_per_block(J);
  const int RT = rows_per_block / (NWV * 16);
  const int rb_count = (J + rows_per_block - 1) / rows_per_block;
  const int64_t base = (int64_t)pairs * rb_count;
  const int tiles = (K + SBC - 1) / SBC;
  const int resident = 256 * (DSIR_SCREEN_WPE * 4 / NWV);   // workgroups the chip holds at once
  int splits = 1;
  double best_eff = -1.0;
  // at most 8 ref splits: every split emits at least one entry per row (its own best column), and a row holds CAP = 16
  // before it overflows into the exhaustive kernel (16 splits: measured 2 x slower end to end at 65536 points)
  for (int sp = 1; sp <= 8 && sp <= tiles; ++sp) {
    const int tiles_per = (tiles + sp - 1) / sp;
    if (sp > 1 && tiles_per < 8) break;
    const int nsp = (tiles + tiles_per - 1) / tiles_per;
    const int64_t blocks = base * nsp;
    const int64_t rounds = (blocks + resident - 1) / resident;
    double eff = (double)blocks / (double)(rounds * resident);
    eff *= (double)tiles / (double)(tiles_per * nsp);
    if (eff > best_eff + 1e-9) { best_eff = eff; splits = sp; }
  }
  // long ref ranges: a second / fourth workgroup per range doubles the lane classes a row's candidates fall into (fewer
  // class collisions -> fewer rows for the exhaustive kernel) and costs one more prologue + epilogue per >= 128 tiles
  // (16384 points: +1.4 % pairs/s, 65536: +3.9 %; at 5000 points - 79 tiles - splitting loses 8 % of the kernel)
  while (splits * 2 <= 4 && tiles / (splits * 2) >= 128) splits *= 2;
  static const int force_splits = (int)tuning_int("DSIR_SCREEN_SPLITS", 0);   // tuning hook
  if (force_splits > 0) splits = force_splits < tiles ? force_splits : tiles;
  const int cols = ((tiles + splits - 1) / splits) * SBC;
  splits = (K + cols - 1) / cols;
  const int total = (int)((int64_t)rb_count * splits * pairs);
  static const int persist = (int)tuning_int("DSIR_SCREEN_PERSIST", 0);   // workgroups per CU; 0 = one per item
  const bool ordered = ord.tlist && ord.queue;         // launch_prune_rows fills every field or none
  const dim3 grid(ordered ? (unsigned)(total < resident ? total : resident)             // persistent, per-XCD item queues (screen_kernel)
                          : (unsigned)(persist > 0 && total > persist * resident ? persist * resident : total));
  const _Float16 *Ah = reinterpret_cast<const _Float16*>(ah), *Al = reinterpret_cast<const _Float16*>(al);
  const _Float16 *Bh = reinterpret_cast<const _Float16*>(bh), *Bl = reinterpret_cast<const _Float16*>(bl);
  if (evk0) (void)hipEventRecord(evk0, st);
  if (ordered) {                                       // the pruned search streams the ref side's copies in column order
    if (ord.cols) { Bh = reinterpret_cast<const _Float16*>(ord.bh); Bl = reinterpret_cast<const _Float16*>(ord.bl); }
    else ord.sbp = sb;
  }
#define DSIR_LAUNCH_SCREEN(RTV, ORDV)                                                                                                  \
  hipLaunchKernelGGL((screen_kernel<RTV, NWV, ORDV>), grid, dim3(NWV * 64), 0, st, Ah, Al, Bh, Bl, sa, sb, J, K, cols, rb_count, splits, \
                     umin, cnt, cand, ovf, rowlist, ovf_min, total, ord)
  if (RT == 4) { if (ordered) DSIR_LAUNCH_SCREEN(4, true); else DSIR_LAUNCH_SCREEN(4, false); }
  else         { if (ordered) DSIR_LAUNCH_SCREEN(2, true); else DSIR_LAUNCH_SCREEN(2, false); }
#undef DSIR_LAUNCH_SCREEN
  if (evk1) (void)hipEventRecord(evk1, st);
  hipLaunchKernelGGL(exact_pick_kernel, dim3((J + 255) / 256, pairs), dim3(256), 0, st, a, b, sa, sb, J, K, umin, cnt, cand, ovf,
                     ovf_min, rowlist, idx);
  launch_nn_match_gated(a, b, sa, sb, pairs, J, K, packed, ovf, ovf_min, rowlist, st);
  hipLaunchKernelGGL(unpack_listed_kernel, dim3((J + 255) / 256, pairs), dim3(256), 0, st, packed, ovf, ovf_min, rowlist, J, idx, acc);
  if (ev1) (void)hipEventRecord(ev1, st);
  static const bool debug = tuning_flag("DSIR_SCREEN_DEBUG");   // diagnostic: rows left to the exhaustive kernel (synchronises)
  if (debug) {
    std::vector<int32_t> h(pairs);
    (void)hipMemcpyAsync(h.data(), ovf, (size_t)pairs * 4, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    long sum = 0; int mx = 0, full = 0;
    for (int v : h) { sum += v; mx = v > mx ? v : mx; full += v >= ovf_min; }
    fprintf(stderr, "[nn_screen] pairs %d J %d K %d splits %d: listed rows %ld (max %d per pair), %d pairs exhaustive\n", pairs, J, K,
            splits, sum, mx, full);
  }
  if (stats) {
    (void)hipMemsetAsync(stats, 0, 16, st);
    hipLaunchKernelGGL(screen_stats_kernel, dim3(256), dim3(256), 0, st, cnt, (int64_t)rows, J, ovf, ovf_min, stats);
  }
}

}  // namespace dsir
