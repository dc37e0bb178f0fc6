// Small device-side helpers shared by the kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsir {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DSIR_GN_WORDS_DEFINED
#define DSIR_GN_WORDS_DEFINED
constexpr int kGnWords = 4;   // 8-byte words per (cloud, group) GroupNorm statistics slot (gn_block_commit / gn_stat_get below)
#endif

// Wave-uniform base + 32-bit per-lane BYTE offset: compiles to the SGPR-base addressing mode
// (global_load_dword v, v_off, s[base:base+1]) — one 32-bit multiply-add per address instead of a 64-bit
// multiply-add chain.  Every per-cloud tensor of the engine is far below 4 GiB.
__device__ __forceinline__ float ld_f32(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ int ld_i32(const int32_t* base, uint32_t byte_off) {
  return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st_f32(float* base, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// XCD-aware work mapping: the hardware deals workgroups round-robin over the 8 XCDs (id & 7), each with its own L2.  This
// bijection of [0, nwg) hands every XCD a CONTIGUOUS range of logical work items, so that items ordered (cloud, block) keep
// a cloud's gathered rows in ONE L2 instead of replicating them through eight.  Speed only: which workgroup computes an
// item never changes its result.
__device__ __forceinline__ int xcd_contiguous(int id, int nwg) {
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
}

// LDS handed to a device body as a byte region (pw_tile_body.h, att_pool_body.h, misc_body.h): the body carves its arrays in a fixed
// order, every piece padded to 16 bytes.  With the region a static __shared__ array of the calling kernel the addresses fold to
// constants exactly as named __shared__ arrays would.
constexpr size_t smem_pad(size_t bytes) { return (bytes + 15) & ~(size_t)15; }
template <typename T>
__device__ __forceinline__ T* smem_carve(char*& p, size_t count) {
  T* r = reinterpret_cast<T*>(p);
  p += smem_pad(sizeof(T) * count);
  return r;
}

// index half of a packed (order-preserving distance bits << 32 | column) arg-min slot.  A slot still at its preset
// (all ones: every distance of the row was NaN, nothing ever won) yields 0, never -1: consumers gather by it.
__device__ __forceinline__ int32_t packed_index(unsigned long long p) {
  return p == ~0ull ? 0 : (int32_t)(p & 0xffffffffull);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// exp(x) for x <= 0 (softmax numerators, x = score - max): 2^(x*log2e) with the hardware v_exp_f32 (1 ulp).
// The rounding of the product x*log2e adds a relative error of at most |x| * 6e-8 — below the noise the scores
// themselves carry (an fp32 GEMM over 32..256 channels), and only on terms that are already e^x of the
// denominator.  Two instructions per element instead of six.
__device__ __forceinline__ float exp_neg(float x) {
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}

// ------------------------------------------------------------------ GroupNorm statistics across workgroups: DETERMINISTIC
// The per-(cloud, group) sums of x and x^2 meet across workgroups in atomics whose result cannot depend on the arrival order:
// every contribution v (a workgroup's partial sum, fp64) is split by a pure function into two INTEGER-VALUED doubles,
//     L1 = rint(v 2^-24)          and          L0 = rint((v - L1 2^24) 2^16),        v = L1 2^24 + L0 2^-16 + r, |r| <= 2^-17,
// which are added to two fp64 counters with the hardware's fp64 atomic add.  |L0| <= 2^39 and (for |v| < 2^64) |L1| < 2^40, so
// with at most 2^12 contributions per statistic every partial sum stays an integer below 2^53 - where IEEE addition is EXACT,
// hence associative and commutative: the totals are the same bits under every order (an atomic add of the raw fp64 partial
// sums, as up to round 3, rounds differently under different orders).  Integer counters do the same but gfx950 has no
// conversion between fp64 and 64-bit integers; decoding them at the head of every consumer workgroup cost 2 % of the step,
// this form costs a multiply-add.  Absolute error per contribution <= 2^-17; a layer's mean and variance carry at most
// (workgroups x 2^-17) / elements <= 6e-9 - eps of GroupNorm is 1e-5.  A non-finite contribution makes L1 non-finite and the
// statistic NaN / Inf, as the plain sum did.  Slot layout per (cloud, group), kGnWords doubles: {L1, L0} sums of x, then of
// x^2; zeroed by the host before the producing launch (one memset per registration).
__device__ __forceinline__ double gn_stat_limb(double v, int limb) {
  const double l1 = rint(v * 0x1p-24);
  return limb ? rint(fma(-l1, 0x1p24, v) * 0x1p16) : l1;
}
// A workgroup's contribution to the statistics of the column range [n0, n0 + ncols) it owns, from its per-column fp32 partial
// sums part[col * 2 + {0: sum x, 1: sum x^2}] (LDS, complete and barrier-separated from this call).  Atomic INSTRUCTIONS are
// what the chip rations (about one wave-instruction per 50 ns per CU, MI355X_MICROARCH.md), so the whole contribution - every
// group the range touches x {sum, sum of squares} x {L1, L0} - leaves in one instruction per wave that holds any of it.
// The order of every addition is fixed by the column index alone.  Call with ALL 256 threads of the block.
__device__ __forceinline__ void gn_block_commit(const float* part, int n0, int ncols, int gw, double* stats_cloud) {
  const int g0 = n0 / gw, ng = (n0 + ncols - 1) / gw - g0 + 1;
  // 16 lanes per (group, statistic): lane j adds columns j, j + 16, ... of the group in ascending order, then a fixed butterfly
  // over the 16 lanes (a serial loop over up to 64 columns was the longest dependent chain of a workgroup's tail)
  const int t = threadIdx.x, j = t & 15;
  for (int q = t >> 4; q < ((ng * 2 + 15) & ~15); q += 16) {   // q = (group, statistic); one trip unless a range holds > 8 groups
    const bool live = q < ng * 2;
    const int g = g0 + (q >> 1), stat = q & 1;
    const int c0 = max(g * gw, n0) - n0, c1 = min((g + 1) * gw, n0 + ncols) - n0;
    double d = 0.0;
    if (live)
      for (int c = c0 + j; c < c1; c += 16) d += (double)part[c * 2 + stat];
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4); d += __shfl_xor(d, 8);
    if (live && j < 2)                                       // lanes 0 / 1 of the quad carry the two limbs
      unsafeAtomicAdd(stats_cloud + (int64_t)g * kGnWords + 2 * stat + j, gn_stat_limb(d, j));   // global_atomic_add_f64, integer-valued operands
  }
}
// 1 / sqrt(var + eps) of GroupNorm (eps = 1e-5, RandLANet.py:93): the hardware's v_rsq_f64 seed (about 2^-27 relative) and two
// Newton steps in fp64 (error squared twice: far below 2^-52), a dozen instructions instead of the ~40 of a correctly rounded
// square root followed by a correctly rounded division - this chain opens EVERY consumer workgroup.  The scale / shift derived
// from it are rounded to fp32 afterwards.
__device__ __forceinline__ double gn_rstd(double var) {
  const double v = var + 1e-5;
  double r = __builtin_amdgcn_rsq(v);
  r = fma(r, fma(-0.5 * v * r, r, 0.5), r);
  r = fma(r, fma(-0.5 * v * r, r, 0.5), r);
  return r;
}
// Inside the deep-level walker (walk.hip defines DSIR_GN_STATS_COHERENT before it includes the tile bodies) the producers of a
// statistic may run in the SAME launch as its reader: the two limbs - results of memory-side atomics - are then read with agent-scope
// (sc1) loads, the pattern of polling a counter.  Everywhere else the producers belong to an earlier launch and plain loads serve
// every workgroup of a cloud from its CU's cache (with sc1 loads the split attentive pooling of level 3, whose prologue decodes 256
// channels' statistics, took 106 us per launch instead of 72).
__device__ __forceinline__ double gn_stat_get(const double* slot) {
#ifdef DSIR_GN_STATS_COHERENT
  const double l1 = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double l0 = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return fma(l1, 0x1p24, l0 * 0x1p-16);
#else
  return fma(slot[0], 0x1p24, slot[1] * 0x1p-16);
#endif
}

// Butterfly steps across the four 16-lane rows of a wave with the gfx950 VALU lane swaps instead of ds_bpermute
// (no LDS round trip, no lgkmcnt wait).  v_permlane16_swap exchanges the odd rows of its first operand with the
// even rows of the second, v_permlane32_swap the upper half of the first with the lower half of the second; fed
// the same value twice they return (x of the lower partner, x of the upper partner) in every lane, so
// op(r0, r1) == op(x, shfl_xor(x, 16 | 32)) bit for bit (op commutative).
__device__ __forceinline__ float xor16_add(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Attentive pooling of one 16x16 MFMA tile (reference RandLANet.py:152-155): column-wise softmax over
// the tile's 16 rows (= the 16 neighbours of a point; C layout: col = lane & 15, row = 4*(lane>>4)+reg),
// then sum_k f[k][c] * a[k][c].  `acc` = scores, `f` = the (normalised) features at the same positions.
// Returns the pooled value of column (lane & 15), valid in lanes 0..15.
// sum_k f_k e_k / sum_k e_k : one division per column instead of one per element.
__device__ __forceinline__ float att_pool_tile(const f32x4& acc, const float (&f)[4]) {
  float mx = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
  mx = xor32_max(xor16_max(mx));
  float se = 0.f, o = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = exp_neg(acc[r] - mx);
    se += e;
    o = fmaf(f[r], e, o);
  }
  se = xor32_add(xor16_add(se));
  o = xor32_add(xor16_add(o));
  return o * __builtin_amdgcn_rcpf(se);   // se >= 1 (the max term contributes e^0): v_rcp_f32 is 1 ulp there
}

// Order-preserving float max through integer atomics (target initialised to -inf).
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else          atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}


// |x|^2 of a 64-channel descriptor row held by 16 consecutive lanes (float4 each), in the one summation order every
// user of the descriptor distance shares (nn_match.hip, nn_screen.hip): the value enters D = (-2 a.b + |a|^2) + |b|^2.
__device__ __forceinline__ float sqnorm_row16(const float4 v) {
  float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
  return s;
}

// Four channels fp32 -> the fp16 operand pair of the screened descriptor search (nn_screen.hip, header: x = xh + 2^-11 xl with BOTH
// parts stored pre-scaled by 2^11, no fp16 subnormals in the high part); *bad is raised when a value is outside the domain of the
// screening's error bound (|x| > 16 or not finite).  Shared by split_norm_kernel and the aggregation chain's epilogue.
typedef _Float16 dsir_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void screen_split4(const float4 v, dsir_h4& h, dsir_h4& l, int32_t* __restrict__ bad) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  if (bad && !(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))) <= 16.f &&
               f[0] == f[0] && f[1] == f[1] && f[2] == f[2] && f[3] == f[3]))
    *bad = 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    _Float16 t = (_Float16)f[k];
    if (fabsf((float)t) < 6.103515625e-05f) t = (_Float16)0.f;          // no fp16 subnormals in the high part
    h[k] = t * (_Float16)2048.f;                                         // exact: |t| <= 16 and t is 0 or normal
    l[k] = (_Float16)((f[k] - (float)t) * 2048.0f);
  }
}

}  // namespace dsir
