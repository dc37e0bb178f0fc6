// Attentive pooling (reference network/RandLANet.py:140-157) of the k = 16 layers of pyramid levels 0 - 2 (d = 16 / 64 / 128):
//   y[i][c] = sum_k softmax_k(S[k][c]) X[k][c],   X[k] = [ fN[nb(i,k)] ; E[i,k] ],   S[k] = fc X[k]
// Up to round 3 these layers ran in pw_stream.hip (EPI_ATT / EPI_ATT2) on 16 x 16 MFMA tiles: the 16 neighbours of a point sat in four
// lane groups x four registers, so every softmax was a chain of cross-lane butterflies, and the kernels were bound by vector-ALU
// issue (~350 instructions per point at d = 64; PMC: 58 % VALU + 30 % MFMA busy; level 0: the exact-fp32 matrix pipe 79 % busy).
// The kernels here are organised around the ACCUMULATOR LAYOUT of v_mfma_f32_32x32x16_f16 instead:
//   * a wave owns units of TWO points = 32 rows = one row tile.  A-row m carries neighbour k = 4 (m >> 3) + (m & 3) of point
//     (m >> 2) & 1: the accumulator rows a lane holds - (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), cdna_hip_programming.md section 3 -
//     are then exactly the 16 neighbours of ONE point (lane half = point, register = neighbour).  Softmax and weighted sum run in
//     registers: no cross-lane instruction at all;
//   * the whole score contraction fc [fN[nb] ; E] runs on the fp16 matrix pipe at fp32 accuracy (x = fp16(x) + fp16(x - fp16(x)), three
//     MFMAs per product: agg_chain_h.hip; weights split at load): lane half h = 0 forms the gathered-feature part of an A row (one
//     index, one row of the raw features, their GroupNorm + LeakyReLU), h = 1 the E part (from memory, or rebuilt from the per-point
//     tables of lse_uv.hip); the normalised values double as the pooled operand through a wave-private LDS tile, so the epilogue
//     gathers nothing.  (A first version kept round 3's split of the scores by linearity - S = G[nb] + W2 E with G = W1 fN a per-point
//     GEMM - and gathered G and fN in the epilogue: 20 KB of L2 reads per two points at d = 64, the matrix pipe 8 % busy, no faster
//     than the kernel it replaced; profiles/README.md, round 4);
//   * software pipeline: the next unit's rows are in flight during the current unit's MFMAs and epilogue, its neighbour indices one
//     unit further ahead (index -> row is a dependent pair of loads).
#include "kernels.h"
#include "device_utils.h"
#include "att_pool_body.h"

namespace dsir {

namespace {

using namespace attp;

// ---------------------------------------------------------------- level 0 (d = 16): the unsplit form, fc [gather(f) ; enc]
// The score GEMM contracts all 16 channels [fN[nb] (8) ; E (8)] (splitting it by linearity would gather 64 more bytes per row
// than it saves).  Same organisation as above with one more twist, because only 16 of a 32-wide MFMA tile's columns exist:
// a unit is FOUR points = two row tiles A (points 4u, 4u + 1) and B (4u + 2, 4u + 3), and both accumulate into ONE
// accumulator - tile A against the weights placed in columns 0 .. 15 of the B operand (zeros elsewhere), tile B against the
// same weights placed in columns 16 .. 31.  C[m][c < 16] is then tile A's result and C[m][c >= 16] tile B's: every lane of
// the wave owns (point, column) work in the epilogue, no accumulator is moved between lanes.  Lane half h = 0 forms the
// gathered-feature part of an A row (one index load, one 32-byte row gather, the producer's GroupNorm + LeakyReLU), h = 1
// the E part; the normalised values double as the pooled operand through the LDS tile.  Round 3's kernel (pw_stream.hip
// EPI_ATT, exact-fp32 16 x 16 x 4 MFMAs) ran the matrix pipe 79 % and the vector ALU 57 % busy; this one issues 6 fp16 MFMAs
// (192 cycles) and ~250 vector instructions per four points.
template <bool UV>   // UV: the E half of an A row rebuilt from the per-point tables of lse_uv.hip; occupancy floor: five (tables: four) waves per SIMD - the next step down spills 64 - 116 bytes and doubles the kernel time
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UV ? 4 : 5, UV ? 4 : 5))) void att_pool16_kernel(const AttPool16Args p) {
  constexpr int LD = 20;                  // LDS row stride (floats): rows 4 apart land 16 banks apart
  constexpr int TB = 32 * LD + 32;        // tile B's offset: 32 banks away from tile A
  __shared__ float s_sc[16];
  __shared__ float s_sh[16];
  __shared__ __attribute__((aligned(16))) float s_t[4][TB + 32 * LD];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  const int bx = wi % p.grid_x;
  const int cloud = wi / p.grid_x;

  float sc[8], sh[8];
  const float slope = (h ? p.enc_act : p.f_act) ? 0.2f : 1.f;

  // B fragments: lane (c = m, h) holds W[col][8 h + j]; tile A's weights live in columns 0 .. 15, tile B's in 16 .. 31
  h8 whA, wlA, whB, wlB;
  {
    const _Float16* Wh = reinterpret_cast<const _Float16*>(p.Wh);
    const _Float16* Wl = reinterpret_cast<const _Float16*>(p.Wl);
    const int col = m & 15;
    const h8 vh = *reinterpret_cast<const h8*>(Wh + col * p.ldw + 8 * h);
    const h8 vl = *reinterpret_cast<const h8*>(Wl + col * p.ldw + 8 * h);
    h8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
    whA = m < 16 ? vh : z; wlA = m < 16 ? vl : z;
    whB = m < 16 ? z : vh; wlB = m < 16 ? z : vl;
  }
  const int pm = (m >> 2) & 1, km = ((m >> 3) << 2) | (m & 3);      // the (point, neighbour) of this lane's A row
  const float* fb = p.f + cloud * p.f_cs;
  const float* eb = p.enc + cloud * p.enc_cs;
  const int32_t* nbb = p.neigh + cloud * p.neigh_cs;
  float* Yb = p.Y + cloud * p.y_cs;
  float* T = &s_t[w][0];
  const int etile = m >> 4, ecol = m & 15;                          // the epilogue's (tile, column) of this lane

  const int units = (p.n + 3) >> 2;
  const int nw = p.grid_x * 4;
  int u = bx * 4 + w;

  // pipeline: rows of unit u + nw and indices of unit u + 2 nw are in flight while unit u is computed
  auto point_of = [&](int uu, int tile) { return min(4 * uu + 2 * tile + pm, p.n - 1); };
  auto load_idx = [&](int uu, int (&j)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) j[t] = nbb[(uint32_t)point_of(uu, t) * 16u + (uint32_t)km];
  };
  const float* uvb = UV ? p.uv + cloud * p.uv_cs : nullptr;
  const float* distb = UV ? p.dist + cloud * p.dist_cs : nullptr;
  float wa[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) wa[c] = UV ? p.w8[c * 8] : 0.f;
  // one pair of 16-byte loads per tile serves both lane halves through per-lane addresses: h = 0 the gathered feature row, h = 1
  // the E row (UV: the neighbour's U row; its V row and dist follow in a second pair that only the h = 1 lanes use)
  auto load_rows = [&](int uu, const int (&j)[2], float (&a)[2][8], float (&v)[2][8], float (&dd)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pt = point_of(uu, t);
      const float* src = h ? (UV ? uvb + (uint32_t)j[t] * 16u : eb + ((uint32_t)(pt * 16 + km)) * 8u) : fb + (uint32_t)j[t] * (uint32_t)p.f_ld;
      const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
      a[t][0] = v0.x; a[t][1] = v0.y; a[t][2] = v0.z; a[t][3] = v0.w; a[t][4] = v1.x; a[t][5] = v1.y; a[t][6] = v1.z; a[t][7] = v1.w;
      if (UV) {
        const float* sv = uvb + (uint32_t)pt * 16u + 8u;
        const float4 z0 = *reinterpret_cast<const float4*>(sv), z1 = *reinterpret_cast<const float4*>(sv + 4);
        v[t][0] = z0.x; v[t][1] = z0.y; v[t][2] = z0.z; v[t][3] = z0.w; v[t][4] = z1.x; v[t][5] = z1.y; v[t][6] = z1.z; v[t][7] = z1.w;
        dd[t] = distb[(uint32_t)(pt * 16 + km)];
      }
    }
  };
  float a[2][8], an[2][8];
  float av[2][8], avn[2][8], ad[2] = {0.f, 0.f}, adn[2] = {0.f, 0.f};
  int jn[2] = {0, 0};
  if (u < units) {
    int j0[2];
    load_idx(u, j0);
    load_rows(u, j0, a, av, ad);
    if (u + nw < units) load_idx(u + nw, jn);
  }
  // GroupNorm scale / shift of both operand halves: decoded while the first unit's loads are in flight
  if (tid < 16) {
    const GnRef& g = tid < 8 ? p.f_gn : p.enc_gn;
    const int c = tid & 7;
    float scale = 1.f, shift = 0.f;
    if (g.stats) {
      const int grp = c / (8 / g.groups);
      const double* st = g.stats + ((int64_t)cloud * g.groups + grp) * kGnWords;
      const double mean = gn_stat_get(st) * g.inv_count;
      double var = gn_stat_get(st + 2) * g.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = gn_rstd(var);
      const double scd = (double)g.gamma[c] * rstd;
      scale = (float)scd;
      shift = (float)((double)g.beta[c] - mean * scd);
    }
    s_sc[tid] = scale;
    s_sh[tid] = shift;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = s_sc[8 * h + j]; sh[j] = s_sh[8 * h + j]; }
  while (u < units) {
    const int un = u + nw;
    if (un < units) {
      load_rows(un, jn, an, avn, adn);
      if (un + nw < units) load_idx(un + nw, jn);
    }
    // ---- A operands of both tiles: normalise, keep fp32 in LDS for the pooled operand, split for the MFMAs
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (UV) {        // the row of lfa.mlp1, as lse_uv.hip formed it (lanes of the E half)
          const float e = __fadd_rn(fmaf(wa[j], ad[t], a[t][j]), av[t][j]);
          a[t][j] = h ? e : a[t][j];
        }
        const float v = fmaf(a[t][j], sc[j], sh[j]);
        a[t][j] = fmaxf(v, slope * v);
      }
      float* row = &T[t * TB + m * LD + 8 * h];
      *reinterpret_cast<float4*>(row) = make_float4(a[t][0], a[t][1], a[t][2], a[t][3]);
      *reinterpret_cast<float4*>(row + 4) = make_float4(a[t][4], a[t][5], a[t][6], a[t][7]);
      h8 ah, al;
      split8f(a[t], ah, al);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, t ? whB : whA, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, t ? wlB : wlA, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, t ? whB : whA, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- epilogue: register i = neighbour i of point 4 u + 2 etile + h, column ecol
    constexpr float L2E = 1.44269504088896340736f;
    float mx = fmaxf(acc[0], acc[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(acc[i], acc[i + 1]));
    const float ml = -mx * L2E;
    float se = 0.f, o = 0.f;
    const float* Tc = &T[etile * TB + 4 * h * LD + ecol];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __builtin_amdgcn_exp2f(fmaf(acc[i], L2E, ml));
      const float x = Tc[(8 * (i >> 2) + (i & 3)) * LD];
      se += e;
      o = fmaf(x, e, o);
    }
    const int pt = 4 * u + 2 * etile + h;
    if (pt < p.n) Yb[(uint32_t)pt * 16u + (uint32_t)ecol] = o * __builtin_amdgcn_rcpf(se);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[t][j] = an[t][j]; av[t][j] = avn[t][j]; }
      ad[t] = adn[t];
    }
    u = un;
  }
}

// levels 1 / 2 (d = 64 / 128), the UNSPLIT form: body in att_pool_body.h (shared with the deep-level walker, walk.hip)
template <int KH, bool UV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KH == 32 ? 3 : 2, KH == 32 ? 3 : 2))) void att_full_kernel(const AttPool16Args p) {
  constexpr int NCB = KH / 32;    // workgroups per row range: 64 output columns each
  __shared__ __attribute__((aligned(16))) char smem[att_full_smem_bytes<KH>()];
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  att_full_body<KH, UV>(p, wi % p.grid_x, (wi / p.grid_x) % NCB, wi / (p.grid_x * NCB), smem);
}

// ---------------------------------------------------------------- level 2 (d = 128): all 128 output columns in ONE workgroup
// att_full_kernel<64> gave each 64-column half of a row range its own workgroup, and each of the two gathered, normalised and split the
// whole 128-channel A row again (~450 of ~680 vector instructions per unit, and every row read twice).  Here a wave forms the A
// fragments of a unit ONCE and walks the four 32-column tiles two at a time (two independent accumulator chains interleave on the
// pipe): per tile 24 MFMAs from the weight fragments in LDS (64 KB for the 128 x 128 fc matrix as fp16 pairs, shared by EIGHT waves -
// 512 threads, so that two waves per SIMD still fit the CU's LDS), the pair's half of the pooled operand goes through the wave's LDS
// tile (32 rows x 64 channels: tiles 0 / 1 are the feature channels, held by lane half 0; tiles 2 / 3 the E channels, lane half 1)
// and the epilogue runs as before.  Every (row, column)
// sees the same MFMA sequence as in att_full_kernel<64>: same bits.
__global__ __launch_bounds__(512) void att_full128_kernel(const AttPool16Args p) {
  constexpr int KH = 64, KC = 64, NS = 8, NT = 4;
  constexpr int LD = 72;          // LDS row stride (floats): rows 4 apart land 32 banks apart
  __shared__ float s_sc[2 * KH];
  __shared__ float s_sh[2 * KH];
  __shared__ __attribute__((aligned(16))) float s_t[8][32 * LD];
  __shared__ h8 s_w[NT * NS * 2 * 64];          // [tile][k-step][high | low][lane]

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  const int bx = wi % p.grid_x;
  const int cloud = wi / p.grid_x;

  // B fragments -> LDS: tile t = columns 32 t + m; the k index of step s, lane half h, element j is channel KH h + 8 s + j
  {
    const _Float16* Wh = reinterpret_cast<const _Float16*>(p.Wh);
    const _Float16* Wl = reinterpret_cast<const _Float16*>(p.Wl);
#pragma unroll
    for (int q = 0; q < NT * NS / 8; ++q) {      // 32 (tile, step) pairs over the 8 waves
      const int ts = q * 8 + w, t = ts / NS, st = ts % NS;
      const int o = (32 * t + m) * p.ldw + KH * h + 8 * st;
      s_w[(ts * 2 + 0) * 64 + lane] = *reinterpret_cast<const h8*>(Wh + o);
      s_w[(ts * 2 + 1) * 64 + lane] = *reinterpret_cast<const h8*>(Wl + o);
    }
  }
  const float slope = (h ? p.enc_act : p.f_act) ? 0.2f : 1.f;
  const int pm = (m >> 2) & 1, km = ((m >> 3) << 2) | (m & 3);      // the (point, neighbour) of this lane's A row
  const float* fb = p.f + cloud * p.f_cs;
  const float* eb = p.enc + cloud * p.enc_cs;
  const int32_t* nbb = p.neigh + cloud * p.neigh_cs;
  float* Yb = p.Y + cloud * p.y_cs;
  float* T = &s_t[w][0];

  const int units = (p.n + 1) >> 1;
  const int nw = p.grid_x * 8;
  int u = bx * 8 + w;

  float a[KC];
  int jn = 0;                                   // neighbour index of this lane's A row, one unit further ahead than the rows
  auto point_of = [&](int uu) { return min(2 * uu + pm, p.n - 1); };
  auto load_idx = [&](int uu) { jn = nbb[(uint32_t)point_of(uu) * 16u + (uint32_t)km]; };
  auto load_rows = [&](int uu) {
    const int pt = point_of(uu);
    const float* src = h ? eb + ((uint32_t)(pt * 16 + km)) * (uint32_t)KH : fb + (uint32_t)jn * (uint32_t)p.f_ld;
#pragma unroll
    for (int q = 0; q < KC / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  };
  if (u < units) {
    load_idx(u);
    load_rows(u);
    if (u + nw < units) load_idx(u + nw);
  }
  if (tid < 2 * KH) {
    const GnRef& g = tid < KH ? p.f_gn : p.enc_gn;
    const int c = tid < KH ? tid : tid - KH;
    float scale = 1.f, shift = 0.f;
    if (g.stats) {
      const int grp = c / (KH / g.groups);
      const double* st = g.stats + ((int64_t)cloud * g.groups + grp) * kGnWords;
      const double mean = gn_stat_get(st) * g.inv_count;
      double var = gn_stat_get(st + 2) * g.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = gn_rstd(var);
      const double scd = (double)g.gamma[c] * rstd;
      scale = (float)scd;
      shift = (float)((double)g.beta[c] - mean * scd);
    }
    s_sc[tid] = scale;
    s_sh[tid] = shift;
  }
  __syncthreads();
  while (u < units) {
    // ---- A operand: normalise (fp32 values stay in registers for the pooled operand), split for the MFMAs
    h8 ah[NS], al[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * s + j;
        const float v = fmaf(a[c], s_sc[KH * h + c], s_sh[KH * h + c]);
        a[c] = fmaxf(v, slope * v);
      }
      split8f(&a[8 * s], ah[s], al[s]);
    }
    const int pt = 2 * u + h;
    const int un = u + nw;
    constexpr float L2E = 1.44269504088896340736f;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      // the pooled operand of this tile pair's 64 columns: the feature half (tp = 0, lane half 0 holds it) or the E half (tp = 1)
      if (h == tp) {
#pragma unroll
        for (int q = 0; q < KC / 4; ++q)
          *reinterpret_cast<float4*>(&T[m * LD + 4 * q]) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        // this lane half's fp32 values are in LDS now (their split copies in ah / al): its rows of the NEXT unit go out here - the
        // feature gathers (index -> row, the latency-critical pair) fly during both tile pairs, the E rows during the second
        if (un < units) load_rows(un);
      }
      // two tiles at a time: their MFMA chains are independent and interleave on the pipe
      f32x16 acc[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        h8 bh[2], bl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          bh[tt] = s_w[(((2 * tp + tt) * NS + s) * 2 + 0) * 64 + lane];
          bl[tt] = s_w[(((2 * tp + tt) * NS + s) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[tt], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[tt], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[tt], acc[tt], 0, 0, 0);
      }
      if (tp == 1 && un + nw < units) load_idx(un + nw);      // the index one unit further ahead (its own jn was consumed above)
      __builtin_amdgcn_wave_barrier();
      // ---- epilogue: register i of the accumulator = neighbour i of point (lane >> 5), column 64 tp + 32 tt + m
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float mx = fmaxf(acc[tt][0], acc[tt][1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(acc[tt][i], acc[tt][i + 1]));
        const float ml = -mx * L2E;
        float se = 0.f, o = 0.f;
        const float* Tc = &T[4 * h * LD + 32 * tt + m];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float e = __builtin_amdgcn_exp2f(fmaf(acc[tt][i], L2E, ml));
          const float x = Tc[(8 * (i >> 2) + (i & 3)) * LD];
          se += e;
          o = fmaf(x, e, o);
        }
        if (pt < p.n) Yb[(uint32_t)pt * (uint32_t)(2 * KH) + (uint32_t)(64 * tp + 32 * tt + m)] = o * __builtin_amdgcn_rcpf(se);
      }
      __builtin_amdgcn_wave_barrier();
    }
    u = un;
  }
}

}  // namespace

bool launch_att_pool16(const AttPool16Args& a, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  if (!a.Wh || !a.Wl || !a.f || !a.neigh || !a.Y) return false;
  if (!a.enc && (!a.uv || !a.dist || !a.w8 || (reinterpret_cast<uintptr_t>(a.uv) % 16) != 0 || (a.uv_cs % 4) != 0)) return false;
  if ((a.ldw % 8) != 0 || (reinterpret_cast<uintptr_t>(a.Wh) % 16) != 0 || (reinterpret_cast<uintptr_t>(a.Wl) % 16) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.f) % 16) != 0 || (a.f_cs % 4) != 0 || (a.f_ld % 4) != 0 || (a.enc && ((reinterpret_cast<uintptr_t>(a.enc) % 16) != 0 || (a.enc_cs % 4) != 0))) return false;
  if ((a.f_gn.stats && (8 % a.f_gn.groups) != 0) || (a.enc_gn.stats && (8 % a.enc_gn.groups) != 0)) return false;
  if ((int64_t)a.n * 16 * 8 * 4 >= ((int64_t)1 << 32)) return false;
  const int units = (a.n + 3) / 4;
  int blocks = (units + 31) / 32;       // ~8 units per wave; no cross-workgroup reduction: the grid may follow the launch size
  const int64_t total = (int64_t)blocks * a.clouds;
  if (total < 512) {
    const int want = (int)((512 + (int64_t)a.clouds - 1) / (int64_t)a.clouds), most = (units + 3) / 4;
    const int nb = want < most ? want : most;
    if (nb > blocks) blocks = nb;
  }
  if (blocks < 1) blocks = 1;
  AttPool16Args b = a;
  b.grid_x = blocks;
  if (a.enc) hipLaunchKernelGGL(att_pool16_kernel<false>, dim3((unsigned)((int64_t)blocks * a.clouds)), dim3(256), 0, st, b);
  else hipLaunchKernelGGL(att_pool16_kernel<true>, dim3((unsigned)((int64_t)blocks * a.clouds)), dim3(256), 0, st, b);
  return true;
}

// levels 1 / 2 (d = 64 / 128), unsplit: the arguments of level 0 with KH-channel halves (f [n][f_ld >= KH], E [n * 16][KH] or - KH = 32 -
// tables [n][64]), fc [2 KH][ldw], Y [n][2 KH]
bool launch_att_full(const AttPool16Args& a, int KH, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  if ((KH != 32 && KH != 64) || !a.Wh || !a.Wl || !a.f || !a.neigh || !a.Y) return false;
  if (!a.enc && (KH != 32 || !a.uv || !a.dist || !a.w8 || (reinterpret_cast<uintptr_t>(a.uv) % 16) != 0 || (a.uv_cs % 4) != 0)) return false;
  if ((a.ldw % 8) != 0 || (reinterpret_cast<uintptr_t>(a.Wh) % 16) != 0 || (reinterpret_cast<uintptr_t>(a.Wl) % 16) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.f) % 16) != 0 || (a.f_cs % 4) != 0 || (a.f_ld % 4) != 0 || (a.enc && ((reinterpret_cast<uintptr_t>(a.enc) % 16) != 0 || (a.enc_cs % 4) != 0))) return false;
  if ((a.f_gn.stats && (KH % a.f_gn.groups) != 0) || (a.enc_gn.stats && (KH % a.enc_gn.groups) != 0)) return false;
  if ((int64_t)a.n * 16 * KH * 4 >= ((int64_t)1 << 32)) return false;
  const int units = (a.n + 1) / 2;
  // d = 128: one 8-wave workgroup per row range (att_full128_kernel) once the launch fills the chip with them; small launches keep the
  // two 4-wave workgroups per row range (a 64 KB weight prologue per workgroup is what a one-unit wave cannot amortise).  Same bits.
  if (KH == 64 && (int64_t)units * a.clouds >= 8192) {
    int blocks = (units + 63) / 64;     // ~8 units per wave
    if ((int64_t)blocks * a.clouds < 256) {
      const int want = (int)((256 + (int64_t)a.clouds - 1) / (int64_t)a.clouds), most = (units + 7) / 8;
      const int nb = want < most ? want : most;
      if (nb > blocks) blocks = nb;
    }
    if (blocks < 1) blocks = 1;
    AttPool16Args b = a;
    b.grid_x = blocks;
    hipLaunchKernelGGL(att_full128_kernel, dim3((unsigned)((int64_t)blocks * a.clouds)), dim3(512), 0, st, b);
    return true;
  }
  const int ncb = KH / 32;
  int blocks = (units + 31) / 32;       // ~8 units per wave; no cross-workgroup reduction: the grid may follow the launch size
  const int64_t total = (int64_t)blocks * ncb * a.clouds;
  if (total < 512) {
    const int want = (int)((512 + (int64_t)ncb * a.clouds - 1) / ((int64_t)ncb * a.clouds)), most = (units + 3) / 4;
    const int nb = want < most ? want : most;
    if (nb > blocks) blocks = nb;
  }
  if (blocks < 1) blocks = 1;
  AttPool16Args b = a;
  b.grid_x = blocks;
  const dim3 grid((unsigned)((int64_t)blocks * ncb * a.clouds));
  if (KH == 64) hipLaunchKernelGGL((att_full_kernel<64, false>), grid, dim3(256), 0, st, b);
  else if (a.enc) hipLaunchKernelGGL((att_full_kernel<32, false>), grid, dim3(256), 0, st, b);
  else hipLaunchKernelGGL((att_full_kernel<32, true>), grid, dim3(256), 0, st, b);
  return true;
}

// launch_att_full's d = 128 form (KH = 64, two 4-wave workgroups per row range) as a phase of the deep-level walker (walk.hip):
// the same envelope checks; the row ranges are cut so that a cloud's tiles about match its workgroups (a unit's result does not
// depend on the cut).
bool walk_plan_att_full(const AttPool16Args& a, int KH, int wpc, WalkJob* out) {
  if (a.n <= 0 || a.clouds <= 0 || KH != 64 || !a.Wh || !a.Wl || !a.f || !a.neigh || !a.Y || !a.enc) return false;
  if ((a.ldw % 8) != 0 || (reinterpret_cast<uintptr_t>(a.Wh) % 16) != 0 || (reinterpret_cast<uintptr_t>(a.Wl) % 16) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.f) % 16) != 0 || (a.f_cs % 4) != 0 || (a.f_ld % 4) != 0 || (reinterpret_cast<uintptr_t>(a.enc) % 16) != 0 || (a.enc_cs % 4) != 0) return false;
  if ((a.f_gn.stats && (KH % a.f_gn.groups) != 0) || (a.enc_gn.stats && (KH % a.enc_gn.groups) != 0)) return false;
  if ((int64_t)a.n * 16 * KH * 4 >= ((int64_t)1 << 32)) return false;
  const int units = (a.n + 1) / 2, ncb = KH / 32;
  int blocks = wpc / ncb, most = (units + 3) / 4;
  blocks = blocks < 1 ? 1 : (blocks > most ? most : blocks);
  out->kind = WK_ATT_FULL64; out->v0 = out->v1 = 0;
  out->att = a;
  out->att.grid_x = blocks;
  out->gx = blocks; out->gy = ncb;
  return true;
}

}  // namespace dsir
