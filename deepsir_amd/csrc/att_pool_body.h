// Device body of the unsplit attentive pooling of levels 1 / 2 (att_pool.hip, att_full_kernel): shared by its own launch and by the
// deep-level walker (walk.hip) - same instructions on the same operands, same bits.
#pragma once
#include "kernels.h"
#include "device_utils.h"

namespace dsir {
namespace attp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split8f(const float* x, h8& h, h8& l) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 t = (_Float16)x[k];
    h[k] = t;
    l[k] = (_Float16)(x[k] - (float)t);
  }
}

// ---------------------------------------------------------------- levels 1 / 2 (d = 64 / 128): the UNSPLIT form on the same organisation
// All d channels [fN[nb] (d/2) ; E (d/2)] are contracted, like level 0 does: lane half h = 0 forms the gathered-feature part of an A row (one index,
// one row of the raw features, their GroupNorm + LeakyReLU), h = 1 the E part (from memory or from the tables of
// lse_uv.hip); the pooled operand comes back from the wave's LDS tile: 8 KB of global reads per two points at d = 64.  d = 128: two workgroups per row range, one per 64 output columns
// (the first owns columns of the feature half, the second of the E half: each keeps only that half of the A rows in its LDS tile).
template <int KH>
constexpr size_t att_full_smem_bytes() {
  return 2 * smem_pad(sizeof(float) * 2 * KH) + smem_pad(sizeof(float) * 4 * 32 * 72) + smem_pad(sizeof(h8) * 2 * (KH / 8) * 2 * 64);
}

// body of att_full_kernel and of a walker job (walk.hip): row-range block bx of p.grid_x, column block cb, cloud; smem:
// att_full_smem_bytes<KH>() bytes of LDS, 16-byte aligned.  No cross-workgroup reduction: the result of a unit does not depend on p.grid_x.
template <int KH, bool UV>
__device__ __forceinline__ void att_full_body(const AttPool16Args& p, const int bx, const int cb, const int cloud, char* smem) {
  constexpr int KC = KH;          // channels of its row a lane holds (h = 0: features, h = 1: E)
  constexpr int NS = KH / 8;      // k-steps of 16 over the 2 KH channels
  constexpr int LD = 72;          // LDS row stride (floats): rows 4 apart land 32 banks apart
  float* s_sc = smem_carve<float>(smem, 2 * KH);
  float* s_sh = smem_carve<float>(smem, 2 * KH);
  auto s_t = smem_carve<float[32 * LD]>(smem, 4);
  h8* s_w = smem_carve<h8>(smem, 2 * NS * 2 * 64);          // [tile][k-step][high | low][lane]

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int col0 = 64 * cb;

  // B fragments -> LDS: tile t = columns 64 cb + 32 t + m; the k index of step s, lane half h, element j is channel KH h + 8 s + j
  if (w == 0) {
    const _Float16* Wh = reinterpret_cast<const _Float16*>(p.Wh);
    const _Float16* Wl = reinterpret_cast<const _Float16*>(p.Wl);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int o = (col0 + 32 * t + m) * p.ldw + KH * h + 8 * s;
        s_w[((t * NS + s) * 2 + 0) * 64 + lane] = *reinterpret_cast<const h8*>(Wh + o);
        s_w[((t * NS + s) * 2 + 1) * 64 + lane] = *reinterpret_cast<const h8*>(Wl + o);
      }
  }
  const float slope = (h ? p.enc_act : p.f_act) ? 0.2f : 1.f;
  const int pm = (m >> 2) & 1, km = ((m >> 3) << 2) | (m & 3);      // the (point, neighbour) of this lane's A row
  const float* fb = p.f + cloud * p.f_cs;
  const float* eb = p.enc + cloud * p.enc_cs;
  const int32_t* nbb = p.neigh + cloud * p.neigh_cs;
  const float* uvb = UV ? p.uv + cloud * p.uv_cs : nullptr;
  const float* distb = UV ? p.dist + cloud * p.dist_cs : nullptr;
  float* Yb = p.Y + cloud * p.y_cs;
  float* T = &s_t[w][0];
  float wa[UV ? KC : 1];
  if (UV) {
#pragma unroll
    for (int c = 0; c < KC; ++c) wa[c] = p.w8[c * 8];
  }

  const int units = (p.n + 1) >> 1;
  const int nw = p.grid_x * 4;
  int u = bx * 4 + w;

  float a[KC], av[UV ? KC : 1], ad = 0.f;
  int jn = 0;                                   // neighbour index of this lane's A row, one unit further ahead than the rows
  auto point_of = [&](int uu) { return min(2 * uu + pm, p.n - 1); };
  auto load_idx = [&](int uu) { jn = nbb[(uint32_t)point_of(uu) * 16u + (uint32_t)km]; };
  auto load_rows = [&](int uu) {
    const int pt = point_of(uu);
    // one instruction stream serves both lane halves through per-lane addresses: h = 0 the gathered feature row, h = 1 the E row
    // (UV: the neighbour's U row; its V row and dist follow in loads only the h = 1 lanes use)
    const float* src = h ? (UV ? uvb + (uint32_t)jn * (uint32_t)(2 * KH) : eb + ((uint32_t)(pt * 16 + km)) * (uint32_t)KH) : fb + (uint32_t)jn * (uint32_t)p.f_ld;
#pragma unroll
    for (int q = 0; q < KC / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    if (UV) {
      const float* sv = uvb + (uint32_t)pt * (uint32_t)(2 * KH) + (uint32_t)KH;
#pragma unroll
      for (int q = 0; q < KC / 4; ++q) {
        const float4 z = *reinterpret_cast<const float4*>(sv + 4 * q);
        av[UV ? 4 * q : 0] = z.x; av[UV ? 4 * q + 1 : 0] = z.y; av[UV ? 4 * q + 2 : 0] = z.z; av[UV ? 4 * q + 3 : 0] = z.w;
      }
      ad = distb[(uint32_t)(pt * 16 + km)];
    }
  };
  if (u < units) {
    load_idx(u);
    load_rows(u);
    if (u + nw < units) load_idx(u + nw);
  }
  // GroupNorm scale / shift of both operand halves (channels 0 .. KH - 1 the features, then E): decoded while the first loads fly
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
    // ---- A operand: normalise, keep fp32 in LDS for the pooled operand, split for the MFMAs
    h8 ah[NS], al[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * s + j;
        if (UV) {        // the row of lfa.mlp1, as lse_uv.hip formed it (lanes of the E half)
          const float e = __fadd_rn(fmaf(wa[UV ? c : 0], ad, a[c]), av[UV ? c : 0]);
          a[c] = h ? e : a[c];
        }
        const float v = fmaf(a[c], s_sc[KH * h + c], s_sh[KH * h + c]);
        a[c] = fmaxf(v, slope * v);
      }
      split8f(&a[8 * s], ah[s], al[s]);
    }
    // the pooled operand of this workgroup's 64 columns: d = 64 both halves (32 + 32), d = 128 the half the columns belong to
    if (KH == 32 || h == cb) {
#pragma unroll
      for (int q = 0; q < KC / 4; ++q)
        *reinterpret_cast<float4*>(&T[m * LD + (KH == 32 ? 32 * h : 0) + 4 * q]) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
    // ---- next unit's rows (and the index one unit further): in flight during the MFMAs and the epilogue
    const int pt = 2 * u + h;
    const int un = u + nw;
    if (un < units) {
      load_rows(un);
      if (un + nw < units) load_idx(un + nw);
    }
    // ---- scores: 3 fp16 MFMAs per (tile, k-step)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // the two tiles' chains are independent: issued alternately they interleave on the pipe (per accumulator the order is unchanged)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      h8 bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = s_w[((t * NS + s) * 2 + 0) * 64 + lane];
        bl[t] = s_w[((t * NS + s) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[t], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- epilogue: register i of the accumulator = neighbour i of point (lane >> 5), column 64 cb + 32 t + m
    constexpr float L2E = 1.44269504088896340736f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float mx = fmaxf(acc[t][0], acc[t][1]);
#pragma unroll
      for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(acc[t][i], acc[t][i + 1]));
      const float ml = -mx * L2E;
      float se = 0.f, o = 0.f;
      const float* Tc = &T[4 * h * LD + 32 * t + m];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f(fmaf(acc[t][i], L2E, ml));
        const float x = Tc[(8 * (i >> 2) + (i & 3)) * LD];
        se += e;
        o = fmaf(x, e, o);
      }
      if (pt < p.n) Yb[(uint32_t)pt * (uint32_t)(2 * KH) + (uint32_t)(col0 + 32 * t + m)] = o * __builtin_amdgcn_rcpf(se);
    }
    __builtin_amdgcn_wave_barrier();
    u = un;
  }
}


}  // namespace attp
}  // namespace dsir
