// Device bodies of the LDS-tiled point-wise GEMM kernels (pw_tile.hip): a workgroup's tile as a function of (row block, column
// block, cloud) and an LDS region, so that the same code runs as its own launch (pw_tile.hip) and as a job of the deep-level walker
// (walk.hip) - same instructions on the same operands, same bits.
#pragma once
#include "kernels.h"
#include "device_utils.h"

namespace dsir {
namespace tile {


constexpr int BK = 32;
constexpr int LDS_LD = BK + 2;
// H = true (round 3): the contraction on the fp16 matrix pipe at fp32 accuracy, by the operand split of agg_chain_h.hip
// (x = fp16(x) + fp16(x - fp16(x)); a.b = ah.bh + ah.bl + al.bh: three v_mfma_f32_16x16x32_f16).  A values are split when
// they are staged (after GroupNorm + LeakyReLU), weights were split at load (GemmArgs::Wh / Wl); the LDS tiles hold the two
// fp16 parts in rows of 32 + 8 halfs (80 bytes: 16-byte aligned, conflict-free ds_read_b128 fragment reads), and a
// 32-channel chunk is ONE k-step: 3 RT NT MFMAs of 16 cycles instead of 8 RT NT of 32.  These layers ran the fp32 pipe
// 31 - 40 % busy with two waves per SIMD contending for it.  Results differ from the H = false kernels (exact fp32,
// k-ordered fmaf chains; DSIR_TILE_F32=1) by ~1e-7 of the layer's scale.
constexpr int LDH = BK + 8;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4f(const float4 v, h4& h, h4& l) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const _Float16 t = (_Float16)f[k];
    h[k] = t;
    l[k] = (_Float16)(f[k] - (float)t);
  }
}
#define DSIR_MMA3H(acc, ah, al, bh, bl)                                      \
  do {                                                                        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);       \
  } while (0)
constexpr int BN = 64;
constexpr int NT = 4;
constexpr int MAXC = 768;

struct RowOff { int64_t o0, o1; };

__device__ __forceinline__ RowOff row_off(const GemmArgs& p, int cloud, int row) {
  RowOff r{-1, -1};
  if (row >= p.M) return r;
  {
    const Seg& s = p.seg[0];
    const int sr = s.idx ? s.idx[cloud * s.idx_cloud_stride + row] : row;
    r.o0 = cloud * s.cloud_stride + (int64_t)sr * s.ld;
  }
  if (p.nseg > 1) {
    const Seg& s = p.seg[1];
    const int sr = s.idx ? s.idx[cloud * s.idx_cloud_stride + row] : row;
    r.o1 = cloud * s.cloud_stride + (int64_t)sr * s.ld;
  }
  return r;
}

// LDS of one pw_tile workgroup (bytes), in the order the body carves it
// SC = 2 (cached scores: no contraction) stages no operand tile at all: its workgroups keep the few KB of scale / shift tables and
// with them the occupancy a latency-bound gather kernel lives on (with the tiles' 60 KB reserved it ran 106 us instead of 72)
template <int RT, int EPI, bool H, int SC = 0>
constexpr size_t pw_tile_smem_bytes() {
  constexpr int BM = 64 * RT;
  constexpr int NB = SC == 2 ? 0 : 1;     // tile buffers present?
  return NB * (smem_pad(sizeof(float) * (H ? 1 : 2) * (H ? 4 : BM * LDS_LD)) + smem_pad(sizeof(float) * (H ? 1 : 2) * (H ? 4 : BN * LDS_LD)) +
               smem_pad(sizeof(_Float16) * (H ? 2 : 1) * 2 * (H ? BM * LDH : 8)) + smem_pad(sizeof(_Float16) * (H ? 2 : 1) * 2 * (H ? BN * LDH : 8))) +
         2 * smem_pad(sizeof(float) * MAXC) + 2 * smem_pad(sizeof(float) * (EPI == EPI_ATT2 ? 128 : 1)) +
         smem_pad(sizeof(float) * (EPI == EPI_GN ? 4 * BN * 2 : 1));
}

// One workgroup's tile (row block bx, column block by) of cloud `cloud`: the body of pw_tile_kernel and of a walker job (walk.hip).
// smem: pw_tile_smem_bytes<RT, EPI, H>() bytes of LDS, 16-byte aligned, free of other users for the duration of the call.
template <int RT, int EPI, int SC = 0, bool H = false>   // SC: GemmArgs::s2_mode (EPI_ATT2 only); H: fp16-split contraction
__device__ __forceinline__ void pw_tile_body(const GemmArgs& p, const int bx, const int by, const int cloud, char* smem) {
  constexpr int BM = 64 * RT;
  constexpr int AV = BM / 32;     // float4 A loads per thread per chunk (BM*32/4/256)
  constexpr int NB = SC == 2 ? 0 : 1;     // SC = 2 never stages a tile (pw_tile_smem_bytes)
  auto As = smem_carve<float[H ? 4 : BM * LDS_LD]>(smem, NB * (H ? 1 : 2));
  auto Ws = smem_carve<float[H ? 4 : BN * LDS_LD]>(smem, NB * (H ? 1 : 2));
  auto AsH = smem_carve<_Float16[2][H ? BM * LDH : 8]>(smem, NB * (H ? 2 : 1));      // [buffer][high | low]
  auto WsH = smem_carve<_Float16[2][H ? BN * LDH : 8]>(smem, NB * (H ? 2 : 1));
  float* s_sc = smem_carve<float>(smem, MAXC);
  float* s_sh = smem_carve<float>(smem, MAXC);
  float* s_fsc = smem_carve<float>(smem, EPI == EPI_ATT2 ? 128 : 1);   // EPI_ATT2: GroupNorm scale/shift of the gathered-feature half
  float* s_fsh = smem_carve<float>(smem, EPI == EPI_ATT2 ? 128 : 1);
  float* s_gn = smem_carve<float>(smem, EPI == EPI_GN ? 4 * BN * 2 : 1);   // GroupNorm partial sums [wave][column][sum, sum of squares]

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = bx * BM;
  const int n0 = by * BN;

  // GroupNorm scale / shift of the operands -> LDS: a dependent chain (statistics load, fixed-point decode, fp64 arithmetic) that opens
  // every workgroup - it runs after the first K chunk's global loads have been issued (below)
  auto stats_to_lds = [&]() {
    if (EPI == EPI_ATT2) {
      const Seg& s = p.fseg;
      for (int c = tid; c < s.C; c += 256) {
        float scale = 1.f, shift = 0.f;
        if (s.gn.stats) {
          const int g = c / (s.C / s.gn.groups);
          const double* st = s.gn.stats + ((int64_t)cloud * s.gn.groups + g) * kGnWords;
          const double mean = gn_stat_get(st) * s.gn.inv_count;
          double var = gn_stat_get(st + 2) * s.gn.inv_count - mean * mean;
          var = var > 0.0 ? var : 0.0;
          const double rstd = gn_rstd(var);
          const double scd = (double)s.gn.gamma[c] * rstd;
          scale = (float)scd;
          shift = (float)((double)s.gn.beta[c] - mean * scd);
        }
        s_fsc[c] = scale;
        s_fsh[c] = shift;
      }
    }

    for (int c = tid; c < p.Cin; c += 256) {
      const Seg& s = (c < p.seg[0].C) ? p.seg[0] : p.seg[1];
      const int lc = (c < p.seg[0].C) ? c : c - p.seg[0].C;
      float scale = 1.f, shift = 0.f;
      if (s.gn.stats) {
        const int g = lc / (s.C / s.gn.groups);
        const double* st = s.gn.stats + ((int64_t)cloud * s.gn.groups + g) * kGnWords;
        const double mean = gn_stat_get(st) * s.gn.inv_count;
        double var = gn_stat_get(st + 2) * s.gn.inv_count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double rstd = gn_rstd(var);
        const double scd = (double)s.gn.gamma[lc] * rstd;
        scale = (float)scd;
        shift = (float)((double)s.gn.beta[lc] - mean * scd);
      }
      s_sc[c] = scale;
      s_sh[c] = shift;
    }
  };

  // staging assignment: thread -> 4 consecutive channels (tid & 7) of rows (tid >> 3) + 32 i
  const int c4 = (tid & 7) * 4;
  const int sr0 = tid >> 3;
  RowOff ro[AV];
#pragma unroll
  // rows past M / columns past Cout are CLAMPED, not predicated: their products are never stored nor counted,
  // and unconditional loads keep exec-mask juggling out of the K loop
  for (int i = 0; i < AV; ++i) ro[i] = row_off(p, cloud, min(m0 + sr0 + 32 * i, p.M - 1));
  const float* wrow[2];
  const _Float16 *wrowh[2], *wrowl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = min(n0 + sr0 + 32 * i, p.Cout - 1);
    const int64_t wo = (int64_t)col * (p.ldw ? p.ldw : p.Cin) + c4;
    wrow[i] = p.W + wo;
    wrowh[i] = reinterpret_cast<const _Float16*>(p.Wh) + wo;
    wrowl[i] = reinterpret_cast<const _Float16*>(p.Wl) + wo;
  }
  const int C0 = p.seg[0].C;
  const int act0 = p.seg[0].act, act1 = p.nseg > 1 ? p.seg[1].act : 0;

  float4 ra[AV], rw[H ? 1 : 2];
  h4 rwh[H ? 2 : 1], rwl[H ? 2 : 1];
  auto gload = [&](int k0) {
    const int c = k0 + c4;
    const bool s1 = c >= C0;
    const float* base = s1 ? p.seg[1].x : p.seg[0].x;
    const int lc = s1 ? c - C0 : c;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int64_t o = s1 ? ro[i].o1 : ro[i].o0;
      ra[i] = *reinterpret_cast<const float4*>(base + o + lc);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (H) { rwh[i] = *reinterpret_cast<const h4*>(wrowh[i] + k0); rwl[i] = *reinterpret_cast<const h4*>(wrowl[i] + k0); }
      else rw[H ? 0 : i] = *reinterpret_cast<const float4*>(wrow[i] + k0);
    }
  };
  auto lstore = [&](int k0, int buf) {
    const int c = k0 + c4;
    const float slope = ((c >= C0) ? act1 : act0) ? 0.2f : 1.f;   // LeakyReLU(v) = max(v, slope * v); slope 1 = identity
    const float4 sc = *reinterpret_cast<const float4*>(&s_sc[c]);
    const float4 sh = *reinterpret_cast<const float4*>(&s_sh[c]);
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      float4 v = ra[i];
      v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
      v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
      if (H) {
        h4 hh, ll;
        split4f(v, hh, ll);
        *reinterpret_cast<h4*>(&AsH[buf][0][(sr0 + 32 * i) * LDH + c4]) = hh;
        *reinterpret_cast<h4*>(&AsH[buf][1][(sr0 + 32 * i) * LDH + c4]) = ll;
      } else {
        float2* d = reinterpret_cast<float2*>(&As[H ? 0 : buf][(sr0 + 32 * i) * LDS_LD + c4]);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (H) {
        *reinterpret_cast<h4*>(&WsH[buf][0][(sr0 + 32 * i) * LDH + c4]) = rwh[i];
        *reinterpret_cast<h4*>(&WsH[buf][1][(sr0 + 32 * i) * LDH + c4]) = rwl[i];
      } else {
        float2* d = reinterpret_cast<float2*>(&Ws[H ? 0 : buf][(sr0 + 32 * i) * LDS_LD + c4]);
        d[0] = make_float2(rw[H ? 0 : i].x, rw[H ? 0 : i].y);
        d[1] = make_float2(rw[H ? 0 : i].z, rw[H ? 0 : i].w);
      }
    }
  };

  if (SC != 2) gload(0);     // in flight while the statistics are decoded
  stats_to_lds();
  __syncthreads();           // s_sc / s_sh ready

  f32x4 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // EPI_ATT2: the epilogue's gathered operands (rows of G = W1 f and of f, or the enc rows themselves) are
  // independent of the contraction: their index loads and the dependent row gathers are issued BEFORE the K loop
  // and land while it runs, instead of forming an exposed index -> row -> use chain after it.
  constexpr int PR = EPI == EPI_ATT2 ? RT : 1, PT = EPI == EPI_ATT2 ? NT : 1;
  float gpre[PR][PT][4], fpre[PR][PT][4];
  const bool gath = EPI == EPI_ATT2 && n0 < p.fseg.C;    // block-uniform: this block pools gathered f (else enc)
  if (EPI == EPI_ATT2) {
    const int ch = p.fseg.C;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(m0 + 16 * RT * w + 16 * rt + 4 * fq + r, p.M - 1);
        const int gi = p.fseg.idx[cloud * p.fseg.idx_cloud_stride + row];
        // G is stored in this kernel's order (engine.hip, up_fc_g): lane fr's column in tiles 0..3 of the block is one float4
        const float4 g4 = *reinterpret_cast<const float4*>(p.g + cloud * p.g_cloud_stride + (int64_t)gi * p.Cout + n0 + 4 * fr);
        const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
        const float* fp = gath ? p.fseg.x + cloud * p.fseg.cloud_stride + (int64_t)gi * p.fseg.ld + n0 + fr
                               : p.seg[0].x + row_off(p, cloud, row).o0 + (n0 - ch) + fr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          gpre[rt][t][r] = gq[t];
          fpre[rt][t][r] = fp[16 * t];
        }
      }
    }
  }

  // cached enc-half scores (kernels.h, GemmArgs::s2): C fragments of row tile (r0 >> 4) + rt, column tiles n0/16 .. +NT
  float4* s2p = (SC != 0) ? reinterpret_cast<float4*>(p.s2 + cloud * p.s2_cloud_stride) +
                                ((int64_t)((m0 + 16 * RT * w) >> 4) * (p.Cout >> 4) + (n0 >> 4)) * 64 + lane
                          : nullptr;
  const int nchunks = SC == 2 ? 0 : p.Cin / BK;      // loaded scores: no contraction, no staging
  if (SC == 2) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool live = m0 + 16 * RT * w + 16 * rt < p.M;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 v = live ? s2p[((int64_t)rt * (p.Cout >> 4) + t) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[rt][t] = f32x4{v.x, v.y, v.z, v.w};
      }
    }
  } else {
    lstore(0, 0);
    __syncthreads();
  }
  int buf = 0;
  for (int kc = 0; kc < nchunks; ++kc) {
    const bool more = kc + 1 < nchunks;
    if (more) gload((kc + 1) * BK);
    if (H) {
      h8 ah[RT], al[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        ah[rt] = *reinterpret_cast<const h8*>(&AsH[buf][0][(16 * RT * w + 16 * rt + fr) * LDH + 8 * fq]);
        al[rt] = *reinterpret_cast<const h8*>(&AsH[buf][1][(16 * RT * w + 16 * rt + fr) * LDH + 8 * fq]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const h8 bh = *reinterpret_cast<const h8*>(&WsH[buf][0][(16 * t + fr) * LDH + 8 * fq]);
        const h8 bl = *reinterpret_cast<const h8*>(&WsH[buf][1][(16 * t + fr) * LDH + 8 * fq]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) DSIR_MMA3H(acc[rt][t], ah[rt], al[rt], bh, bl);
      }
    } else {
      const float* At = As[H ? 0 : buf];
      const float* Wt = Ws[H ? 0 : buf];
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float a[RT], b[NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = At[(16 * RT * w + 16 * rt + fr) * LDS_LD + 4 * s + fq];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = Wt[(16 * t + fr) * LDS_LD + 4 * s + fq];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b[t], acc[rt][t], 0, 0, 0);
      }
    }
    if (more) lstore((kc + 1) * BK, buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  if (SC == 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (m0 + 16 * RT * w + 16 * rt < p.M) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          s2p[((int64_t)rt * (p.Cout >> 4) + t) * 64] = make_float4(acc[rt][t][0], acc[rt][t][1], acc[rt][t][2], acc[rt][t][3]);
      }
  }
  // ---- epilogues.  C layout: col = lane & 15, row = 4 * (lane >> 4) + reg.
  const int r0 = m0 + 16 * RT * w;
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + 16 * t + fr;
    bv[t] = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
  }
  if (EPI == EPI_GN) {
    float* Y = p.Y + cloud * p.y_cloud_stride;
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] = 0.f; s2[t] = 0.f;
      const int col = n0 + 16 * t + fr;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * rt + 4 * fq + r;
          if (row < p.M && col < p.Cout) {
            const float v = acc[rt][t][r] + bv[t];
            Y[(int64_t)row * p.ldy + col] = v;
            s1[t] += v;
            s2[t] += v * v;
          }
        }
    }
    // per-wave column sums (the four lane groups hold different rows of the same column) -> LDS -> the waves' sums in a fixed
    // order -> ONE atomic instruction per workgroup (device_utils.h, gn_block_commit)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] += __shfl_xor(s1[t], 16); s1[t] += __shfl_xor(s1[t], 32);
      s2[t] += __shfl_xor(s2[t], 16); s2[t] += __shfl_xor(s2[t], 32);
      if (fq == 0) {
        s_gn[(w * BN + 16 * t + fr) * 2] = s1[t];
        s_gn[(w * BN + 16 * t + fr) * 2 + 1] = s2[t];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const float v = (s_gn[tid] + s_gn[2 * BN + tid]) + (s_gn[4 * BN + tid] + s_gn[6 * BN + tid]);
      s_gn[tid] = v;          // wave 0's slots are re-used: each thread touches only its own entry
    }
    __syncthreads();
    const int ncols = min(BN, p.Cout - n0);
    gn_block_commit(s_gn, n0, ncols, p.Cout / p.groups_out, p.stats_out + (int64_t)cloud * p.groups_out * kGnWords);
  } else if (EPI == EPI_ACT || EPI == EPI_LINEAR) {
    float* Y = p.Y + cloud * p.y_cloud_stride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + 16 * t + fr;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * rt + 4 * fq + r;
          if (row < p.M && col < p.Cout) {
            float v = acc[rt][t][r] + bv[t];
            if (EPI == EPI_LINEAR && p.residual) v += p.residual[cloud * p.res_cloud_stride + (int64_t)row * p.ldres + col];
            if (EPI == EPI_ACT && v < 0.f) v *= 0.2f;
            Y[(int64_t)row * p.ldy + col] = v;
          }
        }
    }
  } else if (EPI == EPI_ATT) {
    float* Y = p.Y + cloud * p.y_cloud_stride;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int trow = r0 + 16 * rt;
      if (trow >= p.M) continue;
      RowOff er[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) er[r] = row_off(p, cloud, trow + 4 * fq + r);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + fr;
        const bool cs1 = col >= C0;
        const float* base = cs1 ? p.seg[1].x : p.seg[0].x;
        const int lc = cs1 ? col - C0 : col;
        const int act = cs1 ? act1 : act0;
        const float scv = s_sc[col < p.Cin ? col : 0], shv = s_sh[col < p.Cin ? col : 0];
        float f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = 0.f;
          if (col < p.Cin) {
            const float v = fmaf(base[(cs1 ? er[r].o1 : er[r].o0) + lc], scv, shv);
            f[r] = (act && v < 0.f) ? 0.2f * v : v;
          }
        }
        const float o = att_pool_tile(acc[rt][t], f);
        if (lane < 16 && col < p.Cout) Y[(int64_t)(trow >> 4) * p.ldy + col] = o;
      }
    }
  } else if (EPI == EPI_ATT2) {
    // split attentive pooling (kernels.h): scores = acc (enc half) + gathered G rows; pooled operand =
    // [gathered f (column blocks < Cout/2) ; enc (column blocks >= Cout/2)], prefetched above, GroupNorm applied here
    float* Y = p.Y + cloud * p.y_cloud_stride;
    const int ch = p.fseg.C;
    const float pslope = (gath ? p.fseg.act : act0) ? 0.2f : 1.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int trow = r0 + 16 * rt;
      if (trow >= p.M) continue;
      float o[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + fr;
        const float scv = gath ? s_fsc[col] : s_sc[col - ch], shv = gath ? s_fsh[col] : s_sh[col - ch];
        f32x4 sc4 = acc[rt][t];
        float f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc4[r] += gpre[rt][t][r];
          const float v = fmaf(fpre[rt][t][r], scv, shv);
          f[r] = fmaxf(v, pslope * v);
        }
        o[t] = att_pool_tile(sc4, f);
      }
      // every lane holds its column's pooled value for all four tiles: lane group fq stores tile fq (one store)
      const float ov = fq == 0 ? o[0] : (fq == 1 ? o[1] : (fq == 2 ? o[2] : o[3]));
      Y[(int64_t)(trow >> 4) * p.ldy + n0 + lane] = ov;
    }
  }
}

// Small-M variant (per-cloud layers of the deep pyramid levels: M = 312 / 78 / 19 rows at N = 5000): block tile
// 32 rows x 64 columns, the four waves arranged 2 x 2 (wave = 16 rows x 32 columns), same 32-channel K chunks, same
// staging, same k order - every output element is the same fmaf chain as in pw_tile_kernel, bit for bit.  A wave's MFMA
// chain is 4 x shorter (K/2 instead of 2 K for RT = 2), a cloud spreads over 4-8 x more workgroups, and the row padding
// drops from 64-128 to 32 rows (M = 78: 64 % -> 23 %; M = 19: 237 % -> 68 %).  Price: a weight tile is re-read from L2
// once per 32 rows.  GroupNorm statistics: fp32 per column tile and wave, fp64 atomics.
template <int RTS, int EPI, bool H>
constexpr size_t pw_tile_small_smem_bytes() {
  constexpr int BM = 32 * RTS;
  return smem_pad(sizeof(float) * (H ? 1 : 2) * (H ? 4 : BM * LDS_LD)) + smem_pad(sizeof(float) * (H ? 1 : 2) * (H ? 4 : BN * LDS_LD)) +
         smem_pad(sizeof(_Float16) * (H ? 2 : 1) * 2 * (H ? BM * LDH : 8)) + smem_pad(sizeof(_Float16) * (H ? 2 : 1) * 2 * (H ? BN * LDH : 8)) +
         2 * smem_pad(sizeof(float) * MAXC) + smem_pad(sizeof(float) * (EPI == EPI_GN ? 2 * BN * 2 : 1));
}

// body of pw_tile_small_kernel and of a walker job (walk.hip); smem: pw_tile_small_smem_bytes<RTS, EPI, H>() bytes, as for pw_tile_body
template <int RTS, int EPI, bool H = false>
__device__ __forceinline__ void pw_tile_small_body(const GemmArgs& p, const int bx, const int by, const int cloud, char* smem) {
  constexpr int BM = 32 * RTS, NTW = 2;         // RTS row tiles per wave: 32- or 64-row blocks
  auto As = smem_carve<float[H ? 4 : BM * LDS_LD]>(smem, H ? 1 : 2);
  auto Ws = smem_carve<float[H ? 4 : BN * LDS_LD]>(smem, H ? 1 : 2);
  auto AsH = smem_carve<_Float16[2][H ? BM * LDH : 8]>(smem, H ? 2 : 1);      // [buffer][high | low]
  auto WsH = smem_carve<_Float16[2][H ? BN * LDH : 8]>(smem, H ? 2 : 1);
  float* s_sc = smem_carve<float>(smem, MAXC);
  float* s_sh = smem_carve<float>(smem, MAXC);
  float* s_gn = smem_carve<float>(smem, EPI == EPI_GN ? 2 * BN * 2 : 1);   // GroupNorm partial sums [row half of the block][column][sum, sum of squares]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = bx * BM;
  const int n0 = by * BN;

  // staging: thread -> 4 consecutive channels (tid & 7) of A rows (tid >> 3) + 32 i and of W rows (tid >> 3), (tid >> 3) + 32
  const int c4 = (tid & 7) * 4;
  const int sr0 = tid >> 3;
  RowOff ro[RTS];
#pragma unroll
  for (int i = 0; i < RTS; ++i) ro[i] = row_off(p, cloud, min(m0 + sr0 + 32 * i, p.M - 1));
  const float* wrow[2];
  const _Float16 *wrowh[2], *wrowl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = min(n0 + sr0 + 32 * i, p.Cout - 1);
    const int64_t wo = (int64_t)col * (p.ldw ? p.ldw : p.Cin) + c4;
    wrow[i] = p.W + wo;
    wrowh[i] = reinterpret_cast<const _Float16*>(p.Wh) + wo;
    wrowl[i] = reinterpret_cast<const _Float16*>(p.Wl) + wo;
  }
  const int C0 = p.seg[0].C;
  const int act0 = p.seg[0].act, act1 = p.nseg > 1 ? p.seg[1].act : 0;

  // two register sets: a chunk is fetched two iterations before it is stored to LDS (a chunk's 8 MFMA steps hide only a
  // fraction of one L2 round trip; with a single launch on the chip - batch 1 - the K loop runs at load latency)
  struct Pre { float4 a[RTS], w[H ? 1 : 2]; h4 wh[H ? 2 : 1], wl[H ? 2 : 1]; };
  Pre preA, preB;
  const int nchunks = p.Cin / BK;
  auto gload = [&](Pre& pre, int kc) {
    const int k0 = min(kc, nchunks - 1) * BK;    // clamped: a fetch past the end is harmless and never stored
    const int c = k0 + c4;
    const bool s1 = c >= C0;
    const float* base = s1 ? p.seg[1].x : p.seg[0].x;
#pragma unroll
    for (int i = 0; i < RTS; ++i) pre.a[i] = *reinterpret_cast<const float4*>(base + (s1 ? ro[i].o1 : ro[i].o0) + (s1 ? c - C0 : c));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (H) { pre.wh[i] = *reinterpret_cast<const h4*>(wrowh[i] + k0); pre.wl[i] = *reinterpret_cast<const h4*>(wrowl[i] + k0); }
      else pre.w[H ? 0 : i] = *reinterpret_cast<const float4*>(wrow[i] + k0);
    }
  };
  auto lstore = [&](const Pre& pre, int k0, int buf) {
    const int c = k0 + c4;
    const float slope = ((c >= C0) ? act1 : act0) ? 0.2f : 1.f;
    const float4 sc = *reinterpret_cast<const float4*>(&s_sc[c]);
    const float4 sh = *reinterpret_cast<const float4*>(&s_sh[c]);
#pragma unroll
    for (int i = 0; i < RTS; ++i) {
      float4 v = pre.a[i];
      v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
      v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
      if (H) {
        h4 hh, ll;
        split4f(v, hh, ll);
        *reinterpret_cast<h4*>(&AsH[buf][0][(sr0 + 32 * i) * LDH + c4]) = hh;
        *reinterpret_cast<h4*>(&AsH[buf][1][(sr0 + 32 * i) * LDH + c4]) = ll;
      } else {
        float2* d = reinterpret_cast<float2*>(&As[H ? 0 : buf][(sr0 + 32 * i) * LDS_LD + c4]);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (H) {
        *reinterpret_cast<h4*>(&WsH[buf][0][(sr0 + 32 * i) * LDH + c4]) = pre.wh[i];
        *reinterpret_cast<h4*>(&WsH[buf][1][(sr0 + 32 * i) * LDH + c4]) = pre.wl[i];
      } else {
        float2* dw = reinterpret_cast<float2*>(&Ws[H ? 0 : buf][(sr0 + 32 * i) * LDS_LD + c4]);
        dw[0] = make_float2(pre.w[H ? 0 : i].x, pre.w[H ? 0 : i].y);
        dw[1] = make_float2(pre.w[H ? 0 : i].z, pre.w[H ? 0 : i].w);
      }
    }
  };
  f32x4 acc[RTS][NTW];
#pragma unroll
  for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // chunk kc from LDS buffer `buf`; `pre` holds chunk kc + 1 (stored to the other buffer) and is refilled with kc + 3
  auto chunk = [&](int kc, int buf, Pre& pre) {
    if (H) {
      h8 ah[RTS], al[RTS];
#pragma unroll
      for (int rt = 0; rt < RTS; ++rt) {
        ah[rt] = *reinterpret_cast<const h8*>(&AsH[buf][0][(16 * (RTS * wr + rt) + fr) * LDH + 8 * fq]);
        al[rt] = *reinterpret_cast<const h8*>(&AsH[buf][1][(16 * (RTS * wr + rt) + fr) * LDH + 8 * fq]);
      }
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const h8 bh = *reinterpret_cast<const h8*>(&WsH[buf][0][(16 * (NTW * wc + t) + fr) * LDH + 8 * fq]);
        const h8 bl = *reinterpret_cast<const h8*>(&WsH[buf][1][(16 * (NTW * wc + t) + fr) * LDH + 8 * fq]);
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) DSIR_MMA3H(acc[rt][t], ah[rt], al[rt], bh, bl);
      }
    } else {
      const float* At = As[H ? 0 : buf];
      const float* Wt = Ws[H ? 0 : buf];
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float a[RTS], b[NTW];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) a[rt] = At[(16 * (RTS * wr + rt) + fr) * LDS_LD + 4 * s + fq];
#pragma unroll
        for (int t = 0; t < NTW; ++t) b[t] = Wt[(16 * (NTW * wc + t) + fr) * LDS_LD + 4 * s + fq];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b[t], acc[rt][t], 0, 0, 0);
      }
    }
    if (kc + 1 < nchunks) {
      lstore(pre, (kc + 1) * BK, buf ^ 1);
      gload(pre, kc + 3);
    }
    __syncthreads();
  };
  gload(preA, 0);
  // GroupNorm scale / shift of the operands -> LDS: a dependent chain (statistics load, fixed-point decode, fp64 arithmetic) that opens
  // every workgroup - decoded while the first chunk's global loads are in flight
  for (int c = tid; c < p.Cin; c += 256) {
    const Seg& s = (c < p.seg[0].C) ? p.seg[0] : p.seg[1];
    const int lc = (c < p.seg[0].C) ? c : c - p.seg[0].C;
    float scale = 1.f, shift = 0.f;
    if (s.gn.stats) {
      const int g = lc / (s.C / s.gn.groups);
      const double* st = s.gn.stats + ((int64_t)cloud * s.gn.groups + g) * kGnWords;
      const double mean = gn_stat_get(st) * s.gn.inv_count;
      double var = gn_stat_get(st + 2) * s.gn.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = gn_rstd(var);
      const double scd = (double)s.gn.gamma[lc] * rstd;
      scale = (float)scd;
      shift = (float)((double)s.gn.beta[lc] - mean * scd);
    }
    s_sc[c] = scale;
    s_sh[c] = shift;
  }
  __syncthreads();           // s_sc / s_sh ready
  lstore(preA, 0, 0);
  gload(preA, 1);
  gload(preB, 2);
  __syncthreads();
  for (int kc = 0; kc < nchunks; kc += 2) {
    chunk(kc, 0, preA);
    if (kc + 1 < nchunks) chunk(kc + 1, 1, preB);
  }
  // ---- epilogue.  C layout: col = lane & 15, row = 4 * (lane >> 4) + reg.
  // two layers in one launch (GemmArgs::c_split): this workgroup's 64 columns belong to ONE of them (c_split is a multiple of 64)
  const bool sec = EPI == EPI_GN && p.c_split > 0 && n0 >= p.c_split;
  const int cb0 = sec ? p.c_split : 0;                          // first column of the layer
  const int cw = sec ? p.Cout - p.c_split : (p.c_split > 0 && EPI == EPI_GN ? p.c_split : p.Cout);   // its width
  const int ldy = sec ? p.ldy2 : p.ldy;
  float* Y = sec ? p.Y2 + cloud * p.y2_cloud_stride : p.Y + cloud * p.y_cloud_stride;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = n0 + 16 * (NTW * wc + t) + fr;
    const float bv = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rt = 0; rt < RTS; ++rt) {
      const int r0 = m0 + 16 * (RTS * wr + rt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 4 * fq + r;
        if (row < p.M && col < p.Cout) {
          float v = acc[rt][t][r] + bv;
          if (EPI == EPI_LINEAR && p.residual) v += p.residual[cloud * p.res_cloud_stride + (int64_t)row * p.ldres + col];
          if (EPI == EPI_ACT && v < 0.f) v *= 0.2f;
          Y[(int64_t)row * ldy + (col - cb0)] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (EPI == EPI_GN) {
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      if (fq == 0) {
        s_gn[(wr * BN + 16 * (NTW * wc + t) + fr) * 2] = s1;
        s_gn[(wr * BN + 16 * (NTW * wc + t) + fr) * 2 + 1] = s2;
      }
    }
  }
  if (EPI == EPI_GN) {
    // the two row halves' sums in a fixed order, then ONE atomic instruction per workgroup (device_utils.h, gn_block_commit)
    __syncthreads();
    if (tid < 2 * BN) {
      const float v = s_gn[tid] + s_gn[2 * BN + tid];
      s_gn[tid] = v;
    }
    __syncthreads();
    const int groups = sec ? p.groups_out2 : p.groups_out;
    double* stats = sec ? p.stats_out2 : p.stats_out;
    gn_block_commit(s_gn, n0 - cb0, min(BN, cw - (n0 - cb0)), cw / groups, stats + (int64_t)cloud * groups * kGnWords);
  }
}


}  // namespace tile
}  // namespace dsir
