// Row-streaming point-wise GEMM for narrow layers (Cin <= 64): the layers at
// pyramid levels 0/1 and the per-point MLP heads, where a row is 32-256 bytes
// in, 32-256 bytes out and the bound is HBM/L2 bandwidth, not MFMA rate.
//
// No LDS staging and no block-level barrier in the main loop:
//   * every wave owns whole 16-row MFMA tiles and walks them grid-stride;
//   * the MFMA k index is re-ordered so that lane (r = lane & 15, q = lane >> 4)
//     holds the CONTIGUOUS channels [q*KQ, (q+1)*KQ) of row r (KQ = Cin/4): the A
//     fragment of a tile is one 4..64-byte vector load per lane straight from
//     global memory (a wave reads 16 full rows), the producer's GroupNorm +
//     LeakyReLU is applied in registers;
//   * the weight fragments W[col][q*KQ + s] (same k order) and the GroupNorm
//     scale/shift of the lane's channels live in registers for the whole kernel;
//   * GroupNorm statistics of the output are accumulated in registers across all
//     tiles of the wave and reduced once at the end (one fp64 atomic per group per
//     block).
// The sum over k is the same set of products as in pw_gemm.hip, associated in a
// different (fixed) order; results are deterministic.
// Loader modes: vector (aligned segments), element-wise (tiny Cin: xyz / score /
// 6-channel inlier input), and the relative position encoding (RandLANet.py:197-212).
#include <cstdio>
#include "kernels.h"
#include "device_utils.h"
#include <cstdlib>

namespace dsir {

namespace {

enum SMode { S_VEC = 0, S_ELEM = 1, S_LSE = 2, S_UV = 3 };   // S_UV: rows rebuilt from the per-point tables of lse_uv.hip

template <int KQ>
struct Chunk { float v[KQ]; };

template <int KQ>
__device__ __forceinline__ void vec_load(const float* __restrict__ p, float (&v)[KQ]) {
  if constexpr (KQ == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else if constexpr (KQ % 4 == 0) {
#pragma unroll
    for (int i = 0; i < KQ / 4; ++i) {
      const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < KQ; ++i) v[i] = p[i];
  }
}

__device__ __forceinline__ int src_row(const Seg& s, int cloud, int row) {
  return s.idx ? s.idx[cloud * s.idx_cloud_stride + row] : row;
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// x[0..8) -> fp16(x), fp16(x - fp16(x)): the operand split of agg_chain_h.hip (three fp16 MFMAs per fp32 product)
__device__ __forceinline__ void split8(const float* x, h8& h, h8& l) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 t = (_Float16)x[k];
    h[k] = t;
    l[k] = (_Float16)(x[k] - (float)t);
  }
}

// Occupancy floor: the level-1 lfa.mlp2 with the table loader (32 channels in) runs at the latency of its gather -> MFMA -> store
// chain, which a third resident wave per SIMD hides better than 12 bytes of spill cost (244 -> 217 us per launch).  The 64-channel
// streams lose under the same floor (52 bytes of spill: 71 -> 85 us) and keep the allocator's choice.
constexpr int pw_stream_min_waves(int KQ, int NT, int EPI, int MODE) {
  return (EPI != EPI_ATT && EPI != EPI_ATT2 && KQ == 8 && MODE == S_UV) ? 3 : 1;
}

template <int KQ, int NT, int EPI, int MODE, int SC = 0>   // SC: GemmArgs::s2_mode (EPI_ATT2 only)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(pw_stream_min_waves(KQ, NT, EPI, MODE)))) void pw_stream_kernel(const GemmArgs p) {
  constexpr int BN = NT * 16;
  constexpr bool kVec = MODE == S_VEC || MODE == S_UV;   // raw rows loaded first, normalised after the loads have landed
  // Attentive pooling (round 3): the score contraction of a 16-neighbour tile runs on the fp16 matrix pipe at fp32 accuracy -
  // a lane's 8 or 16 contiguous channels ARE the A fragment of v_mfma_f32_16x16x32_f16 (k = 8 fq + j per 32-channel step, the
  // same re-ordered k index as the fp32 form), each operand split into two fp16 numbers, three MFMAs per product: 12 / 24 fp16
  // MFMAs (192 / 384 matrix-pipe cycles) per tile instead of 32 / 64 fp32 ones (1024 / 2048), which two waves per SIMD
  // contend for.  Scores differ from the fp32 form's by ~1e-7 of their scale; the cached enc halves (SC 1 / 2) come from
  // this same arithmetic, so hoisted and recomputed iterations still agree bit for bit.  Compile-time and unconditional: dsir_enable_agg_split(0) /
  // DSIR_AGG_F32 do NOT bring the fp32 form back (include/dsir.h says so); its reference is the oracle.
  constexpr bool kSplit = (EPI == EPI_ATT || EPI == EPI_ATT2) && (KQ == 8 || KQ == 16) && MODE == S_VEC;
  constexpr int NS = kSplit ? KQ / 8 : 1;
  constexpr int CP = KQ * 4;  // padded Cin
  __shared__ float s_sc[CP];
  __shared__ float s_sh[CP];
  __shared__ float s_red[EPI == EPI_GN ? 4 * BN * 2 : 1];
  __shared__ float s_att[(EPI == EPI_ATT || EPI == EPI_ATT2) ? 4 * 16 * (CP + 4) : 1];
  __shared__ float s_fsc[EPI == EPI_ATT2 ? 64 : 1];   // EPI_ATT2: GroupNorm scale/shift of the gathered-feature half
  __shared__ float s_fsh[EPI == EPI_ATT2 ? 64 : 1];

  // the wave index is uniform across the wave, but only readfirstlane lets the compiler KNOW it: tile indices
  // derived from it then live in SGPRs and the per-tile bounds checks become scalar branches instead of
  // exec-mask save/restore sequences
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // XCD-aware work mapping (speed only; per-cloud work and results are unchanged): workgroups are dealt round-robin
  // over the 8 XCDs, each with its own L2.  The bijective remap hands every XCD a CONTIGUOUS range of work items ordered
  // (cloud, column block, row block), i.e. whole clouds, so the rows a cloud's blocks gather (neighbour features, G = W1 f)
  // live in ONE L2 instead of being replicated through eight.
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int wi = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int bx = wi % p.grid_x;
  const int by = (wi / p.grid_x) % p.grid_y;
  const int cloud = wi / (p.grid_x * p.grid_y);
  const int n0 = by * BN;
  const int ldw = p.ldw ? p.ldw : p.Cin;

  // first column of the block's t-th 16-column tile.  EPI_ATT2 (Cout = 2 Cin = 8 KQ, NT = 4): a block owns 32 columns
  // of the gathered-feature half and the matching 32 of the enc half, so "tile t pools a gathered feature" is the
  // compile-time condition t < NT/2 for every block
  auto col0_of = [&](int t) -> int {
    if (EPI == EPI_ATT2) return t < NT / 2 ? (n0 >> 1) + 16 * t : p.fseg.C + (n0 >> 1) + 16 * (t - NT / 2);
    return n0 + 16 * t;
  };
  // The weight fragments are fetched FIRST: their global loads are in flight while the GroupNorm statistics below are decoded
  // (a dependent chain of fp64 arithmetic that opens every workgroup).
  const int c_lo = fq * KQ;                       // first channel of this lane's chunk
  float wf[NT][KQ];
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0_of(t) + fr;
    if ((KQ % 4) == 0 && p.Cin == 4 * KQ) {          // rows of W are 16-byte aligned: vector loads
      if (col < p.Cout) vec_load<KQ>(p.W + (int64_t)col * ldw + c_lo, wf[t]);
      else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) wf[t][j] = 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        const int k = c_lo + j;
        wf[t][j] = (col < p.Cout && k < p.Cin) ? p.W[(int64_t)col * ldw + k] : 0.f;
      }
    }
    bv[t] = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
  }
  // GroupNorm scale / shift of the operands -> LDS.  A dependent chain (statistics load, fixed-point decode, fp64 arithmetic, barrier)
  // that opens every workgroup: in vector mode it runs AFTER the first tile group's loads have been issued (below).
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

    if (MODE != S_LSE) {
      for (int c = tid; c < CP; c += 256) {
        float scale = 1.f, shift = 0.f;
        if (c < p.Cin) {
          const Seg& s = (c < p.seg[0].C) ? p.seg[0] : p.seg[1];
          const int lc = (c < p.seg[0].C) ? c : c - p.seg[0].C;
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
        }
        s_sc[c] = scale;
        s_sh[c] = shift;
      }
      __syncthreads();
    }
  };
  if (!kVec) stats_to_lds();      // the element-wise and position-encoding loaders normalise while they load
  // lane-constant pieces
  float sc[KQ], sh[KQ];
  auto fill_scale_shift = [&]() {
#pragma unroll
    for (int j = 0; j < KQ; ++j) { sc[j] = (MODE != S_LSE) ? s_sc[c_lo + j] : 1.f; sh[j] = (MODE != S_LSE) ? s_sh[c_lo + j] : 0.f; }
  };
  if (!kVec) fill_scale_shift();
  h8 wh[kSplit ? NT : 1][NS], wl[kSplit ? NT : 1][NS];   // kSplit: the weight fragments as fp16 pairs (wf is dead after this)
  if (kSplit) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < NS; ++u) split8(&wf[t][8 * u], wh[t][u], wl[t][u]);
  }
  // vector mode: the chunk lies inside one segment
  const int C0 = p.seg[0].C;
  const bool in_seg1 = (MODE == S_VEC) && (p.nseg > 1) && (c_lo >= C0);
  const Seg& myseg = in_seg1 ? p.seg[1] : p.seg[0];
  const int seg_c = in_seg1 ? c_lo - C0 : c_lo;
  const float my_slope = myseg.act ? 0.2f : 1.f;   // LeakyReLU(v) = max(v, slope * v); slope 1 = identity
  const float* mybase = myseg.x + cloud * myseg.cloud_stride + seg_c;                     // per lane (segment of its chunk)
  const int32_t* myidx = myseg.idx ? myseg.idx + cloud * myseg.idx_cloud_stride : nullptr;
  const uint32_t my_ld = (uint32_t)myseg.ld;
  const float* uvb = MODE == S_UV ? p.seg[0].uv + cloud * p.seg[0].uv_cloud_stride : nullptr;
  const float* distb = MODE == S_UV ? p.seg[0].dist + cloud * p.seg[0].dist_cloud_stride : nullptr;
  float uva[KQ];
#pragma unroll
  for (int j = 0; j < KQ; ++j) uva[j] = MODE == S_UV ? p.seg[0].w8[(c_lo + j) * 8] : 0.f;

  const int ntiles = (p.M + 15) >> 4;
  // EPI_GN: the tile -> wave assignment is that of a VIRTUAL grid of vgrid_x workgroups per cloud (a function of M alone, so a cloud's
  // statistics are summed in the same order alone or in a batch); the physical workgroup bx plays the virtual ones bx, bx + grid_x, ...
  // one after the other - same sums, same commits, but weights, scale / shift and the launch of the workgroup are paid once
  const int vgx = EPI == EPI_GN ? p.vgrid_x : p.grid_x;
  const int nwaves = vgx * 4;

  // source row of this lane's A row in tile `tile` (gathered segments: one index load, issued a tile ahead)
  // Rows past M (last, partial tile) are CLAMPED to row M-1 rather than predicated: their MFMA results are
  // never stored nor counted, and unconditional loads keep the exec mask (and the branch count) out of the loop.
  auto tile_srow = [&](int tile) -> int {
    if (!kVec) return 0;
    const int row = min(tile * 16 + fr, p.M - 1);
    return myidx ? myidx[row] : row;
  };
  auto finish_tile = [&](int tile, Chunk<KQ>& ch) {
    if (!kVec) return;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
      const float v = fmaf(ch.v[j], sc[j], sh[j]);
      ch.v[j] = fmaxf(v, my_slope * v);
    }
  };
  auto load_tile = [&](int tile, int srow, Chunk<KQ>& ch) {
    const int row = tile * 16 + fr;
    const bool ok = row < p.M;
    if (MODE == S_VEC) {
      const float* src = mybase + (uint32_t)srow * my_ld;      // 32-bit row offset: per-cloud tensors are < 4 GiB
      vec_load<KQ>(src, ch.v);          // raw values; normalised by finish_tile() after the MFMA burst
    } else if (MODE == S_UV) {
      // lse_uv.hip: row (point i = row / 16, neighbour j = srow) of the position-encoding layer = a dist + U[j] + V[i], this
      // lane's KQ channels; the same expression - the same bits - its GroupNorm statistics were taken of
      const int rowc = min(row, p.M - 1);
      float u[KQ], v[KQ];
      vec_load<KQ>(uvb + (uint32_t)srow * (uint32_t)(2 * 4 * KQ) + c_lo, u);
      vec_load<KQ>(uvb + (uint32_t)(rowc >> 4) * (uint32_t)(2 * 4 * KQ) + 4 * KQ + c_lo, v);
      const float dd = distb[rowc];
#pragma unroll
      for (int j = 0; j < KQ; ++j) ch.v[j] = __fadd_rn(fmaf(uva[j], dd, u[j]), v[j]);
    } else if (MODE == S_ELEM) {
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        const int c = c_lo + j;
        float v = 0.f;
        if (ok && c < p.Cin) {
          const Seg& s = (c < C0) ? p.seg[0] : p.seg[1];
          const int lc = (c < C0) ? c : c - C0;
          const float x = s.x[cloud * s.cloud_stride + (int64_t)src_row(s, cloud, row) * s.ld + lc];
          v = fmaf(x, sc[j], sh[j]);
          if (s.act && v < 0.f) v *= 0.2f;
        }
        ch.v[j] = v;
      }
    } else {  // S_LSE: KQ == 3; channels [dist, rel(3), pi(3), pj(3), 0, 0]
      float e[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) e[j] = 0.f;
      if (ok) {
        const int i = row >> 4;
        const int jn = p.neigh[cloud * p.neigh_cloud_stride + row];
        const float* pi = p.xyz + cloud * p.xyz_cloud_stride + (int64_t)i * 3;
        const float* pj = p.xyz + cloud * p.xyz_cloud_stride + (int64_t)jn * 3;
        const float ix = pi[0], iy = pi[1], iz = pi[2], jx = pj[0], jy = pj[1], jz = pj[2];
        const float dx = __fsub_rn(jx, ix), dy = __fsub_rn(jy, iy), dz = __fsub_rn(jz, iz);
        e[0] = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        e[1] = dx; e[2] = dy; e[3] = dz; e[4] = ix; e[5] = iy; e[6] = iz; e[7] = jx; e[8] = jy; e[9] = jz;
      }
#pragma unroll
      for (int j = 0; j < KQ; ++j)
        ch.v[j] = fq == 0 ? e[j] : (fq == 1 ? e[3 + j] : (fq == 2 ? e[6 + j] : e[9 + j]));
    }
  };

  float g1[NT], g2[NT];   // GroupNorm partial sums of this lane's columns (reset per virtual workgroup)

  // Memory-level parallelism: a narrow tile is only 16 rows x 8..64 bytes, far too little to cover the
  // HBM/L2 latency with the few waves a CU holds.  Tiles are therefore processed in GROUPS of D: first the
  // gather indices of all D tiles, then all their row loads (D independent vector loads in flight per
  // lane), then the D MFMA + epilogue passes.  D shrinks as the per-tile register footprint grows.
  constexpr bool kAtt = (EPI == EPI_ATT || EPI == EPI_ATT2);
  constexpr int D = kAtt ? (NT == 1 ? 4 : 1) : (KQ * NT <= 4 ? 8 : (KQ * NT <= 16 ? 4 : (KQ * NT <= 32 ? 2 : 1)));
  constexpr int GA = EPI == EPI_ATT2 ? D : 1, GN_ = EPI == EPI_ATT2 ? NT : 1, GF_ = EPI == EPI_ATT2 ? NT / 2 : 1;
  // Two groups are alive at a time (ping-pong): the loads of group g+1 are issued before group g is computed,
  // so index -> row -> use latencies overlap with MFMA / epilogue work even at one wave per SIMD.
  struct Group {
    int srow[D];
    Chunk<KQ> buf[D];
    float gpre_all[GA][GN_][4], fpre_all[GA][GF_][4];
    int gi_all[GA][4];
  };
  // Vector mode issues every load unconditionally on CLAMPED tile / row indices (a group past the end re-reads the
  // last tile; its results are never used): straight-line code, no exec-mask or scalar branches between the loads,
  // so the compiler can keep them all in flight and count them precisely.
  auto issue_group = [&](Group& G, int tile0) {
    int (&srow)[D] = G.srow;
    Chunk<KQ> (&buf)[D] = G.buf;
    float (&gpre_all)[GA][GN_][4] = G.gpre_all;
    float (&fpre_all)[GA][GF_][4] = G.fpre_all;
    int (&gi_all)[GA][4] = G.gi_all;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int tl = tile0 + d * nwaves;
      const int tlc = min(tl, ntiles - 1);
      srow[d] = kVec ? tile_srow(tlc) : (tl < ntiles ? tile_srow(tl) : 0);
      if (EPI == EPI_ATT2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gi_all[d][r] = ld_i32(p.fseg.idx + cloud * p.fseg.idx_cloud_stride, 4u * (uint32_t)min(tlc * 16 + 4 * fq + r, p.M - 1));
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int tl = tile0 + d * nwaves;
      if (kVec) load_tile(min(tl, ntiles - 1), srow[d], buf[d]);
      else if (tl < ntiles) load_tile(tl, srow[d], buf[d]);
      if (EPI == EPI_ATT2) {
        // the gathered rows of G = W1 f (added to the scores; all NT tiles) and of f (pooled operand; the first
        // NT/2 tiles) for this lane's 4 rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* gp = p.g + cloud * p.g_cloud_stride;                 // wave-uniform bases
          const float* fp = p.fseg.x + cloud * p.fseg.cloud_stride;
          // G is stored in this kernel's order (engine.hip, up_fc_g): the block's column of lane fr in tiles 0..3 is one float4
          const uint32_t go = 4u * ((uint32_t)gi_all[d][r] * (uint32_t)p.Cout + (uint32_t)(n0 + 4 * fr));
          const uint32_t fo = 4u * ((uint32_t)gi_all[d][r] * (uint32_t)p.fseg.ld + (uint32_t)fr);
          static_assert(EPI != EPI_ATT2 || NT == 4, "one float4 of G per gathered row");
          const float4 g4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gp) + go);
          gpre_all[d][0][r] = g4.x; gpre_all[d][1 % GN_][r] = g4.y; gpre_all[d][2 % GN_][r] = g4.z; gpre_all[d][3 % GN_][r] = g4.w;
#pragma unroll
          for (int t = 0; t < NT / 2; ++t) fpre_all[d][t][r] = ld_f32(fp + col0_of(t), fo);
        }
      }
    }
  };
  auto compute_group = [&](Group& G, int tile0) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
    const int tile = tile0 + d * nwaves;
    if (tile < ntiles) {
    Chunk<KQ>& cur = G.buf[d];
    finish_tile(tile, cur);
    float (&gpre)[GN_][4] = G.gpre_all[EPI == EPI_ATT2 ? d : 0];
    float (&fpre)[GF_][4] = G.fpre_all[EPI == EPI_ATT2 ? d : 0];
    const int rbase = tile * 16 + 4 * fq;   // C layout: col = lane & 15, row = 4 * (lane >> 4) + reg

    f32x4 acc[NT];
    // cached enc-half scores: C fragments of tile `tile`, column tiles n0/16 .. n0/16 + NT
    float4* s2p = (SC != 0) ? reinterpret_cast<float4*>(p.s2 + cloud * p.s2_cloud_stride) +
                                  ((int64_t)tile * (p.Cout >> 4) + (n0 >> 4)) * 64 + lane
                            : nullptr;
    if (SC == 2) {
#pragma unroll
      for (int t = 0; t < NT; ++t) { const float4 v = s2p[t * 64]; acc[t] = f32x4{v.x, v.y, v.z, v.w}; }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kSplit) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          h8 ah, al;
          split8(&cur.v[8 * u], ah, al);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[t][u], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[t][u], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[t][u], acc[t], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < KQ; ++s)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v[s], wf[t][s], acc[t], 0, 0, 0);
      }
      if (SC == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) s2p[t * 64] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      }
    }

    if (EPI == EPI_GN) {
      float* Y = p.Y + cloud * p.y_cloud_stride;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + r;
          if (row < p.M && col < p.Cout) {
            const float v = acc[t][r] + bv[t];
            st_f32(Y, 4u * ((uint32_t)row * (uint32_t)p.ldy + (uint32_t)col), v);
            g1[t] += v;
            g2[t] += v * v;
          }
        }
      }
    } else if (EPI == EPI_ACT || EPI == EPI_LINEAR) {
      float* Y = p.Y + cloud * p.y_cloud_stride;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + r;
          if (row < p.M && col < p.Cout) {
            float v = acc[t][r] + bv[t];
            if (EPI == EPI_LINEAR && p.residual) v += ld_f32(p.residual + cloud * p.res_cloud_stride, 4u * ((uint32_t)row * (uint32_t)p.ldres + (uint32_t)col));
            if (EPI == EPI_ACT && v < 0.f) v *= 0.2f;
            st_f32(Y, 4u * ((uint32_t)row * (uint32_t)p.ldy + (uint32_t)col), v);
          }
        }
      }
    } else if (EPI == EPI_L2NORM) {
      float* Y = p.Y + cloud * p.y_cloud_stride;
      float ss[4] = {0.f, 0.f, 0.f, 0.f};
      float v[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = n0 + 16 * t + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[t][r] = (col < p.Cout) ? acc[t][r] + bv[t] : 0.f;
          ss[r] += v[t][r] * v[t][r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ss[r] += __shfl_xor(ss[r], 1); ss[r] += __shfl_xor(ss[r], 2);
        ss[r] += __shfl_xor(ss[r], 4); ss[r] += __shfl_xor(ss[r], 8);
        const float den = fmaxf(__fsqrt_rn(ss[r]), 1e-12f);
        const int row = rbase + r;
        if (row < p.M) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int col = n0 + 16 * t + fr;
            if (col < p.Cout) st_f32(Y, 4u * ((uint32_t)row * (uint32_t)p.ldy + (uint32_t)col), v[t][r] / den);
          }
        }
      }
    } else if (EPI == EPI_ATT) {
      // the tile's 16 rows are the 16 neighbours of point `tile` (RandLANet.py:152-155).
      // f[row][col] (already normalised) is in the A fragments of OTHER lanes: transpose it
      // through a wave-private LDS tile [16][CP+4] (conflict-free: rows shift by 4 banks).
      // Requires gridDim.y == 1 (Cout == Cin <= 64), which launch_pw_stream guarantees.
      float* Y = p.Y + cloud * p.y_cloud_stride;
      float* T = &s_att[w * 16 * (CP + 4)];
#pragma unroll
      for (int j = 0; j < KQ; ++j) T[fr * (CP + 4) + c_lo + j] = cur.v[j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + fr;
        float f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = T[(4 * fq + r) * (CP + 4) + col];
        const float o = att_pool_tile(acc[t], f);
        if (lane < 16 && col < p.Cout) st_f32(Y, 4u * ((uint32_t)tile * (uint32_t)p.ldy + (uint32_t)col), o);
      }
      __builtin_amdgcn_wave_barrier();
    } else if (EPI == EPI_ATT2) {
      // split attentive pooling: scores = acc (enc half of the contraction) + gathered G rows;
      // pooled operand = [gathered f (tiles t < NT/2) ; enc (tiles t >= NT/2, from the A fragments via LDS)]
      float* Y = p.Y + cloud * p.y_cloud_stride;
      float* T = &s_att[w * 16 * (CP + 4)];
      const int ch = p.fseg.C;
      const float fslope = p.fseg.act ? 0.2f : 1.f;
#pragma unroll
      for (int j = 0; j < KQ; ++j) T[fr * (CP + 4) + c_lo + j] = cur.v[j];
      __builtin_amdgcn_wave_barrier();
      float o[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = col0_of(t) + fr;
        float f[4];
        f32x4 sc4 = acc[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc4[r] += gpre[t][r];
          if (t < NT / 2) {
            const float v = fmaf(fpre[t < NT / 2 ? t : 0][r], s_fsc[col], s_fsh[col]);
            f[r] = fmaxf(v, fslope * v);
          } else {
            f[r] = T[(4 * fq + r) * (CP + 4) + (col - ch)];
          }
        }
        o[t] = att_pool_tile(sc4, f);
      }
      // every lane holds its column's result for all NT tiles: lane group fq stores tile fq -> one full-wave store
      static_assert(EPI != EPI_ATT2 || NT == 4, "EPI_ATT2 stores one tile per 16-lane group");
      const float ov = fq == 0 ? o[0] : (fq == 1 ? o[1 % NT] : (fq == 2 ? o[2 % NT] : o[3 % NT]));
      st_f32(Y, 4u * ((uint32_t)tile * (uint32_t)p.ldy + (uint32_t)(col0_of(fq) + fr)), ov);
      __builtin_amdgcn_wave_barrier();
    }
    }  // tile < ntiles
    }  // d
  };
  bool first = true;
  for (int vb = bx; vb < vgx; vb += p.grid_x) {
#pragma unroll
  for (int t = 0; t < NT; ++t) { g1[t] = 0.f; g2[t] = 0.f; }
  {
    const int gstride = D * nwaves;
    Group ga, gb;
    int t0 = vb * 4 + w;
    if (t0 < ntiles) issue_group(ga, t0);
    if (kVec && first) { stats_to_lds(); fill_scale_shift(); }     // the first group's loads are in flight meanwhile
    first = false;
    while (t0 < ntiles) {
      const int t1 = t0 + gstride;
      if (t1 < ntiles) issue_group(gb, t1);
      compute_group(ga, t0);
      if (t1 >= ntiles) break;
      const int t2 = t1 + gstride;
      if (t2 < ntiles) issue_group(ga, t2);
      compute_group(gb, t1);
      t0 = t2;
    }
  }

  if (EPI == EPI_GN) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s1 = g1[t], s2 = g2[t];
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      if (lane < 16) {
        s_red[(w * BN + 16 * t + lane) * 2 + 0] = s1;
        s_red[(w * BN + 16 * t + lane) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    const int gw = p.Cout / p.groups_out;
    if (tid < BN) {
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) { c1 += s_red[(ww * BN + tid) * 2]; c2 += s_red[(ww * BN + tid) * 2 + 1]; }
      s_red[tid * 2] = c1;
      s_red[tid * 2 + 1] = c2;
    }
    __syncthreads();
    // ONE atomic instruction per workgroup for all its groups (device_utils.h, gn_block_commit)
    gn_block_commit(s_red, n0, min(BN, p.Cout - n0), gw, p.stats_out + (int64_t)cloud * p.groups_out * kGnWords);
    if (vb + p.grid_x < vgx) __syncthreads();      // s_red is free again before the next virtual workgroup's sums land in it
  }
  }  // vb
}

// Row-block units per cloud and column block: ~8 tiles per wave (32 per workgroup); small layers: at least `floor_blocks` workgroups
// per cloud so that a single pair still spreads over the chip (batch-1 latency: more, shorter waves).  A function of (M, gy) alone.
// For the GroupNorm layers this count is the VIRTUAL grid - it fixes which wave sums which tiles, and it is the number of
// contributions a statistic of the cloud receives (pw_stream_gn_contributions).  DSIR_STREAM_MIN_BLOCKS: tuning hook.
int stream_blocks(int M, int gy, bool big) {
  const int ntiles = (M + 15) / 16;
  int blocks = (ntiles + 31) / 32;
  static const int floor_blocks = (int)tuning_int("DSIR_STREAM_MIN_BLOCKS", 16);
  if (!big && blocks * gy < floor_blocks) {
    const int want = (floor_blocks + gy - 1) / gy, most = (ntiles + 3) / 4;
    blocks = want < most ? want : most;
  }
  return blocks < 1 ? 1 : blocks;
}

template <int KQ, int NT, int EPI, int MODE, int SC = 0>
void launch_s(const GemmArgs& a, hipStream_t st) {
  const int ntiles = (a.M + 15) / 16;
  const int gy = (a.Cout + NT * 16 - 1) / (NT * 16);
  // The grid depends on (M, Cout) only — never on the number of clouds — so the
  // tile->wave assignment, hence the summation order of the GroupNorm statistics, is the same for a
  // cloud whether it is registered alone or inside a batch (bitwise batch invariance).
  const int natural = (ntiles + 31) / 32;
  const bool big = EPI != EPI_GN && (int64_t)natural * gy * a.clouds >= 512;   // no statistics, chip already full: ~8 tiles per wave stand
  int blocks = stream_blocks(a.M, gy, big);
  if (EPI == EPI_GN && blocks > kGnMaxContrib) {     // dsir_create bounds max_points so that this cannot happen (kernels.h)
    fprintf(stderr, "dsir: pw_stream: %d contributions per GroupNorm statistic exceed the exactness bound %d\n", blocks, kGnMaxContrib);
    abort();
  }
  // Epilogues without a cross-workgroup reduction (everything but the GroupNorm statistics) give the same bits under any
  // tile -> wave assignment, so their grid may follow the launch size: with one or two clouds in flight (batch-1 latency)
  // a cloud spreads over enough workgroups to reach ~2 per CU; with many clouds nothing changes.
  if (EPI != EPI_GN) {
    const int64_t total = (int64_t)blocks * gy * a.clouds;
    if (total < 512) {
      const int want = (int)((512 + (int64_t)gy * a.clouds - 1) / ((int64_t)gy * a.clouds)), most = (ntiles + 3) / 4;
      const int nb = want < most ? want : most;
      if (nb > blocks) blocks = nb;
    }
  }
  GemmArgs b = a;
  b.vgrid_x = blocks;
  // EPI_GN: `blocks` fixes the summation order (above); how many PHYSICAL workgroups play them follows the launch size: the natural
  // grid (~8 tiles per wave) once that alone fills the chip, else about one residency round (512 workgroups), never more than the
  // virtual grid.  A physical workgroup walks the virtual ones bx, bx + grid_x, ...: weights, scale / shift and its own launch are
  // paid once, sums and commits are those of the virtual grid - same bits at every launch size.
  if (EPI == EPI_GN) {
    static const int phys_target = (int)tuning_int("DSIR_STREAM_PHYS_BLOCKS", 512);   // tuning hook; 0 = one workgroup per virtual one
    if (phys_target > 0) {
      const int64_t per = (int64_t)gy * a.clouds;
      int64_t want = (int64_t)natural * per >= phys_target ? natural : (phys_target + per - 1) / per;
      if (want < 1) want = 1;
      if (want < blocks) blocks = (int)want;
    }
  }
  b.grid_x = blocks; b.grid_y = gy;
  dim3 grid((unsigned)((int64_t)blocks * gy * a.clouds));
  hipLaunchKernelGGL((pw_stream_kernel<KQ, NT, EPI, MODE, SC>), grid, dim3(256), 0, st, b);
}

template <int KQ, int NT, int MODE>
bool launch_epi(const GemmArgs& a, hipStream_t st) {
  switch (a.epi) {
    case EPI_GN: launch_s<KQ, NT, EPI_GN, MODE>(a, st); return true;
    case EPI_ACT: if (MODE == S_LSE) return false; launch_s<KQ, NT, EPI_ACT, MODE == S_LSE ? S_VEC : MODE>(a, st); return true;
    case EPI_LINEAR: if (MODE == S_LSE) return false; launch_s<KQ, NT, EPI_LINEAR, MODE == S_LSE ? S_VEC : MODE>(a, st); return true;
    default: return false;
  }
}

template <int KQ, int MODE>
bool launch_nt(const GemmArgs& a, hipStream_t st) {
  if (a.Cout <= 16) return launch_epi<KQ, 1, MODE>(a, st);
  if (a.Cout <= 32) return launch_epi<KQ, 2, MODE>(a, st);
  return launch_epi<KQ, 4, MODE>(a, st);
}

bool seg_vec_ok(const Seg& s, int KQ) {
  const int al = KQ >= 4 ? 4 : KQ;
  return (s.ld % al) == 0 && (s.cloud_stride % al) == 0 && (reinterpret_cast<uintptr_t>(s.x) % (al * 4)) == 0;
}

}  // namespace

// workgroups of one cloud that add into one GroupNorm statistic of a layer served here (a group never spans two column blocks:
// its width divides Cout / 4 <= 16 NT)
int pw_stream_gn_contributions(int M, int Cout) {
  const int nt16 = Cout <= 16 ? 16 : (Cout <= 32 ? 32 : 64);
  return stream_blocks(M, (Cout + nt16 - 1) / nt16, false);
}

// Returns false when the layer is outside this kernel's envelope (caller falls back to pw_gemm.hip).
bool launch_pw_stream(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.clouds <= 0) return true;
  if (a.c_split > 0) return false;             // two-layer launches: pw_tile_small_kernel only
  if (a.amode == A_LSE) {
    if (a.epi != EPI_GN) return false;
    return launch_nt<3, S_LSE>(a, st);
  }
  if (a.Cin > 64) return false;
  if (a.seg[0].uv) {     // rows rebuilt from per-point tables (lse_uv.hip): one segment of 8 or 32 channels, GroupNorm epilogue
    if (a.nseg != 1 || a.epi != EPI_GN || !a.seg[0].idx || !a.seg[0].dist || !a.seg[0].w8 || (a.M % 16) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.seg[0].uv) % 16) != 0 || (a.seg[0].uv_cloud_stride % 4) != 0) return false;
    if (a.Cin == 8) return launch_nt<2, S_UV>(a, st);
    if (a.Cin == 32) return launch_nt<8, S_UV>(a, st);
    return false;
  }
  const int C0 = a.seg[0].C;
  // vector mode: Cin = 4 KQ exactly, chunks do not straddle the segment boundary, aligned rows
  for (int KQ : {2, 4, 8, 16}) {
    if (a.Cin != 4 * KQ) continue;
    bool ok = seg_vec_ok(a.seg[0], KQ) && (a.nseg == 1 || (seg_vec_ok(a.seg[1], KQ) && (C0 % KQ) == 0));
    if (!ok) break;
    if (a.epi == EPI_ATT) {
      if (KQ == 4 && a.Cout == 16) { launch_s<4, 1, EPI_ATT, S_VEC>(a, st); return true; }
      if (KQ == 16 && a.Cout == 64) { launch_s<16, 4, EPI_ATT, S_VEC>(a, st); return true; }
      return false;
    }
    if (a.epi == EPI_L2NORM) {
      if (KQ == 16 && a.Cout == 64) { launch_s<16, 4, EPI_L2NORM, S_VEC>(a, st); return true; }
      return false;
    }
    if (a.epi == EPI_ATT2) {   // A = enc (Cin = d/2), Cout = d
      if (a.nseg != 1 || a.Cout != 2 * a.Cin || !a.g || !a.fseg.idx) return false;
      const int sc = a.s2 ? a.s2_mode : 0;
      if (KQ == 8) {                                                            // d = 64
        if (sc == 1) launch_s<8, 4, EPI_ATT2, S_VEC, 1>(a, st);
        else if (sc == 2) launch_s<8, 4, EPI_ATT2, S_VEC, 2>(a, st);
        else launch_s<8, 4, EPI_ATT2, S_VEC>(a, st);
        return true;
      }
      if (KQ == 16) {                                                           // d = 128 (two column blocks)
        if (sc == 1) launch_s<16, 4, EPI_ATT2, S_VEC, 1>(a, st);
        else if (sc == 2) launch_s<16, 4, EPI_ATT2, S_VEC, 2>(a, st);
        else launch_s<16, 4, EPI_ATT2, S_VEC>(a, st);
        return true;
      }
      return false;
    }
    switch (KQ) {
      case 2: return launch_nt<2, S_VEC>(a, st);
      case 4: return launch_nt<4, S_VEC>(a, st);
      case 8: return launch_nt<8, S_VEC>(a, st);
      case 16: return launch_nt<16, S_VEC>(a, st);
    }
  }
  if (a.Cin <= 8 && (a.epi == EPI_GN || a.epi == EPI_ACT || a.epi == EPI_LINEAR)) {
    if (a.Cin <= 4) return launch_nt<1, S_ELEM>(a, st);
    return launch_nt<2, S_ELEM>(a, st);
  }
  return false;
}

}  // namespace dsir
