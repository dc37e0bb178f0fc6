// Attentive pooling (reference network/RandLANet.py:140-157) for the k = 16 layers of pyramid levels 1 / 2 (d = 64 / 128):
//   y[i][c] = sum_k softmax_k(S[k][c]) X[k][c],   X[k] = [ fN[nb(i,k)] ; E[i,k] ],   S[k] = fc X[k]
// with the score GEMM split by linearity: S[k] = G[nb(i,k)] + W2 E[i,k], G = W1 fN a per-POINT GEMM made
// beforehand.  Round 4: the previous kernels (pw_stream.hip EPI_ATT2) were bound by VALU issue - ~350 vector instructions per
// point, most of them the cross-lane softmax butterflies of a 16 x 16 accumulator tile, gather address arithmetic and
// operand plumbing (PMC: 58 % VALU + 30 % MFMA busy).  This kernel is organised around the ACCUMULATOR LAYOUT instead:
//   * a wave owns units of TWO points = 32 rows = one row tile of v_mfma_f32_32x32x16_f16.  A-row m carries neighbour
//     k = 4 (m >> 3) + (m & 3) of point (m >> 2) & 1: the accumulator rows a lane holds - (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
//     cdna_hip_programming.md section 3 - are then exactly the 16 neighbours of ONE point (lane half = point, register =
//     neighbour).  The softmax over the neighbours and the weighted sum run in registers: no cross-lane instruction at all;
//   * a lane's column is lane & 31: every gathered operand of the epilogue - G[nb][c], fN[nb][c] - is one dword load whose 32
//     lanes read 128 contiguous bytes of the neighbour's row.  G and fN live in ONE row buffer gp = [G (d) | fN (d/2)] written
//     by the per-point GEMM (its weight matrix carries an identity block, engine.hip::up_fc_p: fN x 1.0 is exact in the fp32
//     MFMA), so one address per neighbour serves all three loads through immediate offsets;
//   * the contraction W2 E runs on the fp16 matrix pipe at fp32 accuracy (x = fp16(x) + fp16(x - fp16(x)), three MFMAs per
//     product: agg_chain_h.hip); weights split at load (GemmArgs-style blob offsets);
//   * E (normalised: the producer's GroupNorm + LeakyReLU applied while the A operand is formed) reaches the epilogue's
//     column layout through a wave-private LDS tile [32][KH + 8] (conflict-free dword reads);
//   * software pipeline: the next unit's neighbour indices and E rows are in flight during the current unit's epilogue.
// A block owns 64 columns: the 32 of the gathered-feature half that start at 32 cb and the matching 32 of the enc half.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

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

template <int KH>   // enc channels per row = d / 2: 32 (level 1) or 64 (level 2)
__global__ __launch_bounds__(256) void att_pool_kernel(const AttPoolArgs p) {
  constexpr int KC = KH / 2;     // channels of its row a lane holds: [h KC, (h + 1) KC), 8 of them per k-step
  constexpr int NS = KH / 16;    // k-steps
  constexpr int LD = KH + 8;     // LDS row stride in floats: rows 4 apart land 32 banks apart
  constexpr int NCB = KH / 32;   // column blocks
  constexpr uint32_t ROWB = 3u * KH * 4u;   // bytes per row of gp
  __shared__ float s_sc[KH];
  __shared__ float s_sh[KH];
  __shared__ __attribute__((aligned(16))) float s_t[4][32 * LD];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);     // whole clouds per XCD: a cloud's gathered rows stay in one L2
  const int bx = wi % p.grid_x;
  const int cb = (wi / p.grid_x) % NCB;
  const int cloud = wi / (p.grid_x * NCB);
  const int d = 2 * KH;

  // GroupNorm (+ LeakyReLU) of the producer of E, per channel
  for (int c = tid; c < KH; c += 256) {
    float scale = 1.f, shift = 0.f;
    if (p.enc_gn.stats) {
      const int g = c / (KH / p.enc_gn.groups);
      const double* st = p.enc_gn.stats + ((int64_t)cloud * p.enc_gn.groups + g) * kGnWords;
      const double mean = gn_stat_get(st) * p.enc_gn.inv_count;
      double var = gn_stat_get(st + 2) * p.enc_gn.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + 1e-5);
      const double scd = (double)p.enc_gn.gamma[c] * rstd;
      scale = (float)scd;
      shift = (float)((double)p.enc_gn.beta[c] - mean * scd);
    }
    s_sc[c] = scale;
    s_sh[c] = shift;
  }
  __syncthreads();

  // B fragments: tile 0 = columns 32 cb + m of the gathered-feature half, tile 1 = the same of the enc half; the k index of step s,
  // lane half h, element j is channel h KC + 8 s + j of E (any bijection serves as long as A and B agree: this one makes a
  // lane's A chunk contiguous)
  const _Float16* Wh = reinterpret_cast<const _Float16*>(p.Wh);
  const _Float16* Wl = reinterpret_cast<const _Float16*>(p.Wl);
  h8 wh[2][NS], wl[2][NS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = t * KH + 32 * cb + m;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int64_t o = (int64_t)col * p.ldw + p.wcol0 + h * KC + 8 * s;
      wh[t][s] = *reinterpret_cast<const h8*>(Wh + o);
      wl[t][s] = *reinterpret_cast<const h8*>(Wl + o);
    }
  }
  const float slope = p.enc_act ? 0.2f : 1.f;
  const int pm = (m >> 2) & 1, km = ((m >> 3) << 2) | (m & 3);      // the (point, neighbour) of this lane's A row
  const float* encb = p.enc + cloud * p.enc_cs + h * KC;
  const int32_t* nbb = p.neigh + cloud * p.neigh_cs;
  const float* gpb = p.gp + cloud * p.gp_cs;      // [G columns of the gathered-feature half | of the enc half | fN]
  const float* gpbE = gpb + KH;
  const float* gpbX = gpb + 2 * KH;
  float* Yb = p.Y + cloud * p.y_cs;
  float* T = &s_t[w][0];
  const uint32_t coff = 4u * (uint32_t)(32 * cb + m);

  const int units = (p.n + 1) >> 1;
  const int nw = p.grid_x * 4;
  int u = bx * 4 + w;

  float a[KC];          // this lane's raw E chunk of the CURRENT unit
  int nb[16];           // the 16 neighbours of this lane's point
  auto load_unit = [&](int uu) {
    const int pa = min(2 * uu + pm, p.n - 1);                       // clamped: results of a padding point are never stored
    const float* src = encb + ((uint32_t)(pa * 16 + km)) * (uint32_t)KH;
#pragma unroll
    for (int q = 0; q < KC / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    const int pe = min(2 * uu + h, p.n - 1);
    const int4* ip = reinterpret_cast<const int4*>(nbb + (uint32_t)pe * 16u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = ip[q];
      nb[4 * q] = v.x; nb[4 * q + 1] = v.y; nb[4 * q + 2] = v.z; nb[4 * q + 3] = v.w;
    }
  };
  if (u < units) load_unit(u);
  while (u < units) {
    // ---- gathers of the first tile's epilogue (scores' G half and the pooled features): issued first, consumed last.
    // One 32-bit offset per neighbour, wave-uniform bases: global_load_dword v, v_off, s[base]
    uint32_t off[16];
    float gF[16], xF[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      off[i] = __umul24((uint32_t)nb[i], ROWB) + coff;
      gF[i] = ld_f32(gpb, off[i]);
      xF[i] = ld_f32(gpbX, off[i]);
    }
    // ---- A operand: normalise (GroupNorm + LeakyReLU of the producer), keep fp32 for the pooled operand, split for the MFMAs
    h8 ah[NS], al[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * s + j;
        const float v = fmaf(a[c], s_sc[h * KC + c], s_sh[h * KC + c]);
        a[c] = fmaxf(v, slope * v);
      }
      split8f(&a[8 * s], ah[s], al[s]);
    }
#pragma unroll
    for (int q = 0; q < KC / 4; ++q)
      *reinterpret_cast<float4*>(&T[m * LD + h * KC + 4 * q]) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    // ---- next unit's rows and indices: in flight during the MFMAs and the epilogue
    const int pt = 2 * u + h;
    const int un = u + nw;
    if (un < units) load_unit(un);
    // ---- scores of the enc half: 3 fp16 MFMAs per (tile, k-step)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], wh[t][s], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], wl[t][s], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], wh[t][s], acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the second tile's G rows: in flight during the first tile's epilogue
    float gE[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) gE[i] = ld_f32(gpbE, off[i]);
    // ---- epilogue: register i of the accumulator = neighbour i of point (lane >> 5), column lane & 31
    constexpr float L2E = 1.44269504088896340736f;
    float y[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float sc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = acc[t][i] + (t == 0 ? gF[i] : gE[i]);
      float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
      for (int i = 2; i < 16; i += 2) mx = fmaxf(mx, fmaxf(sc[i], sc[i + 1]));     // v_max3_f32
      const float ml = -mx * L2E;
      float se = 0.f, o = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // softmax numerator e^(s - max) = 2^(s log2e - max log2e): one fma + v_exp_f32 (1 ulp), arguments <= 0 up to rounding
        const float e = __builtin_amdgcn_exp2f(fmaf(sc[i], L2E, ml));
        const float x = t == 0 ? xF[i] : T[(8 * (i >> 2) + 4 * h + (i & 3)) * LD + 32 * cb + m];
        se += e;
        o = fmaf(x, e, o);
      }
      y[t] = o * __builtin_amdgcn_rcpf(se);      // se >= ~1: the max term contributes 2^0
    }
    if (pt < p.n) {
      st_f32(Yb, 4u * ((uint32_t)pt * (uint32_t)d) + coff, y[0]);
      st_f32(Yb, 4u * ((uint32_t)pt * (uint32_t)d + (uint32_t)KH) + coff, y[1]);
    }
    __builtin_amdgcn_wave_barrier();
    u = un;
  }
}

template <int KH>
void launch_k(const AttPoolArgs& a, hipStream_t st) {
  const int units = (a.n + 1) / 2;
  // ~8 units per wave; no cross-workgroup reduction in this kernel, so the grid may follow the launch size (same bits under
  // any unit -> wave assignment): a single cloud still spreads over the chip
  int blocks = (units + 31) / 32;
  const int ncb = KH / 32;
  const int64_t total = (int64_t)blocks * ncb * a.clouds;
  if (total < 512) {
    const int want = (int)((512 + (int64_t)ncb * a.clouds - 1) / ((int64_t)ncb * a.clouds)), most = (units + 3) / 4;
    const int nb = want < most ? want : most;
    if (nb > blocks) blocks = nb;
  }
  if (blocks < 1) blocks = 1;
  AttPoolArgs b = a;
  b.grid_x = blocks;
  hipLaunchKernelGGL((att_pool_kernel<KH>), dim3((unsigned)((int64_t)blocks * ncb * a.clouds)), dim3(256), 0, st, b);
}

}  // namespace

bool launch_att_pool(const AttPoolArgs& a, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  if (!a.Wh || !a.Wl || !a.enc || !a.gp || !a.neigh || !a.Y) return false;
  if ((a.ldw % 8) != 0 || (a.wcol0 % 8) != 0 || (reinterpret_cast<uintptr_t>(a.Wh) % 16) != 0 || (reinterpret_cast<uintptr_t>(a.Wl) % 16) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.enc) % 16) != 0 || (a.enc_cs % 4) != 0 || (reinterpret_cast<uintptr_t>(a.neigh) % 16) != 0 || (a.neigh_cs % 4) != 0) return false;
  if (a.enc_gn.stats && (a.KH % a.enc_gn.groups) != 0) return false;
  // 32-bit byte offsets inside a cloud
  if ((int64_t)a.n * 16 * a.KH * 4 >= ((int64_t)1 << 32) || (int64_t)a.n * 3 * a.KH * 4 >= ((int64_t)1 << 32)) return false;
  switch (a.KH) {
    case 32: launch_k<32>(a, st); return true;
    case 64: launch_k<64>(a, st); return true;
    default: return false;
  }
}

}  // namespace dsir
