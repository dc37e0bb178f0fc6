// Point-wise GEMM for the wide layers (Cin a multiple of 32, up to 768): pyramid
// levels 2/3, the decoder and the 128/256-wide aggregation MLP layers.  These are
// real GEMMs (K = 128..768, N = 64..512) over few rows per cloud, so the bound is
// the exact-fp32 MFMA rate and the enemy is latency: no LDS staging and no barrier
// in the K loop.
//
//   * a wave owns a (16*RT rows) x 64 columns output tile (RT*4 accumulators) and
//     walks K in 32-channel chunks; lane (r = lane & 15, q = lane >> 4) holds the
//     contiguous channels [kc + 8q, kc + 8q + 8) of its A rows and of its four
//     weight rows (same k re-ordering as pw_stream.hip), i.e. two 16-byte loads per
//     row per chunk straight from global/L2 — every weight fragment is re-used by RT
//     row tiles, every A fragment by 4 column tiles: (RT+4)*2 loads per RT*32 MFMAs;
//   * the next chunk's fragments are loaded into a second register set before the
//     current chunk's MFMAs (software pipeline, no waits inside the MFMA burst);
//   * the 4 waves of a block take different (row group, column tile) pairs so that
//     A rows / weight rows are shared through L1;
//   * the producer's GroupNorm scale/shift table (<= 768 channels) is the only LDS
//     use; it is built once per block.
// Epilogues as in pw_gemm.hip: GroupNorm statistics (reduced inside the wave, one
// fp64 atomic per group per wave), bias + LeakyReLU, linear (+residual),
// attentive pooling.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr int KC = 32;    // channels per K chunk
constexpr int KQ = 8;     // channels per lane per chunk
constexpr int NT = 4;     // 16-column tiles per wave
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

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[KQ]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <int RT, int EPI>
__global__ __launch_bounds__(256) void pw_deep_kernel(const GemmArgs p, int ctb /* column tiles per block: 1, 2 or 4 */) {
  __shared__ float s_sc[MAXC];
  __shared__ float s_sh[MAXC];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int cloud = blockIdx.z;

  for (int c = tid; c < p.Cin; c += 256) {
    const Seg& s = (c < p.seg[0].C) ? p.seg[0] : p.seg[1];
    const int lc = (c < p.seg[0].C) ? c : c - p.seg[0].C;
    float scale = 1.f, shift = 0.f;
    if (s.gn.stats) {
      const int g = lc / (s.C / s.gn.groups);
      const double* st = s.gn.stats + ((int64_t)cloud * s.gn.groups + g) * 2;
      const double mean = st[0] * s.gn.inv_count;
      double var = st[1] * s.gn.inv_count - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + 1e-5);
      const double scd = (double)s.gn.gamma[lc] * rstd;
      scale = (float)scd;
      shift = (float)((double)s.gn.beta[lc] - mean * scd);
    }
    s_sc[c] = scale;
    s_sh[c] = shift;
  }
  __syncthreads();

  // wave -> (row group, column tile)
  const int rgb = 4 / ctb;
  const int rg = blockIdx.x * rgb + w / ctb;
  const int ct = blockIdx.y * ctb + w % ctb;
  const int r0 = rg * 16 * RT;
  const int n0 = ct * 64;
  if (r0 >= p.M || n0 >= p.Cout) return;   // whole wave idle (no barrier follows)

  RowOff ro[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) ro[rt] = row_off(p, cloud, r0 + 16 * rt + fr);
  const int C0 = p.seg[0].C;
  const int act0 = p.seg[0].act, act1 = p.nseg > 1 ? p.seg[1].act : 0;
  const float* Wl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + 16 * t + fr;
    Wl[t] = (col < p.Cout) ? p.W + (int64_t)col * (p.ldw ? p.ldw : p.Cin) + KQ * fq : nullptr;
  }

  float a_cur[RT][KQ], a_nxt[RT][KQ], w_cur[NT][KQ], w_nxt[NT][KQ];
  // issue the global loads of one K chunk (raw values; nothing here depends on their arrival)
  auto issue_loads = [&](int kc, float (&A)[RT][KQ], float (&Wf)[NT][KQ]) {
    const int c = kc + KQ * fq;               // first channel of this lane's slice
    const bool s1 = c >= C0;
    const float* base = s1 ? p.seg[1].x : p.seg[0].x;
    const int lc = s1 ? c - C0 : c;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int64_t o = s1 ? ro[rt].o1 : ro[rt].o0;
      if (o >= 0) load8(base + o + lc, A[rt]);
      else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) A[rt][j] = 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (Wl[t]) load8(Wl[t] + kc, Wf[t]);
      else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) Wf[t][j] = 0.f;
      }
    }
  };
  // the producer's GroupNorm + LeakyReLU, applied to a chunk once its loads have landed
  auto normalise = [&](int kc, float (&A)[RT][KQ]) {
    const int c = kc + KQ * fq;
    const int act = (c >= C0) ? act1 : act0;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
      const float sc = s_sc[c + j], sh = s_sh[c + j];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float v = fmaf(A[rt][j], sc, sh);
        A[rt][j] = (ro[rt].o0 >= 0) ? ((act && v < 0.f) ? 0.2f * v : v) : 0.f;
      }
    }
  };

  f32x4 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Cin / KC;
  issue_loads(0, a_cur, w_cur);
  normalise(0, a_cur);
  for (int kc = 0; kc < nchunks; ++kc) {
    const bool more = kc + 1 < nchunks;
    if (more) issue_loads((kc + 1) * KC, a_nxt, w_nxt);     // in flight during the MFMA burst below
#pragma unroll
    for (int s = 0; s < KQ; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rt][s], w_cur[t][s], acc[rt][t], 0, 0, 0);
    if (more) {
      normalise((kc + 1) * KC, a_nxt);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < KQ; ++j) a_cur[rt][j] = a_nxt[rt][j];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < KQ; ++j) w_cur[t][j] = w_nxt[t][j];
    }
  }

  // ---- epilogues.  C layout: col = lane & 15, row = 4 * (lane >> 4) + reg.
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
    // reduce inside the wave down to one value per GroupNorm group (gw = 8, 16, 32 or 64 channels)
    const int gw = p.Cout / p.groups_out;
    const int lw = gw < 16 ? gw : 16;     // lanes of a 16-column tile that belong to one group
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] += __shfl_xor(s1[t], 16); s1[t] += __shfl_xor(s1[t], 32);
      s2[t] += __shfl_xor(s2[t], 16); s2[t] += __shfl_xor(s2[t], 32);
      for (int o = 1; o < lw; o <<= 1) { s1[t] += __shfl_xor(s1[t], o); s2[t] += __shfl_xor(s2[t], o); }
    }
    if (fq == 0 && (fr % lw) == 0) {
      const int tpg = gw > 16 ? gw / 16 : 1;   // column tiles per group
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if ((t % tpg) != 0) continue;
        const int col = n0 + 16 * t + fr;
        if (col >= p.Cout) continue;
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int u = 0; u < NT; ++u)
          if (u >= t && u < t + tpg) { d1 += (double)s1[u]; d2 += (double)s2[u]; }
        double* st = p.stats_out + ((int64_t)cloud * p.groups_out + col / gw) * 2;
        atomicAdd(st, d1);
        atomicAdd(st + 1, d2);
      }
    }
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
    // each 16-row tile = the 16 neighbours of one point; f[row][col] is re-read (L1/L2 hit)
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
  }
}

template <int RT, int EPI>
void launch_d(const GemmArgs& a, hipStream_t st) {
  const int CT = (a.Cout + 63) / 64;
  const int ctb = CT >= 4 ? 4 : (CT >= 2 ? 2 : 1);
  const int rgb = 4 / ctb;
  const int row_groups = (a.M + 16 * RT - 1) / (16 * RT);
  dim3 grid((row_groups + rgb - 1) / rgb, (CT + ctb - 1) / ctb, a.clouds);
  hipLaunchKernelGGL((pw_deep_kernel<RT, EPI>), grid, dim3(256), 0, st, a, ctb);
}

template <int RT>
bool launch_e(const GemmArgs& a, hipStream_t st) {
  switch (a.epi) {
    case EPI_GN: launch_d<RT, EPI_GN>(a, st); return true;
    case EPI_ACT: launch_d<RT, EPI_ACT>(a, st); return true;
    case EPI_LINEAR: launch_d<RT, EPI_LINEAR>(a, st); return true;
    case EPI_ATT: launch_d<RT, EPI_ATT>(a, st); return true;
    default: return false;
  }
}

bool seg_ok(const Seg& s) {
  return (s.ld % 4) == 0 && (s.cloud_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(s.x) % 16) == 0;
}

}  // namespace

// Returns false when the layer is outside this kernel's envelope (caller falls back to pw_gemm.hip).
bool launch_pw_deep(const GemmArgs& a, hipStream_t st) {
  if (a.amode != A_SEGS || a.Cin < 64 || a.Cin > MAXC || (a.Cin % KC) != 0 || a.Cout < 64) return false;
  if (!seg_ok(a.seg[0]) || (a.nseg > 1 && (!seg_ok(a.seg[1]) || (a.seg[0].C % KQ) != 0))) return false;
  if ((reinterpret_cast<uintptr_t>(a.W) % 16) != 0) return false;
  if (a.epi == EPI_GN && ((a.Cout / a.groups_out) % 8) != 0) return false;
  // rows per wave tile: a function of M only (batch-invariant tiling)
  if (a.M >= 192) return launch_e<4>(a, st);
  if (a.M >= 24) return launch_e<2>(a, st);
  return launch_e<1>(a, st);
}

}  // namespace dsir
