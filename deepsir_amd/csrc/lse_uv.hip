// The position-encoding layer lfa.mlp1 of pyramid levels 0 / 1 WITHOUT its output in memory (reference network/RandLANet.py:197-212
// relative_pos_encoding + :58-107 MLP2D).
//
// The layer is a 1 x 1 convolution of the 10-channel code [|p_j - p_i|, p_j - p_i, p_i, p_j] of every (point i, neighbour j)
// pair.  Nine of the ten inputs are linear in the two points, so the raw output splits by linearity into per-POINT parts:
//     enc_raw[i, k][c] = a[c] dist(i, j) + U[j][c] + V[i][c],      j = neigh[i][k],
//     a = W[:, 0],   U[j] = (W[:, 1:4] + W[:, 7:10]) p_j,   V[i] = (W[:, 4:7] - W[:, 1:4]) p_i + bias
// (the folded weights are made once at weight load, engine.hip::up_lse_uv).  Up to round 3 the layer's output - 16 rows per
// point, 2.56 MB per cloud at each of the two levels - was written once and re-read by the attentive pooling and by lfa.mlp2 in
// every pass: ~90 MB of the 432 MB a registration moved through HBM.  Now the consumers rebuild a row from dist (4 bytes) and
// two gathered per-point rows that live in L2 (att_pool.hip, pw_stream.hip loader S_UV): two instructions per channel.
// What remains of the layer is this kernel: one pass over the rows that
//   * writes U | V per point and dist per row,
//   * accumulates the GroupNorm statistics of enc_raw (the same expression, the same bits the consumers form) and commits them
//     in the order-independent form of device_utils.h (gn_block_commit).
// Four lanes per row, a quarter of the channels each (KH = 32) or one lane per row (KH = 8); U[j] is evaluated from the neighbour's coordinates by the
// very function that fills the table, so table and on-the-fly values agree bit for bit.
#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

// p -> folded weights . p in a fixed order (every user of U / V goes through here)
__device__ __forceinline__ float dot3(float wx, float wy, float wz, float x, float y, float z) {
  return fmaf(wz, z, fmaf(wy, y, __fmul_rn(wx, x)));
}

// LPR lanes share a row, KH / LPR channels each: 4 at KH = 32; ONE at KH = 8 (round 4: with four lanes per row every lane repeated
// the row's distance and coordinate loads for two channels' worth of work - 120 vector instructions per row against 66)
template <int KH, int LPR>
__global__ __launch_bounds__(256) void lse_uv_stats_kernel(const LseUvArgs p) {
  constexpr int KQ = KH / LPR;             // channels per lane
  constexpr int PP = 16 / LPR;             // points per workgroup step (256 threads = 256 / LPR rows)
  static_assert(KQ <= 16, "a point's table rows are written by the lanes of its first KQ neighbour slots");
  __shared__ float s_red[4 * KH * 2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cloud = blockIdx.y;
  const int q = tid % LPR, r = tid / LPR, pl = r >> 4, k = r & 15;
  // this lane's folded weights {a, ux, uy, uz, vx, vy, vz, b} of its KQ channels: registers for the whole kernel
  float wq[KQ][8];
#pragma unroll
  for (int c = 0; c < KQ; ++c) {
    const float4 lo = *reinterpret_cast<const float4*>(p.w8 + (q * KQ + c) * 8), hi = *reinterpret_cast<const float4*>(p.w8 + (q * KQ + c) * 8 + 4);
    wq[c][0] = lo.x; wq[c][1] = lo.y; wq[c][2] = lo.z; wq[c][3] = lo.w; wq[c][4] = hi.x; wq[c][5] = hi.y; wq[c][6] = hi.z; wq[c][7] = hi.w;
  }
  const float* X = p.xyz + cloud * p.xyz_cs;
  const int32_t* NB = p.neigh + cloud * p.neigh_cs;
  float* UV = p.uv + cloud * p.uv_cs;
  float* D = p.dist + cloud * p.dist_cs;
  float s1[KQ], s2[KQ];
  const int ntile = (p.n + PP - 1) / PP;
  // VIRTUAL workgroups (p.vgrid per cloud, a function of n alone: ~4 steps each) fix which rows are summed together; the physical
  // workgroup plays the virtual ones blockIdx.x, blockIdx.x + gridDim.x, ... (chip-filling launches: few physical workgroups, weights
  // loaded once; one or two clouds: one physical workgroup per virtual one, four dependent gathers deep instead of sixteen)
  for (int vb = blockIdx.x; vb < p.vgrid; vb += gridDim.x) {
#pragma unroll
  for (int c = 0; c < KQ; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  // the next step's index is fetched while the current one is computed (index -> coordinates is a dependent pair of loads)
  int tile = vb;
  int jn = tile < ntile ? NB[(uint32_t)min(tile * PP + pl, p.n - 1) * 16u + (uint32_t)k] : 0;
  for (; tile < ntile; tile += p.vgrid) {
    const int i = tile * PP + pl;
    const bool ok = i < p.n;
    const int ic = ok ? i : p.n - 1;
    const int j = jn;
    if (tile + p.vgrid < ntile) jn = NB[(uint32_t)min((tile + p.vgrid) * PP + pl, p.n - 1) * 16u + (uint32_t)k];
    const float ix = X[(uint32_t)ic * 3u], iy = X[(uint32_t)ic * 3u + 1], iz = X[(uint32_t)ic * 3u + 2];
    const float jx = X[(uint32_t)j * 3u], jy = X[(uint32_t)j * 3u + 1], jz = X[(uint32_t)j * 3u + 2];
    const float dx = __fsub_rn(jx, ix), dy = __fsub_rn(jy, iy), dz = __fsub_rn(jz, iz);
    const float dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    if (ok && q == 0) D[(uint32_t)i * 16u + (uint32_t)k] = dist;
#pragma unroll
    for (int c = 0; c < KQ; ++c) {
      const float u = dot3(wq[c][1], wq[c][2], wq[c][3], jx, jy, jz);
      const float v = __fadd_rn(dot3(wq[c][4], wq[c][5], wq[c][6], ix, iy, iz), wq[c][7]);
      const float e = __fadd_rn(fmaf(wq[c][0], dist, u), v);
      if (ok) { s1[c] += e; s2[c] = fmaf(e, e, s2[c]); }
      // the point's own table rows: neighbour slot k = c of chunk lane q writes channel q KQ + c
      if (ok && c == k) {
        UV[(uint32_t)i * (2u * KH) + q * KQ + c] = dot3(wq[c][1], wq[c][2], wq[c][3], ix, iy, iz);
        UV[(uint32_t)i * (2u * KH) + KH + q * KQ + c] = v;
      }
    }
  }
  // per-channel sums over the workgroup's rows: a fixed butterfly over the lanes of a wave that share a chunk, the four waves in order
#pragma unroll
  for (int c = 0; c < KQ; ++c) {
    float a = s1[c], b = s2[c];
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane < LPR) { s_red[(w * KH + q * KQ + c) * 2] = a; s_red[(w * KH + q * KQ + c) * 2 + 1] = b; }
  }
  __syncthreads();
  if (tid < 2 * KH) {
    const float v = (s_red[tid] + s_red[2 * KH + tid]) + (s_red[4 * KH + tid] + s_red[6 * KH + tid]);
    s_red[tid] = v;
  }
  __syncthreads();
  gn_block_commit(s_red, 0, KH, KH / p.groups, p.stats_out + (int64_t)cloud * p.groups * kGnWords);
  if (vb + (int)gridDim.x < p.vgrid) __syncthreads();      // s_red is free again before the next virtual workgroup's sums land in it
  }  // vb
}

}  // namespace

// VIRTUAL workgroups per cloud: a function of n alone (a cloud's summation order - hence its bits - is the same alone or in a batch);
// each commits once: the contributions a statistic of the cloud receives
int lse_uv_gn_contributions(int n, int KH) {
  const int pp = KH == 8 ? 16 : 4;         // points per workgroup step
  const int ntile = (n + pp - 1) / pp;
  const int v = (ntile + 3) / 4;           // ~4 steps each (the unit of the statistics' fp32 sums)
  return v < 1 ? 1 : v;
}

bool launch_lse_uv_stats(const LseUvArgs& a, hipStream_t st) {
  if (a.n <= 0 || a.clouds <= 0) return true;
  if (!a.xyz || !a.neigh || !a.w8 || !a.uv || !a.dist || !a.stats_out || a.groups < 1 || (a.KH % a.groups) != 0) return false;
  if ((int64_t)a.n * 16 * 4 >= ((int64_t)1 << 32) || (int64_t)a.n * 2 * a.KH * 4 >= ((int64_t)1 << 32)) return false;
  const int pp = a.KH == 8 ? 16 : 4;       // points per workgroup step
  const int ntile = (a.n + pp - 1) / pp;
  LseUvArgs b = a;
  b.vgrid = lse_uv_gn_contributions(a.n, a.KH);
  if (b.vgrid > kGnMaxContrib) return false;      // dsir_create bounds max_points so that this cannot happen (kernels.h)
  // physical workgroups: ~16 steps each once that fills the chip, else about one residency round, never more than the virtual grid
  const int natural = (ntile + 15) / 16;
  int64_t want = (int64_t)natural * a.clouds >= 512 ? natural : (512 + a.clouds - 1) / a.clouds;
  if (want < 1) want = 1;
  const int blocks = (int)(want < b.vgrid ? want : b.vgrid);
  const dim3 grid(blocks, a.clouds);
  switch (a.KH) {
    case 8: hipLaunchKernelGGL((lse_uv_stats_kernel<8, 1>), grid, dim3(256), 0, st, b); return true;
    case 32: hipLaunchKernelGGL((lse_uv_stats_kernel<32, 4>), grid, dim3(256), 0, st, b); return true;
    default: return false;
  }
}

}  // namespace dsir
