// Device body of the pooling ("random sampling") kernel with the residual combine fused in (misc.hip): shared by its own launch and
// by the deep-level walker (walk.hip) - same instructions on the same operands, same bits.
#pragma once
#include "kernels.h"
#include "device_utils.h"

namespace dsir {
namespace miscb {

__device__ __forceinline__ void gn_scale_shift(const GnRef& g, int cloud, int c, int C, float& scale, float& shift) {
  const int grp = c / (C / g.groups);
  const double* st = g.stats + ((int64_t)cloud * g.groups + grp) * kGnWords;
  const double mean = gn_stat_get(st) * g.inv_count;
  double var = gn_stat_get(st + 2) * g.inv_count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = gn_rstd(var);
  const double sc = (double)g.gamma[c] * rstd;
  scale = (float)sc;
  shift = (float)((double)g.beta[c] - mean * sc);
}

__device__ __forceinline__ float lrelu(float v) { return v < 0.f ? 0.2f * v : v; }

// Same pooling with the residual combine fused in: the block output LeakyReLU(GN(a) + GN(b)) is evaluated on
// the fly for the 16 pooled neighbours and never materialised (levels >= 1, where nothing else reads it).
// body of gather_max_combine_kernel and of a walker job (walk.hip): block bx of the bpc that share a cloud's elements; smem: 8 KB
constexpr size_t gmc_smem_bytes() { return 4 * smem_pad(sizeof(float) * 512); }
__device__ __forceinline__ void gather_max_combine_body(const float* __restrict__ a, const GnRef& ga, const float* __restrict__ b, const GnRef& gb,
                                                        int rows_in, const int32_t* __restrict__ idx, int64_t idx_cs, int C, int rows_out,
                                                        float* __restrict__ out, int bpc, const int bx, const int cloud, char* smem) {
  float* sa = smem_carve<float>(smem, 512);
  float* ha = smem_carve<float>(smem, 512);
  float* sb = smem_carve<float>(smem, 512);
  float* hb = smem_carve<float>(smem, 512);
  const int C4 = C >> 2;
  const int64_t total4 = (int64_t)rows_out * C4;
  const float* pa = a + (int64_t)cloud * rows_in * C;
  const float* pb = b + (int64_t)cloud * rows_in * C;
  float4* o4 = reinterpret_cast<float4*>(out + (int64_t)cloud * rows_out * C);
  // the first element's neighbour list is fetched before the statistics chain that opens the workgroup, every next one during
  // the current element's gathers (index -> row is a dependent pair of loads)
  const int64_t e0 = (int64_t)bx * blockDim.x + threadIdx.x, estep = (int64_t)bpc * blockDim.x;
  int nbn[kKnn];
  {
    const int4* ip = reinterpret_cast<const int4*>(idx + cloud * idx_cs + (int64_t)(int)(min(e0, total4 - 1) / C4) * kKnn);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int4 v = ip[q]; nbn[4 * q] = v.x; nbn[4 * q + 1] = v.y; nbn[4 * q + 2] = v.z; nbn[4 * q + 3] = v.w; }
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    gn_scale_shift(ga, cloud, c, C, sa[c], ha[c]);
    gn_scale_shift(gb, cloud, c, C, sb[c], hb[c]);
  }
  __syncthreads();
  for (int64_t e = e0; e < total4; e += estep) {
    const int c = (int)(e % C4) * 4;
    int nb[kKnn];
#pragma unroll
    for (int k = 0; k < kKnn; ++k) nb[k] = nbn[k];
    if (e + estep < total4) {
      const int4* ip = reinterpret_cast<const int4*>(idx + cloud * idx_cs + (int64_t)(int)((e + estep) / C4) * kKnn);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int4 v = ip[q]; nbn[4 * q] = v.x; nbn[4 * q + 1] = v.y; nbn[4 * q + 2] = v.z; nbn[4 * q + 3] = v.w; }
    }
    const float4 s1 = *reinterpret_cast<const float4*>(&sa[c]), h1 = *reinterpret_cast<const float4*>(&ha[c]);
    const float4 s2 = *reinterpret_cast<const float4*>(&sb[c]), h2 = *reinterpret_cast<const float4*>(&hb[c]);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < kKnn; ++k) {
      const int64_t o = (int64_t)nb[k] * C + c;
      const float4 x = *reinterpret_cast<const float4*>(pa + o), z = *reinterpret_cast<const float4*>(pb + o);
      m.x = fmaxf(m.x, lrelu(fmaf(x.x, s1.x, h1.x) + fmaf(z.x, s2.x, h2.x)));
      m.y = fmaxf(m.y, lrelu(fmaf(x.y, s1.y, h1.y) + fmaf(z.y, s2.y, h2.y)));
      m.z = fmaxf(m.z, lrelu(fmaf(x.z, s1.z, h1.z) + fmaf(z.z, s2.z, h2.z)));
      m.w = fmaxf(m.w, lrelu(fmaf(x.w, s1.w, h1.w) + fmaf(z.w, s2.w, h2.w)));
    }
    o4[e] = m;
  }
}

}  // namespace miscb
}  // namespace dsir
