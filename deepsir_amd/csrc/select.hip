// Key-point selection and the small row utilities of the `feat` / `label` pipelines:
//   * top-num_sub points by saliency score          (reference network/model.py:682-697 feat_score: torch.topk +
//                                                     gather_neighbour_V3 of xyz / feat / label)
//   * F.normalize(x, p=2, dim=channels)             (network/model.py:650-651)
// torch.topk leaves the order of equal scores unspecified; the rule owned here (and restated in the oracle):
// descending score, equal scores (+0 == -0) in ascending point index.  The sort itself is a library call
// (hipCUB segmented radix sort, stable), not the hot path.
#include <hipcub/hipcub.hpp>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
inline int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

// ascending key order == descending score order
__global__ void topk_key_kernel(const float* __restrict__ score, int64_t total, int n, uint32_t* __restrict__ key,
                                uint32_t* __restrict__ val) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t b = __float_as_uint(score[i] + 0.0f);            // -0 -> +0
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);           // monotone float -> uint
    key[i] = ~b;
    val[i] = (uint32_t)(i % n);
  }
}

__global__ void seg_offsets_kernel(int* seg, int clouds, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= clouds; i += gridDim.x * blockDim.x) seg[i] = i * n;
}

__global__ void topk_take_kernel(const uint32_t* __restrict__ sorted_val, const float* __restrict__ score, int clouds, int n,
                                 int k, int32_t* __restrict__ idx_out, float* __restrict__ score_out) {
  const int64_t total = (int64_t)clouds * k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cloud = (int)(i / k), j = (int)(i % k);
    const uint32_t src = sorted_val[(int64_t)cloud * n + j];
    idx_out[i] = (int32_t)src;
    if (score_out) score_out[i] = score[(int64_t)cloud * n + src];
  }
}

// out[cloud][j][0..C) = in[cloud][idx[cloud][j]][0..C)   (idx == nullptr: identity)
__global__ void gather_rows_kernel(const float* __restrict__ in, int64_t in_cs, int ld, const int32_t* __restrict__ idx, int C,
                                   int m, int clouds, float* __restrict__ out) {
  const int64_t total = (int64_t)clouds * m * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % C);
    const int64_t r = i / C;
    const int cloud = (int)(r / m), j = (int)(r % m);
    const int src = idx ? idx[(int64_t)cloud * m + j] : j;
    out[i] = in[cloud * in_cs + (int64_t)src * ld + ch];
  }
}

__global__ void gather_i32_kernel(const int32_t* __restrict__ in, int64_t in_cs, const int32_t* __restrict__ idx, int m,
                                  int clouds, int32_t* __restrict__ out) {
  const int64_t total = (int64_t)clouds * m;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cloud = (int)(i / m), j = (int)(i % m);
    out[i] = in[cloud * in_cs + (idx ? idx[i] : j)];
  }
}

// rows of 64 floats: 16 lanes x float4 per row; y = x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void l2norm64_kernel(const float* __restrict__ x, int64_t rows, float* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < rows; r += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + r * 64 + 4 * sub);
    float s = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    const float d = fmaxf(sqrtf(s), 1e-12f);
    *reinterpret_cast<float4*>(y + r * 64 + 4 * sub) = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
  }
}

}  // namespace

size_t topk_scratch_bytes(int clouds, int n) {
  const int64_t total = (int64_t)clouds * n;
  size_t sort_tmp = 0;
  hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sort_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                              (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)total, clouds,
                                              (const int*)nullptr, (const int*)nullptr);
  return 4 * align256((size_t)total * 4) + align256((size_t)(clouds + 1) * 4) + align256(sort_tmp);
}

int launch_topk(const float* score, int clouds, int n, int k, int32_t* idx_out, float* score_out, void* scratch, hipStream_t st) {
  const int64_t total = (int64_t)clouds * n;
  if (total <= 0 || total > 0x7fffffffll || k < 1 || k > n) return 1;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
  uint32_t* k0 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* k1 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* v0 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* v1 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  int* seg = reinterpret_cast<int*>(take((size_t)(clouds + 1) * 4));
  size_t sort_tmp = 0;
  hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sort_tmp, k0, k1, v0, v1, (int)total, clouds, seg, seg + 1);
  hipLaunchKernelGGL(seg_offsets_kernel, dim3(1), dim3(256), 0, st, seg, clouds, n);
  hipLaunchKernelGGL(topk_key_kernel, dim3(grid_for(total)), dim3(256), 0, st, score, total, n, k0, v0);
  if (hipcub::DeviceSegmentedRadixSort::SortPairs(p, sort_tmp, k0, k1, v0, v1, (int)total, clouds, seg, seg + 1, 0, 32, st) != hipSuccess)
    return 2;
  hipLaunchKernelGGL(topk_take_kernel, dim3(grid_for((int64_t)clouds * k)), dim3(256), 0, st, v1, score, clouds, n, k, idx_out,
                     score_out);
  return 0;
}

namespace {
// key = destination row of source e (cloud * n + clamped index), value = e
__global__ __launch_bounds__(256) void plan_key_kernel(const int32_t* __restrict__ idx, int64_t total, int m, int n, uint32_t* __restrict__ key,
                                                       uint32_t* __restrict__ val) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int i = idx[e];
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    key[e] = (uint32_t)((e / m) * n + i);
    val[e] = (uint32_t)e;
  }
}
// offsets[d] = first position of the sorted keys that is >= d, d in [0, dests]
__global__ __launch_bounds__(256) void plan_offsets_kernel(const uint32_t* __restrict__ key, int64_t total, int64_t dests, int32_t* __restrict__ offsets) {
  for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d <= dests; d += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = total;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)key[mid] < d) lo = mid + 1; else hi = mid;
    }
    offsets[d] = (int32_t)lo;
  }
}
}  // namespace

// The inverse of a gather index (train_ops.hip: the backward of gather_neighbour / nearest_interpolation / random_sample is a sum
// over the SOURCES of every destination row): order = the sources grouped by destination, ascending source inside a group (one
// stable radix sort of (destination, source) pairs - a library sort, not the hot path), offsets = the groups' bounds.  With it the
// backward operators sum in a fixed order instead of racing float atomics.
size_t scatter_plan_scratch_bytes(int64_t total) {
  size_t sort_tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (int)total);
  return 3 * align256((size_t)total * 4) + align256(sort_tmp);
}
int launch_scatter_plan(const int32_t* idx, int m, int clouds, int n, int32_t* order, int32_t* offsets, void* scratch, hipStream_t st) {
  const int64_t total = (int64_t)clouds * m, dests = (int64_t)clouds * n;
  if (total <= 0 || total > 0x7fffffffll || dests > 0x7fffffffll) return 1;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
  uint32_t* k0 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* k1 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* v0 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  size_t sort_tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, k0, k1, v0, reinterpret_cast<uint32_t*>(order), (int)total);
  int bits = 1;
  while (((int64_t)1 << bits) < dests) ++bits;
  hipLaunchKernelGGL(plan_key_kernel, dim3(grid_for(total)), dim3(256), 0, st, idx, total, m, n, k0, v0);
  if (hipcub::DeviceRadixSort::SortPairs(p, sort_tmp, k0, k1, v0, reinterpret_cast<uint32_t*>(order), (int)total, 0, bits, st) != hipSuccess) return 2;
  hipLaunchKernelGGL(plan_offsets_kernel, dim3(grid_for(dests + 1)), dim3(256), 0, st, k1, total, dests, offsets);
  return 0;
}

void launch_gather_rows(const float* in, int64_t in_cloud_stride, int ld, const int32_t* idx, int C, int m, int clouds, float* out,
                        hipStream_t st) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)clouds * m * C)), dim3(256), 0, st, in, in_cloud_stride, ld, idx, C,
                     m, clouds, out);
}

void launch_gather_i32(const int32_t* in, int64_t in_cloud_stride, const int32_t* idx, int m, int clouds, int32_t* out,
                       hipStream_t st) {
  hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for((int64_t)clouds * m)), dim3(256), 0, st, in, in_cloud_stride, idx, m, clouds,
                     out);
}

void launch_l2norm64(const float* x, int64_t rows, float* y, hipStream_t st) {
  hipLaunchKernelGGL(l2norm64_kernel, dim3(grid_for(rows * 16)), dim3(256), 0, st, x, rows, y);
}

}  // namespace dsir
