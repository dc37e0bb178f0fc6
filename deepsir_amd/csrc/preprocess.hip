// Pre-processing in front of the KNN pyramid, on ragged cloud batches:
//   * range / height crop                     (reference dataloader/data_base.py:299-312 process_point_cloud)
//   * voxel-grid down-sample, voxel average   (open3d voxel_down_sample as called at threeDMatch_loader.py:168-175,
//                                              kitti_loader.py:335-338; all channels are averaged, as open3d
//                                              averages points and colours)
//   * resample to exactly k points            (dataloader/transformation.py:72-93 Resampler / FixedResampler)
//
// open3d is not installed and its output ORDER is the iteration order of a std::unordered_map, which nothing pins:
// parity at this boundary is "unpinned" (DESIGN.md).  The rule owned here (and restated in oracle/preprocess.py):
// voxel index = floor((p - (min_bound - voxel/2)) / voxel) in float64 like open3d; voxels come out in ascending
// (ix, iy, iz) order; a voxel's points are summed in float64 in ascending input order; the random resampling draws
// its keys from splitmix64(seed, cloud, index) and breaks ties by index.  Everything is deterministic.
//
// Sorting and scanning are library calls (hipCUB device radix sort / scan): plain library ops, not the hot path.
#include <hipcub/hipcub.hpp>

#include "kernels.h"
#include "device_utils.h"

namespace dsir {

namespace {

constexpr uint64_t kInvalidKey = ~0ull;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ bool crop_ok(const float* p, const float4 crop, bool use_crop) {
  if (!use_crop) return true;
  const float r2 = __fadd_rn(__fadd_rn(__fmul_rn(p[0], p[0]), __fmul_rn(p[1], p[1])), __fmul_rn(p[2], p[2]));
  return r2 <= crop.y * crop.y && r2 > crop.x * crop.x && p[2] >= crop.z && p[2] <= crop.w;
}

// one block per cloud: min bound of the points that survive the crop
__global__ __launch_bounds__(1024) void bounds_kernel(const float* __restrict__ pts, const int64_t* __restrict__ off,
                                                      int stride, float4 crop, int use_crop, float* __restrict__ minb) {
  __shared__ float red[3][16];
  const int cloud = blockIdx.x;
  const int64_t a = off[cloud], b = off[cloud + 1];
  float lo[3] = {INFINITY, INFINITY, INFINITY};
  for (int64_t i = a + threadIdx.x; i < b; i += blockDim.x) {
    const float* p = pts + i * stride;
    if (crop_ok(p, crop, use_crop != 0))
      for (int k = 0; k < 3; ++k) lo[k] = fminf(lo[k], p[k]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    float v = lo[k];
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    if (lane == 0) red[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = red[threadIdx.x][0];
    for (int ww = 1; ww < 16; ++ww) v = fminf(v, red[threadIdx.x][ww]);
    minb[cloud * 3 + threadIdx.x] = v;
  }
}

__global__ void voxel_key_kernel(const float* __restrict__ pts, const int64_t* __restrict__ off, int clouds, int stride,
                                 double voxel, float4 crop, int use_crop, const float* __restrict__ minb,
                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = clouds;                       // cloud of point i: offsets are sorted
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    const float* p = pts + i * stride;
    uint64_t key = kInvalidKey;
    if (crop_ok(p, crop, use_crop != 0)) {
      uint64_t q[3];
      bool ok = true;
      for (int k = 0; k < 3; ++k) {
        const double c = floor(((double)p[k] - ((double)minb[lo * 3 + k] - 0.5 * voxel)) / voxel);
        ok = ok && c >= 0.0 && c < 262144.0;
        q[k] = (uint64_t)c;
      }
      if (ok) key = ((uint64_t)lo << 54) | (q[0] << 36) | (q[1] << 18) | q[2];
    }
    keys[i] = key;
    vals[i] = (uint32_t)(i - off[lo]);
  }
}

__global__ void voxel_head_kernel(const uint64_t* __restrict__ keys, int64_t total, int32_t* __restrict__ head) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (keys[i] != kInvalidKey && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}

// first voxel ordinal and voxel count of every cloud
__global__ void voxel_cloud_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ head,
                                   const int32_t* __restrict__ ordinal, int64_t total, int32_t* __restrict__ first,
                                   int32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    if (head[i]) {
      const int cloud = (int)(keys[i] >> 54);
      atomicMin(&first[cloud], ordinal[i]);
      atomicAdd(&counts[cloud], 1);
    }
}

__global__ void voxel_average_kernel(const float* __restrict__ pts, const int64_t* __restrict__ off, int stride,
                                     const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                     const int32_t* __restrict__ head, const int32_t* __restrict__ ordinal,
                                     const int32_t* __restrict__ first, int64_t total, int cap, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (!head[i]) continue;
    const uint64_t key = keys[i];
    const int cloud = (int)(key >> 54);
    const int slot = ordinal[i] - first[cloud];
    if (slot >= cap) continue;
    double sum[16];
    for (int c = 0; c < stride; ++c) sum[c] = 0.0;
    int cnt = 0;
    for (int64_t j = i; j < total && keys[j] == key; ++j) {      // stable sort => ascending input order
      const float* p = pts + (off[cloud] + vals[j]) * stride;
      for (int c = 0; c < stride; ++c) sum[c] += (double)p[c];
      ++cnt;
    }
    float* o = out + ((int64_t)cloud * cap + slot) * stride;
    for (int c = 0; c < stride; ++c) o[c] = (float)(sum[c] / (double)cnt);
  }
}

// ---- resampling
__global__ void resample_key_kernel(const int32_t* __restrict__ counts, int clouds, int cap, uint64_t seed,
                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t total = (int64_t)clouds * cap;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e / cap), i = (int)(e % cap);
    const int n = min(counts[c], cap);
    // 63-bit random key (top bit clear) for real rows, all-ones for padding: padding sorts last
    keys[e] = i < n ? (splitmix64(seed ^ ((uint64_t)c << 40) ^ (uint64_t)i) >> 1) : kInvalidKey;
    vals[e] = (uint32_t)i;
  }
}

__global__ void resample_gather_kernel(const float* __restrict__ in, const int32_t* __restrict__ counts,
                                       const uint32_t* __restrict__ perm, int clouds, int cap, int stride, int k, int mode,
                                       uint64_t seed, float* __restrict__ out) {
  const int64_t total = (int64_t)clouds * k;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e / k), j = (int)(e % k);
    const int n = min(counts[c], cap);
    float* o = out + e * stride;
    if (n <= 0) { for (int ch = 0; ch < stride; ++ch) o[ch] = 0.f; continue; }
    int srow;
    if (mode == 1) srow = j % n;                                             // FixedResampler: tile / prefix
    else if (j < n) srow = (int)perm[(int64_t)c * cap + j];                  // random order, no repeats
    else srow = (int)(splitmix64(~seed ^ ((uint64_t)c << 40) ^ (uint64_t)j) % (uint64_t)n);   // top-up with replacement
    const float* s = in + ((int64_t)c * cap + srow) * stride;
    for (int ch = 0; ch < stride; ++ch) o[ch] = s[ch];
  }
}

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t voxel_downsample_scratch_bytes(int64_t total, int clouds) {
  size_t sort_tmp = 0, scan_tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)total);
  hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr, (int)total);
  size_t b = 0;
  b += align256((size_t)(clouds + 1) * sizeof(int64_t));     // offsets (device copy)
  b += align256((size_t)clouds * 3 * sizeof(float));         // min bounds
  b += 2 * align256((size_t)total * sizeof(uint64_t));       // keys in / out
  b += 2 * align256((size_t)total * sizeof(uint32_t));       // vals in / out
  b += 2 * align256((size_t)total * sizeof(int32_t));        // head, ordinal
  b += align256((size_t)clouds * sizeof(int32_t));           // first ordinal
  b += align256(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
  return b;
}

int launch_voxel_downsample(const float* pts, const int64_t* offsets_host, int clouds, int stride, float voxel,
                            const float* crop_host, int cap, float* out, int32_t* counts, void* scratch, hipStream_t st) {
  const int64_t total = offsets_host[clouds];
  if (total <= 0 || total > 0x7fffffffll || stride < 3 || stride > 16 || clouds < 1 || clouds > 1023 || !(voxel > 0.f)) return 1;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
  int64_t* off = reinterpret_cast<int64_t*>(take((size_t)(clouds + 1) * sizeof(int64_t)));
  float* minb = reinterpret_cast<float*>(take((size_t)clouds * 3 * sizeof(float)));
  uint64_t* k0 = reinterpret_cast<uint64_t*>(take((size_t)total * sizeof(uint64_t)));
  uint64_t* k1 = reinterpret_cast<uint64_t*>(take((size_t)total * sizeof(uint64_t)));
  uint32_t* v0 = reinterpret_cast<uint32_t*>(take((size_t)total * sizeof(uint32_t)));
  uint32_t* v1 = reinterpret_cast<uint32_t*>(take((size_t)total * sizeof(uint32_t)));
  int32_t* head = reinterpret_cast<int32_t*>(take((size_t)total * sizeof(int32_t)));
  int32_t* ordinal = reinterpret_cast<int32_t*>(take((size_t)total * sizeof(int32_t)));
  int32_t* first = reinterpret_cast<int32_t*>(take((size_t)clouds * sizeof(int32_t)));
  size_t sort_tmp = 0, scan_tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, k0, k1, v0, v1, (int)total);
  hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, head, ordinal, (int)total);
  void* tmp = p;
  size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;

  const bool use_crop = crop_host != nullptr;
  const float4 crop = use_crop ? make_float4(crop_host[0], crop_host[1], crop_host[2], crop_host[3]) : make_float4(0, 0, 0, 0);
  if (hipMemcpyAsync(off, offsets_host, (size_t)(clouds + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess) return 2;
  hipLaunchKernelGGL(bounds_kernel, dim3(clouds), dim3(1024), 0, st, pts, off, stride, crop, use_crop ? 1 : 0, minb);
  hipLaunchKernelGGL(voxel_key_kernel, dim3(grid_for(total)), dim3(256), 0, st, pts, off, clouds, stride, (double)voxel, crop,
                     use_crop ? 1 : 0, minb, k0, v0, total);
  size_t tb = tmp_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, v0, v1, (int)total, 0, 64, st) != hipSuccess) return 3;
  hipLaunchKernelGGL(voxel_head_kernel, dim3(grid_for(total)), dim3(256), 0, st, k1, total, head);
  tb = tmp_bytes;
  if (hipcub::DeviceScan::ExclusiveSum(tmp, tb, head, ordinal, (int)total, st) != hipSuccess) return 4;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, first, clouds, 0x7fffffff);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, counts, clouds, 0);
  hipLaunchKernelGGL(voxel_cloud_kernel, dim3(grid_for(total)), dim3(256), 0, st, k1, head, ordinal, total, first, counts);
  hipLaunchKernelGGL(voxel_average_kernel, dim3(grid_for(total)), dim3(256), 0, st, pts, off, stride, k1, v1, head, ordinal,
                     first, total, cap, out);
  return 0;
}

size_t resample_scratch_bytes(int clouds, int cap) {
  const int64_t total = (int64_t)clouds * cap;
  size_t sort_tmp = 0;
  hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                              (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)total, clouds,
                                              (const int*)nullptr, (const int*)nullptr);
  return 2 * align256((size_t)total * 8) + 2 * align256((size_t)total * 4) + align256((size_t)(clouds + 1) * 4) + align256(sort_tmp);
}

__global__ void seg_offsets_kernel(int* seg, int clouds, int cap) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= clouds; i += gridDim.x * blockDim.x) seg[i] = i * cap;
}

int launch_resample(const float* in, const int32_t* counts, int clouds, int cap, int stride, int k, int mode, uint64_t seed,
                    float* out, void* scratch, hipStream_t st) {
  const int64_t total = (int64_t)clouds * cap;
  if (total <= 0 || total > 0x7fffffffll || k < 1 || stride < 1) return 1;
  char* p = reinterpret_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
  uint64_t* k0 = reinterpret_cast<uint64_t*>(take((size_t)total * 8));
  uint64_t* k1 = reinterpret_cast<uint64_t*>(take((size_t)total * 8));
  uint32_t* v0 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  uint32_t* v1 = reinterpret_cast<uint32_t*>(take((size_t)total * 4));
  int* seg = reinterpret_cast<int*>(take((size_t)(clouds + 1) * 4));
  const uint32_t* perm = v0;
  if (mode == 0) {
    size_t sort_tmp = 0;
    hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, sort_tmp, k0, k1, v0, v1, (int)total, clouds, seg, seg + 1);
    hipLaunchKernelGGL(seg_offsets_kernel, dim3(1), dim3(256), 0, st, seg, clouds, cap);
    hipLaunchKernelGGL(resample_key_kernel, dim3(grid_for(total)), dim3(256), 0, st, counts, clouds, cap, seed, k0, v0);
    if (hipcub::DeviceSegmentedRadixSort::SortPairs(p, sort_tmp, k0, k1, v0, v1, (int)total, clouds, seg, seg + 1, 0, 64, st) != hipSuccess)
      return 2;
    perm = v1;
  }
  hipLaunchKernelGGL(resample_gather_kernel, dim3(grid_for((int64_t)clouds * k)), dim3(256), 0, st, in, counts, perm, clouds, cap,
                     stride, k, mode, seed, out);
  return 0;
}

}  // namespace dsir
