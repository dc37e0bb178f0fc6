// Element-wise / gather helpers of the RandLA encoder.
#include "kernels.h"
#include "device_utils.h"
#include "misc_body.h"

namespace dsir {

namespace {

using namespace miscb;

// y = LeakyReLU(GN(a) + GN(b)) ; one thread per 4 channels (16-byte accesses), channels fastest (coalesced).
// C is a multiple of 4 (32..512) and every tensor base is 16-byte aligned (arena allocations).
__global__ __launch_bounds__(256) void residual_combine_kernel(const float* __restrict__ a, GnRef ga,
                                                               const float* __restrict__ b, GnRef gb, int C, int rows,
                                                               float* __restrict__ y) {
  __shared__ float sa[512], ha[512], sb[512], hb[512];
  const int cloud = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t total4 = (int64_t)rows * C4;
  const float4* a4 = reinterpret_cast<const float4*>(a + (int64_t)cloud * rows * C);
  const float4* b4 = reinterpret_cast<const float4*>(b + (int64_t)cloud * rows * C);
  float4* y4 = reinterpret_cast<float4*>(y + (int64_t)cloud * rows * C);
  // the first element's loads are issued before the statistics chain (a dependent load -> fp64 arithmetic -> barrier sequence
  // that opens every workgroup), the next element's during the current one's arithmetic
  const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, estep = (int64_t)gridDim.x * blockDim.x;
  float4 xn = make_float4(0.f, 0.f, 0.f, 0.f), zn = xn;
  if (e0 < total4) { xn = a4[e0]; zn = b4[e0]; }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    gn_scale_shift(ga, cloud, c, C, sa[c], ha[c]);
    gn_scale_shift(gb, cloud, c, C, sb[c], hb[c]);
  }
  __syncthreads();
  for (int64_t e = e0; e < total4; e += estep) {
    const int c = (int)(e % C4) * 4;
    const float4 x = xn, z = zn;
    if (e + estep < total4) { xn = a4[e + estep]; zn = b4[e + estep]; }
    float4 r;
    r.x = lrelu(fmaf(x.x, sa[c], ha[c]) + fmaf(z.x, sb[c], hb[c]));
    r.y = lrelu(fmaf(x.y, sa[c + 1], ha[c + 1]) + fmaf(z.y, sb[c + 1], hb[c + 1]));
    r.z = lrelu(fmaf(x.z, sa[c + 2], ha[c + 2]) + fmaf(z.z, sb[c + 2], hb[c + 2]));
    r.w = lrelu(fmaf(x.w, sa[c + 3], ha[c + 3]) + fmaf(z.w, sb[c + 3], hb[c + 3]));
    y4[e] = r;
  }
}

// out[i][c] = max over the 16 pooled neighbours ("random sampling", RandLANet.py:374-391)
__global__ __launch_bounds__(256) void gather_max_kernel(const float* __restrict__ in, int64_t in_cs,
                                                         const int32_t* __restrict__ idx, int64_t idx_cs, int C,
                                                         int rows_out, float* __restrict__ out, int bpc) {
  // 1-D grid of bpc workgroups per cloud, XCD-aware: a cloud's workgroups share one L2 (every input row is gathered ~4 times)
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  const int cloud = wi / bpc, bx = wi % bpc;
  const int C4 = C >> 2;
  const int64_t total4 = (int64_t)rows_out * C4;
  const float* src = in + cloud * in_cs;
  float4* o4 = reinterpret_cast<float4*>(out + (int64_t)cloud * rows_out * C);
  for (int64_t e = (int64_t)bx * blockDim.x + threadIdx.x; e < total4; e += (int64_t)bpc * blockDim.x) {
    const int i = (int)(e / C4), c = (int)(e % C4) * 4;
    const int32_t* nb = idx + cloud * idx_cs + (int64_t)i * kKnn;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < kKnn; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)nb[k] * C + c);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    o4[e] = m;
  }
}

// Same pooling with the residual combine fused in (body: misc_body.h, shared with the deep-level walker): the block output
// LeakyReLU(GN(a) + GN(b)) is evaluated on the fly for the 16 pooled neighbours and never materialised (levels >= 1).
__global__ __launch_bounds__(256) void gather_max_combine_kernel(const float* __restrict__ a, GnRef ga,
                                                                 const float* __restrict__ b, GnRef gb, int rows_in,
                                                                 const int32_t* __restrict__ idx, int64_t idx_cs, int C,
                                                                 int rows_out, float* __restrict__ out, int bpc) {
  __shared__ __attribute__((aligned(16))) char smem[gmc_smem_bytes()];
  // 1-D grid of bpc workgroups per cloud, XCD-aware: a cloud's workgroups share one L2 (every input row is gathered ~4 times)
  const int wi = xcd_contiguous(blockIdx.x, gridDim.x);
  gather_max_combine_body(a, ga, b, gb, rows_in, idx, idx_cs, C, rows_out, out, bpc, wi % bpc, wi / bpc, smem);
}

__global__ void narrow_i64_kernel(const int64_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = (int32_t)src[e];
}

__global__ void copy_xyz_kernel(const float* __restrict__ pts, int64_t cs, int stride, int n, float* __restrict__ out,
                                int64_t ocs) {
  const int cloud = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * 3; e += gridDim.x * blockDim.x)
    out[cloud * ocs + e] = pts[cloud * cs + (int64_t)(e / 3) * stride + (e % 3)];
}

__global__ void copy_xyz_levels_kernel(const float* __restrict__ pts, int64_t cs, int stride, const PyramidLevels lv,
                                       float* __restrict__ out, int64_t ocs) {
  const int cloud = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < lv.S * 3; e += gridDim.x * blockDim.x) {
    const int p = e / 3;
    int l = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k) l += (k < lv.L && p >= lv.off[k]) ? 1 : 0;
    out[cloud * ocs + e] = pts[cloud * cs + (int64_t)(p - lv.off[l]) * stride + (e % 3)];
  }
}

__global__ void copy_sub_levels_kernel(const int32_t* __restrict__ neigh, int64_t ncs, const PyramidLevels lv,
                                       int32_t* __restrict__ sub, int64_t scs) {
  const int cloud = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < lv.S1 * kKnn; e += gridDim.x * blockDim.x) {
    const int q = e / kKnn;
    int l = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k) l += (k < lv.L && q >= lv.soff[k]) ? 1 : 0;
    sub[cloud * scs + e] = neigh[cloud * ncs + (int64_t)(lv.off[l] + q - lv.soff[l]) * kKnn + (e % kKnn)];
  }
}

__global__ void copy_rows_i32_kernel(const int32_t* __restrict__ src, int64_t scs, int count, int32_t* __restrict__ dst,
                                     int64_t dcs) {
  const int cloud = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x)
    dst[cloud * dcs + e] = src[cloud * scs + e];
}

// Caller-supplied indices (forced correspondences, the reference's pyramids): copied with every entry clamped into
// [0, limit) so that no gather can leave its tensor; an out-of-range entry raises bit 1 of flag[cloud % flag_mod].
__global__ void copy_idx_clamped_kernel(const int32_t* __restrict__ src, int64_t scs, int count, int limit,
                                        int32_t* __restrict__ dst, int64_t dcs, int32_t* __restrict__ flag, int flag_mod) {
  const int cloud = blockIdx.y;
  bool bad = false;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
    const int v = src[cloud * scs + e];
    const bool oob = v < 0 || v >= limit;
    bad |= oob;
    dst[cloud * dcs + e] = oob ? (v < 0 ? 0 : limit - 1) : v;
  }
  if (bad && flag) atomicOr(flag + cloud % flag_mod, 2);
}

// the three index tensors of a KNN pyramid in one launch: level l of neigh / sub holds indices into level l's n_l points,
// level l of interp into the n_{l+1} points of the level below
__global__ void copy_pyramid_idx_kernel(PyramidIdxCopy a) {
  const int cloud = blockIdx.y;
  const int64_t total = (int64_t)a.S * kKnn + (int64_t)a.S1 * kKnn + a.S;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t* src; int32_t* dst; int64_t i; int row; bool is_sub = false, is_interp = false;
    if (e < (int64_t)a.S * kKnn) { i = e; src = a.neigh + cloud * (int64_t)a.S * kKnn; dst = a.neigh_out + cloud * (int64_t)a.S * kKnn; row = (int)(i / kKnn); }
    else if (e < (int64_t)(a.S + a.S1) * kKnn) { i = e - (int64_t)a.S * kKnn; src = a.sub + cloud * (int64_t)a.S1 * kKnn; dst = a.sub_out + cloud * (int64_t)a.S1 * kKnn; row = (int)(i / kKnn); is_sub = true; }
    else { i = e - (int64_t)(a.S + a.S1) * kKnn; src = a.interp + cloud * (int64_t)a.S; dst = a.interp_out + cloud * (int64_t)a.S; row = (int)i; is_interp = true; }
    int lvl = 0;
    if (is_sub) { while (lvl + 1 < a.levels && row >= a.soff[lvl + 1]) ++lvl; }
    else { while (lvl + 1 < a.levels && row >= a.off[lvl + 1]) ++lvl; }
    const int limit = is_interp ? a.nl[lvl + 1] : a.nl[lvl];
    const int v = src[i];
    const bool oob = v < 0 || v >= limit;
    bad |= oob;
    dst[i] = oob ? (v < 0 ? 0 : limit - 1) : v;
  }
  if (bad && a.flag) atomicOr(a.flag + cloud % a.flag_mod, 2);
}

inline int grid_for(int64_t total, int block = 256, int cap = 2048) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

void launch_residual_combine(const float* a, GnRef ga, const float* b, GnRef gb, int C, int rows, int clouds, float* y,
                             hipStream_t st) {
  dim3 grid(grid_for((int64_t)rows * C / 4), clouds);
  hipLaunchKernelGGL(residual_combine_kernel, grid, dim3(256), 0, st, a, ga, b, gb, C, rows, y);
}

void launch_gather_max(const float* in, int64_t in_cs, const int32_t* idx, int64_t idx_cs, int C, int rows_out,
                       int clouds, float* out, hipStream_t st) {
  if (rows_out <= 0) return;
  const int bpc = grid_for((int64_t)rows_out * C / 4);
  hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)((int64_t)bpc * clouds)), dim3(256), 0, st, in, in_cs, idx, idx_cs, C, rows_out, out,
                     bpc);
}

void launch_gather_max_combine(const float* a, GnRef ga, const float* b, GnRef gb, int rows_in, const int32_t* idx,
                               int64_t idx_cs, int C, int rows_out, int clouds, float* out, hipStream_t st) {
  if (rows_out <= 0) return;
  const int bpc = grid_for((int64_t)rows_out * C / 4);
  hipLaunchKernelGGL(gather_max_combine_kernel, dim3((unsigned)((int64_t)bpc * clouds)), dim3(256), 0, st, a, ga, b, gb, rows_in, idx,
                     idx_cs, C, rows_out, out, bpc);
}

// launch_gather_max_combine as a phase of the deep-level walker (walk.hip): at most wpc blocks per cloud (an element's result does not
// depend on the cut)
bool walk_plan_gmc(const GmcArgs& a, int wpc, WalkJob* out) {
  if (a.rows_out <= 0 || a.C > 512 || (a.C % 4) != 0) return false;
  int bpc = grid_for((int64_t)a.rows_out * a.C / 4);
  if (bpc > wpc) bpc = wpc < 1 ? 1 : wpc;
  out->kind = WK_GMC; out->v0 = out->v1 = 0;
  out->gmc = a;
  out->gmc.bpc = bpc;
  out->gx = bpc; out->gy = 1;
  return true;
}

namespace {
// blockIdx.y = operation; a workgroup walks its operation's 16-byte (or, unaligned, 4-byte) words grid-stride
__global__ __launch_bounds__(256) void mem_ops_kernel(const MemOps m) {
  const MemOp o = m.op[blockIdx.y];
  const size_t step = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool wide = ((reinterpret_cast<uintptr_t>(o.dst) | reinterpret_cast<uintptr_t>(o.src) | o.bytes) & 15) == 0;
  if (wide) {
    uint4* d = reinterpret_cast<uint4*>(o.dst);
    const uint4* s = reinterpret_cast<const uint4*>(o.src);
    const uint4 f = make_uint4(o.fill, o.fill, o.fill, o.fill);
    for (size_t i = t0; i < o.bytes / 16; i += step) d[i] = s ? s[i] : f;
  } else {
    uint32_t* d = reinterpret_cast<uint32_t*>(o.dst);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(o.src);
    for (size_t i = t0; i < o.bytes / 4; i += step) d[i] = s ? s[i] : o.fill;
  }
}
}  // namespace

void launch_mem_ops(const MemOps& m, hipStream_t st) {
  if (m.n <= 0) return;
  size_t most = 0;
  for (int i = 0; i < m.n; ++i) most = m.op[i].bytes > most ? m.op[i].bytes : most;
  size_t gx = (most / 16 + 255) / 256;          // one 16-byte word per thread up to 1024 workgroups per operation, grid-stride beyond
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  hipLaunchKernelGGL(mem_ops_kernel, dim3((unsigned)gx, (unsigned)m.n), dim3(256), 0, st, m);
}

void launch_narrow_i64(const int64_t* src, int32_t* dst, int64_t n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(narrow_i64_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, dst, n);
}

void launch_copy_xyz(const float* pts, int64_t cs, int stride, int n, int clouds, float* out, int64_t ocs,
                     hipStream_t st) {
  dim3 grid(grid_for((int64_t)n * 3), clouds);
  hipLaunchKernelGGL(copy_xyz_kernel, grid, dim3(256), 0, st, pts, cs, stride, n, out, ocs);
}

void launch_copy_xyz_levels(const float* pts, int64_t cs, int stride, const PyramidLevels& lv, int clouds, float* xyz, int64_t xyz_cs,
                            hipStream_t st) {
  if (clouds <= 0 || lv.S <= 0) return;
  dim3 grid(grid_for((int64_t)lv.S * 3), clouds);
  hipLaunchKernelGGL(copy_xyz_levels_kernel, grid, dim3(256), 0, st, pts, cs, stride, lv, xyz, xyz_cs);
}

void launch_copy_sub_levels(const int32_t* neigh, int64_t ncs, const PyramidLevels& lv, int clouds, int32_t* sub, int64_t scs,
                            hipStream_t st) {
  if (clouds <= 0 || lv.S1 <= 0) return;
  dim3 grid(grid_for((int64_t)lv.S1 * kKnn), clouds);
  hipLaunchKernelGGL(copy_sub_levels_kernel, grid, dim3(256), 0, st, neigh, ncs, lv, sub, scs);
}

void launch_copy_rows_i32(const int32_t* src, int64_t scs, int rows, int width, int clouds, int32_t* dst, int64_t dcs,
                          hipStream_t st) {
  if (rows <= 0) return;
  dim3 grid(grid_for((int64_t)rows * width), clouds);
  hipLaunchKernelGGL(copy_rows_i32_kernel, grid, dim3(256), 0, st, src, scs, rows * width, dst, dcs);
}

void launch_copy_idx_clamped(const int32_t* src, int64_t scs, int count, int limit, int clouds, int32_t* dst, int64_t dcs,
                             int32_t* flag, int flag_mod, hipStream_t st) {
  if (count <= 0 || clouds <= 0) return;
  dim3 grid(grid_for(count), clouds);
  hipLaunchKernelGGL(copy_idx_clamped_kernel, grid, dim3(256), 0, st, src, scs, count, limit, dst, dcs, flag, flag_mod < 1 ? 1 : flag_mod);
}

void launch_copy_pyramid_idx(const PyramidIdxCopy& a, int clouds, hipStream_t st) {
  if (clouds <= 0) return;
  dim3 grid(grid_for((int64_t)a.S * kKnn + (int64_t)a.S1 * kKnn + a.S), clouds);
  hipLaunchKernelGGL(copy_pyramid_idx_kernel, grid, dim3(256), 0, st, a);
}

}  // namespace dsir
