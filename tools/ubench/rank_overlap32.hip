// Companion of rank_overlap.hip: the same work per wave (a 32-row x 32-column block of the screening: K = 64, three fp16
// products, ranking of the previous block's 16 accumulator elements per lane) issued either as 24 v_mfma_f32_16x16x32_f16
// (4 tiles x 6) or as 12 v_mfma_f32_32x32x16_f16 (one tile x 12).  The wider shape holds the SIMD's issue port for 8 of its
// 32 cycles instead of 8 of 16: does the ranking hide behind it?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void rank_one(float& z1, float& z2, int& k1, float z, int col) {
  asm volatile("v_med3_f32 %1, %0, %1, %3\n\tv_cmp_gt_f32 vcc, %3, %0\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_max_f32 %0, %0, %3"
               : "+v"(z1), "+v"(z2), "+v"(k1) : "v"(z), "v"(col) : "vcc");
}

// MODE 0: 16x16x32, ranking after each MFMA pair;  MODE 1: 32x32x16, ranking spread over the 12 MFMAs;  RANK 0: no ranking
template <int MODE, int RANK>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  h8 a[6], b[6];
  for (int j = 0; j < 6; ++j)
    for (int i = 0; i < 8; ++i) { a[j][i] = (_Float16)(((threadIdx.x * 7 + i * 3 + j) % 61) * 0.01f - 0.3f); b[j][i] = (_Float16)(((threadIdx.x * 5 + i + j * 11) % 53) * 0.02f - 0.5f); }
  float z1[16], z2[16]; int k1[16];
  for (int e = 0; e < 16; ++e) { z1[e] = -1e30f; z2[e] = -1e30f; k1[e] = -1; }
  float zp[16];
  for (int e = 0; e < 16; ++e) zp[e] = 0.f;
  int col = threadIdx.x & 31;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(b[j]));
    if (MODE == 0) {
      f32x4 zn[4];
#pragma unroll
      for (int m = 0; m < 6; ++m) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          zn[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m + t) % 6], b[m], m == 0 ? f32x4{0.5f, 0.25f, -0.5f, 1.f} : zn[t], 0, 0, 0);
          if (RANK && (t & 1)) {
            FENCE();
            const int q = 2 * m + (t >> 1);            // 0..11: 16 elements over the first 8 slots, two per slot
            if (q < 8) { rank_one(z1[2 * q], z2[2 * q], k1[2 * q], zp[2 * q], col); rank_one(z1[2 * q + 1], z2[2 * q + 1], k1[2 * q + 1], zp[2 * q + 1], col); }
            FENCE();
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) zp[4 * t + r] = zn[t][r];
    } else {
      f32x16 zn;
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        f32x16 c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) c0[e] = 0.25f * e;
        zn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 6], b[m % 6], m == 0 ? c0 : zn, 0, 0, 0);
        if (RANK) {
          FENCE();
          if (m < 8) { rank_one(z1[2 * m], z2[2 * m], k1[2 * m], zp[2 * m], col); rank_one(z1[2 * m + 1], z2[2 * m + 1], k1[2 * m + 1], zp[2 * m + 1], col); }
          FENCE();
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) zp[e] = zn[e];
    }
    col += 32;
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += z1[e] + z2[e] + k1[e] + zp[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int RANK>
float run(float* d, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, RANK>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, RANK>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d = nullptr;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = wps * 256;
    printf("waves/SIMD %d: 16x16x32 MFMA only %.3f ms, with ranking %.3f | 32x32x16 MFMA only %.3f ms, with ranking %.3f\n", wps,
           run<0, 0>(d, threads, iters), run<0, 1>(d, threads, iters), run<1, 0>(d, threads, iters), run<1, 1>(d, threads, iters));
  }
  return 0;
}
