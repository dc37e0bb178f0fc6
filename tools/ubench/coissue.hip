// Does gfx950 overlap independent VALU work with v_mfma_f32_16x16x32_f16 from the same wave / from other waves?
// Three kernels: MFMA only, VALU only, both interleaved (1 MFMA : VPM VALU).  One workgroup per CU, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int VPM>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  float z1 = -1e30f, z2 = -1e30f; int k1 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE != 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
      if (MODE != 0) {
#pragma unroll
        for (int q = 0; q < VPM; ++q) {
          // the ranking's instruction mix: fma, med3, cmp, cndmask, max
          float& x = v[(u * VPM + q) & 7];
          const int sel = (u * VPM + q) % 5;
          if (sel == 0) x = fmaf(x, 1.0001f, 0.5f);
          else if (sel == 1) z2 = __builtin_amdgcn_fmed3f(z1, z2, x);
          else if (sel == 2) k1 = x > z1 ? it : k1;
          else if (sel == 3) z1 = fmaxf(z1, x);
          else x = fmaf(x, 0.9999f, -0.5f);
        }
      }
      if (MODE == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      }
    }
  }
  float s = z1 + z2 + k1;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int VPM>
float run(float* d, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VPM>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VPM>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d = nullptr;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int threads = wps * 256;
    const float m = run<0, 3>(d, threads, iters), v3 = run<1, 3>(d, threads, iters), b3 = run<2, 3>(d, threads, iters);
    const float v2 = run<1, 2>(d, threads, iters), b2 = run<2, 2>(d, threads, iters);
    const float v4 = run<1, 4>(d, threads, iters), b4 = run<2, 4>(d, threads, iters);
    const double mf = (double)iters * 16;   // MFMAs per wave
    printf("waves/SIMD %d: MFMA only %.3f ms (%.1f cyc/MFMA/wave @2.4GHz) | VALU x2 %.3f both %.3f | VALU x3 %.3f both %.3f | VALU x4 %.3f both %.3f\n",
           wps, m, m * 1e-3 * 2.4e9 / mf, v2, b2, v3, b3, v4, b4);
  }
  return 0;
}
