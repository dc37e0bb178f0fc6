// Issue cost (cycles per wave64 instruction per SIMD, 4 waves per SIMD) of the VALU instructions the screening's ranking uses.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float v[8], z1[8], z2[8];
  int k1[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; z1[i] = -1e30f; z2[i] = -1e30f; k1[i] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        if (MODE == 1) z2[i] = __builtin_amdgcn_fmed3f(z1[i], z2[i], v[i]), v[i] = z2[i] + 1.f;          // med3 + add
        if (MODE == 2) z1[i] = fmaxf(z1[i], v[i]), v[i] = z1[i] + 1.f;                                   // max + add
        if (MODE == 3) k1[i] = v[i] > z1[i] ? it + u : k1[i], v[i] = v[i] + 1.f;                         // cmp + cndmask + add
        if (MODE == 4) v[i] = v[i] + 1.f;                                                               // add
        if (MODE == 5) {                                                                                 // the whole ranking step
          const float z = fmaf(v[i], 1.0001f, 0.25f);
          z2[i] = __builtin_amdgcn_fmed3f(z1[i], z2[i], z);
          k1[i] = z > z1[i] ? it + u : k1[i];
          z1[i] = fmaxf(z1[i], z);
          v[i] = z;
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + z1[i] + z2[i] + k1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* d = nullptr;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;
  const double steps = (double)iters * 64 * 4;   // element steps per SIMD (4 waves)
  const float t[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
  const char* n[6] = {"fma", "med3+add", "max+add", "cmp+cndmask+add", "add", "ranking step (fma, med3, cmp, cndmask, max)"};
  for (int i = 0; i < 6; ++i) printf("%-45s %.3f ms  %.2f cycles per element step per SIMD @2.25GHz\n", n[i], t[i], t[i] * 1e-3 * 2.25e9 / steps);
  return 0;
}
