// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal INPUTS on gfx950?  (round 3: decides whether the low part of an fp16 split
// may be left unscaled).  A[row][k] = a for k == 0 else 0, B[k][col] = b for k == 0 else 0  ->  C[row][col] = a * b.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* ab, float* out, int n) {
  const int lane = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((lane >> 4) == 0) { a[0] = (_Float16)ab[2 * i]; b[0] = (_Float16)ab[2 * i + 1]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[i] = c[0];
  }
}
int main() {
  const float cases[][2] = {{1.f, 1.f}, {5.9604645e-8f, 1.f}, {5.9604645e-8f, 1024.f}, {3.0517578e-5f, 1.f}, {3.0517578e-5f, 3.0517578e-5f},
                            {6.1035156e-5f, 1.f}, {1.f, 5.9604645e-8f}, {2.9802322e-7f, 4.f}};
  const int n = sizeof(cases) / sizeof(cases[0]);
  float *d_ab, *d_out, out[16];
  hipMalloc(&d_ab, sizeof(cases)); hipMalloc(&d_out, sizeof(float) * n);
  hipMemcpy(d_ab, cases, sizeof(cases), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_ab, d_out, n);
  hipMemcpy(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("a = %.9g (fp16 %s) b = %.9g : mfma %.9g  expected %.9g  %s\n", cases[i][0], cases[i][0] < 6.1e-5f ? "subnormal" : "normal", cases[i][1],
           out[i], cases[i][0] * cases[i][1], out[i] == cases[i][0] * cases[i][1] ? "ok" : "FLUSHED / DIFFERENT");
  return 0;
}
