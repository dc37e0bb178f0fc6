// What schedule lets the screening kernel's ranking (VALU) hide behind its MFMAs on gfx950?
// One "step" = what a wave of screen_kernel does per 16 columns with RT = 2 row tiles: 12 v_mfma_f32_16x16x32_f16
// (two accumulator chains of 6) + the ranking of the PREVIOUS step's 8 accumulator elements (per element: v_med3_f32,
// v_max_f32, v_cmp_gt_f32 + v_cndmask_b32; 8 independent (z1, z2, k1) chains, exactly the kernel's data flow).
// Operands stay in registers (no LDS): this isolates the issue behaviour of the two pipes.
//   mode 0  MFMA only            mode 1  VALU only
//   mode 2  clustered: 12 MFMAs, then the 32 VALU (what hipcc emitted for screen_kernel in round 1)
//   mode 3  interleaved: after every MFMA 3 (first 8) or 2 (last 4) ranking instructions, order pinned by sched_barrier
//   mode 4  clustered + staggered roles: waves 4..7 of a 512-thread block rank BEFORE their MFMAs, waves 0..3 after
// each with 1 or 2 waves per SIMD and with / without a workgroup barrier every 4 steps (the kernel has one per 64-column tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

struct Rank {
  float z1[8], z2[8];
  int k1[8];
};

// the four ranking instructions of one element, pinned as written (opaque to the optimiser)
__device__ __forceinline__ void rank_one(Rank& R, int e, float z, int col) {
  asm volatile("v_med3_f32 %1, %0, %1, %3\n\tv_cmp_gt_f32 vcc, %3, %0\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_max_f32 %0, %0, %3"
               : "+v"(R.z1[e]), "+v"(R.z2[e]), "+v"(R.k1[e]) : "v"(z), "v"(col) : "vcc");
}

template <int MODE, int BAR>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  h8 a[6], b[3];
  for (int j = 0; j < 6; ++j)
    for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)(((threadIdx.x * 7 + i * 3 + j) % 61) * 0.01f - 0.3f);
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(((threadIdx.x * 5 + i + j * 11) % 53) * 0.02f - 0.5f);
  Rank R;
  for (int e = 0; e < 8; ++e) { R.z1[e] = -1e30f; R.z2[e] = -1e30f; R.k1[e] = -1; }
  f32x4 zp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bool late = MODE == 4 && (threadIdx.x >> 6) >= (blockDim.x >> 7);   // upper half of the waves
  int col = threadIdx.x & 15;
  f32x4 c0 = f32x4{0.5f, 0.25f, -0.5f, 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 zn[2];
      // operands "change" every step as far as the compiler can tell (in the kernel they come from LDS): nothing hoists
#pragma unroll
      for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(b[j]));
      asm volatile("" : "+v"(c0));
      if (MODE == 4 && late) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rank_one(R, e, zp[e >> 2][e & 3], col);
        FENCE();
      }
      if (MODE != 1) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            zn[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m + 3 * rt) % 6], b[m % 3], m == 0 ? c0 : zn[rt], 0, 0, 0);
            if (MODE == 3) {
              FENCE();
              const int q = 2 * m + rt;                  // 0..11: 3 ranking instructions after the first 8 MFMAs, 2 after the rest
              // element e's three instructions are spread over consecutive slots; done by issuing whole elements:
              // slots 0..7 take one element each (3 instr), the cmp/cndmask pair counts as 2 -> 4 instr per element;
              // to keep it simple: elements 0..7 after MFMAs 0..7 (4 instr each), nothing after 8..11
              if (q < 8) rank_one(R, q, zp[q >> 2][q & 3], col);
              FENCE();
            }
          }
        }
      } else {
        zn[0] = zp[0] + c0; zn[1] = zp[1] - c0;
      }
      if (MODE == 1 || MODE == 2 || (MODE == 4 && !late)) {
        FENCE();
#pragma unroll
        for (int e = 0; e < 8; ++e) rank_one(R, e, zp[e >> 2][e & 3], col);
        FENCE();
      }
      zp[0] = zn[0]; zp[1] = zn[1];
      col += 16;
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int e = 0; e < 8; ++e) s += R.z1[e] + R.z2[e] + R.k1[e];
  s += zp[0][0] + zp[1][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int BAR>
float run(float* d, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, BAR>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, BAR>), dim3(256), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d = nullptr;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;   // x 4 steps
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = wps * 256;
    const double steps = (double)iters * 4;
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.1e9 / steps; };   // cycles per step per wave at ~2.1 GHz (indicative)
    const float m0 = run<0, 0>(d, threads, iters), m1 = run<1, 0>(d, threads, iters), m2 = run<2, 0>(d, threads, iters),
                m3 = run<3, 0>(d, threads, iters), m4 = run<4, 0>(d, threads, iters);
    const float b0 = run<0, 1>(d, threads, iters), b2 = run<2, 1>(d, threads, iters), b3 = run<3, 1>(d, threads, iters),
                b4 = run<4, 1>(d, threads, iters);
    printf("waves/SIMD %d, no barrier : MFMA %.3f ms (%.0f cyc/step)  VALU %.3f (%.0f)  clustered %.3f (%.0f)  interleaved %.3f (%.0f)  staggered %.3f (%.0f)\n",
           wps, m0, cyc(m0), m1, cyc(m1), m2, cyc(m2), m3, cyc(m3), m4, cyc(m4));
    printf("waves/SIMD %d, barrier/4  : MFMA %.3f ms (%.0f cyc/step)               clustered %.3f (%.0f)  interleaved %.3f (%.0f)  staggered %.3f (%.0f)\n",
           wps, b0, cyc(b0), b2, cyc(b2), b3, cyc(b3), b4, cyc(b4));
  }
  return 0;
}
