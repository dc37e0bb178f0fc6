// What does a dependent PHASE cost inside one launch, and which hand-off forms are correct?  (round 5: the deep-level walker)
//
// Pairs of workgroups (one producer, one consumer, on different CUs) hand a 16 KB tile back and forth R times; the consumer re-reads
// the SAME addresses every round (L1-warm: the case that shows stale lines) and checks every word.  A pair is either on ONE XCD
// (both workgroups read HW_REG_XCC_ID and meet through a per-XCD arrival counter) or on two different XCDs.
//   mode 0: plain stores, s_waitcnt vmcnt(0), barrier, agent-scope atomic add     | consumer: sc1 poll, acquire fence (buffer_inv sc1), barrier
//   mode 1: as 0 with the release fence (buffer_wbl2 sc1) in front of the add     | as 0                       (the documented cross-XCD form)
//   mode 2: as 0, the flag add at WORKGROUP scope (an L2 atomic)                  | as 0
//   mode 3: as 0                                                                 | NO acquire (expected stale: control)
//   mode 4: sc1 (write-through) stores, vmcnt(0), barrier, agent add              | sc1 poll, barrier, sc1 loads, NO acquire
// Output per (placement, mode): stale words of R x 4096, microseconds per one-way hand-off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned f4u __attribute__((ext_vector_type(4)));
constexpr int N = 4096;          // floats per tile (16 KB), 256 threads x float4 x 4
struct Ctl {
  unsigned arrive[8];            // per-XCD arrival counter (role assignment)
  unsigned first_xcc;            // cross-XCD placement: XCD of the producer
  unsigned flag[8][32];          // [pair][..] producer -> consumer round counter (own 128-byte line)
  unsigned ack[8][32];           // consumer -> producer
  unsigned stale[8];
  unsigned long long cycles[8];
  unsigned pairs_formed;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void signal(unsigned* p, int mode) {
  if (mode == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  if (mode == 2) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wait is bounded (a pair that never formed must not hang the launch): ~0.5 s of polling, then the caller gives up
__device__ __forceinline__ bool wait_for(const unsigned* p, unsigned want) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// same_xcd = 1: workgroups 0 / 1 of every XCD form a pair; = 0: pair p = (first arrival on XCD p, first arrival on XCD (p + 1) % 8)
__global__ __launch_bounds__(256) void pingpong(Ctl* c, float* tiles, int rounds, int mode, int same_xcd, float* sink) {
  const unsigned x = xcc_id();
  __shared__ unsigned s_slot;
  if (threadIdx.x == 0) s_slot = atomicAdd(&c->arrive[x], 1u);
  __syncthreads();
  const unsigned slot = s_slot;
  int pair, role;              // role 0 = producer, 1 = consumer
  if (same_xcd) { if (slot > 1) return; pair = (int)x; role = (int)slot; }
  else { if (slot > 1) return; role = (int)slot; pair = role == 0 ? (int)x : (int)((x + 7) & 7); }   // consumer of pair p sits on XCD p + 1
  float4* t = reinterpret_cast<float4*>(tiles + (size_t)pair * N);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)t, 0, N * 4, 0x00020000);
  unsigned* flag = c->flag[pair];
  unsigned* ack = c->ack[pair];
  unsigned stale = 0;
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    if (role == 0) {
      if (threadIdx.x == 0 && r > 1 && !wait_for(ack, (unsigned)(r - 1))) s_dead = 1;
      __syncthreads();
      if (s_dead) return;
      for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k;
        const float v = (float)(r * 7 + i);
        if (mode == 4) __builtin_amdgcn_raw_buffer_store_b128(f4u{__float_as_uint(v), __float_as_uint(v + 0.25f), __float_as_uint(v + 0.5f), __float_as_uint(v + 0.75f)}, rs, i * 16, 0, 16);
        else t[i] = make_float4(v, v + 0.25f, v + 0.5f, v + 0.75f);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) signal(flag, mode);
    } else {
      if (threadIdx.x == 0) {
        if (!wait_for(flag, (unsigned)r)) s_dead = 1;
        if (mode != 3 && mode != 4) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      }
      __syncthreads();
      if (s_dead) { if (threadIdx.x == 0) atomicAdd(&c->stale[pair], 0x40000000u); return; }
      for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k;
        float4 v;
        if (mode == 4) { const f4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16); v = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])); }
        else v = t[i];
        const float w = (float)(r * 7 + i);
        stale += (v.x != w) + (v.y != w + 0.25f) + (v.z != w + 0.5f) + (v.w != w + 0.75f);
      }
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ack, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (role == 1) {
    atomicAdd(&c->stale[pair], stale);
    if (threadIdx.x == 0) { c->cycles[pair] = t1 - t0; atomicAdd(&c->pairs_formed, 1u); }
  }
  if (sink && stale == 0xffffffffu) sink[0] = 1.f;
}

int main() {
  Ctl* c; float* tiles;
  CHECK(hipMalloc(&c, sizeof(Ctl)));
  CHECK(hipMalloc(&tiles, sizeof(float) * N * 8));
  int khz = 100000;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const int rounds = 2000;
  const char* names[5] = {"plain + vmcnt(0), agent add | poll + acquire", "release fence + agent add   | poll + acquire",
                          "plain + vmcnt(0), L2 add    | poll + acquire", "plain + vmcnt(0), agent add | poll, NO acquire",
                          "sc1 stores + vmcnt(0), add  | poll, sc1 loads"};
  for (int same = 1; same >= 0; --same)
    for (int mode = 0; mode < 5; ++mode) {
      CHECK(hipMemset(c, 0, sizeof(Ctl)));
      CHECK(hipMemset(tiles, 0, sizeof(float) * N * 8));
      hipLaunchKernelGGL(pingpong, dim3(256), dim3(256), 0, 0, c, tiles, rounds, mode, same, (float*)nullptr);
      CHECK(hipDeviceSynchronize());
      Ctl h;
      CHECK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
      unsigned long long stale = 0, cyc = 0;
      for (int p = 0; p < 8; ++p) { stale += h.stale[p]; cyc = h.cycles[p] > cyc ? h.cycles[p] : cyc; }
      printf("%-9s %-50s pairs %u  stale words %llu of %llu  one-way hand-off %.2f us\n", same ? "same-XCD" : "cross-XCD", names[mode], h.pairs_formed,
             stale, (unsigned long long)rounds * N * h.pairs_formed, (double)cyc / khz * 1e3 / rounds / 2.0);
      fflush(stdout);
    }
  hipFree(c); hipFree(tiles);
  return 0;
}
