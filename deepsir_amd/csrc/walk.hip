// The deep pyramid levels of RandLA.forward (reference network/RandLANet.py:215-230, :339-359, :374-408) as ONE persistent launch.
//
// Why: with one pair in flight (the reference's own evaluation mode, test.py:56) a registration is a chain of ~300 dependent
// launches, and a dependent kernel node of a replayed graph costs 4.5 - 5 us before it computes anything (the shortest kernels of
// the chain - a 4-byte-per-thread copy - take that long; profiles/r05_b1_timeline.txt).  Levels 2 / 3, mlp_mid and the first two
// decoder blocks are 19 of a pass's 40 launches and their tiles are small (78 - 312 points per cloud): a launch each buys nothing.
//
// What: the walker executes the same tile bodies as PHASES of one launch (kernels.h, WalkProgram).  A phase = a set of independent
// workgroup tiles per cloud (what one of the former launches covered); phase p needs all tiles of phase `dep` (p - 1) published.
//   * a cloud has `wpc` workgroups; the workgroups of a cloud take tiles from the phase's queue with one returning atomic each,
//     so no workgroup ever waits for work that has not been picked up by a RUNNING workgroup: correctness does not depend on how
//     many workgroups are resident or in which order they are dispatched (a tile's predecessors are complete or in execution);
//   * hand-off between workgroups, the documented agent-scope form (MI355X_MICROARCH.md, "Valid forms"): producer - plain stores,
//     every wave s_waitcnt vmcnt(0), workgroup barrier, lane 0: release fence, s_waitcnt, relaxed add to the phase's `done`
//     counter; consumer - lane 0 polls `done` (relaxed sc1 loads), acquire fence, s_waitcnt, workgroup barrier, then plain loads.
//     Measured 1.0 us (same XCD) - 1.4 us (across XCDs) per hand-off of a 16 KB tile, zero stale words in 65 M
//     (tools/ubench/xcd_sync.hip); GroupNorm statistics travel in memory-side atomics and are read with sc1 loads;
//   * every poll is bounded: a workgroup that waited ~seconds raises the cloud's error word and leaves (nothing can hang the box).
// Placement (speed only): the workgroups of a cloud are those with equal blockIdx % 8, i.e. one XCD under the observed round-robin
// dispatch - the cheaper hand-off, and the cloud's activations stay in one L2.
//
// What it measured (MI355X, one 5000-point pair replayed from its graph; tools/walk_trace.py, profiles/r05_walk_trace.txt): the
// registration's launches fall from 305 to 191, a hand-off between phases costs 1.0 - 2.0 us and a publish 0.5 - 1.0 us as predicted -
// and the registration takes 3.46 ms instead of 3.09 ms.  The premise was wrong: the 4.5 - 5 us "floor" of a dependent kernel node
// is not dispatch cost a persistent kernel avoids (that part is ~1.5 us, MI355X_MICROARCH.md "boundary"), it is the latency chain of
// the kernel's own first loads, which a phase pays just the same; a small GEMM tile takes 8 - 9 us inside the walker and 8 - 9 us as a
// launch.  With ~2 us of hand-off per phase against ~1.5 us per launch boundary, and the attentive-pooling phases cut into fewer,
// longer tiles, 20 phases cost 240 - 300 us against 178 us for the 19 launches.  The walker therefore stays OFF by default
// (dsir_enable_walk): what shortens a single pair's registration is FEWER dependent steps and shorter chains inside them, not fewer
// launches.  It is kept, tested bit for bit, as the vehicle for that next step - phases that fuse what are dependent launches today.
//
// Bits: a phase's tiles are computed by the same code on the same operands, each output element by the same chain of operations,
// the statistics meet in exact atomics (device_utils.h) - the walker's results equal the separate launches' bit for bit
// (tests/test_gpu_walk.py), so it may be switched per launch size without changing a pair's result.
#define DSIR_GN_STATS_COHERENT 1      // device_utils.h, gn_stat_get: statistics produced inside this launch are read with sc1 loads
#include "kernels.h"
#include "device_utils.h"
#include "pw_tile_body.h"
#include "att_pool_body.h"
#include "misc_body.h"

namespace dsir {

namespace {

constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t kWalkSmem =
    cmax(cmax(tile::pw_tile_small_smem_bytes<2, EPI_GN, true>(), tile::pw_tile_smem_bytes<2, EPI_ATT2, true>()),
         cmax(attp::att_full_smem_bytes<64>(), miscb::gmc_smem_bytes()));

// bounded wait for *p >= want (lane 0 of the workgroup); false: gave up
__device__ __forceinline__ bool walk_wait(const unsigned* p, unsigned want) {
  for (int spin = 0; spin < (1 << 24); ++spin) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

__global__ __launch_bounds__(256) void walk_kernel(const WalkProgram* __restrict__ prog) {
  __shared__ __attribute__((aligned(16))) char smem[kWalkSmem];
  __shared__ int s_job;
  const int tid = threadIdx.x;
  const int clouds = prog->clouds, wpc = prog->wpc, nph = prog->nphases;
  // cluster q = the workgroups with equal blockIdx % 8 (one XCD under round-robin dispatch), wpc of them per cloud; flags bit 0: the
  // workgroups of a cloud are blockIdx % clouds - every CU of the chip for a single cloud
  const int slot = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int cloud = (prog->flags & 1) ? (int)(blockIdx.x % (unsigned)clouds) : slot + 8 * (k / wpc);
  if (cloud >= clouds) return;
  unsigned* ctr = prog->ctr + (size_t)cloud * kWalkCtrWords;
  if (!(prog->flags & 2)) {
    // the program was written by a host copy: pull its lines towards this CU once, instead of one cold miss per field and phase
    const int words = (int)((offsetof(WalkProgram, job) + (size_t)nph * sizeof(WalkJob)) / 16);
    const uint4* pw = reinterpret_cast<const uint4*>(prog);
    unsigned acc = 0;
    for (int i = tid; i < words; i += 256) { const uint4 v = pw[i]; acc |= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u && nph < 0) s_job = 0;      // never true: keeps the loads
  }
  for (int ph = 0; ph < nph; ++ph) {
    const WalkJob& J = prog->job[ph];
    const int njobs = J.gx * J.gy;
    bool ready = J.dep < 0;
    for (;;) {
      __syncthreads();                                   // the previous tile's LDS (and s_job) is free
      if (tid == 0) s_job = (int)atomicAdd(&ctr[2 * ph], 1u);
      __syncthreads();
      const int j = s_job;
      if (j >= njobs) break;
      unsigned long long* tr = (prog->trace && cloud == 0 && tid == 0) ? prog->trace + 4 * ph : nullptr;
      if (tr) atomicMin(&tr[0], wall_clock64());
      if (!ready) {
        // every tile of the phase this one reads from must be published; they have all been picked up by running workgroups
        if (tid == 0) {
          const WalkJob& D = prog->job[J.dep];
          if (!walk_wait(&ctr[2 * J.dep + 1], (unsigned)(D.gx * D.gy))) atomicExch(&ctr[2 * kWalkMaxPhases], 1u);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        ready = true;
      }
      if (tr) atomicMin(&tr[1], wall_clock64());
      const int bx = j % J.gx, by = j / J.gx;
      // The tile's arguments are COPIED out of the program before the body runs: a by-value copy is what a stand-alone kernel has in its
      // kernel arguments - every field one scalar load at the top, loop invariants in registers.  Read through the program pointer the
      // fields would be reloaded after every barrier and store of the body (they could alias), a load latency per K chunk.
      switch (J.kind) {
        case WK_TILE_SMALL: {
          const GemmArgs a = J.gemm;
          if (J.v1 == EPI_GN) {
            if (J.v0 == 2) tile::pw_tile_small_body<2, EPI_GN, true>(a, bx, by, cloud, smem);
            else tile::pw_tile_small_body<1, EPI_GN, true>(a, bx, by, cloud, smem);
          } else {
            if (J.v0 == 2) tile::pw_tile_small_body<2, EPI_LINEAR, true>(a, bx, by, cloud, smem);
            else tile::pw_tile_small_body<1, EPI_LINEAR, true>(a, bx, by, cloud, smem);
          }
          break;
        }
        case WK_TILE_ATT2: {
          const GemmArgs a = J.gemm;
          if (J.v0 == 2) {
            if (J.v1 == 1) tile::pw_tile_body<2, EPI_ATT2, 1, true>(a, bx, by, cloud, smem);
            else if (J.v1 == 2) tile::pw_tile_body<2, EPI_ATT2, 2, true>(a, bx, by, cloud, smem);
            else tile::pw_tile_body<2, EPI_ATT2, 0, true>(a, bx, by, cloud, smem);
          } else {
            if (J.v1 == 1) tile::pw_tile_body<1, EPI_ATT2, 1, true>(a, bx, by, cloud, smem);
            else if (J.v1 == 2) tile::pw_tile_body<1, EPI_ATT2, 2, true>(a, bx, by, cloud, smem);
            else tile::pw_tile_body<1, EPI_ATT2, 0, true>(a, bx, by, cloud, smem);
          }
          break;
        }
        case WK_ATT_FULL64: {
          const AttPool16Args a = J.att;
          attp::att_full_body<64, false>(a, bx, by, cloud, smem);
          break;
        }
        default: {
          const GmcArgs g = J.gmc;
          miscb::gather_max_combine_body(g.a, g.ga, g.b, g.gb, g.rows_in, g.idx, g.idx_cs, g.C, g.rows_out, g.out, g.bpc, bx, cloud, smem);
        }
      }
      // publish the tile: every wave's stores (and statistics atomics) have left, then ONE release and one counter add per workgroup
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        if (tr) atomicMax(&tr[2], wall_clock64());
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&ctr[2 * ph + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tr) atomicMax(&tr[3], wall_clock64());
      }
    }
  }
}

}  // namespace

void launch_walk(const WalkProgram& host, const WalkProgram* prog, hipStream_t st) {
  if (host.nphases <= 0 || host.clouds <= 0) return;
  const int wpc = host.wpc < 1 ? 1 : host.wpc;
  const int groups = (host.clouds + 7) / 8;             // clouds per XCD slot
  const unsigned grid = (host.flags & 1) ? (unsigned)(wpc * host.clouds) : (unsigned)(8 * wpc * groups);
  hipLaunchKernelGGL(walk_kernel, dim3(grid), dim3(256), 0, st, prog);
}

}  // namespace dsir
