// Fused nearest-descriptor search: for every src descriptor a_j the index of the
// ref descriptor b_k minimising  ((-2 a_j.b_k) + |a_j|^2) + |b_k|^2  in fp32
// (reference network/matchnet.py:96-113 square_distance_V2 + .min(dim=2)[1] at
// network/model.py:558-569).  The reference materialises the [J,K] matrix in
// 6000-row chunks; here it never leaves the MFMA accumulators.
//
// Block = 4 waves; wave w owns RT row tiles (16 src rows each) whose A fragments
// (64 channels = 16 k-steps) stay in registers for the whole kernel.  Ref
// descriptors stream through LDS in tiles of 64 columns ([col][64+2], the pad
// makes fragment reads conflict free).  Per 16x16 tile: 16 exact-fp32 MFMAs
// (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, channels 0..63 ascending),
// then 4 running (min, argmin) updates per lane.  The ref range is split over
// blockIdx.y; partial results meet in a packed 64-bit atomicMin
// (order-preserving distance bits << 32 | index), so ties go to the lower
// index regardless of arrival order (deterministic).
#include "kernels.h"
#include "device_utils.h"
#include <cstdlib>

namespace dsir {

namespace {

constexpr int BC = 64;        // ref columns per LDS tile
constexpr int LDB = 64 + 2;   // padded row (floats)

__device__ __forceinline__ unsigned int order_bits(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// |x|^2 of every descriptor row: one wave per 4 rows... (16 lanes per row, float4 each)
// `init` (optional): the row's packed (distance, index) result slot, set to all ones here instead of by a separate memset
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, int64_t rows, float* __restrict__ out,
                                                     unsigned long long* __restrict__ init) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) v = *reinterpret_cast<const float4*>(x + row * 64 + l * 4);
  const float s = sqnorm_row16(v);
  if (row < rows && l == 0) {
    out[row] = s;
    if (init) init[row] = ~0ull;
  }
}

template <int RT>
__global__ __launch_bounds__(256) void nn_match_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       const float* __restrict__ sa, const float* __restrict__ sb,
                                                       int J, int K, int cols_per_split, int rb_count, int splits,
                                                       unsigned long long* __restrict__ packed,
                                                       unsigned long long* __restrict__ tstamp,
                                                       const int32_t* __restrict__ gate, int gate_min,
                                                       const int32_t* __restrict__ rowlist) {
  // measurement hook: first-wave start / last-wave end on the device's constant-rate clock (what a kernel trace reports)
  if (tstamp && threadIdx.x == 0) atomicMin(tstamp, (unsigned long long)wall_clock64());
  __shared__ float Bs[2][BC * LDB];   // double-buffered ref tile
  __shared__ float sbs[2][BC];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int fr = lane & 15, fq = lane >> 4;
  // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, each with
  // its own L2.  The bijective remap below hands every XCD a CONTIGUOUS range of work items, ordered
  // (pair, ref split, row block) with the row block fastest, so the workgroups that stream the same
  // ref descriptors share one L2 instead of re-fetching them through eight.
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int wi = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  int rb = wi % rb_count;
  int split = (wi / rb_count) % splits;
  int pair = wi / (rb_count * splits);
  if (rowlist) {
    // row-list use: only the first row blocks of a pair have work.  Row block slowest, in plain dispatch order: the
    // working blocks come first and spread over all CUs (with the row block fastest every rb_count-th workgroup works,
    // and round-robin dispatch piles them onto a few CUs)
    const int ps = (int)(gridDim.x / rb_count);   // pairs * splits
    rb = id / ps;
    pair = (id % ps) / splits;
    split = id % splits;
  }
  // fallback use (nn_screen.hip), block-uniform exits before any barrier.  gate[pair] = src rows of the pair whose
  // screening failed.  Without a row list: all rows of the pairs with gate >= gate_min.  With one: only the listed rows
  // (rowlist[pair][0 .. gate)) of the pairs with gate < gate_min.
  int vJ = J;
  if (gate) {
    const int g = gate[pair];
    if (rowlist) {
      if (g >= gate_min || rb * (64 * RT) >= g) return;
      vJ = g;
    } else if (g < gate_min) {
      return;
    }
  }
  const int32_t* lst = rowlist ? rowlist + (int64_t)pair * J : nullptr;
  auto rowof = [&](int v) { return v < vJ ? (lst ? lst[v] : v) : -1; };
  const float* Ap = A + (int64_t)pair * J * 64;
  const float* Bp = B + (int64_t)pair * K * 64;
  const int row0 = (rb * 4 + w) * (16 * RT);

  // A fragments: lane holds A[row = fr][k = 4 s + fq]
  float af[RT][16];
  float san[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = rowof(row0 + rt * 16 + fr);
#pragma unroll
    for (int s = 0; s < 16; ++s) af[rt][s] = row >= 0 ? Ap[(int64_t)row * 64 + 4 * s + fq] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = rowof(row0 + rt * 16 + 4 * fq + r);
      san[rt][r] = rr >= 0 ? sa[(int64_t)pair * J + rr] : 0.f;
    }
  }
  float best[RT][4];
  int bidx[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { best[rt][r] = INFINITY; bidx[rt][r] = 0x7fffffff; }

  const int c_begin = split * cols_per_split;
  const int c_end = min(K, c_begin + cols_per_split);
  // Pipeline: the next 64-column ref tile is fetched into registers while the MFMAs of the current
  // tile run, then written to the other LDS buffer; one barrier per tile.
  float4 pre[4];
  float pre_sb = 0.f;
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int r = f >> 4, c4 = f & 15;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + r < c_end) pre[i] = *reinterpret_cast<const float4*>(Bp + (int64_t)(c0 + r) * 64 + c4 * 4);
    }
    if (tid < BC) pre_sb = (c0 + tid < c_end) ? sb[(int64_t)pair * K + c0 + tid] : 0.f;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int r = f >> 4, c4 = f & 15;
      float2* dst = reinterpret_cast<float2*>(&Bs[buf][r * LDB + c4 * 4]);
      dst[0] = make_float2(pre[i].x, pre[i].y);
      dst[1] = make_float2(pre[i].z, pre[i].w);
    }
    if (tid < BC) sbs[buf][tid] = pre_sb;
  };
  gload(c_begin);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += BC) {
    const bool has_next = c0 + BC < c_end;
    if (has_next) gload(c0 + BC);
    const float* Bt = Bs[buf];
#pragma unroll
    for (int t = 0; t < BC / 16; ++t) {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float b = Bt[(16 * t + fr) * LDB + 4 * s + fq];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt][s], b, acc[rt], 0, 0, 0);
      }
      const int col = c0 + 16 * t + fr;
      const float sbv = sbs[buf][16 * t + fr];
      if (col < c_end) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // fl(fl(-2*dot + |a|^2) + |b|^2): -2*dot is exact, so the fused form rounds exactly like the reference
            const float d = __fadd_rn(__fmaf_rn(acc[rt][r], -2.f, san[rt][r]), sbv);
            if (d < best[rt][r]) { best[rt][r] = d; bidx[rt][r] = col; }
          }
      }
    }
    if (has_next) lstore(buf ^ 1);   // last readers of that buffer finished before the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  // reduce over the 16 lanes that share a row; ties -> lower column
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float d = best[rt][r];
      int ix = bidx[rt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float d2 = __shfl_xor(d, o);
        const int i2 = __shfl_xor(ix, o);
        if (d2 < d || (d2 == d && i2 < ix)) { d = d2; ix = i2; }
      }
      const int row = rowof(row0 + rt * 16 + 4 * fq + r);
      if (fr == 0 && row >= 0 && ix != 0x7fffffff) {
        const unsigned long long key = ((unsigned long long)order_bits(d) << 32) | (unsigned int)ix;
        atomicMin(packed + (int64_t)pair * J + row, key);
      }
    }
  if (tstamp && threadIdx.x == 0) atomicMax(tstamp + 1, (unsigned long long)wall_clock64());
}

// A row whose distances are all NaN (non-finite descriptors) never updates its slot: index 0, like torch.min over an
// all-NaN row (the first NaN), never -1 - the consumers gather by this index.  The pair ends as identity + invalid
// flag in the Kabsch step (model.py:61-64).
__global__ void unpack_idx_kernel(const unsigned long long* __restrict__ packed, int64_t n, int32_t* __restrict__ idx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    idx[e] = packed_index(packed[e]);
}

}  // namespace

void launch_sqnorm(const float* x, int64_t rows, float* out, hipStream_t st) {
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, x, rows, out,
                     (unsigned long long*)nullptr);
}

// scratch layout (floats): sa[pairs*J] | sb[pairs*K] | packed (u64)[pairs*J]
size_t nn_match_scratch_bytes(int pairs, int J, int K) {
  size_t f = ((size_t)pairs * J + (size_t)pairs * K + 3) & ~(size_t)3;
  return f * 4 + (size_t)pairs * J * 8;
}

// the exhaustive kernel over [pairs] x J x K with norms and the packed result slots supplied (slots preset to all ones)
static void launch_core(const float* a, const float* b, const float* sa, const float* sb, int pairs, int J, int K,
                        unsigned long long* packed, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                        unsigned long long* tstamp, const int32_t* gate, int gate_min, const int32_t* rowlist = nullptr) {
  // geometry: 2 row tiles per wave (128-row blocks) once there is enough work, else 64-row blocks; the ref
  // range is split so that the grid is a whole number of residency rounds (256 CUs x 4 blocks: 34 KB LDS each)
  static const int force_rt = (int)tuning_int("DSIR_MATCH_RT", 0);   // tuning hook
  int rt = (int64_t)pairs * ((J + 127) / 128) >= 256 ? 2 : 1;
  if (force_rt == 1 || force_rt == 2 || force_rt == 4) rt = force_rt;
  if (rowlist) rt = 1;                          // few rows per pair: small row blocks, the ref range split as far as it goes
  const int rows_per_block = 64 * rt;
  const int resident = rt == 4 ? 768 : 1024;   // blocks the chip holds at once (VGPR- resp. LDS-limited)
  // a row list holds fewer than gate_min rows
  const int rb_count = ((rowlist ? (gate_min < J ? gate_min : J) : J) + rows_per_block - 1) / rows_per_block;
  const int64_t base = (int64_t)pairs * rb_count;
  const int tiles = (K + BC - 1) / BC;
  int splits = 1;
  double best_eff = -1.0;
  if (rowlist) { splits = tiles / 8 < 1 ? 1 : (tiles / 8 > 16 ? 16 : tiles / 8); best_eff = 2.0; }
  // A launch that fits the chip in ONE residency round (a single pair: 79 row blocks) is bound by the CU that gets the most
  // workgroups: its matrix pipe serves ceil(blocks / 256) of them, each tiles_per tiles long plus ~2 tiles' worth of
  // prologue / epilogue; one or two workgroups per CU run no faster than three (their load latencies no longer overlap).  (Measured at 1 x 5000 x 5000:
  // 9 splits 3.595 ms per registration, 10 splits - what the balance formula below picks - 3.627, 12: 3.60, 8: 3.613, 4: 3.69.)
  int max_nsp = 1;
  for (int sp = 1; sp <= 16 && sp <= tiles; ++sp) {
    const int tiles_per = (tiles + sp - 1) / sp;
    if (sp > 1 && tiles_per < 8) break;
    max_nsp = (tiles + tiles_per - 1) / tiles_per;
  }
  const bool one_round = !rowlist && base * max_nsp <= resident;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 16 && sp <= tiles && !rowlist; ++sp) {
    const int tiles_per = (tiles + sp - 1) / sp;
    if (sp > 1 && tiles_per < 8) break;                    // keep the A-fragment preload amortised
    const int nsp = (tiles + tiles_per - 1) / tiles_per;
    const int64_t blocks = base * nsp;
    if (one_round) {
      const int64_t per_cu = (blocks + 255) / 256;
      const double cost = (double)(per_cu < 3 ? 3 : per_cu) * (tiles_per + 2);   // < 3 workgroups per CU: no faster than 3
      if (cost < best_cost - 1e-9) { best_cost = cost; splits = sp; }
      continue;
    }
    const int64_t rounds = (blocks + resident - 1) / resident;
    double eff = (double)blocks / (double)(rounds * resident);   // residency balance
    eff *= (double)tiles / (double)(tiles_per * nsp);        // padding of the last split
    if (eff > best_eff + 1e-9) { best_eff = eff; splits = sp; }
  }
  static const int force_splits = (int)tuning_int("DSIR_MATCH_SPLITS", 0);   // tuning hook
  if (force_splits > 0) splits = force_splits < tiles ? force_splits : tiles;
  int cols = ((tiles + splits - 1) / splits) * BC;
  splits = (K + cols - 1) / cols;
  dim3 grid((unsigned)((int64_t)rb_count * splits * pairs));
  if (ev0) hipEventRecord(ev0, st);
  if (rt == 4)      hipLaunchKernelGGL((nn_match_kernel<4>), grid, dim3(256), 0, st, a, b, sa, sb, J, K, cols, rb_count, splits, packed, tstamp, gate, gate_min, rowlist);
  else if (rt == 2) hipLaunchKernelGGL((nn_match_kernel<2>), grid, dim3(256), 0, st, a, b, sa, sb, J, K, cols, rb_count, splits, packed, tstamp, gate, gate_min, rowlist);
  else              hipLaunchKernelGGL((nn_match_kernel<1>), grid, dim3(256), 0, st, a, b, sa, sb, J, K, cols, rb_count, splits, packed, tstamp, gate, gate_min, rowlist);
  if (ev1) hipEventRecord(ev1, st);
}

void nn_match_scratch_layout(void* scratch, int pairs, int J, int K, float** sa, unsigned long long** packed) {
  float* s = reinterpret_cast<float*>(scratch);
  const size_t f = ((size_t)pairs * J + (size_t)pairs * K + 3) & ~(size_t)3;
  *sa = s;
  *packed = reinterpret_cast<unsigned long long*>(s + f);
}

void launch_nn_match_ws(const float* a, const float* b, int pairs, int J, int K, int32_t* idx, void* scratch,
                        hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, bool ref_norms_cached, unsigned long long* tstamp,
                        bool src_norms_ready) {
  float* sa = reinterpret_cast<float*>(scratch);
  float* sb = sa + (size_t)pairs * J;
  size_t f = ((size_t)pairs * J + (size_t)pairs * K + 3) & ~(size_t)3;
  unsigned long long* packed = reinterpret_cast<unsigned long long*>(sa + f);
  const int64_t ra = (int64_t)pairs * J, rb = (int64_t)pairs * K;
  if (!src_norms_ready) hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((ra + 15) / 16)), dim3(256), 0, st, a, ra, sa, packed);
  // the ref descriptors are loop invariant in dsir_register: their norms stay in the (persistent) scratch
  if (!ref_norms_cached) hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((rb + 15) / 16)), dim3(256), 0, st, b, rb, sb, nullptr);
  launch_core(a, b, sa, sb, pairs, J, K, packed, st, ev0, ev1, tstamp, nullptr, 0);
  hipLaunchKernelGGL(unpack_idx_kernel, dim3(256), dim3(256), 0, st, packed, ra, idx);
}

// Exhaustive search of what the screening (nn_screen.hip) could not decide: all rows of the pairs with
// gate[pair] >= gate_min, and the rows rowlist[pair][0 .. gate[pair]) of the other pairs (workgroups without work exit at
// once).  Results stay in `packed` (low word = index), whose slots the caller has preset to all ones.
void launch_nn_match_gated(const float* a, const float* b, const float* sa, const float* sb, int pairs, int J, int K,
                           unsigned long long* packed, const int32_t* gate, int gate_min, const int32_t* rowlist,
                           hipStream_t st) {
  launch_core(a, b, sa, sb, pairs, J, K, packed, st, nullptr, nullptr, nullptr, gate, gate_min, nullptr);
  launch_core(a, b, sa, sb, pairs, J, K, packed, st, nullptr, nullptr, nullptr, gate, gate_min, rowlist);
}

}  // namespace dsir
