// LDS-tiled point-wise GEMM for the wide layers (Cin a multiple of 32 in [64,768],
// Cout >= 64): pyramid levels 2/3, decoder, 128/256-wide aggregation MLP layers.
//
// Block = 4 waves, tile (64*RT rows) x 64 columns x 32-channel K chunks:
//   * global -> registers as 16-byte loads (A: RT float4 per thread, same 4 channels of RT*... rows,
//     W: 2 float4 per thread), the producer's GroupNorm + LeakyReLU applied in registers,
//     then ds_write into the OTHER LDS buffer while the MFMAs of the current chunk run
//     (double-buffered LDS, one barrier per 32-channel chunk);
//   * LDS rows are [32+2] floats: the MFMA fragment reads (lane (r,q) reads row r, k = 4s+q) are
//     bank-conflict free (34 r mod 32 = 2 r);
//   * wave w owns rows [16*RT*w, 16*RT*(w+1)) and all four 16-column tiles: per k-step RT A reads +
//     4 B reads feed 4*RT MFMAs (v_mfma_f32_16x16x4_f32, channels ascending = k-ordered fmaf chain);
//   * a weight tile is re-read from L2 once per 64*RT rows.
// Epilogues: GroupNorm statistics (wave-level reduction, one fp64 atomic per group per wave),
// bias + LeakyReLU, linear (+residual), attentive pooling — as in pw_gemm.hip.
#include "kernels.h"
#include "device_utils.h"
#include "pw_tile_body.h"
#include <cstdlib>

namespace dsir {

using namespace tile;

namespace {

template <int RT, int EPI, int SC = 0, bool H = false>
__global__ __launch_bounds__(256) void pw_tile_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[pw_tile_smem_bytes<RT, EPI, H, SC>()];
  pw_tile_body<RT, EPI, SC, H>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

template <int RTS, int EPI, bool H = false>
__global__ __launch_bounds__(256) void pw_tile_small_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[pw_tile_small_smem_bytes<RTS, EPI, H>()];
  pw_tile_small_body<RTS, EPI, H>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// fp16-split contraction when the caller supplied split weights (GemmArgs::Wh / Wl); DSIR_TILE_F32: the exact-fp32 kernels
inline bool use_split(const GemmArgs& a) {
  static const bool f32 = tuning_flag("DSIR_TILE_F32");   // A/B switch
  return !f32 && a.Wh && a.Wl;
}

template <int RTS, int EPI>
void launch_small(const GemmArgs& a, hipStream_t st) {
  dim3 grid((a.M + 32 * RTS - 1) / (32 * RTS), (a.Cout + BN - 1) / BN, a.clouds);
  if (use_split(a)) hipLaunchKernelGGL((pw_tile_small_kernel<RTS, EPI, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((pw_tile_small_kernel<RTS, EPI, false>), grid, dim3(256), 0, st, a);
}

template <int RTS>
bool launch_small_e(const GemmArgs& a, hipStream_t st) {
  switch (a.epi) {
    case EPI_GN: launch_small<RTS, EPI_GN>(a, st); return true;
    case EPI_ACT: launch_small<RTS, EPI_ACT>(a, st); return true;
    default: launch_small<RTS, EPI_LINEAR>(a, st); return true;
  }
}

template <int RT, int EPI, int SC = 0>
void launch_t(const GemmArgs& a, hipStream_t st) {
  dim3 grid((a.M + 64 * RT - 1) / (64 * RT), (a.Cout + BN - 1) / BN, a.clouds);
  if (use_split(a)) hipLaunchKernelGGL((pw_tile_kernel<RT, EPI, SC, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((pw_tile_kernel<RT, EPI, SC, false>), grid, dim3(256), 0, st, a);
}

template <int RT>
bool launch_e(const GemmArgs& a, hipStream_t st) {
  switch (a.epi) {
    case EPI_GN: launch_t<RT, EPI_GN>(a, st); return true;
    case EPI_ACT: launch_t<RT, EPI_ACT>(a, st); return true;
    case EPI_LINEAR: launch_t<RT, EPI_LINEAR>(a, st); return true;
    case EPI_ATT: launch_t<RT, EPI_ATT>(a, st); return true;
    case EPI_ATT2:
      if (a.nseg != 1 || a.Cout != 2 * a.Cin || a.Cin > 128 || !a.g || !a.fseg.idx) return false;
      if (a.s2 && a.s2_mode == 1) launch_t<RT, EPI_ATT2, 1>(a, st);
      else if (a.s2 && a.s2_mode == 2) launch_t<RT, EPI_ATT2, 2>(a, st);
      else launch_t<RT, EPI_ATT2>(a, st);
      return true;
    default: return false;
  }
}

bool seg_ok(const Seg& s) {
  return (s.ld % 4) == 0 && (s.cloud_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(s.x) % 16) == 0;
}

}  // namespace

// does launch_pw_tile hand this layer to pw_tile_small_kernel?  (engine.hip asks before it fuses two layers into one launch)
bool pw_tile_small_serves(const GemmArgs& a) {
  static const int min_cout_h = (int)tuning_int("DSIR_TILE_MIN_COUT", 32);
  static const int small_m = (int)tuning_int("DSIR_TILE_SMALL_M", 320);
  if (a.amode != A_SEGS || a.Cin < 64 || a.Cin > MAXC || (a.Cin % BK) != 0 || a.Cout < (use_split(a) ? min_cout_h : 64)) return false;
  if (!seg_ok(a.seg[0]) || (a.nseg > 1 && (!seg_ok(a.seg[1]) || (a.seg[0].C % 4) != 0))) return false;
  if ((reinterpret_cast<uintptr_t>(a.W) % 16) != 0) return false;
  return a.M <= small_m && a.epi == EPI_GN;
}

// workgroups of one cloud that add into one GroupNorm statistic of a layer served here: row blocks (the smallest block any of the
// kernels above uses has 32 rows) x the 64-column blocks a group spans
int pw_tile_gn_contributions(int M, int Cout, int groups) {
  const int gw = Cout / (groups > 0 ? groups : 1);
  const int span = (gw % BN) == 0 ? gw / BN : ((BN % gw) == 0 ? 1 : (gw + BN - 1) / BN + 1);   // groups start at multiples of their width
  return ((M + 31) / 32) * span;
}

// Would launch_pw_gemm serve this layer with one of the kernels the deep-level walker holds (walk.hip) - the fp16-split small-M
// kernels (GroupNorm statistics or linear epilogue) or the split attentive pooling?  Mirrors launch_pw_gemm's order (pw_stream.hip
// takes Cin <= 64 first) and launch_pw_tile's choices below; fills the phase's kernel variant and tile grid.
bool walk_plan_gemm(const GemmArgs& a, WalkJob* out) {
  static const bool off = tuning_flag("DSIR_NO_STREAM") || tuning_flag("DSIR_NO_TILE");
  if (off || !use_split(a) || a.M <= 0 || a.amode != A_SEGS || a.Cin <= 64 || a.Cin > MAXC || (a.Cin % BK) != 0 || a.Cout < 64) return false;
  if (!seg_ok(a.seg[0]) || (a.nseg > 1 && (!seg_ok(a.seg[1]) || (a.seg[0].C % 4) != 0))) return false;
  if ((reinterpret_cast<uintptr_t>(a.W) % 16) != 0 || a.seg[0].uv || (a.nseg > 1 && a.seg[1].uv)) return false;
  if (a.epi == EPI_GN && a.c_split == 0 && ((a.Cout / a.groups_out) % 8) != 0) return false;
  if (a.c_split > 0 && ((a.c_split % BN) != 0 || a.c_split >= a.Cout || !a.Y2 || !a.stats_out2 || a.groups_out2 < 1 ||
                        ((a.c_split / a.groups_out) % 8) != 0 || (((a.Cout - a.c_split) / a.groups_out2) % 8) != 0))
    return false;
  static const int small_m = (int)tuning_int("DSIR_TILE_SMALL_M", 320);
  static const int rt2_min = (int)tuning_int("DSIR_TILE_SMALL_RT2", 128);
  out->gemm = a;
  out->gy = (a.Cout + BN - 1) / BN;
  if (a.epi == EPI_GN || a.epi == EPI_LINEAR) {
    if (a.M > small_m) return false;
    const int p32 = ((a.M + 31) / 32) * 32, p64 = ((a.M + 63) / 64) * 64;
    const int rts = (a.M >= rt2_min && p64 == p32) ? 2 : 1;
    out->kind = WK_TILE_SMALL; out->v0 = rts; out->v1 = a.epi;
    out->gx = (a.M + 32 * rts - 1) / (32 * rts);
    return true;
  }
  if (a.epi == EPI_ATT2) {
    if (a.c_split > 0 || a.nseg != 1 || a.Cout != 2 * a.Cin || a.Cin > 128 || !a.g || !a.fseg.idx) return false;
    const int pad128 = ((a.M + 127) / 128) * 128, pad64 = ((a.M + 63) / 64) * 64;
    const int rt = (a.M >= 128 && pad128 * 3 <= pad64 * 4) ? 2 : 1;
    out->kind = WK_TILE_ATT2; out->v0 = rt; out->v1 = a.s2 ? a.s2_mode : 0;
    out->gx = (a.M + 64 * rt - 1) / (64 * rt);
    return true;
  }
  return false;
}

// Returns false when the layer is outside this kernel's envelope (caller falls back to pw_gemm.hip).
bool launch_pw_tile(const GemmArgs& a, hipStream_t st) {
  if (a.c_split > 0 && !pw_tile_small_serves(a)) return false;      // two-layer launches exist for the small-M kernel only
  // Cout >= 64 - or >= 32 with the fp16-split contraction, where the unused half of the 64-column tile costs next to nothing
  // (the level-0 decoder layer 160 -> 32, otherwise left to the generic pw_gemm.hip kernel)
  static const int min_cout_h = (int)tuning_int("DSIR_TILE_MIN_COUT", 32);   // A/B hook
  if (a.amode != A_SEGS || a.Cin < 64 || a.Cin > MAXC || (a.Cin % BK) != 0 || a.Cout < (use_split(a) ? min_cout_h : 64)) return false;
  if (!seg_ok(a.seg[0]) || (a.nseg > 1 && (!seg_ok(a.seg[1]) || (a.seg[0].C % 4) != 0))) return false;
  if ((reinterpret_cast<uintptr_t>(a.W) % 16) != 0) return false;
  if (a.epi == EPI_GN && a.c_split == 0 && ((a.Cout / a.groups_out) % 8) != 0) return false;
  if (a.c_split > 0 && ((a.c_split % BN) != 0 || a.c_split >= a.Cout || !a.Y2 || !a.stats_out2 || a.groups_out2 < 1 ||
                        ((a.c_split / a.groups_out) % 8) != 0 || (((a.Cout - a.c_split) / a.groups_out2) % 8) != 0))
    return false;
  // rows per block: a function of M only (batch-invariant tiling)
  static const int small_m = (int)tuning_int("DSIR_TILE_SMALL_M", 320);   // tuning hook; 0 = off
  if (a.M <= small_m && (a.epi == EPI_GN || a.epi == EPI_ACT || a.epi == EPI_LINEAR)) {
    // 64-row blocks (two row tiles per wave: W fragments reused twice) when they pad no more than 32-row blocks
    static const int rt2_min = (int)tuning_int("DSIR_TILE_SMALL_RT2", 128);   // tuning hook
    const int p32 = ((a.M + 31) / 32) * 32, p64 = ((a.M + 63) / 64) * 64;
    if (a.M >= rt2_min && p64 == p32) return launch_small_e<2>(a, st);
    return launch_small_e<1>(a, st);
  }
  // 128-row tiles unless they waste > 25 %
  const int pad128 = ((a.M + 127) / 128) * 128, pad64 = ((a.M + 63) / 64) * 64;
  if (a.M >= 128 && pad128 * 3 <= pad64 * 4) return launch_e<2>(a, st);
  return launch_e<1>(a, st);
}

}  // namespace dsir
