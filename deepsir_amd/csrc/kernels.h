// Internal host-side launch API of the HIP kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace dsir {

constexpr int kKnn = 16;  // neighbours per point (args.num_knn; one MFMA row tile)

// Measurement / A-B switches (DSIR_* environment variables) are read through ONE gate (engine.hip): nullptr unless the gate is
// open (DSIR_TUNING=1, or dsir_set_tuning(1) before the first call that reads a switch).
const char* tuning_env(const char* name);
inline bool tuning_flag(const char* name) { return tuning_env(name) != nullptr; }
inline long long tuning_int(const char* name, long long dflt) { const char* e = tuning_env(name); return e ? atoll(e) : dflt; }

// GroupNorm of the PRODUCER, applied lazily in the consumer's prologue:
// the producer wrote raw conv outputs plus per-(cloud,group) sum / sum-of-squares.
#ifndef DSIR_GN_WORDS_DEFINED
#define DSIR_GN_WORDS_DEFINED
constexpr int kGnWords = 4;   // 8-byte words per (cloud, group) statistics slot: device_utils.h, gn_block_commit / gn_stat_get
#endif
struct GnRef {
  const double* stats;   // [clouds][groups][kGnWords] (two integer-valued fp64 limbs per sum, device_utils.h); nullptr => no normalisation
  const float* gamma;    // [C]
  const float* beta;     // [C]
  int groups;
  double inv_count;      // 1 / ((C/groups) * rows_per_cloud)
};

// One channel segment of the A operand (rows = points or (point,neighbour) pairs).
struct Seg {
  const float* x;        // [clouds][rows_src][ld]
  int64_t cloud_stride;  // floats
  int C;                 // channels taken from this segment
  int ld;                // row stride in floats
  const int32_t* idx;    // optional row gather index [clouds][M]
  int64_t idx_cloud_stride;
                         // source row = idx ? idx[r] : r
  GnRef gn;
  int act;               // 1 => LeakyReLU(0.2) after the normalisation
  // lse_uv.hip: the segment is the position-encoding layer of a level whose rows are NOT in memory (x == nullptr): row r =
  // (point i = r / 16, neighbour j = idx[r]) is rebuilt as a[c] dist[r] + U[j][c] + V[i][c] from the tables below
  const float* uv = nullptr;        // [clouds][n][2 C] = [U | V] per point
  int64_t uv_cloud_stride = 0;
  const float* dist = nullptr;      // [clouds][n * 16]
  int64_t dist_cloud_stride = 0;
  const float* w8 = nullptr;        // [C][8] folded weights {a, ux, uy, uz, vx, vy, vz, b} (engine.hip, up_lse_uv)
};

enum AMode { A_SEGS = 0, A_LSE = 1 };
// EPI_ATT2 = attentive pooling with the score GEMM split by linearity:
//   fc [gather(f); enc] = gather(W1 f) + W2 enc.   G = W1 f is a per-POINT GEMM (16x fewer rows) done
//   beforehand; this launch contracts only the enc half (A = enc, W = fc[:, d/2:], ldw = d), adds the
//   gathered G rows to the scores, and pools [gather(f); enc] exactly as EPI_ATT does.
enum Epilogue { EPI_GN = 0, EPI_ACT = 1, EPI_LINEAR = 2, EPI_L2NORM = 3, EPI_ATT = 4, EPI_ATT2 = 5 };

struct GemmArgs {
  int amode = A_SEGS;
  int nseg = 1;
  Seg seg[2] = {};
  // A_LSE: relative position encoding from xyz + neighbour index (RandLANet.py:197-212)
  const float* xyz = nullptr;   // [clouds][n][3]
  int64_t xyz_cloud_stride = 0;
  const int32_t* neigh = nullptr;  // [clouds][n][16]
  int64_t neigh_cloud_stride = 0;

  const float* W = nullptr;     // [Cout][ldw] row-major, Cin columns used
  int ldw = 0;                  // weight row stride in floats (0 => Cin)
  // optional fp16 split of W (x -> fp16(x), fp16(x - fp16(x)); same layout and strides, made at weight load): kernels with an
  // fp16-split form (pw_tile.hip) take it when both are set
  const void* Wh = nullptr;
  const void* Wl = nullptr;
  const float* bias = nullptr;  // [Cout] or nullptr
  int Cin = 0, Cout = 0;
  // EPI_ATT2 only: G = W1 f  [clouds][n][Cout] and the gathered-feature half of the pooled operand
  const float* g = nullptr;
  int64_t g_cloud_stride = 0;
  Seg fseg = {};                // f [clouds][n][Cout/2] with its lazy GroupNorm; fseg.idx = neighbour index of every A row
  // EPI_ATT2 only, optional: the enc half of the scores (W2 enc, the launch's own contraction) kept across launches.  In the
  // inlier model it depends on the pyramid and the weights alone, so iteration 0 stores the accumulators
  // (s2_mode 1) and the later iterations load them instead of contracting (s2_mode 2): same values, bit for bit.
  // Layout [clouds][M/16][Cout/16][64 lanes][4]: the MFMA C fragments as they sit in registers.
  float* s2 = nullptr;
  int64_t s2_cloud_stride = 0;
  int s2_mode = 0;
  int M = 0;                    // rows per cloud
  int clouds = 1;
  int epi = EPI_GN;
  float* Y = nullptr;           // [clouds][M (or M/16 for EPI_ATT)][ldy]
  int64_t y_cloud_stride = 0;
  int ldy = 0;
  double* stats_out = nullptr;  // EPI_GN: [clouds][groups_out][kGnWords]
  int groups_out = 0;
  const float* residual = nullptr;  // EPI_LINEAR: added before the store
  int64_t res_cloud_stride = 0;
  int ldres = 0;
  int grid_x = 0, grid_y = 0;   // filled by the launchers that flatten their grid (XCD-aware work mapping)
  // pw_tile_small_kernel, EPI_GN: TWO convolutions of the same input in one launch (mlp1 + mlp_skip of a dilated residual block,
  // RandLANet.py:226 / :229): W / bias are the two layers' rows one after the other, columns [0, c_split) are the first layer's
  // (Y, ldy, stats_out, groups_out as usual), columns [c_split, Cout) the second's (below).  c_split is a multiple of 64 or 0 (off).
  int c_split = 0;
  float* Y2 = nullptr; int64_t y2_cloud_stride = 0; int ldy2 = 0;
  double* stats_out2 = nullptr; int groups_out2 = 0;
  int vgrid_x = 0;              // pw_stream_kernel, EPI_GN: row-block UNITS per cloud (a function of M alone: they fix the summation order of the
                                // statistics); a workgroup walks the units bx, bx + grid_x, ... (filled by the launcher)
};

// GroupNorm statistics meet across workgroups in exact atomics (device_utils.h, gn_block_commit): the proof that every partial
// total stays an integer below 2^53 holds for at most kGnMaxContrib contributions per (cloud, group) statistic.  Every launcher
// that commits statistics states how many workgroups of ONE cloud add into one statistic (a function of the layer's shape alone);
// dsir_create bounds max_points by the largest of them over the schedule (engine.hip, gn_max_contributions).
constexpr int kGnMaxContrib = 1 << 12;
int pw_stream_gn_contributions(int M, int Cout);        // pw_stream.hip: the VIRTUAL workgroups of a column block
int pw_tile_gn_contributions(int M, int Cout, int groups);   // pw_tile.hip: row blocks x column blocks a group spans
int lse_uv_gn_contributions(int n, int KH);             // lse_uv.hip: virtual workgroups per cloud

void launch_pw_gemm(const GemmArgs& a, hipStream_t st);
bool pw_gemm_serves_pair(const GemmArgs& a);   // would launch_pw_gemm serve this launch with GemmArgs::c_split set? (ask before fusing two layers)
// narrow-layer fast path (pw_stream.hip); false => not applicable
bool launch_pw_stream(const GemmArgs& a, hipStream_t st);
// wide-layer path, LDS-tiled 128/64 x 64 x 32 (pw_tile.hip): Cin a multiple of 32 in [64,768], Cout >= 64
bool launch_pw_tile(const GemmArgs& a, hipStream_t st);
// does launch_pw_tile serve this layer with pw_tile_small_kernel (the one kernel that takes GemmArgs::c_split)?
bool pw_tile_small_serves(const GemmArgs& a);

// att_pool.hip - attentive pooling of the k = 16 layers of levels 0 - 2 on v_mfma_f32_32x32x16_f16: softmax and weighted sum in registers
// level 0 (d = 16), unsplit: scores = fc [gather(f) ; E] with both halves 8 channels wide; four points per wave
struct AttPool16Args {
  const float* f = nullptr;          // [clouds][n][f_ld], 8 channels used: the features that are gathered (raw conv outputs)
  int64_t f_cs = 0; int f_ld = 8;
  GnRef f_gn = {nullptr, nullptr, nullptr, 0, 0.0}; int f_act = 1;
  const float* enc = nullptr;        // E [clouds][n * 16][8], or nullptr: rebuilt from the tables of lse_uv.hip
  int64_t enc_cs = 0;
  const float* uv = nullptr; int64_t uv_cs = 0;       // [clouds][n][16] = [U | V]
  const float* dist = nullptr; int64_t dist_cs = 0;   // [clouds][n * 16]
  const float* w8 = nullptr;                          // [8][8] folded weights
  GnRef enc_gn = {nullptr, nullptr, nullptr, 0, 0.0}; int enc_act = 1;
  const int32_t* neigh = nullptr;    // [clouds][n][16]
  int64_t neigh_cs = 0;
  const void* Wh = nullptr;          // fp16 split of fc [16][ldw]
  const void* Wl = nullptr;
  int ldw = 16;
  float* Y = nullptr;                // [clouds][n][16]
  int64_t y_cs = 0;
  int n = 0, clouds = 0;
  int grid_x = 0;                    // filled by the launcher
};
bool launch_att_pool16(const AttPool16Args& a, hipStream_t st);
// levels 1 / 2 (d = 64 / 128) in the same unsplit form: KH-channel halves (f [n][f_ld], E [n * 16][KH] or - KH = 32 - tables [n][64]),
// fc [2 KH][ldw], Y [n][2 KH]
bool launch_att_full(const AttPool16Args& a, int KH, hipStream_t st);

// lse_uv.hip - lfa.mlp1 of levels 0 / 1 split by linearity into per-point tables: writes U | V and dist, commits the layer's GroupNorm
// statistics; the layer's output itself is never stored (consumers: att_pool.hip, pw_stream.hip loader S_UV)
struct LseUvArgs {
  const float* xyz = nullptr; int64_t xyz_cs = 0;        // [clouds][n][3]
  const int32_t* neigh = nullptr; int64_t neigh_cs = 0;  // [clouds][n][16]
  const float* w8 = nullptr;                             // [KH][8] folded weights
  float* uv = nullptr; int64_t uv_cs = 0;                // [clouds][n][2 KH]
  float* dist = nullptr; int64_t dist_cs = 0;            // [clouds][n * 16]
  double* stats_out = nullptr; int groups = 0;           // [clouds][groups][kGnWords]
  int n = 0, clouds = 0, KH = 0;
  int vgrid = 0;                                         // virtual workgroups per cloud (filled by the launcher; a function of n alone)
};
bool launch_lse_uv_stats(const LseUvArgs& a, hipStream_t st);   // KH = 8 or 32; false => outside the envelope

// mlp_out + fc_label fused (head_mlp.hip): x[32] -> feat[64] -> 64 -> 32 -> ncls   (RandLANet.py:363-367)
struct HeadArgs {
  Seg in = {};                  // last decoder block, [clouds][M][32], lazy GroupNorm + LeakyReLU
  const float *W1 = nullptr;    // mlp_out      [64][32]
  const float *W2 = nullptr, *b2 = nullptr;   // fc_label.0 (BN folded) [64][64]
  const float *W3 = nullptr, *b3 = nullptr;   // fc_label.3 (BN folded) [32][64]
  const float *W4 = nullptr, *b4 = nullptr;   // fc_label.6 [ncls][32]
  int ncls = 0, M = 0, clouds = 1;
  float* feat_out = nullptr;    // [clouds][M][64] or nullptr
  float* logits_out = nullptr;  // [clouds][M][ncls]
  // head_mlp_h.hip only: the fp16 split (high, low; same [Cout][Cin] layout) of W1 .. W4, made at weight load
  const void* Wh[4] = {nullptr, nullptr, nullptr, nullptr};
  const void* Wl[4] = {nullptr, nullptr, nullptr, nullptr};
};
bool launch_head_mlp(const HeadArgs& a, hipStream_t st);   // false => shape outside the fused envelope
// the same head as fp16-split products on the fp16 matrix pipe (fp32 accuracy); false => split weights missing / outside the envelope
bool launch_head_mlp_h(const HeadArgs& a, hipStream_t st);

// agg_chain.hip — mlp_att chain + residual + mlp_proj + L2 normalise in one launch (model.py:223-233)
struct AggArgs {
  const float* xyz = nullptr; int64_t xyz_cs = 0;   // [clouds][n][3], cloud stride in floats
  const float* score = nullptr;                     // [clouds][n]
  const float* F = nullptr;                         // [clouds][n][64] = mlp_feat(feat0)
  const float *W1 = nullptr, *b1 = nullptr;         // mlp_att, BN folded: [32][4]
  const float *W2 = nullptr, *b2 = nullptr;         // [64][32]
  const float *W3 = nullptr, *b3 = nullptr;         // [128][64]
  const float *W4 = nullptr, *b4 = nullptr;         // [256][128]
  const float *W5 = nullptr, *b5 = nullptr;         // [64][256]
  const float *W6 = nullptr, *b6 = nullptr;         // mlp_proj [64][64]
  float* desc = nullptr;                            // [clouds][n][64]
  int n = 0, clouds = 0;
  // agg_chain_h.hip only, optional: what the descriptor search needs of the descriptors, written by the same epilogue instead of a
  // second pass over them (nn_screen.hip split_norm_kernel / nn_match.hip sqnorm_kernel: same arithmetic, same bits) -
  // sq [clouds * n] = |desc|^2; hi / lo [clouds * n][64] fp16 = the screening's operand pair; packed_init [clouds * n] u64 = all ones
  // (the exhaustive search's result slots); each may be nullptr
  float* sq = nullptr;
  void* hi = nullptr;
  void* lo = nullptr;
  unsigned long long* packed_init = nullptr;
  // agg_chain_h.hip only: the fp16 split (high, low part; same [Cout][Cin] layout) of W2 .. W6, made at weight load
  const void* Wh[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const void* Wl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};
bool launch_agg_chain(const AggArgs& a, hipStream_t st);
// the same chain with its five wide layers as three fp16 MFMAs per product (fp32 accuracy, 3/16 of the fp32 MFMA time);
// false => split weights missing
bool launch_agg_chain_h(const AggArgs& a, hipStream_t st);
void split_weights_f16(const float* w, size_t n, uint16_t* hi, uint16_t* lo);   // host: x -> fp16(x), fp16(x - fp16(x))

// select.hip — feat / label pipelines (model.py:650-651, :682-697)
size_t topk_scratch_bytes(int clouds, int n);
// idx_out [clouds][k] = the k highest scores of every cloud, descending, ties in ascending index; score_out optional
int launch_topk(const float* score, int clouds, int n, int k, int32_t* idx_out, float* score_out, void* scratch, hipStream_t st);
// inverse of a gather index [clouds][m] into [clouds][n]: order [clouds * m] (sources by destination, ascending inside), offsets [clouds * n + 1]
size_t scatter_plan_scratch_bytes(int64_t total);
int launch_scatter_plan(const int32_t* idx, int m, int clouds, int n, int32_t* order, int32_t* offsets, void* scratch, hipStream_t st);
void launch_gather_rows(const float* in, int64_t in_cloud_stride, int ld, const int32_t* idx, int C, int m, int clouds, float* out,
                        hipStream_t st);
void launch_gather_i32(const int32_t* in, int64_t in_cloud_stride, const int32_t* idx, int m, int clouds, int32_t* out,
                       hipStream_t st);
void launch_l2norm64(const float* x, int64_t rows, float* y, hipStream_t st);   // F.normalize over 64 channels

// y = LeakyReLU(GN_a(a) + GN_b(b))   (RandLANet.py:228-230)
void launch_residual_combine(const float* a, GnRef ga, const float* b, GnRef gb, int C, int rows, int clouds,
                             float* y, hipStream_t st);
// out[i][c] = max_k in[idx[i][k]][c]   (RandLANet.py:374-391)
void launch_gather_max(const float* in, int64_t in_cloud_stride, const int32_t* idx, int64_t idx_cloud_stride, int C,
                       int rows_out, int clouds, float* out, hipStream_t st);

// both fused: out[i][c] = max_k LeakyReLU(GN_a(a)[idx[i][k]][c] + GN_b(b)[idx[i][k]][c]); a, b: [clouds][rows_in][C]
void launch_gather_max_combine(const float* a, GnRef ga, const float* b, GnRef gb, int rows_in, const int32_t* idx,
                               int64_t idx_cloud_stride, int C, int rows_out, int clouds, float* out, hipStream_t st);

// walk.hip - the deep pyramid levels of RandLA.forward (levels >= 2, mlp_mid, the first decoder blocks) as ONE launch.
// With a few clouds in flight (the reference's batch of one, served batches of up to 8 pairs) those layers are ~20 dependent launches
// of 5 - 16 us each, most of it the fixed cost of a launch (4.5 - 5 us per dependent kernel node of a replayed graph, measured:
// profiles/README.md round 5).  The walker runs the same tile bodies (pw_tile_body.h, att_pool_body.h, misc_body.h) as PHASES of one
// persistent launch: a phase is a set of independent workgroup tiles per cloud; the workgroups of a cloud take tiles from the
// phase's queue (one returning atomic each - no workgroup ever waits for one that is not running) and start phase p when every
// tile of phase p - 1 has been published (plain stores -> s_waitcnt -> barrier -> agent-scope release fence -> counter add;
// consumer: relaxed poll -> agent-scope acquire fence -> barrier): 1.0 - 1.4 us per hand-off (tools/ubench/xcd_sync.hip).
// Same code on the same operands in the same order per output element, GroupNorm statistics in exact atomics: same bits as the
// separate launches.
struct GmcArgs {            // launch_gather_max_combine's operands
  const float* a; GnRef ga; const float* b; GnRef gb; int rows_in; const int32_t* idx; int64_t idx_cs; int C, rows_out; float* out; int bpc;
};
enum WalkKind { WK_TILE_SMALL = 0, WK_TILE_ATT2 = 1, WK_ATT_FULL64 = 2, WK_GMC = 3 };
struct WalkJob {            // one phase
  int kind = 0;
  int v0 = 0, v1 = 0;       // WK_TILE_SMALL: RTS (1 | 2), epilogue (EPI_GN | EPI_LINEAR); WK_TILE_ATT2: RT (1 | 2), GemmArgs::s2_mode
  int gx = 1, gy = 1;       // gx * gy tiles per cloud; tile j = (bx = j % gx, by = j / gx)
  int dep = -1;             // the phase whose tiles must all be published before a tile of this one starts (-1: none)
  GemmArgs gemm;
  AttPool16Args att;
  GmcArgs gmc = {};
};
constexpr int kWalkMaxPhases = 32;
constexpr int kWalkCtrWords = 2 * kWalkMaxPhases + 2;      // per cloud: {next, done} per phase, then {error flag, pad}
struct WalkProgram {
  int nphases = 0, clouds = 0, wpc = 1;    // wpc: workgroups per cloud
  int flags = 0;                           // bit 0: a cloud's workgroups are blockIdx % clouds (all XCDs) instead of one XCD slot; bit 1: no program prefetch
  unsigned* ctr = nullptr;                 // [clouds][kWalkCtrWords], zero before the launch
  unsigned long long* trace = nullptr;     // measurement (DSIR_WALK_TRACE): [kWalkMaxPhases][4] device-clock stamps of cloud 0 - earliest tile
                                           // picked up, earliest tile past its wait, latest body end, latest publish - preset to ~0 / 0
  WalkJob job[kWalkMaxPhases];
};
// can these launches be phases?  (the same predicates the stand-alone launchers apply, plus the instantiations walk.hip holds)
bool walk_plan_gemm(const GemmArgs& a, WalkJob* out);      // pw_tile.hip: the small-M GroupNorm / linear kernels, split attentive pooling
bool walk_plan_att_full(const AttPool16Args& a, int KH, int wpc, WalkJob* out);      // att_pool.hip (KH = 64); wpc: workgroups a cloud will have
bool walk_plan_gmc(const GmcArgs& a, int wpc, WalkJob* out);                         // misc.hip
// prog: DEVICE copy of the program (host copy for the grid geometry)
void launch_walk(const WalkProgram& host, const WalkProgram* prog, hipStream_t st);

void launch_narrow_i64(const int64_t* src, int32_t* dst, int64_t n, hipStream_t st);

// Several device fills / copies in ONE launch (misc.hip): what opens a registration - zeroing flags and the statistics arena,
// staging the two input clouds side by side, presetting reduction targets - was six memset / memcpy launches (round 4).
// Sizes and addresses are multiples of 4 bytes; 16-byte accesses where an operation is aligned for them.
struct MemOp { void* dst; const void* src; size_t bytes; uint32_t fill; };     // src == nullptr: every 32-bit word = fill
struct MemOps {
  static constexpr int kMax = 10;
  int n = 0;
  MemOp op[kMax];
  void fill(void* dst, size_t bytes, uint32_t word = 0) { if (dst && bytes) op[n++] = MemOp{dst, nullptr, bytes, word}; }
  void copy(void* dst, const void* src, size_t bytes) { if (dst && src && bytes) op[n++] = MemOp{dst, src, bytes, 0u}; }
  bool full() const { return n >= kMax; }
};
void launch_mem_ops(const MemOps& m, hipStream_t st);    // no-op when m.n == 0

// Caller-supplied index tensors are copied with every entry clamped into its valid range (no gather can leave its
// tensor); an out-of-range entry raises bit 1 (value 2) of flag[cloud % flag_mod] (flag may be nullptr).
void launch_copy_idx_clamped(const int32_t* src, int64_t src_cloud_stride, int count, int limit, int clouds, int32_t* dst,
                             int64_t dst_cloud_stride, int32_t* flag, int flag_mod, hipStream_t st);
struct PyramidIdxCopy {
  const int32_t *neigh, *sub, *interp;      // [clouds][S][16], [clouds][S1][16], [clouds][S]
  int32_t *neigh_out, *sub_out, *interp_out;
  int S, S1, levels;
  int nl[6], off[6], soff[6];               // level sizes / offsets (data_base.py:178-181)
  int32_t* flag; int flag_mod;
};
void launch_copy_pyramid_idx(const PyramidIdxCopy& a, int clouds, hipStream_t st);

// KNN (data_base.py:153-183): one level.  support = first n_support points of `pts`.
void launch_knn16(const float* pts, int64_t cloud_stride, int stride, int n, int clouds, int32_t* out,
                  int64_t out_cloud_stride, hipStream_t st);
// the interpolation searches of all levels and the 16-NN searches of the levels without a grid in ONE launch (knn.hip)
struct KnnSmallJobs {
  static constexpr int kMax = 8;
  struct Job {
    int kind;            // 0: nearest support point (launch_nn1); 16-NN (launch_knn16): 1 one wave per query (small levels), 2 one lane per query
    int n, n_support;    // queries (= the level's points); kind 0: support = the first n_support points
    int32_t* out;        // cloud 0's output
    int64_t ocs;         // ints between clouds in out
    int b0;              // first workgroup (filled by the launcher)
  } job[kMax];
  int njobs;
};
bool knn16_takes_wave_kernel(int n, int clouds);      // launch_knn16 would run the one-wave-per-query kernel
void launch_knn_small_levels(const float* pts, int64_t cloud_stride, int stride, int clouds, KnnSmallJobs& jobs, hipStream_t st);
// exact grid-pruned variant for large levels (knn_grid.hip); scratch from knn_grid_scratch_bytes
size_t knn_grid_scratch_bytes(int clouds, int n);
void launch_knn16_grid(const float* pts, int64_t cloud_stride, int stride, int n, int clouds, int32_t* out,
                       int64_t out_cloud_stride, void* scratch, hipStream_t st);
// several levels (each knn16_grid_can_merge: its grid is built in one launch) in two launches - all grids, all searches; at most
// GridLevelsArgs::kMax = 4 levels; scratch[l] from knn_grid_scratch_bytes(clouds, n[l]), left as launch_knn16_grid leaves it
bool knn16_grid_can_merge(int n);
void launch_knn16_grid_levels(const float* pts, int64_t cloud_stride, int stride, int nlev, const int* n, int clouds, int32_t* const* out,
                              int64_t out_cloud_stride, void* const* scratch, hipStream_t st);
void launch_nn1(const float* pts, int64_t cloud_stride, int stride, int n_query, int n_support, int clouds,
                int32_t* out, int64_t out_cloud_stride, hipStream_t st);
// the same search through the grid launch_knn16_grid has just built over the first n_support points (its scratch); same bits
// query_scratch (optional): the scratch launch_knn16_grid(.., n = n_query, ..) left for the QUERY level - its points in cell order are
// then taken as the queries (neighbouring lanes walk neighbouring cells), each result written at the query's original index
void launch_nn1_grid(const float* pts, int64_t cloud_stride, int stride, int n_query, int n_support, int clouds, int32_t* out,
                     int64_t out_cloud_stride, const void* grid_scratch, hipStream_t st, const void* query_scratch = nullptr);
void launch_copy_xyz(const float* pts, int64_t cloud_stride, int stride, int n, int clouds, float* out,
                     int64_t out_cloud_stride, hipStream_t st);
// all levels of a pyramid in one launch each (every level is a prefix of the level above, data_base.py:166-172):
// xyz[off[l] + i] = points[i], i < nl[l];  sub[soff[l] + i] = neigh[off[l] + i], i < nl[l + 1]
constexpr int kMaxLevels = 4;   // = DSIR_MAX_LEVELS (include/dsir.h)
struct PyramidLevels { int L; int nl[kMaxLevels + 1]; int off[kMaxLevels + 1]; int soff[kMaxLevels + 1]; int S, S1; };
void launch_copy_xyz_levels(const float* pts, int64_t cloud_stride, int stride, const PyramidLevels& lv, int clouds, float* xyz,
                            int64_t xyz_cs, hipStream_t st);
void launch_copy_sub_levels(const int32_t* neigh, int64_t neigh_cs, const PyramidLevels& lv, int clouds, int32_t* sub, int64_t sub_cs,
                            hipStream_t st);
void launch_copy_rows_i32(const int32_t* src, int64_t src_cloud_stride, int rows, int width, int clouds, int32_t* dst,
                          int64_t dst_cloud_stride, hipStream_t st);

// score_fun (model.py:701-757)
struct ScoreScratch {  // per cloud: [0]=max feat, [1]=max label weight, [2]=max prob   (float bits, atomics)
  float* red;          // [clouds][4]
  float* prob;         // [clouds][n]
  int32_t* label;      // [clouds][n]
};
// red_preset: s.red already holds -inf in every word (a registration presets it in its opening launch_mem_ops)
void launch_score(const float* feat, const float* logits, int ncls, const float* xyz, int64_t xyz_cloud_stride,
                  const int32_t* neigh, int64_t neigh_cloud_stride, int clouds, int n, ScoreScratch s, float* score,
                  int32_t* label_out, hipStream_t st, bool red_preset = false);

// fused distance GEMM + row arg-min (matchnet.py:96-113 + model.py:566); ev0/ev1 (optional) bracket the main kernel
size_t nn_match_scratch_bytes(int pairs, int J, int K);
void launch_nn_match_ws(const float* a, const float* b, int pairs, int J, int K, int32_t* idx, void* scratch,
                        hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, bool ref_norms_cached = false,
                        unsigned long long* tstamp = nullptr,    // tstamp: {min start, max end} device-clock slot or nullptr
                        bool src_norms_ready = false);           // the src norms and the preset result slots are already in the scratch
// where launch_nn_match_ws keeps the src norms and the packed result slots inside its scratch (a producer may fill them: AggArgs)
void nn_match_scratch_layout(void* scratch, int pairs, int J, int K, float** sa, unsigned long long** packed);
// exhaustive search of all rows of the pairs with gate[pair] >= gate_min and of the rows rowlist[pair][0 .. gate[pair]) of
// the others; results left in `packed` (preset to all ones)
void launch_nn_match_gated(const float* a, const float* b, const float* sa, const float* sb, int pairs, int J, int K,
                           unsigned long long* packed, const int32_t* gate, int gate_min, const int32_t* rowlist,
                           hipStream_t st);

void launch_sqnorm(const float* x, int64_t rows, float* out, hipStream_t st);   // |x|^2 of [rows][64], nn_match's order

// nn_screen.hip — the same arg-min, screened with fp16 MFMAs under a rigorous bound and decided in exact fp32
size_t nn_screen_scratch_bytes(int pairs, int J);
// Pruned search (nn_prune.hip): the screening may walk the src rows and the ref columns in a given ORDER and, per row block, only
// a LIST of column tiles.  launch_prune_rows fills the fields (all of them); default-constructed: the dense search.  Results are
// reported in original indices.
struct ScreenOrder {
  const int32_t* rows = nullptr;     // [pairs][J]: row order (position -> src row)
  const int32_t* cols = nullptr;     // [pairs][K]: column order (position -> ref row)
  const void* bh = nullptr;          // with cols: the ref side's fp16 pairs and seeds ALREADY in column order (rows of 64 halves / floats
  const void* bl = nullptr;          //   per position): the search streams contiguous tiles, `cols` only names the winners
  const float* sbp = nullptr;
  const int32_t* tlist = nullptr;    // [pairs][row blocks][tl_stride]: tiles (of 64 positions of the column order) a row block must visit
  const int32_t* tcount = nullptr;   // [pairs][row blocks]
  int tl_stride = 0;
  const int32_t* rborder = nullptr;  // [pairs][row blocks]: the row blocks by descending tile count (the order items are taken in)
  int32_t* queue = nullptr;          // [8], zeroed before every launch: the XCDs' item counters (with tlist)
  long long cand_rs = 0;             // filled by launch_nn_screen: rows of the launch (pairs x J) = the stride between the slots of the entry lists
};
int nn_screen_rows_per_block(int J);   // rows a workgroup of the screening owns for this J (what tile lists are built for)
int nn_screen_max_bound_tiles();
void launch_centroid_argmin(const void* ah, const void* al, const void* ch, const void* cl, const float* cn2, int pairs, int J, int nt,
                            int32_t* tstar, hipStream_t st);
void launch_tile_T(const void* ah, const void* al, const float* sa, const int32_t* rows, const int32_t* tstar, const void* bh, const void* bl,
                   const float* sbp, int pairs, int J, int K, int nt, float* T, hipStream_t st);
void launch_tile_bound(const void* ah, const void* al, const float* sa, const int32_t* rows, const float* T, const void* ch, const void* cl,
                       const float* cn2, const float* rad, int pairs, int J, int nt, int32_t* tlist, int32_t* tcount, int tl_stride,
                       int32_t* rborder, hipStream_t st);
// nn_prune.hip: column order (Morton order of the ref points) + tile bounds once per registration; row order, upper bounds and
// tile lists per iteration (acc, optional: device 2 x u64 running totals {tile products kept, tile products in all})
bool nn_prune_supported(int pairs, int J, int K);
size_t nn_prune_scratch_bytes(int pairs, int J, int K);
int launch_prune_ref(const float* ref_xyz, int64_t xyz_cloud_stride, const float* desc_ref, const void* bh, const void* bl, const float* sb,
                     int pairs, int J, int K, void* scratch, hipStream_t st);
int launch_prune_rows(const float* desc_src, const float* desc_ref, const void* ah, const void* al, const float* sa, const float* sb,
                      const int32_t* idx_prev, int pairs, int J, int K, void* scratch, hipStream_t st, ScreenOrder* ord, unsigned long long* acc);
// fp32 [rows][64] -> fp16 hi / lo; `bad` (optional device flag) is set when an element is outside the screening's domain
void launch_split16(const float* x, int64_t rows, void* hi, void* lo, hipStream_t st, int32_t* bad = nullptr);
// the same split plus launch_sqnorm's |x|^2 per row, one pass
void launch_split16_norm(const float* x, int64_t rows, void* hi, void* lo, float* sq, hipStream_t st, int32_t* bad = nullptr);
void launch_nn_screen(const float* a, const float* b, const void* ah, const void* al, const void* bh, const void* bl,
                      const float* sa, const float* sb, int pairs, int J, int K, int32_t* idx, void* scratch, hipStream_t st,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, unsigned long long* stats = nullptr,
                      bool keep_gate = false, const int32_t* bad = nullptr, unsigned long long* acc = nullptr,
                      hipEvent_t evk0 = nullptr, hipEvent_t evk1 = nullptr,    // evk0/evk1 bracket screen_kernel alone
                      const ScreenOrder& ord = ScreenOrder());
// diagnostics of the screening on ONE small pair (dsir_screen_bounds): lower / upper / exact [J][K]; the candidate lists of the
// product launch that ran on `scratch` (pairs = 1): thresh [J], count [J], code / lower [J][nn_screen_cap()]
int nn_screen_cap();
void launch_screen_bounds(const float* a, const float* b, const void* ah, const void* al, const void* bh, const void* bl,
                          const float* sa, const float* sb, int J, int K, float* lower, float* upper, float* exact, float* zacc,
                          hipStream_t st);
void launch_screen_export(const void* scratch, int J, float* thresh, int32_t* count, int32_t* code, float* lower, hipStream_t st);
// keep_gate: pairs found not selective by the previous call on this scratch stay exhaustive; bad: launch_split16's flag;
// acc (device, 4 x u64, optional): running totals {searches, rows, rows left to the exhaustive kernel, pairs searched
// exhaustively as a whole}

// weighted Kabsch + SE(3) bookkeeping (model.py:22-66, :586-595; se3_torch.py:28-77)
struct KabschArgs {
  const float* src;      // [pairs][m][3]  current (transformed) src points
  const float* ref;      // [pairs][K][3]  ref points
  const int32_t* idx;    // [pairs][m] correspondences (nullptr => ref is already matched, [pairs][m][3])
  const float* w;        // [pairs][m] weights, or logits when sigmoid != 0
  int64_t src_stride, ref_stride;  // floats between pairs
  int sigmoid;
  int pairs, m;
  float* T;              // [pairs][3][4] this iteration's transform
  int32_t* invalid;      // [pairs], OR-ed
  // optional SE(3) bookkeeping
  float* src_out;        // transformed src [pairs][m][3] (may alias src)
  int64_t src_out_stride;
  const float* T_prev;   // [pairs][.][3][4] previous cumulative (nullptr on iteration 0)
  float* T_cum;          // cumulative out
  int64_t T_stride;      // floats between pairs in T_prev / T_cum
  float* matched_out;    // [pairs][m][3] gathered ref points (or nullptr)
  int ref_ld;            // floats between ref points (0 => 3)
  const int32_t* skip;   // [pairs] or nullptr: non-zero => this pair's update is the identity, src_out untouched (ICP)
  double* part;          // kabsch_part_bytes(pairs, m, chunk_min) of scratch, or nullptr: clouds of chunk_min points and more are
                         // then reduced in chunks by several workgroups per pair (same formulas; the fp64 sums in another order)
  int chunk_min;         // 0 => kKabschChunkedMin
};
constexpr int kKabschChunkedMin = 16384;
size_t kabsch_part_bytes(int pairs, int m, int chunk_min = 0);   // 0 below the threshold
void launch_kabsch(const KabschArgs& a, hipStream_t st);

// icp.hip — point-to-point ICP refinement (test.py:241-258 / open3d registration_icp), all pairs at once
size_t icp_scratch_bytes(int pairs, int J);
void launch_icp_refine(const float* src, const float* ref, int pairs, int J, int K, int stride, float max_corr_dist,
                       int max_iter, float rel_fitness, float rel_rmse, const float* T_init, float* T_out,
                       double* stats_out, void* scratch, hipStream_t st);

// finetune.hip — Adam fine-tune of the pose on matched points (test.py:159-207), one workgroup per pair
void launch_pose_finetune(const float* src, const float* ref, const float* w, int sigmoid, int pairs, int m, const float* T_init,
                          float quant, int max_iter, float break_ratio, int max_break, float* T_out, double* stats,
                          hipStream_t st);

// align_loss.hip — ScanAlignmentLoss and its gradient down to the inlier logits (loss.py:705-851, model.py:22-66, :571-595);
// losses [n_iter][2] float64 on device (point-distance term, confidence term), summed over pairs; returns 0 on success
int launch_align_loss(const float* src, const float* ref, const int32_t* idx, const float* logits, const float* labels,
                      const float* T_gt, int P, int J, int K, int n_iter, int mse, float wt_pt, float wt_in, float discount,
                      float* T_out, double* losses, float* grad, hipStream_t st, double* loss_part);   // loss_part: [P][n_iter][2] scratch (with losses)

// pre-processing on ragged batches (preprocess.hip); return 0 on success
size_t voxel_downsample_scratch_bytes(int64_t total, int clouds);
int launch_voxel_downsample(const float* pts, const int64_t* offsets_host, int clouds, int stride, float voxel,
                            const float* crop_host, int cap, float* out, int32_t* counts, void* scratch, hipStream_t st);
size_t resample_scratch_bytes(int clouds, int cap);
int launch_resample(const float* in, const int32_t* counts, int clouds, int cap, int stride, int k, int mode, uint64_t seed,
                    float* out, void* scratch, hipStream_t st);

// evaluation metrics (metrics_util.py:27-85); out [pairs][8] float64
void launch_eval_metrics(const float* pred, int64_t pred_stride, const float* gt, const float* src, const float* ref,
                         int pairs, int n, int stride, float rte_thresh, float rre_thresh, double* out, hipStream_t st);

}  // namespace dsir
