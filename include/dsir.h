/*
 * dsir.h — C ABI of the MI355X-native registration engine (libdsir.so).
 *
 * Drop-in boundary for the hot path of LeoQLi/DeepSIR.  Every
 * entry point names the reference interface it replaces (file:line into the
 * reference tree).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers (HBM) unless a parameter says
 *     "host".  Calls are asynchronous on the context's HIP stream
 *     (dsir_stream) unless documented otherwise; dsir_sync waits for it.
 *   - Activations are POINT-MAJOR: a feature tensor is [clouds][points][C]
 *     fp32 (the reference is channel-major [B,C,N]; the Python wrapper
 *     transposes where it returns such tensors to the caller).
 *   - Indices are int32 on this side (the reference's int64 index tensors are
 *     narrowed with dsir_narrow_i64).
 *   - A "cloud batch" is a set of clouds with the same point count; src and
 *     ref clouds of P pairs are 2P clouds for the shared feature extractor.
 *   - Return value: 0 on success, non-zero on error; the message is then
 *     available from dsir_last_error.  A degenerate Kabsch covariance is NOT
 *     an error: identity is returned and the pair's invalid flag is set
 *     (reference network/model.py:61-64).
 *   - A context is not thread-safe: one context per (thread, device).
 */
#ifndef DSIR_H
#define DSIR_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIR_MAX_LEVELS 4
#define DSIR_KNN 16

typedef struct dsir_ctx dsir_ctx;

/* The args fields the path reads: reference arguments.py:27-82,
 * network/model.py:122-126, network/RandLANet.py:240-247. */
typedef struct dsir_cfg {
  int32_t feat_len;                       /* 3 (xyz) or 4 (xyz+reflectance)   */
  int32_t num_knn;                        /* must be 16                        */
  int32_t num_layers;                     /* must be 4                         */
  int32_t sub_sampling_ratio[DSIR_MAX_LEVELS];
  int32_t d_out[DSIR_MAX_LEVELS];         /* 16,64,128,256                     */
  int32_t out_feat_dim;                   /* 64                                */
  int32_t num_classes;                    /* 19 (semantic head of feat_extractor) */
  int32_t max_points;                     /* workspace sizing: points per cloud (1024 .. dsir_max_points_limit(): 131072, the bound up
                                           * to which the GroupNorm statistics are provably order independent) */
  int32_t max_pairs;                      /* workspace sizing: pairs per call   */
  int32_t pipeline;                       /* args.pipeline (network/model.py:122,131): DSIR_PIPELINE_*; selects
                                           * which sub-networks exist, i.e. which state-dict keys are expected  */
} dsir_cfg;

#define DSIR_PIPELINE_ALIGN 0   /* feat_extractor + mlp_feat/att/proj + inlier_model (370 keys): the whole path  */
#define DSIR_PIPELINE_FEAT  1   /* feat_extractor + mlp_feat/att/proj: key-point descriptors (dsir_forward_pair)  */
#define DSIR_PIPELINE_LABEL 2   /* feat_extractor only: semantic head (dsir_forward_pair)                         */

/* ---- lifecycle ------------------------------------------------------------ */

/* Replaces Network.__init__ + .to(device) (network/model.py:119-195, test.py:609-611). */
int dsir_create(int device, const dsir_cfg* cfg, dsir_ctx** out);
void dsir_destroy(dsir_ctx* ctx);
const char* dsir_last_error(const dsir_ctx* ctx);   /* ctx may be NULL: creation errors */
/* The HIP stream (hipStream_t) every call is ordered on. */
void* dsir_stream(dsir_ctx* ctx);
int dsir_sync(dsir_ctx* ctx);
/* Order every later call of this context on a CALLER-OWNED hipStream_t instead (restore_own != 0: back to the context's own
 * stream; a NULL hip_stream with restore_own == 0 is the legacy default stream, which is what torch's default stream is) - e.g.
 * the host framework's current stream, so that the engine's operators interleave with the caller's kernels without host
 * synchronisation and can be captured into the caller's hipGraph.  The context never destroys a caller's stream.  Work enqueued earlier stays
 * on the stream it was enqueued on, and the new stream WAITS for it on the device (an event recorded on the old stream): every
 * call of a context re-uses its one workspace arena, so launches on the new stream must not overtake the old stream's.  The
 * one exception is a stream under capture, which cannot wait on outside work: synchronise before the capture begins. */
int dsir_set_stream(dsir_ctx* ctx, void* hip_stream, int restore_own);
int dsir_num_weights(const dsir_ctx* ctx);
/* i-th expected state-dict key and its element count (for host-side validation). */
const char* dsir_weight_name(const dsir_ctx* ctx, int i, int64_t* numel);

/* Replaces Network.load_state_dict (test.py:614): called once per state-dict
 * key with a HOST fp32 pointer; the engine copies.  Unknown keys and shape
 * mismatches are errors (strict loading); '*.num_batches_tracked' is accepted
 * and ignored.  dsir_finalize_weights folds eval-mode BatchNorm1d into the
 * preceding Conv1d (RandLANet.py:34-55) and uploads everything; it fails if a
 * key is missing. */
int dsir_load_weight(dsir_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
int dsir_finalize_weights(dsir_ctx* ctx);

/* ---- stage entry points (each mirrors one reference operator) ------------- */

/* torch int64 index tensor -> int32 (device to device). */
int dsir_narrow_i64(dsir_ctx* ctx, const int64_t* src, int32_t* dst, int64_t n);

/* Replaces DataBase.nn_search (dataloader/data_base.py:153-183) incl. the
 * third-party torch_points_kernels.knn: 4-level KNN pyramid of `clouds`
 * clouds of n points each.  points: [clouds][n][stride] fp32, xyz in the
 * first three columns.
 * Outputs (per cloud, levels concatenated; S = sum n_l, S1 = sum n_{l+1}):
 *   xyz_multi [clouds][S][3] f32, neigh_idx [clouds][S][16] i32,
 *   sub_idx [clouds][S1][16] i32, interp_idx [clouds][S] i32.
 * Tie rule: (fp32 squared distance, then lower index) — oracle/knn.py. */
int dsir_knn_pyramid(dsir_ctx* ctx, const float* points, int stride, int clouds, int n,
                     float* xyz_multi, int32_t* neigh_idx, int32_t* sub_idx, int32_t* interp_idx);

/* Replaces RandLA.forward (network/RandLANet.py:311-372).
 * which: 0 = feat_extractor (Cin = feat_len, num_classes logits),
 *        1 = inlier_model   (Cin = 6, 1 logit).
 * features [clouds][n][cin]; pyramids as produced by dsir_knn_pyramid.
 * Outputs: feat [clouds][n][64], logits [clouds][n][ncls] (either may be NULL). */
int dsir_randla_forward(dsir_ctx* ctx, int which, const float* features, int cin, int clouds, int n,
                        const float* xyz_multi, const int32_t* neigh_idx, const int32_t* sub_idx,
                        const int32_t* interp_idx, float* feat, float* logits);

/* Replaces torch.max(logits,1) + Network.feat_score/score_fun with num_sub<=0
 * (network/model.py:638-644, :668-757).
 * feat [clouds][n][64], logits [clouds][n][ncls], xyz [clouds][n][3],
 * neigh_idx = level-0 rows of the pyramid with row stride `neigh_stride` ints
 * between clouds.  Outputs: score [clouds][n] f32, label [clouds][n] i32 (may be NULL). */
int dsir_score(dsir_ctx* ctx, const float* feat, const float* logits, const float* xyz, int64_t xyz_cloud_stride,
               const int32_t* neigh_idx, int64_t neigh_cloud_stride, int clouds, int n, float* score, int32_t* label);

/* Replaces one cloud's half of Network.aggregation (network/model.py:209-235):
 * desc = normalize(mlp_proj(mlp_feat(feat0) + mlp_att([xyz; score]))).
 * xyz [clouds][n][3], feat0 [clouds][n][64], score [clouds][n] -> desc [clouds][n][64]. */
int dsir_aggregate(dsir_ctx* ctx, const float* xyz, int64_t xyz_cloud_stride, const float* feat0, const float* score,
                   int clouds, int n, float* desc);

/* Replaces match_features_V2 + min(dim=2)[1] incl. the 6000-row chunking
 * (network/matchnet.py:96-144, network/model.py:558-569): for every src
 * descriptor the index of the nearest ref descriptor, distance evaluated as
 * fp32 ((-2 a.b) + |a|^2) + |b|^2, ties to the lower index.
 * desc_src [pairs][J][64], desc_ref [pairs][K][64] -> idx [pairs][J] i32. */
int dsir_nn_match(dsir_ctx* ctx, const float* desc_src, const float* desc_ref, int pairs, int J, int K, int32_t* idx);

/* The same arg-min as dsir_nn_match, bit for bit, the way dsir_register computes it: columns are first discarded by an
 * fp16-split MFMA screening with a rigorous error bound, the exact fp32 formula then decides among the survivors
 * (csrc/nn_screen.hip); rows the screening cannot decide are searched by the exhaustive fp32 kernel of dsir_nn_match.
 * Any finite input gives the exact result: components outside the screening's domain (|x| > 16) or not finite switch the whole
 * call to the exhaustive kernel.
 * stats (HOST, optional): [0] = candidate entries emitted by the screening, [1] = rows left to the exhaustive kernel. */
int dsir_nn_match_screened(dsir_ctx* ctx, const float* desc_src, const float* desc_ref, int pairs, int J, int K,
                           int32_t* idx, int64_t* stats);

/* Replaces compute_rigid_transform_2 (network/model.py:22-66): weighted
 * Kabsch with fp64 3x3 SVD on device.  src/tgt [pairs][m][3], w [pairs][m]
 * -> T [pairs][3][4] f32, invalid [pairs] i32 (1 = non-finite covariance,
 * identity returned). */
int dsir_kabsch(dsir_ctx* ctx, const float* src, const float* tgt, const float* w, int pairs, int m,
                float* T, int32_t* invalid);

/* ---- the whole path -------------------------------------------------------- */

typedef struct dsir_pair_batch {
  int32_t pairs;                 /* P */
  int32_t n_src, n_ref;          /* J, K points per cloud */
  const float* points_src;       /* [P][J][feat_len] */
  const float* points_ref;       /* [P][K][feat_len] */
  /* Optional caller-supplied pyramids (reference data dict keys
   * points_{src,ref}_{xyz,neigh_idx,sub_idx,interp_idx}, data_base.py:178-181),
   * already int32.  All NULL => built on device (dsir_knn_pyramid). */
  const float* src_xyz;   const int32_t* src_neigh;   const int32_t* src_sub;   const int32_t* src_interp;
  const float* ref_xyz;   const int32_t* ref_neigh;   const int32_t* ref_sub;   const int32_t* ref_interp;
  /* Optional teacher forcing (tests): correspondences per iteration [n_iter][P][J] i32.
   * Caller-supplied indices (these and the pyramids) are clamped into their valid range on device, so a bad index
   * can never fault a gather; the pair's invalid flag then carries bit 1 (value 2). */
  const int32_t* forced_idx;
} dsir_pair_batch;

typedef struct dsir_pair_result {
  float* transforms;     /* [P][n_iter][3][4] cumulative src->ref (model.py:595)        required */
  int32_t* idx;          /* [n_iter][P][J] arg-min correspondences (pred_pairs[...,1])   or NULL  */
  float* logits;         /* [n_iter][P][J] inlier logits (endpoints['perm_matrices'])    or NULL  */
  float* pt_ref_new;     /* [P][J][3] matched ref points of the last iteration           or NULL  */
  int32_t* invalid;      /* [P] OR over iterations: bit 0 = non-finite Kabsch covariance, identity returned
                          *     (endpoints['invalid_gradient'], model.py:61-64; also the outcome of a non-finite
                          *     input point: that pair alone, the other pairs of the batch are unaffected);
                          *     bit 1 = a caller-supplied index was out of range and clamped       or NULL  */
  /* test aids (parity of the arg-min on the engine's OWN descriptors, tests/test_gpu_large_configs.py): the aggregated,
   * L2-normalised descriptors Network.aggregation returns (model.py:552) as the search of each iteration saw them */
  float* desc_src;       /* [n_iter][P][J][64]                                                    or NULL  */
  float* desc_ref;       /* [P][K][64] (loop invariant)                                           or NULL  */
} dsir_pair_result;

/* Replaces Network.forward -> forward_align_4 (network/model.py:297-298,
 * :520-607) for P independent pairs in one call: KNN pyramids (if not
 * supplied), 2x feature RandLA + score, then n_iter x {aggregation, NN match,
 * inlier RandLA, weighted Kabsch, SE(3) update}. */
int dsir_register(dsir_ctx* ctx, const dsir_pair_batch* in, int n_iter, const dsir_pair_result* out);

/* ---- the `feat` / `label` pipelines of the same Network ---- */

typedef struct dsir_cloud_out {   /* one side (src or ref) of endpoints, point-major; M = num_sub > 0 ? num_sub : N */
  float* xyz;       /* [P][M][3]   endpoints['pt_*']                                                   or NULL */
  float* feat;      /* [P][M][64]  endpoints['feat_*'] (ALIGN ctx: raw feat0 of forward_pair's 8-tuple)  or NULL */
  float* logits;    /* [P][N][num_classes] endpoints['logits_*'] (always all N points)                 or NULL */
  float* score;     /* [P][M] endpoints['score_*'] (descending when num_sub > 0); not for LABEL        or NULL */
  int32_t* label;   /* [P][M] arg-max class of the (selected) points; not for LABEL                    or NULL */
  int32_t* index;   /* [P][M] selected point indices (only when num_sub > 0)                           or NULL */
} dsir_cloud_out;

/* Replaces Network.forward -> forward_pair (network/model.py:609-666) incl. feat_score's top-num_sub key-point
 * selection (:682-697, torch.topk; equal scores are taken in ascending index here) for P pairs:
 *   LABEL ctx: feat = normalize(feat_extractor features), logits.
 *   FEAT  ctx: score -> optional top-num_sub -> aggregation (:209-235) -> normalize (:650-651), logits, score.
 *   ALIGN ctx: forward_pair as forward_align_4 calls it (return_flag, :646-648): raw feat0, xyz, label, score;
 *              num_sub must be <= 0 (the inlier model needs the full pyramid, :575).
 * The KNN pyramids are built on device unless supplied in `in`. */
int dsir_forward_pair(dsir_ctx* ctx, const dsir_pair_batch* in, int num_sub, const dsir_cloud_out* src,
                      const dsir_cloud_out* ref);

/* ---- in front of the path: pre-processing ----------- */

/* Replaces the range/height crop of process_point_cloud (dataloader/data_base.py:299-312) and open3d's
 * voxel_down_sample as called at dataloader/threeDMatch_loader.py:168-175 / kitti_loader.py:335-338, for a
 * RAGGED batch: points [total][stride] (xyz first, every channel is voxel-averaged), offsets = HOST array
 * [clouds+1] of row offsets, crop = HOST [r_min, r_max, z_min, z_max] or NULL.
 * out [clouds][cap][stride] (voxels in ascending (ix,iy,iz) order; rows beyond cap are dropped),
 * counts [clouds] i32 on device = number of voxels of every cloud (may exceed cap: caller's overflow check).
 * The output order / arithmetic rule is this engine's own (open3d is unpinned): oracle/preprocess.py. */
int dsir_voxel_downsample(dsir_ctx* ctx, const float* points, const int64_t* offsets, int clouds, int stride,
                          float voxel_size, const float* crop, int cap, float* out, int32_t* counts);

/* Replaces Resampler._resample (mode 0: seeded random order, no repeats while points last, then draws with
 * replacement) and FixedResampler._resample (mode 1: tile / prefix) of dataloader/transformation.py:72-93.
 * in [clouds][cap][stride] with counts [clouds] (device, as written by dsir_voxel_downsample)
 * -> out [clouds][k][stride].  The random order makes the prefix sub-sampling of the pyramid a random sample. */
int dsir_resample(dsir_ctx* ctx, const float* in, const int32_t* counts, int clouds, int cap, int stride, int k,
                  int mode, uint64_t seed, float* out);

/* ---- after the path: evaluation metrics ------------- */

/* Replaces common/metrics_util.py:27-85 compute_metrics as called per iteration by test.py:308-355
 * evaluate_align.  pred_T: [pairs] transforms of 12 floats each, `pred_stride` floats apart (so one
 * iteration of a [P][n_iter][3][4] result can be addressed in place); gt_T [pairs][3][4];
 * points_src / points_ref [pairs][n][stride] (xyz first; the first min(n,2048) points are used).
 * out [pairs][8] float64: r_mse, r_mae, t_mse, t_mae, err_r_deg, err_t, succ (0/1), chamfer_dist. */
int dsir_eval_metrics(dsir_ctx* ctx, const float* pred_T, int64_t pred_stride, const float* gt_T,
                      const float* points_src, const float* points_ref, int pairs, int n, int stride,
                      float rte_thresh, float rre_thresh, double* out);

/* ---- after the path: pose refinement --------------- */

/* Replaces the `use_icp` branch of pose_optimization (test.py:241-258): open3d
 *   registration_icp(src, tgt, max_correspondence_distance, T_init, TransformationEstimationPointToPoint(),
 *                    ICPConvergenceCriteria(relative_fitness, relative_rmse, max_iteration))
 * for P pairs at once, entirely on device.  points_src [P][J][stride], points_ref [P][K][stride] (xyz first);
 * T_init / T_out [P][3][4]; stats [P][4] float64 {fitness, inlier_rmse, converged (0/1), iterations} or NULL.
 * The branch is disabled in the reference and open3d is not pinned: the rule (open3d's RegistrationICP loop, exact
 * brute-force nearest neighbours in fp32 with ties to the lower index) is restated in oracle/icp.py. */
int dsir_icp_refine(dsir_ctx* ctx, const float* points_src, const float* points_ref, int pairs, int J, int K, int stride,
                    float max_corr_dist, int max_iter, float rel_fitness, float rel_rmse, const float* T_init,
                    float* T_out, double* stats);

/* Replaces the `use_tune` branch of pose_optimization (test.py:209-239): transformation_finetune (test.py:159-207) with
 * HighDimSmoothL1Loss (test.py:103-131) over the network's last correspondences, the pose re-parametrised as
 * network/DGR.py's Transformation (6-D rotation + translation, :60-132) and fitted by Adam (lr 0.1, ExponentialLR 0.999)
 * until the loss is below 1e-7, max_iter steps are done, or the relative loss change was below break_threshold_ratio
 * max_break_count times - for P pairs at once, each pair's whole optimisation inside one workgroup.
 * xyz_src / xyz_ref [P][m][3] MATCHED points (endpoints['pt_src'] and 'pt_ref_new' = dsir_pair_result.pt_ref_new);
 * weights [P][m] or NULL (unweighted mean); weights_are_logits != 0: sigmoid applied on the fly (perm_matrices[-1]);
 * T_init / T_out [P][3][4]; stats [P][3] float64 {iterations, loss, break_count} (opt_result) or NULL.
 * The branch is disabled in the reference (use_tune = False) and test.py / DGR.py cannot be imported offline (open3d):
 * the rule is restated with the same torch calls in oracle/finetune.py (parity unpinned). */
int dsir_pose_finetune(dsir_ctx* ctx, const float* xyz_src, const float* xyz_ref, const float* weights, int weights_are_logits,
                       int pairs, int m, const float* T_init, float quantization_size, int max_iter, float break_threshold_ratio,
                       int max_break_count, float* T_out, double* stats);

/* ---- the training slice (backward half) ------- */

/* Replaces ScanAlignmentLoss.forward with reduction='mean' (network/loss.py:705-851; defaults of arguments.py:51-61:
 * loss_type mae, wt_ptDist_loss 1, wt_inlier_loss 1, wt_pose_loss 0, loss_discount_factor 0.5; called at train.py:401)
 * AND torch autograd's backward of it down to the inlier logits: through se3_torch.concatenate (model.py:595) and the
 * SVD of compute_rigid_transform_2 (model.py:22-66).  In forward_align_4 the matching runs under no_grad and the src
 * cloud is moved by R_t.detach(), so d total / d logits is the whole gradient the network receives from this loss: the
 * input of the inlier RandLA's backward pass (not built).
 * pt_src [P][J][3], pt_ref [P][K][3] (endpoints['pt_src'/'pt_ref']); idx [n_iter][P][J] i32 (pred_pairs[...,1]);
 * logits [n_iter][P][J] (perm_matrices); labels [n_iter][P][J] f32 0/1 = find_correct_correspondence(matches, pred_pairs)
 * (loss.py:723-749, host work in the reference too) or NULL = no confidence term; transform_gt [P][3][4].
 * loss_type 0 = mae, 1 = mse; wt_ptDist_loss only gates the point-distance term (> 0), as in the reference.
 * Outputs: transforms [P][n_iter][3][4] cumulative poses replayed from the logits (or NULL); losses = HOST float64
 * [n_iter][2] {mae_i | mse_i, outlier_i} (total = sum_i discount^(n_iter-1-i) (term_i + outlier_i)) or NULL;
 * grad_logits [n_iter][P][J] = d total / d logits.  n_iter <= 8.
 * dsir_align_loss_backward2: the same call with one more output, losses_per_pair = HOST float64 [P][n_iter][2] or NULL: every pair's own
 * terms as reduction='none' reports them (loss.py:779, :836: the mean over that pair's points / rows alone; validate_align, train.py:136). */
int dsir_align_loss_backward(dsir_ctx* ctx, const float* pt_src, const float* pt_ref, const int32_t* idx, const float* logits,
                             const float* labels, const float* transform_gt, int pairs, int J, int K, int n_iter, int loss_type,
                             float wt_ptDist_loss, float wt_inlier_loss, float loss_discount_factor, float* transforms,
                             double* losses, float* grad_logits);
int dsir_align_loss_backward2(dsir_ctx* ctx, const float* pt_src, const float* pt_ref, const int32_t* idx, const float* logits,
                              const float* labels, const float* transform_gt, int pairs, int J, int K, int n_iter, int loss_type,
                              float wt_ptDist_loss, float wt_inlier_loss, float loss_discount_factor, float* transforms,
                              double* losses, float* grad_logits, double* losses_per_pair);

/* Launch-bound small batches: capture the whole dsir_register launch sequence into a hipGraph once per
 * call signature (sizes and buffer addresses) and replay it; the context keeps the graphs of its 16 most recent signatures
 * (a server that coalesces 1 .. K single-pair requests per call replays K graphs in turn).  Off by default.
 * ROCm 7.x: the process must run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (in the environment before the HIP runtime
 * initialises): the runtime's graph packet capture replays a captured registration wrongly from its third launch on once
 * the host has waited between launches (tools/graph_replay_check.py).  The Python host sets it on import. */
int dsir_enable_graph(dsir_ctx* ctx, int enable);
/* Launch census of the registration captured LAST by this context (dsir_enable_graph): out (HOST, 4 x i64) = graph nodes in all,
 * kernel nodes, memset nodes, memcpy nodes - what one dsir_register call of that signature costs in launches.  Zeros before the
 * first capture. */
int dsir_graph_stats(dsir_ctx* ctx, int64_t* out);
/* A/B switch (measurement / test): the deep pyramid levels of RandLA.forward (RandLANet.py:215-230, :339-359; levels 2 / 3, mlp_mid,
 * the first two decoder blocks) as ONE persistent launch whose phases are the former launches' tiles (csrc/walk.hip), for launches of
 * up to 16 clouds.  OFF by default: it removes 114 of a single-pair registration's 305 launches but is slower than they are (3.46 ms
 * against 3.09 ms per registration on MI355X: a dependent launch costs ~1.5 us, a phase hand-off ~2 us, and the tile bodies' own
 * latency chains - what the time is made of - are the same).  Same kernels' code on the same operands, statistics in exact atomics:
 * same bits either way (tests/test_gpu_walk.py). */
int dsir_enable_walk(dsir_ctx* ctx, int enable);
/* A/B switch (measurement / test): launches of up to 16 clouds run the independent branches of the schedule - the KNN searches of the
 * four pyramid levels (data_base.py:153-183), a block's position-encoding branch and its mlp_skip (RandLANet.py:176-186, :229), the
 * loop-invariant halves of the aggregation (model.py:552) - on auxiliary streams beside the main chain, joined through events
 * (parallel branches of a captured registration).  OFF by default: measured slower on this runtime (a replayed graph with parallel
 * branches costs 0.3 - 1.0 ms more per single-pair registration than the linear chain, whatever the branches save).  Same kernels on
 * the same operands: same bits either way (tests/test_gpu_walk.py). */
int dsir_enable_fork(dsir_ctx* ctx, int enable);
/* Measurement: with DSIR_TUNING=1 DSIR_WALK_TRACE=1 in the environment at dsir_create the walker stamps, per program of a call
 * (12 slots) and phase (32), the device clock of cloud 0's earliest tile picked up, earliest tile past its wait, latest tile body
 * end and latest publish: out (HOST, 12 x 32 x 4 i64), clock_khz the clock's rate.  reset: re-arm the stamps.  Synchronises. */
int dsir_walk_trace(dsir_ctx* ctx, int reset, int64_t* out, int64_t* clock_khz);

/* Test/measurement hooks. */
/* Wall-clock-free kernel timing of the dominant kernel (nn_match) accumulated
 * with HIP events on the engine stream since the last reset: total ms and
 * launch count. */
int dsir_match_timer(dsir_ctx* ctx, int reset, double* total_ms, int64_t* launches);
/* The same launches with the operation's dominant kernel (screen_kernel of csrc/nn_screen.hip, or nn_match_kernel when
 * the exhaustive path runs) bracketed on its own: op_ms = every kernel of the operation, kernel_ms = that kernel. */
int dsir_match_timer2(dsir_ctx* ctx, int reset, double* op_ms, double* kernel_ms, int64_t* launches);
/* A/B switch (measurement): 0 = dsir_register always takes the exhaustive exact-fp32 arg-min kernel, 1 (default) = the
 * fp16-screened path for large problems.  Both return the same bits.  Initialised from DSIR_NO_SCREEN (behind the tuning gate, below). */
int dsir_enable_screen(dsir_ctx* ctx, int enable);
/* A/B switch (measurement / test): 1 (default) = the five wide layers of the aggregation chain (mlp_att 32 -> 64 -> 128 -> 256 ->
 * 64, mlp_proj; network/model.py:223-233) run as fp16-split products on the fp16 matrix pipe (csrc/agg_chain_h.hip: each fp32
 * operand = two fp16 numbers, three MFMAs per product, fp32 accumulation - fp32 accuracy, descriptors within ~1e-7 of the
 * fp32 kernel's); 0 = the exact-fp32 chain (csrc/agg_chain.hip), bit-identical to the unfused layer-by-layer launches.
 * Initialised from DSIR_AGG_F32 (set => 0).
 * What the switch covers: the aggregation chain, the per-point head (csrc/head_mlp_h.hip) and the LDS-tiled GEMMs (csrc/pw_tile.hip,
 * through GemmArgs::Wh staying unset).  What it does NOT cover: the attentive-pooling score contraction, which is an fp16-split
 * product unconditionally - csrc/att_pool.hip (levels 0 - 2) and the kSplit epilogues of csrc/pw_stream.hip (level 3 and the
 * DSIR_NO_ATT_POOL reference) have no exact-fp32 twin at run time; their fp32 reference is the oracle (tests/test_gpu_parity.py). */
int dsir_enable_agg_split(dsir_ctx* ctx, int enable);
/* The operand split those layers rest on, HOST buffers, no context, no GPU: hi[i] = fp16(x[i]) and lo[i] = fp16(x[i] - hi[i])
 * as IEEE binary16 bit patterns, both rounded to nearest even - what dsir_finalize_weights applies to every weight matrix (the
 * kernels apply the same rule to activations).  x = hi + lo + r with |r| <= max(2^-22 |x|, 2^-25) for |x| <= 65504. */
void dsir_split_f16(const float* x, int64_t n, uint16_t* hi, uint16_t* lo);
/* Same launches, bracketed on the DEVICE's constant-rate clock inside the kernel (first wave start .. last wave end,
 * the quantity a kernel trace reports): unlike the HIP-event bracket it does not include time the launch spends
 * queued behind other streams' kernels when several engines share the GPU. */
int dsir_match_timer_device(dsir_ctx* ctx, int reset, double* total_ms, int64_t* launches);
int dsir_enable_match_timer(dsir_ctx* ctx, int enable);
/* Which arg-min path dsir_register took since the last reset, and how selective the screening was.  out (HOST, 5 x i64):
 * [0] searches through the screened path (csrc/nn_screen.hip), [1] src rows they searched, [2] of those the rows the
 * screening left undecided (searched by the exhaustive exact-fp32 kernel), [3] pairs searched exhaustively as a whole,
 * [4] searches that took the exhaustive kernel directly (small problems, forced runs excluded).  Synchronises.
 * Not counted while a hipGraph replays (dsir_enable_graph): [4] is a host-side counter. */
int dsir_screen_stats(dsir_ctx* ctx, int reset, int64_t* out);
/* The pruned search of long ref ranges (csrc/nn_prune.hip; clouds of 8192 points and more in launches of 65536 src rows and more,
 * every registration iteration): out (HOST, 2 x i64) = (row block, column tile) products the screening visited, and the number it
 * would have visited without pruning, since the last reset.  Synchronises. */
int dsir_prune_stats(dsir_ctx* ctx, int reset, int64_t* out);
/* A/B switch (measurement / test): the pruned search runs for ref clouds of min_points points and more (default 8192, initialised
 * from DSIR_PRUNE_MIN_K; 0 = never) in launches of min_rows src rows (pairs x points) and more (default 65536: below that the
 * launch does not fill the chip with work items and the preparation is pure cost).  Pruned and unpruned searches return the
 * same bits: only products that cannot hold a row's arg-min - nor tie with it - are skipped. */
int dsir_set_prune_thresholds(dsir_ctx* ctx, int min_points, int64_t min_rows);
/* A/B switch (test): clouds of `min_points` points and more solve their pose in chunks over several workgroups per pair
 * (csrc/kabsch.hip; default 16384, <= 0 restores it), in dsir_register and dsir_kabsch.  Same formulas either way; the fp64
 * partial sums are taken in another order. */
int dsir_set_kabsch_chunked_min(dsir_ctx* ctx, int min_points);

/* The tuning gate.  The library has a number of measurement / A-B switches named DSIR_* (kernel-family selection, tile
 * geometry, thresholds).  They are environment variables, read in ONE function (csrc/engine.hip,
 * tuning_env) and ONLY while the gate is open: DSIR_TUNING=1 in the environment, or dsir_set_tuning(1) called before the
 * first library call that reads a switch (most are read once per process).  With the gate closed - the default - every
 * DSIR_* variable is ignored, so a stray one in a user's environment cannot change what runs.  dsir_tuning() reports
 * the gate's state.  Every variant behind a switch is the HIP path; none is a fallback. */
void dsir_set_tuning(int on);
int dsir_tuning(void);

/* GroupNorm (RandLANet.py:90-107) statistics of a layer meet across workgroups in exact, order-independent atomics
 * (csrc/device_utils.h, gn_block_commit); the proof needs at most dsir_gn_contribution_limit() contributions per (cloud, group)
 * statistic.  dsir_gn_contributions: the most any statistic receives for a cloud of n_points points under this configuration (host
 * arithmetic over the launchers' grid rules; -1: bad arguments); dsir_max_points_limit: the largest max_points dsir_create accepts
 * for it.  No GPU needed. */
int dsir_gn_contributions(const dsir_cfg* cfg, int n_points);
int dsir_gn_contribution_limit(void);
int dsir_max_points_limit(const dsir_cfg* cfg);

/* Diagnostics of the fp16 screening (csrc/nn_screen.hip) on ONE pair, all pointers DEVICE memory: for every (row, column)
 * the screening's lower bound L, its upper bound U = L + 2 d and the exact fp32 distance D of dsir_nn_match
 * (lower / upper / exact [J][K]; zacc [J][K], optional: the raw fp32 accumulator 2^22 (c + a.b) the six chained
 * v_mfma_f32_16x16x32_f16 leave, for measuring the matrix core's accumulation error against an fp64 sum of the same
 * fp16 products) - computed by the same MFMA chain on the same fp16 operands as the product kernel -,
 * plus what the product path did with the same inputs: idx [J] (its arg-min), thresh [J] (the row's final threshold),
 * cand_count [J] (entries emitted; > dsir_screen_cap() = the list overflowed), cand_code / cand_lower
 * [J][dsir_screen_cap()] (code >= 0: a column and the bits of its L; code < 0: a lane class), out_of_domain [1]
 * (non-zero: a component was outside the bound's domain |x| <= 16 - the product path would not screen such input).
 * The screened search is exact iff L <= D <= U for every entry; tests/test_gpu_screen_bound.py asserts it on
 * adversarial inputs.  J * K <= 2^26. */
int dsir_screen_bounds(dsir_ctx* ctx, const float* desc_src, const float* desc_ref, int J, int K, float* lower, float* upper,
                       float* exact, float* zacc, int32_t* idx, float* thresh, int32_t* cand_count, int32_t* cand_code, float* cand_lower,
                       int32_t* out_of_domain);
int dsir_screen_cap(void);

#ifdef __cplusplus
}
#endif
#endif /* DSIR_H */
