/*
 * dsir_train.h — C ABI of the training operators (libdsir.so), SURVEY.md section 8(f) rank 4: the backward half.
 *
 * The reference trains with torch autograd (train.py:396-448): every operator below is the forward or the backward of
 * one ATen call that RandLA.forward (network/RandLANet.py:311-372) makes, so that the inlier model's training step
 * (forward with saved activations, backward from d loss / d logits - dsir_align_loss_backward - to every parameter,
 * Adam) runs on the device without autograd.  deepsir_amd/train.py strings them together in the reference's module
 * order; this header is what a non-Python host would bind.
 *
 * Conventions: device pointers, fp32, POINT-MAJOR rows ([rows][channels], leading dimension `ld` in floats), int32
 * indices; `stream` is a hipStream_t (NULL = default stream); calls are asynchronous; return 0 or a hipError_t.
 * Reductions are deterministic (fixed partition, fixed order) except the two scatter-adds, which use fp32 atomics
 * exactly like ATen's index/gather backward on a GPU.
 */
#ifndef DSIR_TRAIN_H
#define DSIR_TRAIN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* nn.Conv2d 1x1 / nn.Conv1d / nn.Linear (RandLANet.py:77-88, :148, :270, :41): Y[r][n] = beta Y[r][n] + bias[n] +
 * sum_k X[r][k] W[n wn + k wk].  Forward: W = weight [Cout][Cin], wn = Cin, wk = 1.  Backward w.r.t. the input
 * (dX = dY W): X = dY, wn = 1, wk = Cin, N = Cin, K = Cout; beta = 1 accumulates into an existing gradient.
 * Exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), k ascending. */
int dsir_t_gemm(void* stream, const float* X, int ldx, const float* W, int wn, int wk, const float* bias, float* Y, int ldy,
                int64_t rows, int K, int N, float beta);

/* Backward of the same call w.r.t. weight and bias: dW[n][k] += sum_r dY[r][n] X[r][k], db[n] += sum_r dY[r][n]
 * (db may be NULL).  scratch: dsir_t_gemm_dw_scratch(rows, N, K) bytes. */
size_t dsir_t_gemm_dw_scratch(int64_t rows, int N, int K);
int dsir_t_gemm_dw(void* stream, const float* dY, int ldy, const float* X, int ldx, int64_t rows, int N, int K, float* dW,
                   float* db, void* scratch);

/* nn.GroupNorm(groups, C) (+ LeakyReLU 0.2 when act) over one cloud's [M][C] block (RandLANet.py:90-107), also
 * nn.BatchNorm1d in training mode (clouds = 1, M = all rows, groups = C; RandLANet.py:44): biased variance, eps 1e-5.
 * stats [clouds][groups][2] = {mean, rstd} are kept for the backward.  scratch: dsir_t_gn_scratch(clouds, M, C) bytes
 * (partial sums of up to 64 row chunks per cloud, reduced in chunk order). */
size_t dsir_t_gn_scratch(int clouds, int M, int C);
int dsir_t_gn_fwd(void* stream, const float* Y, int clouds, int M, int C, int groups, const float* gamma, const float* beta,
                  int act, float* out, float* stats, void* scratch);
/* dY (may alias dOut) = d loss / d Y; dgamma / dbeta accumulate. */
int dsir_t_gn_bwd(void* stream, const float* dOut, const float* Y, const float* stats, int clouds, int M, int C, int groups,
                  const float* gamma, const float* beta, int act, float* dY, float* dgamma, float* dbeta, void* scratch);

/* The running statistics nn.BatchNorm1d keeps in training mode (momentum 0.1, unbiased variance); stats = dsir_t_gn_fwd's
 * with groups = C over M rows. */
int dsir_t_bn_running(void* stream, const float* stats, int C, int64_t M, float momentum, float* running_mean, float* running_var);

/* gather_neighbour_V2 / nearest_interpolation (network/tools.py:197-221, RandLANet.py:393-408):
 * Y[cloud][j][col_off + c] = X[cloud][idx[cloud][j]][c], j < m, c < C; X [clouds][n][C]; backward = scatter-add. */
int dsir_t_gather(void* stream, const float* X, int n, int C, const int32_t* idx, int m, int clouds, float* Y, int ldy, int col_off);
/* The backward of a gather is a sum over the SOURCES of every destination row.  It is taken in a fixed order - ascending source row -
 * through the inverse of the index (a "plan": one stable sort of (destination, source) pairs per index tensor, re-used by every
 * operator that shares the index), so a gradient has the same bits on every run (up to round 4: float atomics, 2e-6 of its scale
 * between runs).  dsir_t_scatter_plan: idx [clouds][m] into [clouds][n] -> order [clouds * m] i32 (sources grouped by destination),
 * offsets [clouds * n + 1] i32; scratch: dsir_t_scatter_plan_scratch(clouds * m) bytes.
 * dsir_t_scatter_add: dX[cloud][i][c] = sum over the sources j of row i of dY[cloud m + j][col_off + c]  (overwrites dX). */
size_t dsir_t_scatter_plan_scratch(int64_t total);
int dsir_t_scatter_plan(void* stream, const int32_t* idx, int m, int clouds, int n, int32_t* order, int32_t* offsets, void* scratch);
int dsir_t_scatter_add(void* stream, const float* dY, int ldy, int col_off, const int32_t* order, const int32_t* offsets, int clouds,
                       float* dX, int n, int C);

/* Building_block.relative_pos_encoding (RandLANet.py:197-212): out[cloud][i k + j][10] = {|pj - pi|, pj - pi, pi, pj};
 * xyz [clouds][n][3], idx [clouds][n][k].  No backward: the coordinates are data. */
int dsir_t_relpos(void* stream, const float* xyz, const int32_t* idx, int n, int k, int clouds, float* out);

/* The inlier model's input of one registration iteration (network/model.py:571-573; the src cloud moved by the previous
 * cumulative pose, :587 with R_t.detach()): out[pair][j] = {T x_src[j], x_ref[idx[pair][j]]}; xyz_src [P][J][3], xyz_ref
 * [P][K][3], T = NULL (iteration 0) or the pose of pair p at T + p t_stride ([3][4] row-major), out [P][J][6]. */
int dsir_t_inlier_input(void* stream, const float* xyz_src, const float* xyz_ref, const int32_t* idx, const float* T, int t_stride,
                        int pairs, int J, int K, float* out);

/* Att_pooling (RandLANet.py:148-155) after its fc: S [points][k][C] scores in, softmax over k written back in place (kept
 * for the backward), out[point][c] = sum_k cat[point][k][c] S[point][k][c].  Backward: dCat = direct part (the caller adds
 * dS W_fc with dsir_t_gemm beta = 1), dS = d loss / d scores. */
int dsir_t_attpool_fwd(void* stream, const float* cat, float* S, int64_t points, int k, int C, float* out);
int dsir_t_attpool_bwd(void* stream, const float* dOut, const float* cat, const float* A, int64_t points, int k, int C, float* dCat,
                       float* dS);

/* RandLA.random_sample (RandLANet.py:374-391): out[cloud][j][c] = max_t X[cloud][pool[cloud][j][t]][c]; arg = the row
 * that won (first of equals); backward adds dOut to that row, in ascending order of the outputs (no float atomics). */
int dsir_t_maxpool_fwd(void* stream, const float* X, int n, int C, const int32_t* pool, int m, int k, int clouds, float* out,
                       int32_t* arg);
/* backward: order / offsets = the plan of the pool index viewed as [clouds][m k] (dsir_t_scatter_plan); overwrites dX [clouds][n][C] */
int dsir_t_maxpool_bwd(void* stream, const float* dOut, const int32_t* arg, const int32_t* order, const int32_t* offsets, int m, int k, int C,
                       int clouds, float* dX, int n);

/* SemanticLoss.compute_loss (network/loss.py:930-960, :919-928): F.cross_entropy(weight = class weights, reduction 'mean')
 * over the points whose label is not 0 ("unlabeled"), class = label - 1; logits [rows][C] point-major, labels [rows] in 0..C.
 * out (device, 4 doubles) = {loss = sum w_y nll / sum w_y, sum w_y, correct arg-max predictions, valid rows};
 * dlogits [rows][C] = grad_scale * d loss / d logits (ignored rows 0).  scratch: dsir_t_weighted_ce_scratch(rows) bytes.
 * The reference passes the weights as a [1, C] tensor, which F.cross_entropy of the torch in this image rejects; the rule
 * here is the documented one for a [C] weight vector (oracle/train.py restates it; parity unpinned). */
size_t dsir_t_weighted_ce_scratch(int64_t rows);
int dsir_t_weighted_ce(void* stream, const float* logits, const int32_t* labels, const float* class_weights, int64_t rows, int C,
                       float grad_scale, float* dlogits, double* out, void* scratch);

/* F.normalize(x, p = 2, dim = channels) (model.py:232-233, :651-652) and its backward; norms [rows] = max(|x|, 1e-12). */
int dsir_t_l2norm_fwd(void* stream, const float* x, int64_t rows, int C, float* y, float* norms);
int dsir_t_l2norm_bwd(void* stream, const float* dy, const float* y, const float* norms, int64_t rows, int C, float* dx);

/* DetDesLoss.forward (network/loss.py:667-702) = CircleLoss.forward(feat_ref, feat_src, pt_ref, T_gt pt_src, score_ref, .)
 * (:500-571) with chamfer_loss_weight 0 (arguments.py:46), AND its backward w.r.t. both descriptor sets - what the `feat`
 * pipeline's loss.backward() hands the aggregation MLPs (the feature extractor is frozen there, model.py:136).
 * feat_* [P][M][C] point-major, pt_* [P][M][3], score_ref [P][M], transform_gt [P][3][4]; N1 = N2 = M as the reference's
 * element-wise sum of row and column terms requires.  out (device, 4 doubles) = {total, loss_feat, loss_det, accuracy}.
 * The reference's arithmetic is kept as it executes (see oracle/train.py::det_des_loss, pinned by its autograd). */
size_t dsir_t_det_des_loss_scratch(int pairs, int M);
int dsir_t_det_des_loss(void* stream, const float* feat_ref, const float* feat_src, const float* pt_ref, const float* pt_src,
                        const float* score_ref, const float* transform_gt, int pairs, int M, int C, float thres_radius, float det_loss_weight,
                        double* out, float* d_feat_ref, float* d_feat_src, void* scratch);

/* F.leaky_relu(a + b, 0.2) (RandLANet.py:230) and its backward (d a = d b = dOut * slope(out)). */
int dsir_t_add_leaky_fwd(void* stream, const float* a, const float* b, int64_t n, float* out);
int dsir_t_add_leaky_bwd(void* stream, const float* dOut, const float* out, int64_t n, float* d);

/* nn.Dropout (RandLANet.py:366) with the mask supplied: y = x * mask * scale (forward and backward alike). */
int dsir_t_mul_mask(void* stream, const float* x, const uint8_t* mask, float scale, int64_t n, float* y);
/* torch.topk(score, k) of feat_score (model.py:692): the k best key points per cloud, descending score, equal scores in
 * ascending index (csrc/select.hip, the operator dsir_forward_pair uses); idx [clouds][k], score_out [clouds][k]. */
size_t dsir_t_topk_scratch(int clouds, int n);
int dsir_t_topk(void* stream, const float* score, int clouds, int n, int k, int32_t* idx, float* score_out, void* scratch);

/* logit.sigmoid() (model.py:577): the correspondence weights of the weighted Kabsch step */
int dsir_t_sigmoid(void* stream, const float* x, int64_t n, float* y);
/* y += a x */
int dsir_t_axpy(void* stream, float a, const float* x, int64_t n, float* y);

/* "Check if any of the gradients is NaN" (train.py:437-441): flag[0] (device int32) = 1 if any of the n floats is NaN, else 0. */
int dsir_t_any_nan(void* stream, const float* x, int64_t n, int32_t* flag);

/* torch.optim.Adam.step (train.py:323, :446; betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad):
 * m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps). */
int dsir_t_adam(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int step);

#ifdef __cplusplus
}
#endif
#endif
