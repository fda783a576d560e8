/*
 * creid.h -- C ABI of libcreid_hip.so: the MI355X (gfx950) implementation of the
 * centroids-reid embedding-and-retrieval hot path.
 *
 * The upstream project (mikwieczorek/centroids-reid) is pure Python and has no FFI layer;
 * its boundary for this path is the Python object surface listed in SURVEY.md section 8b.
 * Each entry point below names the reference function (file:line, relative to the
 * upstream repository root) whose arithmetic it replaces; the Python mirror of that
 * surface (package `centroids-reid_amd`) binds these symbols with ctypes
 * (see INTEGRATION.md for the binding a maintainer would add upstream).
 *
 * Conventions (all functions):
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch owns all memory);
 *     the library allocates nothing; scratch comes in through `ws` + `*_workspace_bytes()`;
 *   - sizes / leading dimensions are int64_t element counts, row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - returns 0 on success, <0 for an argument error (CREID_E_*), >0 = hipError_t of the
 *     launch; never throws, never synchronises the stream, re-entrant.  Process-global state is limited to immutable
 *     kernel handles / zero pages and the optional launch-plan registry (creid_tune_set / creid_tune_clear below: host-side,
 *     written before the first launch, read-only afterwards).
 */
#ifndef CREID_H
#define CREID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CREID_ABI_VERSION 1

enum { CREID_F32 = 0, CREID_BF16 = 1, CREID_F16 = 2 };
enum { CREID_E_ARG = -1, CREID_E_DTYPE = -2, CREID_E_WS = -3, CREID_E_SHAPE = -4 };

int creid_abi_version(void);

/* ------------------------------------------------------------------ stage D: distance */

/* utils/reid_metric.py:113-115 (F.normalize p=2 dim=1): y[r,:] = x[r,:] / max(||x[r]||, eps).
 * x fp32 [rows, D]; y has dtype `out_dtype` (CREID_F32 / CREID_BF16 / CREID_F16);
 * sqnorm (nullable) receives sum_k y[r,k]^2 accumulated in fp32 from the ROUNDED y. D % 4 == 0. */
int creid_l2norm_rows(const float* x, void* y, float* sqnorm, int64_t rows, int64_t D,
                      int out_dtype, float eps, void* stream);

/* torch.pow(x,2).sum(dim=1) of utils/reid_metric.py:28-30 / losses/triplet_loss.py:35-36. */
int creid_row_sqnorm(const void* x, float* out, int64_t rows, int64_t D, int dtype, void* stream);

/* utils/reid_metric.py:25-33 get_euclidean: out[i,j] = (qq[i] + gg[j]) - 2 * <q_i, g_j>
 * (squared L2, no clamp, no sqrt).  q [m,D], g [n,D] of `dtype`; qq [m], gg [n] fp32 row
 * square-norms; out fp32 [m, ldo].  fp32 inputs run on v_mfma_f32_32x32x2_f32 (exact f32
 * FMA chain); bf16/f16 inputs on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulate. */
int creid_sqdist_matrix(const void* q, const void* g, const float* qq, const float* gg,
                        int64_t m, int64_t n, int64_t D, int dtype, float* out, int64_t ldo,
                        void* stream);

/* ------------------------------------------------------------------ stage D/E: rank */

/* utils/reid_metric.py:129,132 np.argsort(distmat, axis=1): per-row ascending order of
 * (distance, gallery index) -- ties broken by index (the reference leaves ties undefined).
 * dist fp32 [m, ld]; out_idx int64 [m, n]. */
size_t creid_rank_rows_workspace_bytes(int64_t m, int64_t n);
int creid_rank_rows(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx,
                    void* ws, size_t ws_bytes, void* stream);
/* np.argsort + eval_func's per-query loop (utils/reid_metric.py:129-132 + utils/eval_reid.py:36-90, plain camera ids) in ONE
 * pass: the ranked row is evaluated while it is still in LDS, so the int64 index matrix (8 B per pair) is written for the
 * caller but never read back; rows the one-pass rank kernel cannot take are evaluated by creid_cmc_ap_ranked's kernel.
 * out_idx / ws as creid_rank_rows, out_valid / out_ap / out_first as creid_cmc_ap_ranked; results identical to the two calls. */
int creid_rank_rows_eval(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx, void* ws, size_t ws_bytes,
                         const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_camids, const int64_t* g_camids,
                         uint8_t* out_valid, double* out_ap, int32_t* out_first, void* stream);

/* ------------------------------------------------------------------ stage E: CMC / mAP */

/* utils/eval_reid.py:44-84 (the per-query loop of eval_func, respect_camids=False), one
 * workgroup per query over the ranked row: drop gallery entries with the query's pid AND
 * camid (:57); valid = any match left (:63-65); ap = sum_k [match_k] * cum_k / (k+1) / n_rel
 * in float64 (:75-79); first = 0-based kept-rank of the first match (defines the clipped
 * CMC row :67-70 and the top-k flags :18-22).  idx int64 [m, n]; pids/camids int64. */
int creid_cmc_ap_ranked(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids,
                        const int64_t* g_pids, const int64_t* q_camids, const int64_t* g_camids,
                        uint8_t* out_valid, double* out_ap, int32_t* out_first, void* stream);

/* The same scan for camera-set gallery entries (respect_camids=True, utils/eval_reid.py:51-55, fed by
 * modelling/bases.py:205-236): g_cam_masks[j] = bitmask of the camera ids (< 64) of gallery entry j; an entry
 * is dropped for a query iff it has the query's pid AND bit q_camids[i] is set in its mask. */
int creid_cmc_ap_ranked_camsets(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids,
                                const int64_t* g_pids, const int64_t* q_camids, const int64_t* g_cam_masks,
                                uint8_t* out_valid, double* out_ap, int32_t* out_first, void* stream);

/* The k smallest (distance, index) pairs of every row of dist fp32 [m, ld >= n], ascending -- what
 * inference/get_similar.py:114-119 keeps of its full argsort (`indices[:, :topk]`) -- without ranking the rest.
 * k <= min(n, 1024).  out_idx int64 [m][k]; out_dist fp32 [m][k] (nullable) the corresponding distances; ties resolve by
 * index like creid_rank_rows.  flags uint8 [m]: 1 = the row's candidate set did not fit (massive ties at the k-th
 * distance): that row's outputs are unwritten, use creid_rank_rows for it. */
int creid_topk_rows(const float* dist, int64_t m, int64_t n, int64_t ld, int32_t k, int64_t* out_idx,
                    float* out_dist, uint8_t* flags, void* stream);

/* utils/eval_reid.py:86-90: means over valid queries.  out_cmc float32[max_rank]
 * (= count(first<=r)/n_valid in float32), out_map float64[1], out_topk float64[5] for
 * k in {1,5,10,20,50}, out_nvalid int64[1]. */
int creid_eval_reduce(const uint8_t* valid, const double* ap, const int32_t* first, int64_t m,
                      int32_t max_rank, float* out_cmc, double* out_map, double* out_topk,
                      int64_t* out_nvalid, void* stream);


/* ---- metric-only evaluation with NO m x n matrix (csrc/stream_eval.hip): what utils/reid_metric.py:93-151 +
 * utils/eval_reid.py:25-92 compute, for plain camera ids and the squared-L2 distance.  Three launches:
 *  poslist : per query, distances to its positives (gallery rows of the same pid, other camera) in the arithmetic
 *            of creid_sqdist_matrix, sorted by (distance, gallery index).  The gallery is given grouped by pid:
 *            g_order int32 [n] (gallery indices grouped by pid; any order inside a group -- the kernel re-ranks its
 *            candidates by (distance, gallery index)), csr_off int64 [n_groups + 1], q_slot int32 [m] (the query
 *            pid's group, -1 if absent) -- built on the device by creid_stream_plan.  cap = list capacity per query
 *            (power of two, 2..128); pos_key uint32 [m][cap] (order-preserving image of the fp32 distance, padded
 *            with 0xffffffff), pos_idx int32 [m][cap], npos int32 [m] (-1: more than cap positives -- such queries
 *            must take creid_sqdist_matrix + creid_rank_rows + creid_cmc_ap_ranked instead).
 *  count   : the full contraction, tile by tile on the f32 MFMA pipe; every negative (other pid) bumps
 *            hist[q][#positives ranked before it] -- hist uint32 [m][cap] must be ZERO on entry.
 *  finalize: valid uint8 [m] (0: no positive, 1: ok, 2: overflow), ap float64 [m], first int32 [m] with the
 *            meaning of creid_cmc_ap_ranked; feed creid_eval_reduce. */
/* The index of the three launches above, built on the device from the raw label vectors (the per-query `matches` /
 * `remove` bookkeeping of utils/eval_reid.py:36-65 done once for all queries): counting sort of the gallery by pid over the
 * dense range [pmin, pmin + R) (R = max pid - min pid + 1 of the gallery; group = pid - pmin), then per query its group,
 * its number of positives n_pos (same pid, other camera) and stats int32[2] = {max n_pos over queries with n_pos <= 128,
 * number of queries with more}.  csr_off int64[R + 1], g_order int32[n], q_slot / n_pos int32[m], scratch int32[2 R].
 * No host synchronisation; the caller reads `stats` (8 bytes) to choose cap. */
int creid_stream_plan(const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_cams, const int64_t* g_cams,
                      int64_t m, int64_t n, int64_t pmin, int64_t R, int64_t* csr_off, int32_t* g_order,
                      int32_t* q_slot, int32_t* n_pos, int32_t* stats, int32_t* scratch, void* stream);
int creid_stream_poslist(const float* q, const float* g, const float* qq, const float* gg, int64_t m, int64_t n,
                         int64_t D, const int32_t* q_slot, const int64_t* csr_off, const int32_t* g_order,
                         const int64_t* q_cams, const int64_t* g_cams, int32_t cap, uint32_t* pos_key,
                         int32_t* pos_idx, int32_t* npos, void* stream);
int creid_stream_count(const float* q, const float* g, const float* qq, const float* gg, int64_t m, int64_t n,
                       int64_t D, const int64_t* q_pids, const int64_t* g_pids, int32_t cap,
                       const uint32_t* pos_key, const int32_t* pos_idx, const int32_t* npos, uint32_t* hist,
                       void* stream);
int creid_stream_finalize(const int32_t* npos, const uint32_t* hist, int64_t m, int32_t cap, uint8_t* valid,
                          double* ap, int32_t* first, void* stream);

/* Measured launch plans (optional).  kind 0 = weight gradient: key (M = batch*out_h*out_w, out_c, K = kh*kw*in_c, 0) ->
 * (tile rows 64|128, tile cols 64|128, pixel splits | ring depth << 16 | producer/consumer waves << 20 | two k-groups << 21);
 * kind 1 = implicit-GEMM forward / data gradient: key (GEMM rows M, GEMM cols N, K, transposed 0|1 | stride << 1) ->
 * (N tile 64|128, LDS ring depth 2|3|4, kernel 0 producer/consumer | 1 four-wave DMA | 2 persistent 1x1 | 3 256-row tiles |
 * 4 persistent 1x1, second form: forward only, K = 64|128|256, the ring-depth slot caps its column slab: 2 widest, 3 128, 4 64 |
 * 5 all-waves-multiply persistent kernel (conv_pipe.hip): forward only, K % 64 = 0, the first slot is then the variant word
 * tile rows/128 | (tile cols/128) << 2 | k-tiles per phase << 4 | loop form << 8 (0 ping-pong wave groups, 2 free-running),
 * second slot 0).  Key bit 3 of the last component (| 8) marks a plan that applies to the eval-mode epilogue only (folded affine /
 * residual / ReLU): such launches look it up first and fall back to the key without the bit.  A plan only selects among kernel variants
 * the library already has; shapes without an entry use the built-in rules.  Not thread-safe against running launches:
 * register before the first convolution (the Python binding does it at load time from tuned_plans.json). */
int creid_tune_set(int32_t kind, int64_t a, int64_t b, int64_t c, int64_t d, int32_t p0, int32_t p1, int32_t p2);
int creid_tune_clear(void);

/* ------------------------------------------------------------------ stage B: centroids */

/* train_ctl_model.py:79-104: leave-one-out per-PID centroids of a PID-contiguous [P,K] batch.
 * centroids[i,p,:] = sum_{s != i, is_real[p,s]} feat[p*K+s,:] / max(cnt,1) if is_real[p,i] else 0;
 * valid[i,p] = cnt.  feat fp32 [P*K, D]; is_real uint8 [P*K]; centroids fp32 [K,P,D]; valid int32 [K,P]. */
int creid_loo_centroids_fwd(const float* feat, const uint8_t* is_real, int64_t P, int64_t K, int64_t D,
                            float* centroids, int32_t* valid, void* stream);
/* adjoint of the above: dfeat_accum[P*K, D] += J^T dcentroids. */
int creid_loo_centroids_bwd(const float* dcentroids, const uint8_t* is_real, int64_t P, int64_t K,
                            int64_t D, float* dfeat_accum, void* stream);
/* The centroid rounds' operands and scalars without framework glue (train_ctl_model.py:79-124, 143-177):
 * creid_loo_emb_fwd = creid_loo_centroids_fwd that ALSO lays out the K stacked triplet problems: emb [K][2P][D] (round i:
 * the P i-th instances, then the P leave-one-out centroids), lab int64 [K][2P]; creid_loo_emb_bwd adds demb's query rows and
 * the leave-one-out backward of its centroid rows into dfeat [P*K][D]; cnorm [K][P] = L2 norm of every centroid row;
 * creid_ctl_step_stats: out[0..n) = scal * weights, out[n..n+7) = {sum, sum of the K round losses (slots 4, 8, ..), mean over
 * the K rows of scal[4..4K+4) (4 values), mean of the `rows` entries of cnorm}. */
int creid_loo_emb_fwd(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                      float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, void* stream);
int creid_loo_emb_bwd(const float* demb, const uint8_t* is_real, int64_t P, int64_t K, int64_t D, float* dfeat_accum,
                      void* stream);
int creid_ctl_step_stats(const float* scal, const float* weights, int64_t n, int64_t K, const float* cnorm,
                         int64_t rows, float* out, void* stream);

/* ---- the whole head section of a training step in SIX launches (train_ctl_model.py:59-152: everything between the backbone
 * forward and its backward).  The kernels above are latency-bound on a 64 x 2048 feature matrix and four of their chains are
 * independent (query triplet / center loss / BNNeck -> classifier -> cross entropy / centroid rounds): each launch here runs one
 * step of every chain side by side (workgroup roles), in the accumulation order of the sequential calls -- results are
 * bit-identical to them (tests/test_heads_fused_gpu.py).  The last launch also performs creid_gap_bwd of the finished feature
 * gradient, i.e. it hands the backbone its incoming gradient directly.
 *   masked = 0: every row is real (is_real must still be given: all ones); masked = 1: the device-mask schedule below
 *   (creid_*_masked / *_rows semantics, `lonely` required).
 *   loss_weights [4(K+1)+2] / stats [4(K+1)+2+7]: as creid_ctl_step_stats(_rows).  w_centroid: the centroid-triplet weight
 *   (divided by K inside for masked = 0, by the number of valid rounds on the device for masked = 1).
 *   split_logits / split_dbnf: K-splits of the classifier forward / data-gradient GEMMs (1 = single pass, run-to-run identical;
 *   > 1: partial products combined by fp32 atomics like creid_gemm_f32).
 *   d_* are ACCUMULATED into (NULL = not wanted, except d_bn_weight); bn_batches_tracked (nullable) is incremented by one;
 *   amp_state (nullable): f16 loss scale {scale, 1/scale} -- g is multiplied by scale; dfeat_out (nullable, fp32 [B, D]) receives
 *   the unscaled feature gradient; g: [B * HW, D] in g_dtype.  workspace: creid_ctl_heads_workspace_bytes(), 256-byte aligned.
 *   Limits: K <= 16, B <= 256, D % 8 == 0 (else CREID_E_SHAPE: use the separate calls). */
typedef struct creid_ctl_heads {
  int64_t B, P, K, D, num_classes, num_centers, HW;
  int32_t g_dtype, masked, split_logits, split_dbnf;
  float margin, xent_eps, w_query, w_center, w_xent, w_centroid, bn_momentum, bn_eps;
  const float* feat;
  const int64_t* labels;
  const uint8_t* is_real;
  const float* centers;
  const float* bn_weight;
  const float* bn_bias;
  float* bn_running_mean;
  float* bn_running_var;
  const float* fc_weight;
  const float* loss_weights;
  const float* amp_state;
  float* d_centers;
  float* d_bn_weight;
  float* d_bn_bias;
  float* d_fc_weight;
  int64_t* bn_batches_tracked;
  int32_t* lonely;
  float* stats;
  void* g;
  float* dfeat_out;
  void* workspace;
  size_t workspace_bytes;
  /* optional (all five or none; 16-bit g_dtype, D % 256 == 0): the column sums of the BatchNorm backward that consumes g first
   * (the last bottleneck's bn3, modelling/backbones/resnet.py:81-83) are produced by the launch that writes g --
   * bn_partial [creid_bn2d_bwd_rows(B * HW)][2][D], bit-identical to creid_bn2d_bwd's own column pass (partial_ready = 1):
   * bn_x = that layer's raw convolution output [B * HW, D], bn_mask = the ReLU bits of its output, bn_mean / bn_invstd [D]. */
  const void* bn_x;
  const uint8_t* bn_mask;
  const float* bn_mean;
  const float* bn_invstd;
  float* bn_partial;
} creid_ctl_heads;
size_t creid_ctl_heads_workspace_bytes(int64_t B, int64_t P, int64_t K, int64_t D, int64_t num_classes);
int creid_ctl_heads_fused(const creid_ctl_heads* a, void* stream);

/* ---- the same training step for batches with padded samples (isReal = False, datasets/bases.py:346-406), driven by a DEVICE
 * mask so that the step stays free of host synchronisation (hipGraph-capturable for any pattern of fakes):
 *  creid_loo_emb_fwd_rows    = creid_loo_emb_fwd + row_exists uint8 [K][2P]: identity p takes part in round i iff its i-th
 *                              instance is real and it has another real instance (train_ctl_model.py:112-122 keeps these rows);
 *  creid_triplet_fwd_batched_rows: rows with row_exists = 0 are neither anchors nor candidates; a problem with fewer than
 *                              min_rows rows is skipped (out4 = 0, coef = 0: the round with <= 1 valid identity, :113);
 *  creid_ctl_round_scale     : inv_rounds[0] = 1 / #rounds with out4[k][3] >= 4 (the mean over valid rounds, :143-146), the
 *                              gscale_dev of the rounds' creid_triplet_bwd_batched;
 *  creid_ctl_step_stats_rows : creid_ctl_step_stats with the round terms / means over the VALID rounds only and the centroid
 *                              norm averaged per round over its kept rows; weights[4k] must hold the full centroid weight;
 *  *_masked                  : center loss, label-smoothed cross entropy and the BNNeck over the rows with row_mask != 0 only
 *                              (means / statistics over their count), zero gradient and zero output on the others
 *                              (train_ctl_model.py:69-77 feeds features[isReal] to these three). */
int creid_loo_emb_fwd_rows(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                           float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, uint8_t* row_exists,
                           void* stream);
/* creid_loo_emb_fwd_rows that also COUNTS the real instances without a real partner (an identity with exactly one real
 * instance: the reference raises there, train_ctl_model.py:80-104 -> losses/triplet_loss.py:88) into lonely_accum[0] (int32,
 * accumulated with an atomic: the caller zeroes it once per epoch and reads it at the epoch end -- no sync inside the step). */
int creid_loo_emb_fwd_rows_lonely(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                                  float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, uint8_t* row_exists,
                                  int32_t* lonely_accum, void* stream);
int creid_triplet_fwd_batched_rows(const float* x, const int64_t* labels, const uint8_t* row_exists, int64_t nb, int64_t N,
                                   int64_t D, float margin, int32_t min_rows, float* dist_ap, float* dist_an,
                                   int32_t* p_idx, int32_t* n_idx, float* coef, float* out4, void* stream);
int creid_ctl_round_scale(const float* out4_rounds, int64_t K, float* inv_rounds, void* stream);
int creid_ctl_step_stats_rows(const float* scal, const float* weights, int64_t n, int64_t K, int64_t P, const float* cnorm,
                              const uint8_t* row_exists, float* out, void* stream);
int creid_center_loss_fwd_masked(const float* x, const int64_t* labels, const float* centers, const uint8_t* row_mask,
                                 int64_t B, int64_t C, int64_t D, float* row_sq, float* loss, void* stream);
int creid_center_loss_bwd_masked(const float* x, const int64_t* labels, const float* centers, const float* row_sq,
                                 const uint8_t* row_mask, int64_t B, int64_t D, const float* gscale_dev, float gscale,
                                 float* dx_accum, float* dcenters_accum, void* stream);
int creid_xent_ls_masked(const float* logits, const int64_t* targets, const uint8_t* row_mask, int64_t B, int64_t C,
                         float eps, float gscale, float* row_loss, float* loss, float* dlogits, void* stream);
int creid_bn1d_fwd_masked(const float* x, const uint8_t* row_mask, int64_t B, int64_t D, const float* weight,
                          const float* bias, float* running_mean, float* running_var, float momentum, float eps, float* y,
                          float* save_mean, float* save_invstd, void* stream);
int creid_bn1d_bwd_masked(const float* x, const float* dy, const uint8_t* row_mask, int64_t B, int64_t D,
                          const float* weight, const float* save_mean, const float* save_invstd, float* dx_accum,
                          float* dweight_accum, float* dbias_accum, void* stream);

/* ------------------------------------------------------------------ stage C: losses */

/* losses/triplet_loss.py:27-41 (euclidean_dist), :68-119 (hard_example_mining), :139-173
 * (TripletLoss.__call__).  x fp32 [N,D]; labels int64 [N]; anchor_mask uint8 [N] or NULL (applied
 * AFTER mining, :148-151).  margin >= 0 -> MarginRankingLoss(margin); margin < 0 -> SoftMarginLoss.
 * Outputs: dist_ap/dist_an fp32 [N] (all anchors, unmasked), p_idx/n_idx int32 [N] (first index wins
 * ties), coef fp32 [N] (= dloss/d dist_ap, nullable), out4 = {loss, mean ap, mean an, #anchors} over
 * masked anchors, dist_mat fp32 [N,N] (nullable; the sqrt(clamp(.,1e-12)) matrix). */
int creid_triplet_fwd(const float* x, const int64_t* labels, const uint8_t* anchor_mask, int64_t N,
                      int64_t D, float margin, float* dist_ap, float* dist_an, int32_t* p_idx,
                      int32_t* n_idx, float* coef, float* out4, float* dist_mat, void* stream);
/* dx_accum[N,D] += gscale * (*gscale_dev if non-NULL) * dloss/dx (zero through an active clamp). */
int creid_triplet_bwd(const float* x, int64_t N, int64_t D, const float* dist_ap, const float* dist_an,
                      const int32_t* p_idx, const int32_t* n_idx, const float* coef,
                      const float* gscale_dev, float gscale, float* dx_accum, void* stream);

/* nb independent problems of the same size in one launch each (the K centroid rounds of
 * train_ctl_model.py:112-148): every array of creid_triplet_fwd / _bwd gains a leading [nb] dimension
 * (x [nb][N][D], labels / dist_* / *_idx / coef / anchor_mask [nb][N], out4 [nb][4], dx_accum [nb][N][D]). */
int creid_triplet_fwd_batched(const float* x, const int64_t* labels, const uint8_t* anchor_mask, int64_t nb,
                              int64_t N, int64_t D, float margin, float* dist_ap, float* dist_an,
                              int32_t* p_idx, int32_t* n_idx, float* coef, float* out4, float* dist_mat,
                              void* stream);
int creid_triplet_bwd_batched(const float* x, int64_t nb, int64_t N, int64_t D, const float* dist_ap,
                              const float* dist_an, const int32_t* p_idx, const int32_t* n_idx,
                              const float* coef, const float* gscale_dev, float gscale, float* dx_accum,
                              void* stream);

/* SOLVER.DISTANCE_FUNC = 'cosine' (losses/triplet_loss.py:44-65,134-137): the same mining / loss / backward on
 * the cosine distance clamp(|1 - x_i.x_j|, 1e-12) of rows already scaled to unit length (x_unit =
 * creid_rownorm_fwd(mode 0) of the features).  Argument meaning as creid_triplet_fwd / _bwd; the backward is
 * with respect to x_unit (chain creid_rownorm_bwd for the raw features). */
int creid_triplet_cosine_fwd(const float* x_unit, const int64_t* labels, const uint8_t* anchor_mask, int64_t N,
                             int64_t D, float margin, float* dist_ap, float* dist_an, int32_t* p_idx,
                             int32_t* n_idx, float* coef, float* out4, float* dist_mat, void* stream);
int creid_triplet_cosine_bwd(const float* x_unit, int64_t N, int64_t D, const float* dist_ap,
                             const float* dist_an, const int32_t* p_idx, const int32_t* n_idx,
                             const float* coef, const float* gscale_dev, float gscale, float* dx_accum,
                             void* stream);

/* Row scaling to unit length: mode 0  y = x / max(|x|_2, eps)  (cosine_similarity, losses/triplet_loss.py:50-52);
 * mode 1  y = x / (|x|_2 + eps)  (`normalize`, losses/triplet_loss.py:16-24, used by normalize_feature=True).
 * norm[N] receives |x|_2 (saved for the backward).  bwd: dx = d(y)/d(x)^T dy (overwrites dx). */
int creid_rownorm_fwd(const float* x, int64_t N, int64_t D, int mode, float eps, float* y, float* norm,
                      void* stream);
int creid_rownorm_bwd(const float* x, const float* norm, const float* dy, int64_t N, int64_t D, int mode,
                      float eps, float* dx, void* stream);

/* hard_example_mining(dist_mat, labels, return_inds=True) of losses/triplet_loss.py:68-119 on a GIVEN
 * square distance matrix fp32 [N,N]: hardest positive (max, self included) / hardest negative (min) per
 * anchor row, first index wins ties. */
int creid_hard_mine_from_dist(const float* dist_mat, const int64_t* labels, int64_t N, float* dist_ap,
                              float* dist_an, int32_t* p_idx, int32_t* n_idx, void* stream);

/* d[i] = sqrt(max(d[i], lo)) in place: the `clamp(min=1e-12).sqrt()` tail of euclidean_dist(x, y)
 * (losses/triplet_loss.py:40) after creid_sqdist_matrix. */
int creid_clamp_sqrt_inplace(float* d, int64_t n, float lo, void* stream);

/* losses/center_loss.py:26-46.  row_sq[b] = |x_b|^2 + |c_y|^2 - 2 x_b.c_y (unclamped, saved for bwd);
 * loss[0] = (sum_b clamp(row_sq[b],1e-12,1e12) + B*(C-1)*1e-12) / B. */
int creid_center_loss_fwd(const float* x, const int64_t* labels, const float* centers, int64_t B,
                          int64_t C, int64_t D, float* row_sq, float* loss, void* stream);
/* dx_accum[B,D] += g*(2/B)(x_b - c_y); dcenters_accum[C,D] += g*(2/B) sum_{b:y_b=y}(c_y - x_b) on the
 * touched rows only (either output may be NULL). */
int creid_center_loss_bwd(const float* x, const int64_t* labels, const float* centers,
                          const float* row_sq, int64_t B, int64_t D, const float* gscale_dev,
                          float gscale, float* dx_accum, float* dcenters_accum, void* stream);

/* losses/triplet_loss.py:194-205 CrossEntropyLabelSmooth: loss[0] = mean_b(-sum_c t_c logp_c),
 * t = (1-eps)*onehot + eps/C; dlogits (nullable) = gscale * (softmax - t) / B. */
int creid_xent_ls(const float* logits, const int64_t* targets, int64_t B, int64_t C, float eps,
                  float gscale, float* row_loss, float* loss, float* dlogits, void* stream);

/* modelling/bases.py:83-84 BNNeck = nn.BatchNorm1d(D) on [B,D] (momentum 0.1, eps 1e-5):
 * training: batch stats (biased var for y, unbiased into running_var); eval: running stats. */
int creid_bn1d_fwd(const float* x, int64_t B, int64_t D, const float* weight, const float* bias,
                   float* running_mean, float* running_var, int training, float momentum, float eps,
                   float* y, float* save_mean, float* save_invstd, void* stream);
int creid_bn1d_bwd(const float* x, const float* dy, int64_t B, int64_t D, const float* weight,
                   const float* save_mean, const float* save_invstd, float* dx_accum,
                   float* dweight_accum, float* dbias_accum, void* stream);

/* modelling/bases.py:92-95,238-241 (validation_create_centroids): out[s,:] = mean of the rows
 * emb[order[j],:], j in [offsets[s], offsets[s+1]) (summed in list order). */
int creid_gather_mean_rows(const float* emb, const int64_t* order, const int64_t* offsets,
                           int64_t n_seg, int64_t D, float* out, void* stream);

/* ------------------------------------------------------------------ optimiser steps */

/* solver/build.py:36-39 torch.optim.Adam (L2 weight decay added to the gradient, bias correction,
 * eps outside the sqrt) over a flat fp32 buffer; the gradient is multiplied by grad_scale first. */
int creid_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                    void* stream);
/* Same update with the step-dependent scalars resident on the device: hyper_dev = float[8]
 * {lr, step, 1-beta1^step, sqrt(1-beta2^step), ticket (int32, must be 0 between calls), 3 reserved}; the
 * call uses step + 1 and its bias corrections and leaves them in hyper_dev (written by the last
 * workgroup of the one launch), so a captured hipGraph replays correctly; the host only rewrites
 * hyper_dev[0] when the learning rate changes.  n % 4 == 0. */
int creid_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev,
                        float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                        void* stream);
/* train_ctl_model.py:157-159 + solver/build.py:44: g *= grad_mul (in place); p -= lr * g. */
int creid_sgd_scaled_step(float* p, float* g, int64_t n, float lr, float grad_mul, void* stream);

/* f16 mixed-precision training -- the reference's own mixed precision (utils/misc.py:111 `precision=16`, i.e. native AMP with
 * torch.cuda.amp.GradScaler under pytorch-lightning 1.1.4) -- with the dynamic loss scale RESIDENT ON THE DEVICE, so that the
 * step contains no host synchronisation and a captured hipGraph replays through overflow steps:
 *   amp_state = float[2] {scale, 1 / scale}; amp_flags = int32[3] {found_inf of this step, clean steps in a row, steps
 *   skipped in total}.
 * creid_amp_scale          y = x * scale (the head gradient entering the f16 backbone backward);
 * creid_amp_unscale_check  g *= 1 / scale in place over n (% 4 == 0) floats, any non-finite element sets amp_flags[0];
 * creid_adam_step_dev_amp / creid_sgd_scaled_step_amp  = the plain steps, skipped entirely (Adam's step counter included) while
 *                          *skip_flag != 0 (GradScaler.step);
 * creid_amp_update         found_inf ? scale *= backoff : (after growth_interval clean steps) scale *= growth; clears found_inf
 *                          (GradScaler.update; scale clamped to [1, 2^24]). */
int creid_amp_scale(const float* x, int64_t n, const float* amp_state, float* y, void* stream);
int creid_amp_unscale_check(float* g, int64_t n, const float* amp_state, int32_t* amp_flags, void* stream);
int creid_amp_update(float* amp_state, int32_t* amp_flags, float growth_factor, float backoff_factor, int32_t growth_interval,
                     void* stream);
int creid_adam_step_dev_amp(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, float beta1, float beta2,
                            float eps, float weight_decay, float grad_scale, const int32_t* skip_flag, void* stream);
int creid_sgd_scaled_step_amp(float* p, float* g, int64_t n, float lr, float grad_mul, const int32_t* skip_flag, void* stream);

/* ------------------------------------------------------------------ small fp32 GEMM */

/* Classifier of the BNNeck head (modelling/bases.py:86-87 fc_query = Linear(D -> C, bias=False);
 * used at train_ctl_model.py:75): C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
 * + beta * C[m*ldc + n], exact-f32 MFMA.  split_k > 1 combines K-slices with fp32 atomics
 * (non-deterministic order); split_k == 1 is deterministic. */
int creid_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                   float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                   int32_t split_k, void* stream);

/* ------------------------------------------------------------------ stage A: backbone layers */

/* Convolution geometry (NHWC activations; square kernels 1x1 / 3x3, stride 1 or 2; channel counts
 * powers of two >= 64 -- every non-stem convolution of modelling/backbones/resnet.py:51-120). */
typedef struct {
  int64_t batch, in_h, in_w, in_c, out_h, out_w, out_c;
  int32_t kh, kw, stride, pad;
} creid_conv_desc;

/* nn.Conv2d forward (resnet.py:56-61,109): y[b,oy,ox,n] = sum_{r,s,c} x[b,oy*st+r-pad,ox*st+s-pad,c] *
 * w_krsc[n,r,s,c]; im2col-free implicit GEMM on the MFMA pipe (bf16 -> fp32 accumulate, or exact f32).
 * bn_partial (nullable) receives per-128-row-tile (sum, sumsq) of the fp32 accumulators,
 * float [creid_conv2d_bn_partial_rows(d)][2][out_c], for the BatchNorm that follows. */
int64_t creid_conv2d_bn_partial_rows(const creid_conv_desc* d);
int creid_conv2d_fwd_nhwc(const creid_conv_desc* d, const void* x, const void* w_krsc, void* y,
                          float* bn_partial, int dtype, void* stream);
/* Eval-mode forward of conv -> BatchNorm -> (+ residual) -> (ReLU) in ONE launch (modelling/backbones/resnet.py:67-87 under
 * model.eval(), the path of validation_step, modelling/bases.py:169-177, and of inference/inference_utils.py:104-113): with
 * running statistics BatchNorm is the per-channel affine scale_shift = float[2][out_c] {gamma / sqrt(var + eps),
 * beta - mean * scale} (creid_bn2d_fold_multi), applied to the fp32 accumulators in the epilogue:
 * y = act(conv(x) * scale + shift (+ residual)), act = ReLU when relu != 0; residual (nullable) has y's shape and dtype.
 * Same arithmetic as creid_conv2d_fwd_nhwc + creid_bn2d_finalize(training = 0) + creid_bn2d_apply, without the two extra
 * passes (fp32 mode: bit-identical; bf16 mode: the conv output is rounded once, after the affine, instead of twice). */
int creid_conv2d_fwd_affine_nhwc(const creid_conv_desc* d, const void* x, const void* w_krsc, void* y,
                                 const float* scale_shift, const void* residual, int relu, int dtype, void* stream);
/* Training forward of a 1 x 1 stride-1 convolution whose INPUT is still the raw output of the previous convolution: that layer's
 * training-mode BatchNorm + ReLU (scale_shift [2][K] from creid_bn2d_finalize: a = max(x * scale + shift, 0), the arithmetic of
 * creid_bn2d_apply_mask) is applied on the operand path -- conv3 of a Bottleneck consuming conv2's raw output
 * (modelling/backbones/resnet.py:73-78) without the stand-alone apply pass.  y [M, N] (+ bn_partial like creid_conv2d_fwd_nhwc);
 * a_out [M, K] and mask_out [M * K / 8] (both or neither): the normalised tensor and its ReLU bits, bit-identical to
 * creid_bn2d_apply_mask's (the backward reads them).  16-bit dtypes, K in {64, 128}, N % 64 == 0; else CREID_E_SHAPE. */
int creid_conv1x1_bnrelu_fwd(const void* x_raw, const float* scale_shift, const void* w_krsc, int64_t M, int64_t K, int64_t N,
                             void* y, float* bn_partial, void* a_out, uint8_t* mask_out, int dtype, void* stream);
/* Folds MANY BatchNorm layers in one launch.  table_dev = device array of n_entries records { const float* gamma (nullable);
 * const float* beta (nullable); const float* running_mean; const float* running_var; float* out (float[2][C]); int32 C;
 * float eps } (48 bytes, creid_bn2d_fold_entry_bytes()). */
int64_t creid_bn2d_fold_entry_bytes(void);
int creid_bn2d_fold_multi(const void* table_dev, int64_t n_entries, void* stream);
/* data gradient: dx = conv_transpose(dy, w) (+ add_src if non-NULL); w_crsk is [in_c][kh][kw][out_c]. */
int creid_conv2d_dgrad_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx,
                            const void* add_src, int dtype, void* stream);
/* The same data gradient with the column reduction of the NEXT BatchNorm backward fused into the epilogue
 * (bf16 only): dx is g = dL/da of the layer whose raw conv output is bn_x and post-ReLU activation bn_act
 * (nullable); bn_partial[ceil(M/128)][2][in_c] receives (sum dy, sum dy*xhat), dy = g*[bn_act > 0].
 * bn_stat_image_rows = 0: bn_mean/bn_invstd are float[in_c] (BatchNorm); = in_h*in_w (a multiple of 128): they
 * are float[batch][in_c], per-(image, channel) statistics of an IBN layer (creid_ibn_bwd, partial_ready).
 * add_src_stride = 1: add_src is [batch, in_h, in_w, in_c]; = 2: add_src is the COMPACT gradient of a stride-2
 * 1x1 downsample branch, [batch, in_h/2, in_w/2, in_c], added at even (y, x) only (Bottleneck with stride,
 * resnet.py:87-88: its full-resolution data gradient is never materialised). */
int creid_conv2d_dgrad_bnred_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx,
                                  const void* add_src, const void* bn_x, const void* bn_act,
                                  const float* bn_mean, const float* bn_invstd, float* bn_partial,
                                  int64_t bn_stat_image_rows, int add_src_stride, int dtype, void* stream);

/* The data gradient with everything that can ride in the same launch: "+ add_src" (add_src_stride as above), the
 * NEXT BatchNorm-backward's column reduction (bn_x != NULL, arguments as creid_conv2d_dgrad_bnred_nhwc), and the
 * split reduction of the PREVIOUS weight-gradient launch: wred_desc != NULL names the convolution whose
 * creid_conv2d_wgrad_partials call filled wred_ws; the first workgroups of this launch sum those partials into
 * wred_dw (OIHW fp32; wred_accumulate as in creid_conv2d_wgrad_nhwc) while the rest compute gradient tiles, so the
 * 5-25 us stand-alone reduce launch per layer disappears.  In fp32 parity mode the reduction runs as its own launch
 * first (same result).  bn_mask != NULL: the ReLU mask of that BatchNorm's output as bits (creid_bn2d_apply_mask) --
 * one byte per 8 channels is read instead of the 16 bytes of bn_act.  add_mask != NULL (add_src_stride 1, bf16): add_src
 * is multiplied by those ReLU bits before the add -- the residual branch then takes the block's UNMASKED incoming
 * gradient, and the masked copy creid_bn2d_bwd would write through gm_out is never materialised. */
int creid_conv2d_dgrad_fused_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx,
                                  const void* add_src, int add_src_stride, const uint8_t* add_mask, const void* bn_x,
                                  const void* bn_act, const uint8_t* bn_mask, const float* bn_mean, const float* bn_invstd,
                                  float* bn_partial, int64_t bn_stat_image_rows, const creid_conv_desc* wred_desc, float* wred_dw,
                                  int wred_accumulate, const void* wred_ws, size_t wred_ws_bytes, int dtype,
                                  void* stream);
/* weight gradient into the fp32 OIHW tensor (the reference's nn.Parameter layout), optionally
 * accumulating; the pixel reduction is split over workgroups through `ws`. */
size_t creid_conv2d_wgrad_workspace_bytes(const creid_conv_desc* d, int dtype);
int creid_conv2d_wgrad_nhwc(const creid_conv_desc* d, const void* x, const void* dy, float* dw_oihw,
                            int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream);
/* The two halves of the call above as separate launches (same workspace contents in between): the split
 * reduce is a short, memory-light kernel that a caller may put on a second stream beside the next data gradient. */
int creid_conv2d_wgrad_partials(const creid_conv_desc* d, const void* x, const void* dy, void* ws,
                                size_t ws_bytes, int dtype, void* stream);
/* A BatchNorm-backward finalize (arguments as in creid_conv2d_wgrad_partials_bnfin) and the split reduction of a weight
 * gradient (arguments as in creid_conv2d_wgrad_reduce_job) in ONE launch: independent pieces of work, one latency-bound and
 * one bandwidth-bound.  Afterwards creid_bn2d_bwd_mask(..., partial_ready = 2, ...). */
int creid_bn2d_bwd_finalize_wred(const float* bn_partial, int64_t bn_rows, int64_t bn_C, int64_t bn_count,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_gamma, float* bn_sums,
                                 float* bn_dgamma, float* bn_dbeta, const creid_conv_desc* wred_desc, float* wred_dw,
                                 int wred_accumulate, const void* wred_ws, size_t wred_ws_bytes, int dtype, void* stream);
/* creid_conv2d_wgrad_reduce with the summation order of the carried form (the job creid_conv2d_dgrad_fused_nhwc runs in its
 * last workgroups): a reduction that finds no carrier gives bit-identical gradients to one that did. */
int creid_conv2d_wgrad_reduce_job(const creid_conv_desc* d, float* dw_oihw, int accumulate, const void* ws,
                                  size_t ws_bytes, int dtype, void* stream);
/* creid_conv2d_wgrad_partials whose launch also carries a BatchNorm-backward FINALIZE in its first workgroups: the weight
 * gradient is independent of the chain dgrad -> finalize -> apply -> dgrad, so issued between a data gradient and the next
 * BatchNorm's apply it hides that 4-128-workgroup, latency-bound step (bn_*: the arguments creid_bn2d_bwd's finalize
 * takes -- partial [bn_rows][2][bn_C] as written by the fused data gradient, count = rows of the statistics).  Afterwards
 * call creid_bn2d_bwd_mask(..., partial_ready = 2, sums = bn_sums, ...): apply only. */
int creid_conv2d_wgrad_partials_bnfin(const creid_conv_desc* d, const void* x, const void* dy, void* ws, size_t ws_bytes,
                                      int dtype, const float* bn_partial, int64_t bn_rows, int64_t bn_C, int64_t bn_count,
                                      const float* bn_mean, const float* bn_invstd, const float* bn_gamma, float* bn_sums,
                                      float* bn_dgamma, float* bn_dbeta, void* stream);
int creid_conv2d_wgrad_reduce(const creid_conv_desc* d, float* dw_oihw, int accumulate, const void* ws,
                              size_t ws_bytes, int dtype, void* stream);

/* Stem (resnet.py:94): Conv2d(3, 64, 7, stride 2, pad 3) on the zero-padded NHWC4 image
 * xpad [B, H+8, W+6, 4] made by creid_image_to_nhwc4_pad; w_stem [64][8][32] from creid_stem_weight_prep. */
int creid_stem_conv_fwd(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                        float* bn_partial, int dtype, void* stream);
/* the stem with its eval-mode BatchNorm (scale_shift float[2][64]) and optional ReLU (resnet_ibn_a.py:129) folded in */
int creid_stem_conv_fwd_affine(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                               const float* scale_shift, int relu, int dtype, void* stream);
/* The whole eval-mode stem in one launch (resnet.py:95-98,123-126: conv1 -> bn1 -> [relu] -> maxpool): creid_stem_conv_fwd_affine +
 * creid_maxpool3x3s2_fwd without the full-resolution tensor in between; y [batch, H/4, W/4, 64], bit-identical to the two calls.
 * 16-bit dtypes; CREID_E_SHAPE for image sizes outside the kernel's tiles (W = 128 or 320 with H % 8 = 0): the
 * caller then makes the two calls. */
int creid_stem_conv_pool_fwd_affine(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                                    const float* scale_shift, int relu, int dtype, void* stream);
/* Eval-mode block boundary of layer1 in one launch (resnet.py:77-87 of bottleneck i, then :69-71 of bottleneck i + 1):
 * out3 = relu(conv3(a2) * scale3 + shift3 + residual)   [M, c_out]   (= creid_conv2d_fwd_affine_nhwc with residual and ReLU)
 * out1 = relu(conv1_next(out3) * scale1 + shift1)       [M, c_next]  (= creid_conv2d_fwd_affine_nhwc on out3)
 * both 1 x 1 stride 1; a2 [M, c_mid]; weights in the [O][I] layout of creid_weight_prep; fold3 float[2][c_out], fold1
 * float[2][c_next].  The block output is written once and never read back; both outputs are bit-identical to the two calls.
 * 16-bit dtypes, (c_mid, c_out, c_next) = (64, 256, 64) only -- anything else: CREID_E_SHAPE, the caller makes the two calls. */
int creid_bottleneck_c3_c1_fwd_affine(int64_t M, int64_t c_mid, int64_t c_out, int64_t c_next, const void* a2, const void* w3_krsc,
                                      const float* fold3, const void* residual, void* out3, const void* w1_krsc,
                                      const float* fold1, void* out1, int dtype, void* stream);
/* The same launch when the next block's conv1 feeds an IBN layer (resnet_ibn_a.py:27-32): out1_raw [M, c_next] is the raw
 * convolution output and bn_partial1 float[M / 128][2][c_next] its per-128-row (sum, sum of squares) -- what creid_conv2d_fwd_nhwc
 * hands to creid_ibn_fwd (rows of a tile belong to one image when H * W % 128 = 0); M % 128 = 0 required. */
int creid_bottleneck_c3_c1_fwd_stats(int64_t M, int64_t c_mid, int64_t c_out, int64_t c_next, const void* a2, const void* w3_krsc,
                                     const float* fold3, const void* residual, void* out3, const void* w1_krsc, void* out1_raw,
                                     float* bn_partial1, int dtype, void* stream);
size_t creid_stem_conv_wgrad_workspace_bytes(int64_t batch, int64_t H, int64_t W, int dtype);
int creid_stem_conv_wgrad(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* dy,
                          float* dw_oihw, int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream);
int creid_image_to_nhwc4_pad(const float* x_nchw, int64_t B, int64_t H, int64_t W, int dtype, void* xpad,
                             void* stream);
/* Input transforms after the Resize (datasets/transforms/build.py:16-31, datasets/transforms/random_erasing.py:31-55) on a whole
 * uint8 batch: RandomHorizontalFlip -> Pad(pad, 0) -> RandomCrop(H, W) -> ToTensor -> Normalize -> RandomErasing, as a pure
 * function of the pixels and of the host-made draws.  src_hwc: uint8 [B, H, W, 3]; params: int32 [B, 8] =
 * {flip, crop_top, crop_left, erase, x1, y1, h, w} (crop offsets in the (H + 2 pad) x (W + 2 pad) padded frame; the erased block
 * is rows x1..x1+h, columns y1..y1+w of the output and receives (erase0, erase1, erase2) AS IS -- the reference writes
 * PIXEL_MEAN into the normalised tensor); params NULL = the test transform (ToTensor + Normalize only, pad must be 0).
 * layout 0: fp32 NCHW [B, 3, H, W] (dtype must be CREID_F32); layout 1: the stem's operand, zero-padded NHWC4
 * [B, H + 8, W + 6, 4] in `dtype` (f32 | bf16), identical to creid_image_to_nhwc4_pad of the layout-0 result. */
int creid_augment_u8(const uint8_t* src_hwc, const int32_t* params, int64_t B, int64_t H, int64_t W, int64_t pad,
                     float mean0, float mean1, float mean2, float std0, float std1, float std2, float erase0, float erase1,
                     float erase2, int32_t layout, int32_t dtype, void* out, void* stream);
/* fp32 OIHW master weights -> compute-dtype [O][r][s][I] (forward) and [I][r][s][O] (dgrad, nullable). */
int creid_weight_prep(const float* w_oihw, int64_t O, int64_t I, int64_t kh, int64_t kw, int dtype,
                      void* w_krsc, void* w_crsk, void* stream);
int creid_stem_weight_prep(const float* w_oihw, int dtype, void* w_stem, void* stream);
/* The same transform for MANY convolutions in one launch (LDS-tiled transposes, coalesced writes).
 * table_dev = device array of n_entries records { const float* w_oihw; void* w_krsc; void* w_crsk (nullable);
 * int32 O, I, kh, kw; int64 start } (48 bytes, creid_weight_prep_entry_bytes()); tile_start_dev = int32[n_entries]
 * = cumulative count of 32x32 (o, c) tiles, ceil(O/32)*ceil(I/32) per entry; total_tiles = their sum. */
int64_t creid_weight_prep_entry_bytes(void);
int creid_weight_prep_multi(const void* table_dev, const int32_t* tile_start_dev, int64_t n_entries,
                            int64_t total_tiles, int dtype, void* stream);

/* nn.BatchNorm2d (resnet.py:57-62,96,111; momentum 0.1, eps 1e-5) split in three steps:
 * finalize: partial (sum,sumsq) rows -> mean / invstd (+ running-stat update, unbiased variance) when
 * training, or mean = running_mean, invstd = rsqrt(running_var + eps) in eval; also emits the
 * per-channel affine scale_shift = float[2][C] {gamma*invstd, beta - mean*gamma*invstd};
 * apply: y = x * scale + shift (+ residual) (ReLU if relu) -- the fused
 * BN + residual-add + ReLU tail of Bottleneck.forward (resnet.py:72-85);
 * bwd: dy = g * [act > 0] (act nullable); dgamma += sum dy*xhat; dbeta += sum dy;
 *      dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)); gm_out (nullable) = dy.
 *      partial = float[creid_bn2d_bwd_rows(M)][2][C] scratch (partial_ready != 0: already filled by
 *      creid_conv2d_dgrad_bnred_nhwc, the column pass is skipped), sums = float[3][C] scratch. */
int creid_bn2d_finalize(const float* partial, int64_t rows, int64_t C, int64_t count, float* running_mean,
                        float* running_var, int training, float momentum, float eps, const float* gamma,
                        const float* beta, float* mean_out, float* invstd_out, float* scale_shift,
                        void* stream);
int64_t creid_col_stats_rows(int64_t M);
int creid_col_stats(const void* x, int64_t M, int64_t C, int dtype, float* partial, void* stream);
int creid_bn2d_apply(const void* x, const float* scale_shift, const void* residual, int relu, int64_t M,
                     int64_t C, int dtype, void* y, void* stream);
/* creid_bn2d_apply that also writes the ReLU mask the backward needs as BITS: mask_out (nullable; bf16 / f16 only) =
 * uint8[M * C / 8], bit k of byte i = (y[8 i + k] > 0).  The backward then reads 1 byte per 8 channels instead of
 * re-reading y (modelling/backbones/resnet.py:71,75,85 -- the three ReLUs of a Bottleneck). */
int creid_bn2d_apply_mask(const void* x, const float* scale_shift, const void* residual, int relu, int64_t M,
                          int64_t C, int dtype, void* y, uint8_t* mask_out, void* stream);
/* Training-mode creid_bn2d_finalize + creid_bn2d_apply_mask in ONE launch (round 5), for layers with few statistic rows
 * (rows <= 1024; meant for M <= 32768): every apply workgroup sums the partial rows of its own 64-channel strip in fp64 (the
 * finalize's expressions, so mean / invstd / scale_shift / the running statistics / y / mask_out come out bit-identical to the
 * two launches), the row-block-0 workgroups publish the statistics.  count = M (the statistics cover every row of x).
 * row_blocks <= 0: chosen by the library.  C % 64 == 0 (bf16 / f16) or % 32 (f32), else CREID_E_SHAPE (use the two calls). */
int creid_bn2d_finalize_apply_mask(const float* partial, int64_t rows, int64_t C, int64_t M, float* running_mean,
                                   float* running_var, float momentum, float eps, const float* gamma, const float* beta,
                                   float* mean_out, float* invstd_out, float* scale_shift, const void* x, const void* residual,
                                   int relu, int dtype, void* y, uint8_t* mask_out, int row_blocks, void* stream);
/* bn3 + downsample BatchNorm + add + ReLU of a block with a downsample branch (resnet.py:80-85) in ONE pass:
 * y = act(x * scale + shift + (x_res * scale_res + shift_res)); x_res is the RAW downsample convolution output -- its
 * normalised tensor is never written (fp32: bit-identical to creid_bn2d_apply twice; bf16: one rounding fewer). */
int creid_bn2d_apply_dual_mask(const void* x, const float* scale_shift, const void* x_res, const float* scale_shift_res,
                               int relu, int64_t M, int64_t C, int dtype, void* y, uint8_t* mask_out, void* stream);
int64_t creid_bn2d_bwd_rows(int64_t M);
int creid_bn2d_bwd(const void* x, const void* g, const void* act, const float* mean, const float* invstd,
                   const float* gamma, int64_t M, int64_t C, int dtype, float* partial, int partial_ready,
                   float* sums, float* dgamma_accum, float* dbeta_accum, void* dx, void* gm_out, void* stream);
/* creid_bn2d_bwd with the ReLU mask as bits (mask != NULL: creid_bn2d_apply_mask's output, act is then not read). */
int creid_bn2d_bwd_mask(const void* x, const void* g, const void* act, const uint8_t* mask, const float* mean,
                        const float* invstd, const float* gamma, int64_t M, int64_t C, int dtype, float* partial,
                        int partial_ready, float* sums, float* dgamma_accum, float* dbeta_accum, void* dx,
                        void* gm_out, void* stream);
/* creid_bn2d_bwd_mask for the bn3 of a DOWNSAMPLE block (16-bit types; no act / gm_out): the pass that writes dx also produces the
 * column sums of the downsample branch's BatchNorm backward over the same masked gradient -- partial2
 * [creid_bn2d_bwd_rows(M)][2][C] from that branch's raw output x2 and its (mean2, invstd2) -- bit-identical to that layer's own
 * column pass (hand it to creid_bn2d_bwd_mask with partial_ready = 1); dx is bit-identical to creid_bn2d_bwd_mask's. */
int creid_bn2d_bwd_mask_reduce2(const void* x, const void* g, const uint8_t* mask, const float* mean, const float* invstd,
                                const float* gamma, int64_t M, int64_t C, int dtype, float* partial, int partial_ready, float* sums,
                                float* dgamma_accum, float* dbeta_accum, void* dx, const void* x2, const float* mean2,
                                const float* invstd2, float* partial2, void* stream);

/* IBN of ResNet50-IBN-a (modelling/backbones/resnet_ibn_a.py:18-32): channels [0, c_in) InstanceNorm2d
 * (affine, per-(image, channel) statistics over H*W, eps 1e-5, no running stats), channels [c_in, C)
 * BatchNorm2d; then ReLU if relu.  x, y NHWC [B*HW, C]; partial = float[B*creid_ibn_rows_per_image(HW)][2][C]
 * scratch -- or, with partial_ready = 1 and HW % 128 == 0, the (sum, sumsq) partials the producing
 * creid_conv2d_fwd_nhwc already wrote (its 128-row tiles are then whole row blocks of one image);
 * mean_out/invstd_out float[B][C]; scale_shift float[B][2][C] (saved for the backward).
 * bwd: dy = g*[act>0]; coef float[B][3][C] and per_img float[B][2][c_in] scratch; parameter
 * gradients are accumulated; partial_ready = 1: `partial` was filled by creid_conv2d_dgrad_bnred_nhwc
 * (bn_stat_image_rows = HW, HW % 128 == 0) and the column pass is skipped. */
int64_t creid_ibn_rows_per_image(int64_t HW);
int creid_ibn_fwd(const void* x, int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w,
                  const float* in_b, const float* bn_w, const float* bn_b, float* running_mean,
                  float* running_var, int training, float momentum, float eps, int relu, int dtype,
                  float* partial, int partial_ready, float* mean_out, float* invstd_out, float* scale_shift,
                  void* y, void* stream);
int creid_ibn_bwd(const void* x, const void* g, const void* act, const float* mean, const float* invstd,
                  int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w, const float* bn_w,
                  int dtype, float* partial, int partial_ready, float* coef, float* per_img, float* d_in_w,
                  float* d_in_b, float* d_bn_w, float* d_bn_b, void* dx, void* stream);
/* the same with the ReLU mask as bits (see creid_bn2d_apply_mask): mask_out / mask = uint8[B*HW*C/8], bf16 / f16 only */
int creid_ibn_fwd_mask(const void* x, int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w,
                       const float* in_b, const float* bn_w, const float* bn_b, float* running_mean,
                       float* running_var, int training, float momentum, float eps, int relu, int dtype,
                       float* partial, int partial_ready, float* mean_out, float* invstd_out, float* scale_shift,
                       void* y, uint8_t* mask_out, void* stream);
int creid_ibn_bwd_mask(const void* x, const void* g, const void* act, const uint8_t* mask, const float* mean,
                       const float* invstd, int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w,
                       const float* bn_w, int dtype, float* partial, int partial_ready, float* coef, float* per_img,
                       float* d_in_w, float* d_in_b, float* d_bn_w, float* d_bn_b, void* dx, void* stream);

/* nn.MaxPool2d(3, 2, 1) (resnet.py:98) NHWC, with the argmax tap saved for the backward (idx NULL: inference, no taps written). */
/* The stem's tail without its full-resolution intermediates (modelling/backbones/resnet.py:123-126, conv1 -> bn1 -> [relu] ->
 * maxpool): creid_bn2d_apply_maxpool3x3s2 = creid_bn2d_apply (no residual) + creid_maxpool3x3s2_fwd in one pass over the raw
 * conv output x [B, H, W, C] -- the normalised tensor is never written; y [B, H/2, W/2, C], idx as creid_maxpool3x3s2_fwd;
 * bit-identical to the two separate calls.  creid_bn2d_bwd_pooled = creid_maxpool3x3s2_bwd + creid_bn2d_bwd (no ReLU
 * mask unless act is given) with the pool gradient gathered inside the BatchNorm passes: dy_pooled [B, H/2, W/2, C] and
 * idx go in, the full-resolution gradient is never written; partial / sums as creid_bn2d_bwd; dx [B, H, W, C]. */
int creid_bn2d_apply_maxpool3x3s2(const void* x, const float* scale_shift, int relu, int64_t B, int64_t H, int64_t W,
                                  int64_t C, int dtype, void* y, uint8_t* idx, void* stream);
int creid_bn2d_bwd_pooled(const void* x, const void* dy_pooled, const uint8_t* idx, int64_t B, int64_t H, int64_t W,
                          const void* act, const float* mean, const float* invstd, const float* gamma, int64_t C,
                          int dtype, float* partial, float* sums, float* dgamma_accum, float* dbeta_accum, void* dx,
                          void* stream);
int creid_maxpool3x3s2_fwd(const void* x, int64_t B, int64_t H, int64_t W, int64_t C, int dtype, void* y,
                           uint8_t* idx, void* stream);
int creid_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C,
                           int dtype, void* dx, void* stream);
/* nn.AdaptiveAvgPool2d(1) (modelling/baseline.py:89,93): feat fp32 [B, C]. */
int creid_gap_fwd(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* feat, void* stream);
/* creid_gap_fwd that also adds 1 to *forward_counter (int64 on the device): the training forward's count of batches seen, folded
 * into every BatchNorm2d.num_batches_tracked before a state_dict (one counter per step instead of 53 increments). */
int creid_gap_fwd_count(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* feat, int64_t* forward_counter,
                        void* stream);
int creid_gap_bwd(const float* dfeat, int64_t B, int64_t HW, int64_t C, int dtype, void* dx, void* stream);
/* NHWC compute dtype -> NCHW fp32 (to return `base_out` in the reference's layout). */
int creid_nhwc_to_nchw_f32(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CREID_H */
