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
 *     launch; never throws, never synchronises the stream, re-entrant, no global state.
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

/* ------------------------------------------------------------------ stage E: CMC / mAP */

/* utils/eval_reid.py:44-84 (the per-query loop of eval_func, respect_camids=False), one
 * workgroup per query over the ranked row: drop gallery entries with the query's pid AND
 * camid (:57); valid = any match left (:63-65); ap = sum_k [match_k] * cum_k / (k+1) / n_rel
 * in float64 (:75-79); first = 0-based kept-rank of the first match (defines the clipped
 * CMC row :67-70 and the top-k flags :18-22).  idx int64 [m, n]; pids/camids int64. */
int creid_cmc_ap_ranked(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids,
                        const int64_t* g_pids, const int64_t* q_camids, const int64_t* g_camids,
                        uint8_t* out_valid, double* out_ap, int32_t* out_first, void* stream);

/* utils/eval_reid.py:86-90: means over valid queries.  out_cmc float32[max_rank]
 * (= count(first<=r)/n_valid in float32), out_map float64[1], out_topk float64[5] for
 * k in {1,5,10,20,50}, out_nvalid int64[1]. */
int creid_eval_reduce(const uint8_t* valid, const double* ap, const int32_t* first, int64_t m,
                      int32_t max_rank, float* out_cmc, double* out_map, double* out_topk,
                      int64_t* out_nvalid, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CREID_H */
