// Stages B/C: leave-one-out per-PID centroids, pairwise-distance + batch-hard mining triplet,
// center loss, label-smoothed cross-entropy, BNNeck (BatchNorm1d), optimiser steps.
//
// Reference arithmetic: train_ctl_model.py:79-104 (centroids), losses/triplet_loss.py:27-173,
// losses/center_loss.py:26-46, losses/triplet_loss.py:194-205 (xent), modelling/bases.py:83-84
// (BNNeck), solver/build.py:36-44 + train_ctl_model.py:154-159 (Adam / center SGD).
// All tensors here are tiny (B = 64..512 rows x 2048): every kernel is latency-bound, so the
// design goal is few launches, wave-level reductions (no atomics -> deterministic) and
// features streamed once per workgroup with 16-byte loads.
#include "common.hpp"
#include "gemm_f32_body.hpp"

// ======================================================================================
// B2: leave-one-out centroids.  grid (P, K); centroid[i,p,:] = sum_{s!=i, real} f[p,s,:]/max(cnt,1)
// if slot i of pid p is real, else 0.
// ======================================================================================
__global__ __launch_bounds__(256) void loo_centroids_fwd_kernel(const float* __restrict__ feat,
                                                                const uint8_t* __restrict__ is_real, int P, int K,
                                                                int D, float* __restrict__ cent,
                                                                int32_t* __restrict__ valid) {
  const int p = blockIdx.x, i = blockIdx.y;
  const bool qreal = is_real[p * K + i] != 0;
  int cnt = 0;
  if (qreal)
    for (int s = 0; s < K; ++s) cnt += (s != i && is_real[p * K + s]) ? 1 : 0;
  if (threadIdx.x == 0) valid[i * P + p] = cnt;
  const float inv_den = (float)max(cnt, 1);
  float* out = cent + ((int64_t)i * P + p) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    if (qreal)
      for (int s = 0; s < K; ++s)   // same s-order as the reference's sum(-2)
        if (s != i && is_real[p * K + s]) acc += feat[((int64_t)p * K + s) * D + d];
    out[d] = acc / inv_den;
  }
}

// dfeat[p,s,:] += sum_{i != s, real(i), real(s)} dcent[i,p,:] / max(cnt_i,1).  grid (P, K=s)
__global__ __launch_bounds__(256) void loo_centroids_bwd_kernel(const float* __restrict__ dcent,
                                                                const uint8_t* __restrict__ is_real, int P, int K,
                                                                int D, float* __restrict__ dfeat) {
  const int p = blockIdx.x, s = blockIdx.y;
  if (!is_real[p * K + s]) return;
  float* out = dfeat + ((int64_t)p * K + s) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int i = 0; i < K; ++i) {
      if (i == s || !is_real[p * K + i]) continue;
      int cnt = 0;
      for (int t = 0; t < K; ++t) cnt += (t != i && is_real[p * K + t]) ? 1 : 0;
      acc += dcent[((int64_t)i * P + p) * D + d] / (float)max(cnt, 1);
    }
    out[d] += acc;
  }
}

// ======================================================================================
// C1-C3: triplet.  One 1024-thread workgroup per anchor: distances to all N rows (wave per row, 16-B
// loads, wave reduction; 16 waves because N is only 32..128 and the kernel is pure latency), then
// hardest-positive / hardest-negative mining by wave 0.
// ======================================================================================
// KIND 0: euclidean (sqrt of the clamped expanded square, triplet_loss.py:27-41); KIND 1: cosine distance
// clamp(|1 - x.y|, 1e-12) on rows that the caller has already scaled to unit length (triplet_loss.py:44-65).
constexpr int TM_T = 1024, TM_W = TM_T / 64;
template <int KIND>
__device__ __forceinline__ void triplet_mine_body(const float* __restrict__ x,
                                                           const int64_t* __restrict__ labels, int N, int D,
                                                           float* __restrict__ dist_ap, float* __restrict__ dist_an,
                                                           int32_t* __restrict__ p_idx, int32_t* __restrict__ n_idx,
                                                           float* __restrict__ dist_row_out /* nullable [N,N] */,
                                                           const uint8_t* __restrict__ exists /* nullable [N]: rows that are
                                                           part of the problem at all (neither anchors nor candidates otherwise) */, const int bx_, const int by_) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [D] anchor row, then [N] distances
  float* xa = sm;
  float* drow = sm + D;
  {                                                            // by_ = independent problem (centroid round)
    const int64_t bo = (int64_t)by_ * N;
    x += bo * D; labels += bo; dist_ap += bo; dist_an += bo; p_idx += bo; n_idx += bo;
    if (dist_row_out) dist_row_out += bo * N;
    if (exists) exists += bo;
  }
  const int a = bx_, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xap = x + (int64_t)a * D;
  float saa = 0.f;
  for (int d = tid; d < D; d += TM_T) { float v = xap[d]; xa[d] = v; }
  __syncthreads();
  for (int d = lane; d < D; d += 64) saa = fmaf(xa[d], xa[d], saa);
  saa = wave_sum(saa);
  for (int j = wave; j < N; j += TM_W) {
    const float* xj = x + (int64_t)j * D;
    float dot = 0.f, sjj = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      float4 v = *reinterpret_cast<const float4*>(xj + d);
      float4 u = *reinterpret_cast<const float4*>(xa + d);
      dot = fmaf(u.x, v.x, dot); dot = fmaf(u.y, v.y, dot); dot = fmaf(u.z, v.z, dot); dot = fmaf(u.w, v.w, dot);
      sjj = fmaf(v.x, v.x, sjj); sjj = fmaf(v.y, v.y, sjj); sjj = fmaf(v.z, v.z, sjj); sjj = fmaf(v.w, v.w, sjj);
    }
    dot = wave_sum(dot); sjj = wave_sum(sjj);
    if (lane == 0) {
      if (KIND == 0) {
        const float sq = fmaf(-2.0f, dot, saa + sjj);       // xx + yy - 2 x.y   (triplet_loss.py:35-39)
        drow[j] = sqrtf(fmaxf(sq, 1e-12f));                  // clamp(min=1e-12).sqrt() (:40)
      } else {
        drow[j] = fmaxf(fabsf(1.0f - dot), 1e-12f);          // abs(1 - sim).clamp(min=eps) (:65)
      }
    }
  }
  __syncthreads();
  if (dist_row_out)
    for (int j = tid; j < N; j += TM_T) dist_row_out[(int64_t)a * N + j] = drow[j];
  if (wave == 0) {
    const int64_t la = labels[a];
    float bp = -INFINITY, bn = INFINITY;
    int ip = 0x7fffffff, in_ = 0x7fffffff;
    for (int j = lane; j < N; j += 64) {
      if (exists && !exists[j]) continue;
      const float d = drow[j];
      if (labels[j] == la) { if (d > bp) { bp = d; ip = j; } }
      else { if (d < bn) { bn = d; in_ = j; } }
    }
    // wave arg-max / arg-min with first-index tie-break (torch.max/min return the first)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float obp = __shfl_xor(bp, o, 64); int oip = __shfl_xor(ip, o, 64);
      if (obp > bp || (obp == bp && oip < ip)) { bp = obp; ip = oip; }
      float obn = __shfl_xor(bn, o, 64); int oin = __shfl_xor(in_, o, 64);
      if (obn < bn || (obn == bn && oin < in_)) { bn = obn; in_ = oin; }
    }
    if (lane == 0) { dist_ap[a] = bp; dist_an[a] = bn; p_idx[a] = ip; n_idx[a] = in_; }
  }
}
template <int KIND>
__global__ __launch_bounds__(TM_T) void triplet_mine_kernel(const float* __restrict__ x,
                                                           const int64_t* __restrict__ labels, int N, int D,
                                                           float* __restrict__ dist_ap, float* __restrict__ dist_an,
                                                           int32_t* __restrict__ p_idx, int32_t* __restrict__ n_idx,
                                                           float* __restrict__ dist_row_out /* nullable [N,N] */,
                                                           const uint8_t* __restrict__ exists /* nullable [N]: rows that are
                                                           part of the problem at all (neither anchors nor candidates otherwise) */) {
  triplet_mine_body<KIND>(x, labels, N, D, dist_ap, dist_an, p_idx, n_idx, dist_row_out, exists, (int)blockIdx.x, (int)blockIdx.y);
}

// loss over (masked) anchors; coef[a] = d loss / d dist_ap[a] ( = - d loss / d dist_an[a]).
// margin >= 0: MarginRankingLoss mean(max(0, ap - an + margin)); margin < 0: SoftMarginLoss.
// out[0]=loss, out[1]=mean ap, out[2]=mean an, out[3]=#anchors.
__device__ __forceinline__ void triplet_loss_body(const float* __restrict__ dist_ap,
                                                           const float* __restrict__ dist_an,
                                                           const uint8_t* __restrict__ mask, int N, float margin,
                                                           float* __restrict__ out, float* __restrict__ coef,
                                                           int min_anchors /* fewer anchors: the problem is skipped (a centroid
                                                           round with <= 1 valid identity, train_ctl_model.py:113) */, const int bx_, const int by_) {
  __shared__ float s[4][4];
  {
    const int64_t bo = (int64_t)by_ * N;
    dist_ap += bo; dist_an += bo; out += (int64_t)by_ * 4;
    if (mask) mask += bo;
    if (coef) coef += bo;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float cnt = 0.f;
  for (int a = tid; a < N; a += 256) cnt += (!mask || mask[a]) ? 1.f : 0.f;
  cnt = wave_sum(cnt);
  if (lane == 0) s[wave][3] = cnt;
  __syncthreads();
  const float nm = s[0][3] + s[1][3] + s[2][3] + s[3][3];
  __syncthreads();
  if (nm < (float)min_anchors || nm == 0.f) {          // (nm == 0 only with a mask: nothing to average)
    if (coef) for (int a = tid; a < N; a += 256) coef[a] = 0.f;
    if (tid == 0) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f; }
    return;
  }
  float l = 0.f, sap = 0.f, san = 0.f;
  for (int a = tid; a < N; a += 256) {
    const bool on = !mask || mask[a];
    float c = 0.f;
    if (on) {
      const float ap = dist_ap[a], an = dist_an[a];
      sap += ap; san += an;
      if (margin >= 0.f) {
        const float v = ap - an + margin;           // -(an - ap) + margin
        if (v > 0.f) { l += v; c = 1.f / nm; }
      } else {
        const float z = ap - an;                    // log(1 + exp(-(an - ap)))
        l += (z > 0.f) ? z + log1pf(expf(-z)) : log1pf(expf(z));
        c = (1.f / (1.f + expf(-z))) / nm;
      }
    }
    if (coef) coef[a] = c;
  }
  l = wave_sum(l); sap = wave_sum(sap); san = wave_sum(san);
  if (lane == 0) { s[wave][0] = l; s[wave][1] = sap; s[wave][2] = san; }
  __syncthreads();
  if (tid == 0) {
    out[0] = ((s[0][0] + s[1][0]) + (s[2][0] + s[3][0])) / nm;
    out[1] = ((s[0][1] + s[1][1]) + (s[2][1] + s[3][1])) / nm;
    out[2] = ((s[0][2] + s[1][2]) + (s[2][2] + s[3][2])) / nm;
    out[3] = nm;
  }
}
__global__ __launch_bounds__(256) void triplet_loss_kernel(const float* __restrict__ dist_ap,
                                                           const float* __restrict__ dist_an,
                                                           const uint8_t* __restrict__ mask, int N, float margin,
                                                           float* __restrict__ out, float* __restrict__ coef,
                                                           int min_anchors /* fewer anchors: the problem is skipped (a centroid
                                                           round with <= 1 valid identity, train_ctl_model.py:113) */) {
  triplet_loss_body(dist_ap, dist_an, mask, N, margin, out, coef, min_anchors, (int)blockIdx.x, (int)blockIdx.y);
}

// dx[r,:] += g * sum_a coef[a] * ( [a==r]((xa-xp)/dap - (xa-xn)/dan) + [p_a==r](xp-xa)/dap - [n_a==r](xn-xa)/dan )
// (gradient of sqrt(clamp(.)) is zero where the clamp was active: dist <= 1e-6).  One workgroup per row.
template <int KIND>
__device__ __forceinline__ void triplet_bwd_body(const float* __restrict__ x, int N, int D,
                                                          const float* __restrict__ dist_ap,
                                                          const float* __restrict__ dist_an,
                                                          const int32_t* __restrict__ p_idx,
                                                          const int32_t* __restrict__ n_idx,
                                                          const float* __restrict__ coef,
                                                          const float* __restrict__ gscale_ptr, float gscale,
                                                          float* __restrict__ dx, const int bx_, const int by_) {
  // phase 1: the (few) anchors that touch this row, compacted IN ANCHOR ORDER into LDS as
  // (other row, signed coefficient) terms:  dx[r] += sum_terms w * (x[r] - x[other])
  extern __shared__ __attribute__((aligned(16))) float sm[];
  int* t_other = reinterpret_cast<int*>(sm);           // [<= 4N] capacity 4 terms per anchor
  float* t_w = sm + 4 * N;
  __shared__ int n_terms;
  {
    const int64_t bo = (int64_t)by_ * N;
    x += bo * D; dist_ap += bo; dist_an += bo; p_idx += bo; n_idx += bo; coef += bo; dx += bo * D;
  }
  const int r = bx_;
  // one thread per anchor (all loads in flight at once; a single thread walking the N anchors spent ~10 us in dependent
  // loads), then the terms are compacted in anchor order by an exclusive scan of the per-anchor counts
  __shared__ int t_cnt[256];
  for (int a0 = 0; a0 < N; a0 += 256) {                    // N <= 256 in every caller: one trip
    const int a = a0 + (int)threadIdx.x;
    int oth[4]; float wv[4]; int k = 0;
    if (a < N) {
      const float c = coef[a];
      const int p = p_idx[a], n = n_idx[a];
      if (c != 0.f && (a == r || p == r || n == r)) {
        const float dap = dist_ap[a], dan = dist_an[a];
        // cosine: d|1 - u.v| / du = -sign(1 - u.v) v (zero where the clamp is active); the sign is applied below
        const float ip = KIND == 0 ? (dap > 1e-6f ? c / dap : 0.f) : (dap > 1e-12f ? -c : 0.f);
        const float in_ = KIND == 0 ? (dan > 1e-6f ? c / dan : 0.f) : (dan > 1e-12f ? -c : 0.f);
        if (a == r) { oth[k] = p; wv[k++] = ip; oth[k] = n; wv[k++] = -in_; }
        if (p == r) { oth[k] = a; wv[k++] = ip; }
        if (n == r) { oth[k] = a; wv[k++] = -in_; }
      }
    }
    if (a0 == 0 && threadIdx.x == 0) n_terms = 0;
    t_cnt[threadIdx.x] = k;
    __syncthreads();
    int off = n_terms;
    for (int q = 0; q < (int)threadIdx.x; ++q) off += t_cnt[q];
    for (int q = 0; q < k; ++q) { t_other[off + q] = oth[q]; t_w[off + q] = wv[q]; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = n_terms;
      for (int q = 0; q < 256; ++q) tot += t_cnt[q];
      n_terms = tot;
    }
    __syncthreads();
  }
  const int nt = n_terms;
  if (nt == 0) return;
  const float g = gscale * (gscale_ptr ? *gscale_ptr : 1.f);
  const float* xr = x + (int64_t)r * D;
  float* out = dx + (int64_t)r * D;
  if (KIND == 1) {
    // every term pairs row r with t_other[k]: fold sign(1 - x_r . x_other) into its weight, then
    // dx[r] += g * sum_k w_k * x[other_k]
    __shared__ float part[4];
    for (int k = 0; k < nt; ++k) {
      const float* xo = x + (int64_t)t_other[k] * D;
      float dot = 0.f;
      for (int d = threadIdx.x; d < D; d += 256) dot = fmaf(xr[d], xo[d], dot);
      dot = wave_sum(dot);
      if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float v = 1.0f - ((part[0] + part[1]) + (part[2] + part[3]));
        t_w[k] *= v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
      }
      __syncthreads();
    }
    for (int d = threadIdx.x; d < D; d += 256) {
      float acc = 0.f;
      for (int k = 0; k < nt; ++k) acc += t_w[k] * x[(int64_t)t_other[k] * D + d];
      out[d] += g * acc;
    }
    return;
  }
  for (int d = threadIdx.x; d < D; d += 256) {
    const float xv = xr[d];
    float acc = 0.f;
    for (int k = 0; k < nt; ++k) acc += t_w[k] * (xv - x[(int64_t)t_other[k] * D + d]);
    out[d] += g * acc;
  }
}
template <int KIND>
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ x, int N, int D,
                                                          const float* __restrict__ dist_ap,
                                                          const float* __restrict__ dist_an,
                                                          const int32_t* __restrict__ p_idx,
                                                          const int32_t* __restrict__ n_idx,
                                                          const float* __restrict__ coef,
                                                          const float* __restrict__ gscale_ptr, float gscale,
                                                          float* __restrict__ dx) {
  triplet_bwd_body<KIND>(x, N, D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_ptr, gscale, dx, (int)blockIdx.x, (int)blockIdx.y);
}

// ======================================================================================
// C4: center loss.  row_loss[b] = clamp(|x_b|^2 + |c_y|^2 - 2 x_b.c_y, 1e-12, 1e12) (expanded form
// like the reference); loss = (sum_b row_loss + B*(C-1)*1e-12)/B.
// ======================================================================================
__device__ __forceinline__ void center_row_body(const float* __restrict__ x,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ centers, int D,
                                                         float* __restrict__ row_loss, const int bx_, const int by_) {
  __shared__ float s[4][3];
  const int b = bx_, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (int64_t)b * D;
  const float* c = centers + labels[b] * D;
  float xx = 0.f, cc = 0.f, xc = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float u = xb[d], v = c[d];
    xx = fmaf(u, u, xx); cc = fmaf(v, v, cc); xc = fmaf(u, v, xc);
  }
  xx = wave_sum(xx); cc = wave_sum(cc); xc = wave_sum(xc);
  if (lane == 0) { s[wave][0] = xx; s[wave][1] = cc; s[wave][2] = xc; }
  __syncthreads();
  if (tid == 0) {
    xx = (s[0][0] + s[1][0]) + (s[2][0] + s[3][0]);
    cc = (s[0][1] + s[1][1]) + (s[2][1] + s[3][1]);
    xc = (s[0][2] + s[1][2]) + (s[2][2] + s[3][2]);
    const float dv = fmaf(-2.0f, xc, xx + cc);
    row_loss[b] = dv;   // unclamped; the reduce kernel clamps (and bwd needs the clamp state)
  }
}
__global__ __launch_bounds__(256) void center_row_kernel(const float* __restrict__ x,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ centers, int D,
                                                         float* __restrict__ row_loss) {
  center_row_body(x, labels, centers, D, row_loss, (int)blockIdx.x, (int)blockIdx.y);
}

// number of rows with mask != 0 (all B when mask is NULL), by every thread of a 256-thread workgroup; `sc` = 4 ints of LDS
__device__ __forceinline__ int masked_row_count(const uint8_t* __restrict__ mask, int B, int* sc) {
  if (!mask) return B;
  int c = 0;
  for (int b = threadIdx.x; b < B; b += 256) c += mask[b] ? 1 : 0;
  c = wave_sum_i(c);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
  __syncthreads();
  return (sc[0] + sc[1]) + (sc[2] + sc[3]);
}

// mask (nullable, uint8 [B]): the rows the loss is taken over -- train_ctl_model.py:69-73 feeds features[isReal] only
__device__ __forceinline__ void center_reduce_body(const float* __restrict__ row_loss, int B, int C,
                                                            float* __restrict__ out, const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  __shared__ float s[4];
  __shared__ int sc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = masked_row_count(mask, B, sc);
  float l = 0.f;
  for (int b = tid; b < B; b += 256) l += (!mask || mask[b]) ? fminf(fmaxf(row_loss[b], 1e-12f), 1e12f) : 0.f;
  l = wave_sum(l);
  if (lane == 0) s[wave] = l;
  __syncthreads();
  if (tid == 0) out[0] = nb > 0 ? (((s[0] + s[1]) + (s[2] + s[3])) + (float)nb * (float)(C - 1) * 1e-12f) / (float)nb : 0.f;
}
__global__ __launch_bounds__(256) void center_reduce_kernel(const float* __restrict__ row_loss, int B, int C,
                                                            float* __restrict__ out, const uint8_t* __restrict__ mask) {
  center_reduce_body(row_loss, B, C, out, mask, (int)blockIdx.x, (int)blockIdx.y);
}

// dx[b] += g*(2/B)(x_b - c_y) ; dcenters[y] (+)= g*(2/B) sum_{b:y_b=y}(c_y - x_b), written once per
// distinct label by the workgroup of its FIRST occurrence (deterministic, no atomics).
__device__ __forceinline__ void center_bwd_body(const float* __restrict__ x,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ centers,
                                                         const float* __restrict__ row_loss, int B, int D,
                                                         const float* __restrict__ gscale_ptr, float gscale,
                                                         float* __restrict__ dx, float* __restrict__ dcenters,
                                                         const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  const int b = bx_;
  const int64_t y = labels[b];
  __shared__ int sc[4];
  const int nb = masked_row_count(mask, B, sc);
  if (mask && !mask[b]) return;                        // a padded (isReal = False) row: no loss term, owns no class
  const float g = gscale * (gscale_ptr ? *gscale_ptr : 1.f) * 2.0f / (float)nb;
  const bool on = row_loss[b] >= 1e-12f && row_loss[b] <= 1e12f;
  const float* c = centers + y * D;
  const float* xb = x + (int64_t)b * D;
  // members of this row's class (in batch order) are listed once in LDS instead of being re-discovered by every
  // thread for every feature: [0] = first occurrence decides which workgroup owns the class's center gradient
  __shared__ int s_members[1024];
  __shared__ unsigned char s_flag[1024];
  __shared__ int s_nmem, s_first;
  const bool listed = B <= 1024;
  if (listed) {
    for (int t = threadIdx.x; t < B; t += 256)
      s_flag[t] = (labels[t] == y && (!mask || mask[t])) ? ((row_loss[t] >= 1e-12f && row_loss[t] <= 1e12f) ? 2 : 1) : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      int n = 0, first = -1;
      for (int t = 0; t < B; ++t) {
        if (s_flag[t] && first < 0) first = t;
        if (s_flag[t] == 2) s_members[n++] = t;
      }
      s_nmem = n; s_first = first;
    }
    __syncthreads();
  }
  bool first = true;
  if (listed) first = s_first == b;
  else for (int t = 0; t < b; ++t) first = first && (labels[t] != y || (mask && !mask[t]));
  for (int d = threadIdx.x; d < D; d += 256) {
    const float cv = c[d];
    if (dx) dx[(int64_t)b * D + d] += on ? g * (xb[d] - cv) : 0.f;
    if (first && dcenters) {
      float acc = 0.f;
      if (listed) {
        for (int k = 0; k < s_nmem; ++k) acc += cv - x[(int64_t)s_members[k] * D + d];
      } else {
        for (int t = b; t < B; ++t)
          if (labels[t] == y && (!mask || mask[t]) && row_loss[t] >= 1e-12f && row_loss[t] <= 1e12f) acc += cv - x[(int64_t)t * D + d];
      }
      dcenters[y * D + d] += g * acc;
    }
  }
}
__global__ __launch_bounds__(256) void center_bwd_kernel(const float* __restrict__ x,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ centers,
                                                         const float* __restrict__ row_loss, int B, int D,
                                                         const float* __restrict__ gscale_ptr, float gscale,
                                                         float* __restrict__ dx, float* __restrict__ dcenters,
                                                         const uint8_t* __restrict__ mask) {
  center_bwd_body(x, labels, centers, row_loss, B, D, gscale_ptr, gscale, dx, dcenters, mask, (int)blockIdx.x, (int)blockIdx.y);
}

// ======================================================================================
// C6: label-smoothed cross entropy, fwd + bwd in one pass.  One workgroup per row.
// row_loss[b] = -sum_c t_c logp_c, t = (1-eps) onehot + eps/C ; dlogits = (softmax - t) * g / B.
// ======================================================================================
__device__ __forceinline__ void xent_ls_body(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ targets, int B, int C, float eps,
                                                      float* __restrict__ row_loss, float* __restrict__ dlogits,
                                                      float gscale, const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  __shared__ float s[4];
  __shared__ float bc;
  __shared__ int sc[4];
  const int b = bx_, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrows = masked_row_count(mask, B, sc);     // the mean is over the real rows (train_ctl_model.py:74-77)
  if (mask && !mask[b]) {
    if (tid == 0) row_loss[b] = 0.f;
    if (dlogits) for (int c = tid; c < C; c += 256) dlogits[(int64_t)b * C + c] = 0.f;
    return;
  }
  const float* z = logits + (int64_t)b * C;
  float mx = -INFINITY;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, z[c]);
  mx = wave_max(mx);
  if (lane == 0) s[wave] = mx;
  __syncthreads();
  if (tid == 0) bc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
  __syncthreads();
  mx = bc;
  float se = 0.f, sz = 0.f;
  for (int c = tid; c < C; c += 256) { se += expf(z[c] - mx); sz += z[c] - mx; }
  se = wave_sum(se);
  __syncthreads();
  if (lane == 0) s[wave] = se;
  __syncthreads();
  if (tid == 0) bc = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  se = bc;
  const float lse = logf(se);
  sz = wave_sum(sz);
  __syncthreads();
  if (lane == 0) s[wave] = sz;
  __syncthreads();
  const int64_t y = targets[b];
  if (tid == 0) {
    const float sum_logp = ((s[0] + s[1]) + (s[2] + s[3])) - (float)C * lse;   // sum_c (z_c - mx - lse)
    const float logp_y = z[y] - mx - lse;
    row_loss[b] = -((1.f - eps) * logp_y + (eps / (float)C) * sum_logp);
  }
  if (dlogits) {
    const float gb = gscale / (float)nrows;
    for (int c = tid; c < C; c += 256) {
      const float p = expf(z[c] - mx) / se;
      const float t = (c == y ? (1.f - eps) : 0.f) + eps / (float)C;
      dlogits[(int64_t)b * C + c] = (p - t) * gb;
    }
  }
}
__global__ __launch_bounds__(256) void xent_ls_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ targets, int B, int C, float eps,
                                                      float* __restrict__ row_loss, float* __restrict__ dlogits,
                                                      float gscale, const uint8_t* __restrict__ mask) {
  xent_ls_body(logits, targets, B, C, eps, row_loss, dlogits, gscale, mask, (int)blockIdx.x, (int)blockIdx.y);
}

__device__ __forceinline__ void mean_rows_body(const float* __restrict__ v, int n, float scale,
                                                        float* __restrict__ out, const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  __shared__ float s[4];
  __shared__ int sc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (mask) { const int c = masked_row_count(mask, n, sc); scale = c > 0 ? 1.0f / (float)c : 0.f; }   // v is 0 on masked rows
  float l = 0.f;
  for (int i = tid; i < n; i += 256) l += v[i];
  l = wave_sum(l);
  if (lane == 0) s[wave] = l;
  __syncthreads();
  if (tid == 0) out[0] = ((s[0] + s[1]) + (s[2] + s[3])) * scale;
}
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ v, int n, float scale,
                                                        float* __restrict__ out, const uint8_t* __restrict__ mask) {
  mean_rows_body(v, n, scale, out, mask, (int)blockIdx.x, (int)blockIdx.y);
}

// ======================================================================================
// A4: BNNeck = BatchNorm1d over [B, D] (B small): thread per feature, loop over rows (coalesced).
// ======================================================================================
// 32 channels x 8 row lanes per workgroup (the [B, D] problem is tiny: D/32 workgroups instead of D/256, B/8
// dependent steps per thread instead of B); row-lane partials meet in LDS in a fixed order.
__device__ __forceinline__ void bn1d_fwd_body(const float* __restrict__ x, int B, int D,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                                       int training, float momentum, float eps,
                                                       float* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  // mask (nullable, uint8 [B]): the batch the statistics are taken over -- the BNNeck sees features[isReal] only
  // (train_ctl_model.py:69-75); masked rows of y are written as 0
  __shared__ float red[8][32];
  __shared__ float s_mean[32], s_inv[32];
  __shared__ int sc[4];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int d = bx_ * 32 + cl;
  const bool live = d < D;
  const int nb = masked_row_count(mask, B, sc);
  if (training) {
    float s = 0.f;
    if (live) for (int b = rl; b < B; b += 8) s += (!mask || mask[b]) ? x[(int64_t)b * D + d] : 0.f;
    red[rl][cl] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mean += red[q][cl];
    mean /= (float)nb;
    __syncthreads();
    float m2 = 0.f;
    if (live) for (int b = rl; b < B; b += 8) { const float t = (!mask || mask[b]) ? x[(int64_t)b * D + d] - mean : 0.f; m2 = fmaf(t, t, m2); }
    red[rl][cl] = m2;
    __syncthreads();
    if (rl == 0 && live) {
      m2 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) m2 += red[q][cl];
      const float var = m2 / (float)nb;
      const float invstd = 1.0f / sqrtf(var + eps);
      if (rmean) rmean[d] = (1.f - momentum) * rmean[d] + momentum * mean;
      if (rvar) rvar[d] = (1.f - momentum) * rvar[d] + momentum * (nb > 1 ? m2 / (float)(nb - 1) : var);
      if (save_mean) { save_mean[d] = mean; save_invstd[d] = invstd; }
      s_mean[cl] = mean; s_inv[cl] = invstd;
    }
  } else if (rl == 0 && live) {
    s_mean[cl] = rmean[d];
    s_inv[cl] = 1.0f / sqrtf(rvar[d] + eps);
  }
  __syncthreads();
  if (!live) return;
  const float mean = s_mean[cl], invstd = s_inv[cl];
  const float g = w ? w[d] : 1.f, be = bias ? bias[d] : 0.f;
  for (int b = rl; b < B; b += 8)
    y[(int64_t)b * D + d] = (!mask || mask[b]) ? (x[(int64_t)b * D + d] - mean) * invstd * g + be : 0.f;
}
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, int B, int D,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                                       int training, float momentum, float eps,
                                                       float* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, const uint8_t* __restrict__ mask) {
  bn1d_fwd_body(x, B, D, w, bias, rmean, rvar, training, momentum, eps, y, save_mean, save_invstd, mask, (int)blockIdx.x, (int)blockIdx.y);
}

__device__ __forceinline__ void bn1d_bwd_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                       int B, int D, const float* __restrict__ w,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd,
                                                       float* __restrict__ dx, float* __restrict__ dw,
                                                       float* __restrict__ dbias, const uint8_t* __restrict__ mask, const int bx_, const int by_) {
  __shared__ float red[8][2][32];
  __shared__ int sc[4];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int d = bx_ * 32 + cl;
  const bool live = d < D;
  const int nb = masked_row_count(mask, B, sc);
  const float mean = live ? save_mean[d] : 0.f, invstd = live ? save_invstd[d] : 0.f, g = (live && w) ? w[d] : 1.f;
  float sdy = 0.f, sdyx = 0.f;
  if (live)
    for (int b = rl; b < B; b += 8) {
      if (mask && !mask[b]) continue;
      const float t = dy[(int64_t)b * D + d];
      sdy += t; sdyx = fmaf(t, (x[(int64_t)b * D + d] - mean) * invstd, sdyx);
    }
  red[rl][0][cl] = sdy; red[rl][1][cl] = sdyx;
  __syncthreads();
  sdy = 0.f; sdyx = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { sdy += red[q][0][cl]; sdyx += red[q][1][cl]; }
  if (!live) return;
  if (rl == 0) {
    if (dw) dw[d] += sdyx;
    if (dbias) dbias[d] += sdy;
  }
  const float k = g * invstd / (float)nb;
  for (int b = rl; b < B; b += 8) {
    if (mask && !mask[b]) continue;
    const float xh = (x[(int64_t)b * D + d] - mean) * invstd;
    dx[(int64_t)b * D + d] += k * ((float)nb * dy[(int64_t)b * D + d] - sdy - xh * sdyx);
  }
}
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       int B, int D, const float* __restrict__ w,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd,
                                                       float* __restrict__ dx, float* __restrict__ dw,
                                                       float* __restrict__ dbias, const uint8_t* __restrict__ mask) {
  bn1d_bwd_body(x, dy, B, D, w, save_mean, save_invstd, dx, dw, dbias, mask, (int)blockIdx.x, (int)blockIdx.y);
}

// ======================================================================================
// C5 + optimiser: Adam with L2 weight decay (torch.optim.Adam) over a flat fp32 buffer; center SGD.
// ======================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                   float gscale) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i = i0; i < n; i += stride) {
    if (i + 3 < n) {
      float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = fmaf(wd, pp[k], gp[k] * gscale);
        mp[k] = b1 * mp[k] + (1.f - b1) * gg;
        vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vp[k]) / bc2s + eps;
        pp[k] = pp[k] - (lr / bc1) * (mp[k] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        const float gg = fmaf(wd, p[j], g[j] * gscale);
        m[j] = b1 * m[j] + (1.f - b1) * gg;
        v[j] = b2 * v[j] + (1.f - b2) * gg * gg;
        p[j] = p[j] - (lr / bc1) * (m[j] / (sqrtf(v[j]) / bc2s + eps));
      }
    }
  }
}

// device-resident hyper-parameters {lr, step, bc1, bc2s}: lets a captured hipGraph replay the step with
// the correct bias correction / learning rate (host scalars would be frozen into the graph).
__global__ void adam_advance_kernel(float* __restrict__ hyper, float b1, float b2) {
  const float step = hyper[1] + 1.0f;
  hyper[1] = step;
  hyper[2] = 1.0f - powf(b1, step);
  hyper[3] = sqrtf(1.0f - powf(b2, step));
}

// The kernel ADVANCES the device-resident step itself: every workgroup derives step = hyper[1] + 1 and the bias corrections
// from the not-yet-advanced counter, and the LAST workgroup to finish (a ticket in hyper[4]: by then every other one has read
// hyper[1]) writes {step, bc1, bc2s} back and clears the ticket -- the separate one-thread launch in front of every Adam step
// (adam_advance_kernel, ~5 us of dependent launch) is gone.  A skipped (overflow) step returns before the ticket: the counter
// does not advance (GradScaler.step skips optimizer.step).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       float* __restrict__ hyper, float b1, float b2, float eps,
                                                       float wd, float gscale, const int32_t* __restrict__ skip = nullptr) {
  if (skip && skip[0]) return;                                // f16 training: a non-finite gradient skips the whole step
  const float step = hyper[1] + 1.0f;
  const float lr = hyper[0], bc1 = 1.0f - powf(b1, step), bc2s = sqrtf(1.0f - powf(b2, step));
  const float step_size = lr / bc1;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i = i0; i + 3 < n; i += stride) {      // n is padded to a multiple of 4 by the caller
    float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
    float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = fmaf(wd, pp[k], gp[k] * gscale);
      mp[k] = b1 * mp[k] + (1.f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
      pp[k] = pp[k] - step_size * (mp[k] / (sqrtf(vp[k]) / bc2s + eps));
    }
    *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv;
    *reinterpret_cast<float4*>(v + i) = vv;
  }
  __syncthreads();                                            // every thread of this workgroup has read hyper[1]
  if (threadIdx.x == 0) {
    int* ticket = reinterpret_cast<int*>(hyper + 4);
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      hyper[1] = step; hyper[2] = bc1; hyper[3] = bc2s;
      *ticket = 0;
    }
  }
}

__global__ __launch_bounds__(256) void sgd_scaled_kernel(float* __restrict__ p, float* __restrict__ g, int64_t n,
                                                         float lr, float gmul, const int32_t* __restrict__ skip = nullptr) {
  if (skip && skip[0]) return;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gv = g[i] * gmul;     // train_ctl_model.py:157-158 rescale (kept in .grad like the reference)
    g[i] = gv;
    p[i] = p[i] - lr * gv;
  }
}


// ======================================================================================
// f16 mixed-precision training (the reference's precision=16, utils/misc.py:111): dynamic loss scale RESIDENT ON THE DEVICE with
// torch.cuda.amp.GradScaler's rule (x backoff on a non-finite gradient, x growth after `interval` clean steps), so that a
// captured hipGraph of the step replays correctly through overflow steps -- no host synchronisation anywhere.
//   amp_state = float[2] {scale, 1 / scale};  amp_flags = int32[3] {found_inf of the current step, clean steps in a row, steps skipped in total}
// ======================================================================================
__global__ __launch_bounds__(256) void amp_scale_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ state,
                                                        float* __restrict__ y) {
  const float s = state[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * s;
}

// g *= 1 / scale in place; any non-finite element raises flags[0] (the step is then skipped by the *_amp optimiser kernels)
__global__ __launch_bounds__(256) void amp_unscale_check_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ state,
                                                                int32_t* __restrict__ flags) {
  const float inv = state[1];
  bool bad = false;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  for (int64_t i = i0; i + 3 < n; i += stride) {             // n % 4 == 0 (flat buffer offsets are 16-byte aligned)
    float4 v = *reinterpret_cast<float4*>(g + i);
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    // (x - x) is 0 for finite x and NaN for +-inf / NaN: one test per element without the class intrinsics
    bad |= !((v.x - v.x) == 0.f) | !((v.y - v.y) == 0.f) | !((v.z - v.z) == 0.f) | !((v.w - v.w) == 0.f);
    *reinterpret_cast<float4*>(g + i) = v;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

__global__ void amp_update_kernel(float* __restrict__ state, int32_t* __restrict__ flags, float growth, float backoff,
                                  int interval) {
  float s = state[0];
  if (flags[0]) { s *= backoff; flags[1] = 0; flags[2] += 1; }   // flags[2]: steps skipped so far (host-visible: a scale pinned at 1 with every step skipped must not pass unnoticed)
  else if (++flags[1] >= interval) { s *= growth; flags[1] = 0; }
  s = fminf(fmaxf(s, 1.0f), 16777216.0f);                     // keep 1 / scale and scale * gradient representable
  state[0] = s; state[1] = 1.0f / s;
  flags[0] = 0;
}

__global__ void adam_advance_amp_kernel(float* __restrict__ hyper, float b1, float b2, const int32_t* __restrict__ skip) {
  if (skip[0]) return;                                        // an overflow step does not count (GradScaler.step skips optimizer.step)
  const float step = hyper[1] + 1.0f;
  hyper[1] = step;
  hyper[2] = 1.0f - powf(b1, step);
  hyper[3] = sqrtf(1.0f - powf(b2, step));
}


// ======================================================================================
// D2: per-PID gallery centroids for evaluation (modelling/bases.py:92-95,238-241):
// out[s,:] = sum_{j in [off[s], off[s+1])} emb[order[j],:] / count   (rows summed in list order)
// ======================================================================================
__global__ __launch_bounds__(256) void gather_mean_rows_kernel(const float* __restrict__ emb,
                                                               const int64_t* __restrict__ order,
                                                               const int64_t* __restrict__ offsets, int D,
                                                               float* __restrict__ out) {
  const int s = blockIdx.x;
  const int64_t j0 = offsets[s], j1 = offsets[s + 1];
  const float cnt = (float)(j1 - j0);
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int64_t j = j0; j < j1; ++j) acc += emb[order[j] * D + d];
    out[(int64_t)s * D + d] = acc / cnt;
  }
}

// ======================================================================================
// Row scaling to unit length with its backward (losses/triplet_loss.py:16-24 `normalize`, :44-54 the two
// normalisations inside cosine_similarity).  mode 0: y = x / max(|x|, eps); mode 1: y = x / (|x| + eps).
// One workgroup per row.
// ======================================================================================
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, int D, int mode, float eps,
                                                          float* __restrict__ y, float* __restrict__ norm) {
  __shared__ float sh[4];
  const float* xr = x + (int64_t)blockIdx.x * D;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s = fmaf(xr[d], xr[d], s);
  const float n = sqrtf(block_sum_256(s, sh));
  const float den = mode == 0 ? fmaxf(n, eps) : n + eps;
  for (int d = threadIdx.x; d < D; d += 256) y[(int64_t)blockIdx.x * D + d] = xr[d] / den;
  if (threadIdx.x == 0) norm[blockIdx.x] = n;
}

__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                          const float* __restrict__ dy, int D, int mode, float eps,
                                                          float* __restrict__ dx) {
  __shared__ float sh[4];
  const float* xr = x + (int64_t)blockIdx.x * D;
  const float* gr = dy + (int64_t)blockIdx.x * D;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s = fmaf(xr[d], gr[d], s);
  const float xg = block_sum_256(s, sh);
  const float n = norm[blockIdx.x];
  float a, b;                                   // dx = a * dy - b * x
  if (mode == 0) {
    if (n > eps) { a = 1.f / n; b = xg / (n * n * n); } else { a = 1.f / eps; b = 0.f; }
  } else {
    const float den = n + eps;
    a = 1.f / den; b = n > 0.f ? xg / (n * den * den) : 0.f;
  }
  for (int d = threadIdx.x; d < D; d += 256) dx[(int64_t)blockIdx.x * D + d] = a * gr[d] - b * xr[d];
}

// hard_example_mining on a GIVEN distance matrix (losses/triplet_loss.py:68-119): wave per anchor row.
__global__ __launch_bounds__(64) void mine_from_dist_kernel(const float* __restrict__ dist, const int64_t* __restrict__ labels,
                                                            int N, float* __restrict__ dist_ap, float* __restrict__ dist_an,
                                                            int32_t* __restrict__ p_idx, int32_t* __restrict__ n_idx) {
  const int a = blockIdx.x, lane = threadIdx.x;
  const int64_t la = labels[a];
  float bp = -INFINITY, bn = INFINITY;
  int ip = 0x7fffffff, in_ = 0x7fffffff;
  for (int j = lane; j < N; j += 64) {
    const float d = dist[(int64_t)a * N + j];
    if (labels[j] == la) { if (d > bp) { bp = d; ip = j; } }
    else { if (d < bn) { bn = d; in_ = j; } }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float obp = __shfl_xor(bp, o, 64); int oip = __shfl_xor(ip, o, 64);
    if (obp > bp || (obp == bp && oip < ip)) { bp = obp; ip = oip; }
    float obn = __shfl_xor(bn, o, 64); int oin = __shfl_xor(in_, o, 64);
    if (obn < bn || (obn == bn && oin < in_)) { bn = obn; in_ = oin; }
  }
  if (lane == 0) { dist_ap[a] = bp; dist_an[a] = bn; p_idx[a] = ip; n_idx[a] = in_; }
}

// d <- sqrt(max(d, lo)) in place (the tail of euclidean_dist, triplet_loss.py:40)
__global__ __launch_bounds__(256) void clamp_sqrt_kernel(float* __restrict__ d, int64_t n, float lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    d[i] = sqrtf(fmaxf(d[i], lo));
}

// ---- the centroid rounds' operands in one pass (train_ctl_model.py:79-124): emb[i] = cat(queries of round i, centroids of
// round i) = [K][2P][D], lab[i] = labels of the P identities twice.  grid (P, K): workgroup (p, i) writes the query row
// emb[i][p] = feat[p][i], the leave-one-out centroid emb[i][P + p] (same s-order sum as loo_centroids_fwd_kernel, also to
// cent[i][p]) and the two label slots -- what torch did with two strided copies, a cat and a contiguous().
__device__ __forceinline__ void loo_emb_fwd_body(const float* __restrict__ feat, const uint8_t* __restrict__ is_real,
                                                          const int64_t* __restrict__ labels, int P, int K, int D,
                                                          float* __restrict__ cent, int32_t* __restrict__ valid,
                                                          float* __restrict__ emb, int64_t* __restrict__ lab,
                                                          float* __restrict__ cnorm, uint8_t* __restrict__ exists,
                                                          int32_t* __restrict__ lonely, const int bx_, const int by_) {
  __shared__ float wsum[4];
  const int p = bx_, i = by_;
  const bool qreal = is_real[p * K + i] != 0;
  int cnt = 0;
  if (qreal)
    for (int s = 0; s < K; ++s) cnt += (s != i && is_real[p * K + s]) ? 1 : 0;
  if (threadIdx.x == 0) {
    // exists (nullable, uint8 [K][2P]): identity p takes part in round i -- its i-th instance is real (the query) AND it has
    // another real instance (a non-zero centroid); train_ctl_model.py:112-122 keeps exactly these rows
    if (exists) { const uint8_t e = (qreal && cnt > 0) ? 1 : 0; exists[i * 2 * P + p] = e; exists[i * 2 * P + P + p] = e; }
    // a real instance without a real partner: the reference fails on such a batch (labels.expand, losses/triplet_loss.py:88);
    // a step driven by a DEVICE mask cannot raise without a host sync, so it counts them and the epoch end raises
    if (lonely && qreal && cnt == 0) atomicAdd(lonely, 1);
    valid[i * P + p] = cnt;
    const int64_t l = labels[p * K + i];
    lab[(int64_t)i * 2 * P + p] = l;
    lab[(int64_t)i * 2 * P + P + p] = l;
  }
  const float den = (float)max(cnt, 1);
  float* oc = cent + ((int64_t)i * P + p) * D;
  float* eq = emb + ((int64_t)i * 2 * P + p) * D;
  float* ec = emb + ((int64_t)i * 2 * P + P + p) * D;
  const float* fq = feat + ((int64_t)p * K + i) * D;
  float nrm = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    if (qreal)
      for (int s = 0; s < K; ++s)
        if (s != i && is_real[p * K + s]) acc += feat[((int64_t)p * K + s) * D + d];
    const float c = acc / den;
    oc[d] = c; ec[d] = c;
    eq[d] = fq[d];
    nrm = fmaf(c, c, nrm);
  }
  nrm = wave_sum(nrm);                                            // L2 norm of the centroid row (logged as l2_mean_centroid)
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = nrm;
  __syncthreads();
  if (threadIdx.x == 0) cnorm[i * P + p] = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}
__global__ __launch_bounds__(256) void loo_emb_fwd_kernel(const float* __restrict__ feat, const uint8_t* __restrict__ is_real,
                                                          const int64_t* __restrict__ labels, int P, int K, int D,
                                                          float* __restrict__ cent, int32_t* __restrict__ valid,
                                                          float* __restrict__ emb, int64_t* __restrict__ lab,
                                                          float* __restrict__ cnorm, uint8_t* __restrict__ exists,
                                                          int32_t* __restrict__ lonely = nullptr) {
  loo_emb_fwd_body(feat, is_real, labels, P, K, D, cent, valid, emb, lab, cnorm, exists, lonely, (int)blockIdx.x, (int)blockIdx.y);
}

// dfeat[p][s] += demb[s][p]  (the round's query rows)  +  sum_{i != s, real} demb[i][P + p] / max(cnt_i, 1)  (its share of the
// other rounds' centroids) -- in that order, i.e. the torch add_ followed by loo_centroids_bwd_kernel.  grid (P, K = s).
__device__ __forceinline__ void loo_emb_bwd_body(const float* __restrict__ demb, const uint8_t* __restrict__ is_real,
                                                          int P, int K, int D, float* __restrict__ dfeat, const int bx_, const int by_) {
  const int p = bx_, s = by_;
  float* out = dfeat + ((int64_t)p * K + s) * D;
  const float* dq = demb + ((int64_t)s * 2 * P + p) * D;
  const bool sreal = is_real[p * K + s] != 0;
  // per round i: divisor max(cnt_i, 1), or 0 when round i contributes nothing to slot s -- once per thread, not per element
  constexpr int KMAX = 16;
  float den[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    den[i] = 0.f;
    if (i < K && i != s && sreal && is_real[p * K + i]) {
      int cnt = 0;
      for (int t = 0; t < K; ++t) cnt += (t != i && is_real[p * K + t]) ? 1 : 0;
      den[i] = (float)max(cnt, 1);
    }
  }
  for (int d = threadIdx.x; d < D; d += 256) {
    float o = out[d] + dq[d];
    if (sreal) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        if (den[i] != 0.f) acc += demb[((int64_t)i * 2 * P + P + p) * D + d] / den[i];
      o += acc;
    }
    out[d] = o;
  }
}
__global__ __launch_bounds__(256) void loo_emb_bwd_kernel(const float* __restrict__ demb, const uint8_t* __restrict__ is_real,
                                                          int P, int K, int D, float* __restrict__ dfeat) {
  loo_emb_bwd_body(demb, is_real, P, K, D, dfeat, (int)blockIdx.x, (int)blockIdx.y);
}

// Scalars of one training step (train_ctl_model.py:143-177) in one launch: terms = scal * w, total = sum(terms),
// step = sum of the K round losses (terms[4], terms[8], ...), rounds = mean over the K rows of out4[1:], l2 = mean of the
// centroid row norms (loo_emb_fwd_kernel's cnorm).  out: [n] terms, then {total, step, rounds[0..3], l2}.  One wave.
__device__ __forceinline__ void ctl_step_stats_body(const float* __restrict__ scal, const float* __restrict__ w, int n,
                                                            int K, const float* __restrict__ cnorm, int rows,
                                                            float* __restrict__ out, const int bx_, const int by_) {
  const int lane = threadIdx.x;
  float l2 = 0.f;
  for (int r = lane; r < rows; r += 64) l2 += cnorm[r];
  l2 = wave_sum(l2);
  if (lane == 0) {
    float total = 0.f, step = 0.f;
    for (int i = 0; i < n; ++i) {
      const float t = scal[i] * w[i];
      out[i] = t;
      total += t;
      if (i >= 4 && i < 4 * (K + 1) && (i & 3) == 0) step += t;
    }
    out[n] = total; out[n + 1] = step;
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
      for (int k = 1; k <= K; ++k) a += scal[4 * k + c];
      out[n + 2 + c] = a / (float)K;
    }
    out[n + 6] = l2 / (float)rows;
  }
}
__global__ __launch_bounds__(64) void ctl_step_stats_kernel(const float* __restrict__ scal, const float* __restrict__ w, int n,
                                                            int K, const float* __restrict__ cnorm, int rows,
                                                            float* __restrict__ out) {
  ctl_step_stats_body(scal, w, n, K, cnorm, rows, out, (int)blockIdx.x, (int)blockIdx.y);
}

// Batches with padded (isReal = False) samples: a centroid round counts only if it kept >= 2 identities (out4 slot 3 = its
// number of anchors = 2 x identities; triplet_loss_kernel zeroed the round otherwise).  inv_rounds[0] = 1 / #valid rounds
// (0 if none) -- the device-side factor of the rounds' backward (train_ctl_model.py:143-146: mean over the valid rounds).
__device__ __forceinline__ void ctl_round_scale_body(const float* __restrict__ out4_rounds, int K, float* __restrict__ inv_rounds, const int bx_, const int by_) {
  int nv = 0;
  for (int k = 0; k < K; ++k) nv += out4_rounds[4 * k + 3] >= 4.f ? 1 : 0;
  inv_rounds[0] = nv > 0 ? 1.0f / (float)nv : 0.f;
}
__global__ void ctl_round_scale_kernel(const float* __restrict__ out4_rounds, int K, float* __restrict__ inv_rounds) {
  ctl_round_scale_body(out4_rounds, K, inv_rounds, (int)blockIdx.x, (int)blockIdx.y);
}

// ctl_step_stats_kernel for such batches: w[4k] (round losses) must hold the FULL centroid weight (not weight / K); the round
// means divide by the number of valid rounds; l2 = mean over valid rounds of the mean centroid norm of the round's kept rows.
__device__ __forceinline__ void ctl_step_stats_rows_body(const float* __restrict__ scal, const float* __restrict__ w, int n,
                                                                 int K, int P, const float* __restrict__ cnorm,
                                                                 const uint8_t* __restrict__ exists, float* __restrict__ out, const int bx_, const int by_) {
  if (threadIdx.x != 0) return;
  int nv = 0;
  for (int k = 1; k <= K; ++k) nv += scal[4 * k + 3] >= 4.f ? 1 : 0;
  const float inv = nv > 0 ? 1.0f / (float)nv : 0.f;
  float total = 0.f, step = 0.f;
  for (int i = 0; i < n; ++i) {
    const bool round_loss = i >= 4 && i < 4 * (K + 1) && (i & 3) == 0;
    const float t = scal[i] * w[i] * (round_loss ? inv : 1.f);
    out[i] = t;
    total += t;
    if (round_loss) step += t;
  }
  out[n] = total; out[n + 1] = step;
  for (int c = 0; c < 4; ++c) {
    float a = 0.f;
    for (int k = 1; k <= K; ++k) a += scal[4 * k + 3] >= 4.f ? scal[4 * k + c] : 0.f;
    out[n + 2 + c] = a * inv;
  }
  float l2 = 0.f;
  for (int k = 0; k < K; ++k) {
    if (scal[4 * (k + 1) + 3] < 4.f) continue;
    float a = 0.f; int c = 0;
    for (int p = 0; p < P; ++p)
      if (exists[k * 2 * P + P + p]) { a += cnorm[k * P + p]; ++c; }
    l2 += a / (float)c;
  }
  out[n + 6] = l2 * inv;
}
__global__ __launch_bounds__(64) void ctl_step_stats_rows_kernel(const float* __restrict__ scal, const float* __restrict__ w, int n,
                                                                 int K, int P, const float* __restrict__ cnorm,
                                                                 const uint8_t* __restrict__ exists, float* __restrict__ out) {
  ctl_step_stats_rows_body(scal, w, n, K, P, cnorm, exists, out, (int)blockIdx.x, (int)blockIdx.y);
}

// ======================================================================================
// The whole head section of one training step (train_ctl_model.py:59-152 between the backbone forward and its backward) as SIX
// launches.  Every kernel above is latency-bound (64 x 2048 features): run back to back, the 23 launches of the hand-scheduled
// step cost ~205 us at B = 64, almost all of it dependent-launch latency.  Four chains are independent of each other -- query
// triplet, center loss, BNNeck -> classifier -> cross entropy, centroid rounds -- so each STAGE below is one launch whose
// workgroups take ROLES (the bodies above, by block index); a stage boundary is a real dependency.  The accumulation order into
// dfeat is the sequential schedule's (query triplet, center, BNNeck, centroid rounds): results are bit-identical to it.
//   S1 zero the accumulators | mine (query) | center rows | BNNeck forward | leave-one-out centroids + round operands
//   S2 classifier forward | query triplet loss | center reduce | mine (K rounds)
//   S3 cross entropy (+ dlogits) | query triplet backward -> dfeat | round losses
//   S4 classifier dgrad | classifier wgrad | center backward -> dfeat | cross-entropy mean | rounds backward (unmasked) / round count (masked)
//   S5 BNNeck backward -> dfeat | step scalars | rounds backward (masked)
//   S6 leave-one-out backward -> dfeat, and the global-average-pool backward of the finished dfeat rows (the backbone's incoming
//      gradient [B * HW, D] in its compute dtype, times the f16 loss scale if one is given)
// ======================================================================================
struct HeadsWs {                                       // carved out of creid_ctl_heads.workspace (see heads_carve)
  float *dfeat, *demb, *logits, *dbnf;                 // zeroed by S1 (one contiguous region of zero_count floats)
  int64_t zero_count;
  float *dlogits, *bnf, *save_mean, *save_invstd, *row_c, *row_x, *scal, *inv_rounds;
  float *dap_q, *dan_q, *coef_q, *dap_r, *dan_r, *coef_r;
  int32_t *pi_q, *ni_q, *pi_r, *ni_r, *valid;
  float *cent, *emb, *cnorm;
  int64_t* lab;
  uint8_t* rows;
};

static size_t heads_carve(int64_t B, int64_t P, int64_t K, int64_t D, int64_t C, char* base, HeadsWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  const int64_t R = K * 2 * P;
  // the zeroed region: sizes rounded so that every member starts 16-byte aligned and the region is one float4 sweep
  const int64_t n_dfeat = B * D, n_demb = R * D, n_logits = (B * C + 3) / 4 * 4, n_dbnf = B * D;
  float* z = (float*)take((size_t)(n_dfeat + n_demb + n_logits + n_dbnf) * 4);
  HeadsWs t;
  t.dfeat = z; t.demb = z ? z + n_dfeat : nullptr; t.logits = z ? z + n_dfeat + n_demb : nullptr;
  t.dbnf = z ? z + n_dfeat + n_demb + n_logits : nullptr;
  t.zero_count = n_dfeat + n_demb + n_logits + n_dbnf;
  t.dlogits = (float*)take((size_t)B * C * 4); t.bnf = (float*)take((size_t)B * D * 4);
  t.save_mean = (float*)take((size_t)D * 4); t.save_invstd = (float*)take((size_t)D * 4);
  t.row_c = (float*)take((size_t)B * 4); t.row_x = (float*)take((size_t)B * 4);
  t.scal = (float*)take((size_t)(4 * (K + 1) + 2) * 4); t.inv_rounds = (float*)take(4);
  t.dap_q = (float*)take((size_t)B * 4); t.dan_q = (float*)take((size_t)B * 4); t.coef_q = (float*)take((size_t)B * 4);
  t.pi_q = (int32_t*)take((size_t)B * 4); t.ni_q = (int32_t*)take((size_t)B * 4);
  t.dap_r = (float*)take((size_t)R * 4); t.dan_r = (float*)take((size_t)R * 4); t.coef_r = (float*)take((size_t)R * 4);
  t.pi_r = (int32_t*)take((size_t)R * 4); t.ni_r = (int32_t*)take((size_t)R * 4);
  t.cent = (float*)take((size_t)K * P * D * 4); t.valid = (int32_t*)take((size_t)K * P * 4);
  t.emb = (float*)take((size_t)R * D * 4); t.lab = (int64_t*)take((size_t)R * 8);
  t.cnorm = (float*)take((size_t)K * P * 4); t.rows = (uint8_t*)take((size_t)R);
  if (w) *w = t;
  return off;
}

struct HeadsCtx {
  creid_ctl_heads a;
  HeadsWs w;
  const uint8_t* mask;                                 // a.is_real on the masked schedule, NULL otherwise
  int gl_tn, gl_tm, gd_tn, gd_tm, gw_tn, gw_tm;        // 64 x 64 output tiles of the three classifier GEMMs
};

constexpr int HEADS_ZERO_WGS = 48;

__global__ __launch_bounds__(1024) void heads_stage1_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int B = (int)a.B, P = (int)a.P, K = (int)a.K, D = (int)a.D;
  int b = (int)blockIdx.x;
  if (b < HEADS_ZERO_WGS) {
    float4* z = reinterpret_cast<float4*>(c.w.dfeat);
    const int64_t n4 = c.w.zero_count / 4;
    for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < n4; i += (int64_t)HEADS_ZERO_WGS * 1024) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  b -= HEADS_ZERO_WGS;
  if (b < B) {                                         // query triplet: distances of anchor b to every row + hardest pos / neg
    triplet_mine_body<0>(a.feat, a.labels, B, D, c.w.dap_q, c.w.dan_q, c.w.pi_q, c.w.ni_q, nullptr, nullptr, b, 0);
    return;
  }
  b -= B;
  if (threadIdx.x >= 256) return;                      // the remaining roles are 256-thread workgroups
  if (b < B) { center_row_body(a.feat, a.labels, a.centers, D, c.w.row_c, b, 0); return; }
  b -= B;
  const int nbn = (D + 31) / 32;
  if (b < nbn) {
    if (b == 0 && threadIdx.x == 0 && a.bn_batches_tracked) *a.bn_batches_tracked += 1;
    bn1d_fwd_body(a.feat, B, D, a.bn_weight, a.bn_bias, a.bn_running_mean, a.bn_running_var, 1, a.bn_momentum, a.bn_eps, c.w.bnf,
                  c.w.save_mean, c.w.save_invstd, c.mask, b, 0);
    return;
  }
  b -= nbn;
  loo_emb_fwd_body(a.feat, a.is_real, a.labels, P, K, D, c.w.cent, c.w.valid, c.w.emb, c.w.lab, c.w.cnorm,
                   a.masked ? c.w.rows : nullptr, a.masked ? a.lonely : nullptr, b % P, b / P);
}

__global__ __launch_bounds__(1024) void heads_stage2_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int B = (int)a.B, P = (int)a.P, K = (int)a.K, D = (int)a.D, C = (int)a.num_classes;
  int b = (int)blockIdx.x;
  const int R = 2 * P;
  if (b < K * R) {                                     // the K centroid rounds: anchor b % 2P of round b / 2P
    triplet_mine_body<0>(c.w.emb, c.w.lab, R, D, c.w.dap_r, c.w.dan_r, c.w.pi_r, c.w.ni_r, nullptr,
                         a.masked ? c.w.rows : nullptr, b % R, b / R);
    return;
  }
  b -= K * R;
  if (threadIdx.x >= 256) return;
  const int ngl = c.gl_tn * c.gl_tm * a.split_logits;
  if (b < ngl) {                                       // logits = bnf . W^T   (modelling/bases.py:86 fc_query)
    gemm_f32_body(c.w.bnf, D, 1, a.fc_weight, 1, D, c.w.logits, C, B, C, D, 1.f, 0.f, a.split_logits, b % c.gl_tn,
                  (b / c.gl_tn) % c.gl_tm, b / (c.gl_tn * c.gl_tm));
    return;
  }
  b -= ngl;
  if (b == 0) { triplet_loss_body(c.w.dap_q, c.w.dan_q, c.mask, B, a.margin, c.w.scal, c.w.coef_q, 0, 0, 0); return; }
  center_reduce_body(c.w.row_c, B, (int)a.num_centers, c.w.scal + 4 * (K + 1), c.mask, 0, 0);
}

__global__ __launch_bounds__(256) void heads_stage3_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int B = (int)a.B, P = (int)a.P, D = (int)a.D, C = (int)a.num_classes;
  int b = (int)blockIdx.x;
  if (b < B) { xent_ls_body(c.w.logits, a.labels, B, C, a.xent_eps, c.w.row_x, c.w.dlogits, a.w_xent, c.mask, b, 0); return; }
  b -= B;
  if (b < B) {
    triplet_bwd_body<0>(a.feat, B, D, c.w.dap_q, c.w.dan_q, c.w.pi_q, c.w.ni_q, c.w.coef_q, nullptr, a.w_query, c.w.dfeat, b, 0);
    return;
  }
  b -= B;                                              // round b: loss over its (kept) anchors
  triplet_loss_body(c.w.dap_r, c.w.dan_r, a.masked ? c.w.rows : nullptr, 2 * P, a.margin, c.w.scal + 4, c.w.coef_r,
                    a.masked ? 4 : 0, 0, b);
}

__global__ __launch_bounds__(256) void heads_stage4_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int B = (int)a.B, P = (int)a.P, K = (int)a.K, D = (int)a.D, C = (int)a.num_classes;
  int b = (int)blockIdx.x;
  const int ngd = c.gd_tn * c.gd_tm * a.split_dbnf;
  if (b < ngd) {                                       // dbnf = dlogits . W
    gemm_f32_body(c.w.dlogits, C, 1, a.fc_weight, D, 1, c.w.dbnf, D, B, D, C, 1.f, 0.f, a.split_dbnf, b % c.gd_tn,
                  (b / c.gd_tn) % c.gd_tm, b / (c.gd_tn * c.gd_tm));
    return;
  }
  b -= ngd;
  const int ngw = a.d_fc_weight ? c.gw_tn * c.gw_tm : 0;
  if (b < ngw) {                                       // dW += dlogits^T . bnf
    gemm_f32_body(c.w.dlogits, 1, C, c.w.bnf, D, 1, a.d_fc_weight, D, C, D, B, 1.f, 1.f, 1, b % c.gw_tn, b / c.gw_tn, 0);
    return;
  }
  b -= ngw;
  if (b < B) {
    center_bwd_body(a.feat, a.labels, a.centers, c.w.row_c, B, D, nullptr, a.w_center, c.w.dfeat, a.d_centers, c.mask, b, 0);
    return;
  }
  b -= B;
  if (b == 0) {
    mean_rows_body(c.w.row_x, B, a.masked ? 0.f : 1.0f / (float)B, c.w.scal + 4 * (K + 1) + 1, c.mask, 0, 0);
    return;
  }
  b -= 1;
  if (a.masked) { if (threadIdx.x == 0) ctl_round_scale_body(c.w.scal + 4, K, c.w.inv_rounds, 0, 0); return; }
  const int R = 2 * P;
  triplet_bwd_body<0>(c.w.emb, R, D, c.w.dap_r, c.w.dan_r, c.w.pi_r, c.w.ni_r, c.w.coef_r, nullptr, a.w_centroid / (float)K,
                      c.w.demb, b % R, b / R);
}

__global__ __launch_bounds__(256) void heads_stage5_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int B = (int)a.B, P = (int)a.P, K = (int)a.K, D = (int)a.D;
  int b = (int)blockIdx.x;
  const int nbn = (D + 31) / 32;
  if (b < nbn) {
    bn1d_bwd_body(a.feat, c.w.dbnf, B, D, a.bn_weight, c.w.save_mean, c.w.save_invstd, c.w.dfeat, a.d_bn_weight, a.d_bn_bias,
                  c.mask, b, 0);
    return;
  }
  b -= nbn;
  const int n = 4 * (K + 1) + 2;
  if (b == 0) {
    if (threadIdx.x >= 64) return;
    if (a.masked) ctl_step_stats_rows_body(c.w.scal, a.loss_weights, n, K, P, c.w.cnorm, c.w.rows, a.stats, 0, 0);
    else ctl_step_stats_body(c.w.scal, a.loss_weights, n, K, c.w.cnorm, K * P, a.stats, 0, 0);
    return;
  }
  b -= 1;                                              // (masked schedule only: the grid has no such blocks otherwise)
  const int R = 2 * P;
  triplet_bwd_body<0>(c.w.emb, R, D, c.w.dap_r, c.w.dan_r, c.w.pi_r, c.w.ni_r, c.w.coef_r, c.w.inv_rounds, a.w_centroid, c.w.demb,
                      b % R, b / R);
}

// S6: grid (B, hw_parts).  Every workgroup of image b re-derives the finished dfeat row (three reads per channel from L2) and
// writes its share of the HW copies of it / HW: creid_loo_emb_bwd + (creid_amp_scale) + creid_gap_bwd of the sequential schedule,
// same expressions in the same order.
template <int DT>
__global__ __launch_bounds__(256) void heads_stage6_kernel(HeadsCtx c, int hw_parts) {
  const creid_ctl_heads& a = c.a;
  const int P = (int)a.P, K = (int)a.K, D = (int)a.D, HW = (int)a.HW;
  const int b = (int)blockIdx.x / hw_parts, part = (int)blockIdx.x % hw_parts;
  const int p = b / K, s = b % K;
  const float* demb = c.w.demb;
  const float* dq = demb + ((int64_t)s * 2 * P + p) * D;
  const float* acc_in = c.w.dfeat + (int64_t)b * D;
  const bool sreal = a.is_real[p * K + s] != 0;
  constexpr int KMAX = 16;
  float den[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    den[i] = 0.f;
    if (i < K && i != s && sreal && a.is_real[p * K + i]) {
      int cnt = 0;
      for (int t = 0; t < K; ++t) cnt += (t != i && a.is_real[p * K + t]) ? 1 : 0;
      den[i] = (float)max(cnt, 1);
    }
  }
  const float scale = a.amp_state ? a.amp_state[0] : 1.f;
  const float inv = 1.0f / (float)HW;
  const int hw0 = (int)((int64_t)HW * part / hw_parts), hw1 = (int)((int64_t)HW * (part + 1) / hw_parts);
  for (int d0 = (int)threadIdx.x * 8; d0 < D; d0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int d = d0 + k;
      float o = acc_in[d] + dq[d];
      if (sreal) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
          if (den[i] != 0.f) acc += demb[((int64_t)i * 2 * P + P + p) * D + d] / den[i];
        o += acc;
      }
      v[k] = o;
    }
    if (part == 0 && a.dfeat_out) {
      float* o = a.dfeat_out + (int64_t)b * D + d0;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (a.amp_state) v[k] = v[k] * scale; v[k] = v[k] * inv; }
    for (int hw = hw0; hw < hw1; ++hw) {
      const int64_t off = ((int64_t)b * HW + hw) * D + d0;
      if (DT == CREID_F32) {
        float* g = reinterpret_cast<float*>(a.g) + off;
        *reinterpret_cast<float4*>(g) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(g + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else if (DT == CREID_BF16) {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.g) + off) =
            make_uint4(Bf16T::pack2(v[0], v[1]), Bf16T::pack2(v[2], v[3]), Bf16T::pack2(v[4], v[5]), Bf16T::pack2(v[6], v[7]));
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.g) + off) =
            make_uint4(F16T::pack2(v[0], v[1]), F16T::pack2(v[2], v[3]), F16T::pack2(v[4], v[5]), F16T::pack2(v[6], v[7]));
      }
    }
  }
}

// S6 with the column sums of the LAST bottleneck's bn3 backward folded in (16-bit compute types): the first thing the backbone's
// backward does with g is bn2d_bwd_reduce_kernel -- read g, the raw conv3 output and the ReLU bits again, only to sum
// (g', g' * xhat) per channel.  Here the workgroup that writes a 128-row x 256-channel piece of g holds exactly the values
// that kernel would load (the 16-bit-rounded g), so it accumulates the same sums over the same (row lane, channel octet) thread
// map, in the same order, into the same partial-row layout: the partials are BIT-IDENTICAL to the stand-alone reduction and one
// launch + one read of g are gone.  grid (C / 256, ceil(M / 128)); thread = (octet cch of 32, row lane rl of 8).
template <typename ET>
__global__ __launch_bounds__(256) void heads_stage6r_kernel(HeadsCtx c) {
  const creid_ctl_heads& a = c.a;
  const int P = (int)a.P, K = (int)a.K, D = (int)a.D, HW = (int)a.HW;
  const int64_t M = a.B * a.HW;
  __shared__ float red[2][256 * 8];
  const int cch = (int)threadIdx.x & 31, rl = (int)threadIdx.x >> 5;
  const int c0 = ((int)blockIdx.x * 32 + cch) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * 128, r1 = min(M, r0 + 128);
  const float* demb = c.w.demb;
  const unsigned short* xq = reinterpret_cast<const unsigned short*>(a.bn_x);
  unsigned short* gq = reinterpret_cast<unsigned short*>(a.g);
  float s1[8], s2[8], mu[8], is[8], v[8];
  unsigned gw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < 8; k += 4) {
    const float4 m4 = *reinterpret_cast<const float4*>(a.bn_mean + c0 + k);
    const float4 i4 = *reinterpret_cast<const float4*>(a.bn_invstd + c0 + k);
    mu[k] = m4.x; mu[k + 1] = m4.y; mu[k + 2] = m4.z; mu[k + 3] = m4.w;
    is[k] = i4.x; is[k + 1] = i4.y; is[k + 2] = i4.z; is[k + 3] = i4.w;
  }
  const float scale = a.amp_state ? a.amp_state[0] : 1.f;
  const float inv = 1.0f / (float)HW;
  int cur_b = -1;
  for (int64_t r = r0 + rl; r < r1; r += 8) {
    const int b = (int)(r / HW);
    if (b != cur_b) {                                  // a new image: its finished dfeat octet (HW >= 128: once per workgroup)
      cur_b = b;
      const int p = b / K, s = b % K;
      const bool sreal = a.is_real[p * K + s] != 0;
      const float4 f0 = *reinterpret_cast<const float4*>(c.w.dfeat + (int64_t)b * D + c0);
      const float4 f1 = *reinterpret_cast<const float4*>(c.w.dfeat + (int64_t)b * D + c0 + 4);
      const float4 q0 = *reinterpret_cast<const float4*>(demb + ((int64_t)s * 2 * P + p) * D + c0);
      const float4 q1 = *reinterpret_cast<const float4*>(demb + ((int64_t)s * 2 * P + p) * D + c0 + 4);
      v[0] = f0.x + q0.x; v[1] = f0.y + q0.y; v[2] = f0.z + q0.z; v[3] = f0.w + q0.w;
      v[4] = f1.x + q1.x; v[5] = f1.y + q1.y; v[6] = f1.z + q1.z; v[7] = f1.w + q1.w;
      if (sreal) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int i = 0; i < K; ++i) {                  // same round order and divisions as loo_emb_bwd_kernel
          if (i == s || !a.is_real[p * K + i]) continue;
          int cnt = 0;
          for (int t = 0; t < K; ++t) cnt += (t != i && a.is_real[p * K + t]) ? 1 : 0;
          const float den = (float)max(cnt, 1);
          const float4 e0 = *reinterpret_cast<const float4*>(demb + ((int64_t)i * 2 * P + P + p) * D + c0);
          const float4 e1 = *reinterpret_cast<const float4*>(demb + ((int64_t)i * 2 * P + P + p) * D + c0 + 4);
          acc[0] += e0.x / den; acc[1] += e0.y / den; acc[2] += e0.z / den; acc[3] += e0.w / den;
          acc[4] += e1.x / den; acc[5] += e1.y / den; acc[6] += e1.z / den; acc[7] += e1.w / den;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += acc[k];
      }
      if (a.dfeat_out && r == (int64_t)b * HW) {        // (the thread that owns the image's first row publishes the fp32 row)
        float* o = a.dfeat_out + (int64_t)b * D + c0;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { if (a.amp_state) v[k] = v[k] * scale; v[k] = v[k] * inv; }
#pragma unroll
      for (int k = 0; k < 4; ++k) gw[k] = ET::pack2(v[2 * k], v[2 * k + 1]);
    }
    const int64_t off = r * D + c0;
    *reinterpret_cast<uint4*>(gq + off) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
    const uint4 xv = *reinterpret_cast<const uint4*>(xq + off);
    const unsigned m = a.bn_mask[off >> 3];
    const unsigned* xw = &xv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = ((m >> (2 * k)) & 1u) ? ET::lo(gw[k]) : 0.f, g1 = ((m >> (2 * k + 1)) & 1u) ? ET::hi(gw[k]) : 0.f;
      s1[2 * k] += g0; s2[2 * k] = fmaf(g0, (ET::lo(xw[k]) - mu[2 * k]) * is[2 * k], s2[2 * k]);
      s1[2 * k + 1] += g1; s2[2 * k + 1] = fmaf(g1, (ET::hi(xw[k]) - mu[2 * k + 1]) * is[2 * k + 1], s2[2 * k + 1]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][(rl * 32 + cch) * 8 + k] = s1[k]; red[1][(rl * 32 + cch) * 8 + k] = s2[k]; }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < 2 * 32 * 8; i += 256) {
    const int which = i / (32 * 8), cl = i - which * 32 * 8;
    float acc = 0.f;
    for (int q = 0; q < 8; ++q) acc += red[which][q * 32 * 8 + cl];
    const int ch = (int)blockIdx.x * 32 * 8 + cl;
    if (ch < D) a.bn_partial[((int64_t)blockIdx.y * 2 + which) * D + ch] = acc;
  }
}

// ======================================================================================
extern "C" {

int creid_loo_centroids_fwd(const float* feat, const uint8_t* is_real, int64_t P, int64_t K, int64_t D,
                            float* centroids, int32_t* valid, void* stream) {
  CREID_CHECK_ARG(feat && is_real && centroids && valid && P > 0 && K > 0 && D > 0);
  hipLaunchKernelGGL(loo_centroids_fwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream), feat,
                     is_real, (int)P, (int)K, (int)D, centroids, valid);
  CREID_LAUNCH_RET();
}

int creid_loo_centroids_bwd(const float* dcentroids, const uint8_t* is_real, int64_t P, int64_t K, int64_t D,
                            float* dfeat_accum, void* stream) {
  CREID_CHECK_ARG(dcentroids && is_real && dfeat_accum && P > 0 && K > 0 && D > 0);
  hipLaunchKernelGGL(loo_centroids_bwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream),
                     dcentroids, is_real, (int)P, (int)K, (int)D, dfeat_accum);
  CREID_LAUNCH_RET();
}

int creid_loo_emb_fwd(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                      float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, void* stream) {
  CREID_CHECK_ARG(feat && is_real && labels && centroids && valid && emb && lab && cnorm && P > 0 && K > 0 && D > 0);
  hipLaunchKernelGGL(loo_emb_fwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream), feat, is_real, labels,
                     (int)P, (int)K, (int)D, centroids, valid, emb, lab, cnorm, (uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_loo_emb_fwd_rows(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                           float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, uint8_t* row_exists,
                           void* stream) {
  CREID_CHECK_ARG(feat && is_real && labels && centroids && valid && emb && lab && cnorm && row_exists && P > 0 && K > 0 && D > 0);
  hipLaunchKernelGGL(loo_emb_fwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream), feat, is_real, labels,
                     (int)P, (int)K, (int)D, centroids, valid, emb, lab, cnorm, row_exists);
  CREID_LAUNCH_RET();
}

int creid_loo_emb_fwd_rows_lonely(const float* feat, const uint8_t* is_real, const int64_t* labels, int64_t P, int64_t K, int64_t D,
                                  float* centroids, int32_t* valid, float* emb, int64_t* lab, float* cnorm, uint8_t* row_exists,
                                  int32_t* lonely_accum, void* stream) {
  CREID_CHECK_ARG(feat && is_real && labels && centroids && valid && emb && lab && cnorm && row_exists && lonely_accum && P > 0 &&
                  K > 0 && D > 0);
  hipLaunchKernelGGL(loo_emb_fwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream), feat, is_real, labels,
                     (int)P, (int)K, (int)D, centroids, valid, emb, lab, cnorm, row_exists, lonely_accum);
  CREID_LAUNCH_RET();
}

int creid_ctl_round_scale(const float* out4_rounds, int64_t K, float* inv_rounds, void* stream) {
  CREID_CHECK_ARG(out4_rounds && inv_rounds && K > 0);
  hipLaunchKernelGGL(ctl_round_scale_kernel, dim3(1), dim3(1), 0, as_stream(stream), out4_rounds, (int)K, inv_rounds);
  CREID_LAUNCH_RET();
}

int creid_ctl_step_stats_rows(const float* scal, const float* weights, int64_t n, int64_t K, int64_t P, const float* cnorm,
                              const uint8_t* row_exists, float* out, void* stream) {
  CREID_CHECK_ARG(scal && weights && cnorm && row_exists && out && n >= 4 * (K + 1) && K > 0 && P > 0);
  hipLaunchKernelGGL(ctl_step_stats_rows_kernel, dim3(1), dim3(64), 0, as_stream(stream), scal, weights, (int)n, (int)K, (int)P,
                     cnorm, row_exists, out);
  CREID_LAUNCH_RET();
}

int creid_loo_emb_bwd(const float* demb, const uint8_t* is_real, int64_t P, int64_t K, int64_t D, float* dfeat_accum,
                      void* stream) {
  CREID_CHECK_ARG(demb && is_real && dfeat_accum && P > 0 && K > 0 && D > 0);
  if (K > 16) return CREID_E_SHAPE;
  hipLaunchKernelGGL(loo_emb_bwd_kernel, dim3((unsigned)P, (unsigned)K), dim3(256), 0, as_stream(stream), demb, is_real, (int)P,
                     (int)K, (int)D, dfeat_accum);
  CREID_LAUNCH_RET();
}

int creid_ctl_step_stats(const float* scal, const float* weights, int64_t n, int64_t K, const float* cnorm, int64_t rows,
                         float* out, void* stream) {
  CREID_CHECK_ARG(scal && weights && cnorm && out && n >= 4 * (K + 1) && K > 0 && rows > 0);
  hipLaunchKernelGGL(ctl_step_stats_kernel, dim3(1), dim3(64), 0, as_stream(stream), scal, weights, (int)n, (int)K, cnorm,
                     (int)rows, out);
  CREID_LAUNCH_RET();
}

static int triplet_fwd_impl(int kind, const float* x, const int64_t* labels, const uint8_t* anchor_mask, int64_t nb, int64_t N,
                            int64_t D, float margin, float* dist_ap, float* dist_an, int32_t* p_idx, int32_t* n_idx,
                            float* coef, float* out4, float* dist_mat, void* stream, const uint8_t* row_exists = nullptr,
                            int min_anchors = 0) {
  CREID_CHECK_ARG(x && labels && dist_ap && dist_an && p_idx && n_idx && out4 && N > 0 && D > 0 && nb > 0);
  if (D % 4 != 0 || nb > 65535) return CREID_E_SHAPE;
  const size_t smem = (size_t)(D + N) * sizeof(float);
  if (smem > 64 * 1024) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  if (kind == 0)
    hipLaunchKernelGGL(triplet_mine_kernel<0>, dim3((unsigned)N, (unsigned)nb), dim3(TM_T), smem, s, x, labels, (int)N,
                       (int)D, dist_ap, dist_an, p_idx, n_idx, dist_mat, row_exists);
  else
    hipLaunchKernelGGL(triplet_mine_kernel<1>, dim3((unsigned)N, (unsigned)nb), dim3(TM_T), smem, s, x, labels, (int)N,
                       (int)D, dist_ap, dist_an, p_idx, n_idx, dist_mat, row_exists);
  hipLaunchKernelGGL(triplet_loss_kernel, dim3(1, (unsigned)nb), dim3(256), 0, s, dist_ap, dist_an,
                     row_exists ? row_exists : anchor_mask, (int)N, margin, out4, coef, min_anchors);
  CREID_LAUNCH_RET();
}

int creid_triplet_fwd_batched(const float* x, const int64_t* labels, const uint8_t* anchor_mask, int64_t nb, int64_t N,
                              int64_t D, float margin, float* dist_ap, float* dist_an, int32_t* p_idx, int32_t* n_idx,
                              float* coef, float* out4, float* dist_mat, void* stream) {
  return triplet_fwd_impl(0, x, labels, anchor_mask, nb, N, D, margin, dist_ap, dist_an, p_idx, n_idx, coef, out4, dist_mat,
                          stream);
}

int creid_triplet_fwd_batched_rows(const float* x, const int64_t* labels, const uint8_t* row_exists, int64_t nb, int64_t N,
                                   int64_t D, float margin, int32_t min_rows, float* dist_ap, float* dist_an, int32_t* p_idx,
                                   int32_t* n_idx, float* coef, float* out4, void* stream) {
  CREID_CHECK_ARG(row_exists && coef && min_rows >= 0);
  return triplet_fwd_impl(0, x, labels, nullptr, nb, N, D, margin, dist_ap, dist_an, p_idx, n_idx, coef, out4, nullptr, stream,
                          row_exists, (int)min_rows);
}

int creid_triplet_cosine_fwd(const float* x_unit, const int64_t* labels, const uint8_t* anchor_mask, int64_t N, int64_t D,
                             float margin, float* dist_ap, float* dist_an, int32_t* p_idx, int32_t* n_idx, float* coef,
                             float* out4, float* dist_mat, void* stream) {
  return triplet_fwd_impl(1, x_unit, labels, anchor_mask, 1, N, D, margin, dist_ap, dist_an, p_idx, n_idx, coef, out4,
                          dist_mat, stream);
}

int creid_triplet_fwd(const float* x, const int64_t* labels, const uint8_t* anchor_mask, int64_t N, int64_t D,
                      float margin, float* dist_ap, float* dist_an, int32_t* p_idx, int32_t* n_idx, float* coef,
                      float* out4, float* dist_mat, void* stream) {
  return creid_triplet_fwd_batched(x, labels, anchor_mask, 1, N, D, margin, dist_ap, dist_an, p_idx, n_idx, coef, out4,
                                   dist_mat, stream);
}

static int triplet_bwd_impl(int kind, const float* x, int64_t nb, int64_t N, int64_t D, const float* dist_ap,
                            const float* dist_an, const int32_t* p_idx, const int32_t* n_idx, const float* coef,
                            const float* gscale_dev, float gscale, float* dx_accum, void* stream) {
  CREID_CHECK_ARG(x && dist_ap && dist_an && p_idx && n_idx && coef && dx_accum && N > 0 && D > 0 && nb > 0);
  if ((size_t)N * 8 * sizeof(float) > 48 * 1024 || nb > 65535) return CREID_E_SHAPE;
  if (kind == 0)
    hipLaunchKernelGGL(triplet_bwd_kernel<0>, dim3((unsigned)N, (unsigned)nb), dim3(256), (size_t)N * 8 * sizeof(float),
                       as_stream(stream), x, (int)N, (int)D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_dev, gscale, dx_accum);
  else
    hipLaunchKernelGGL(triplet_bwd_kernel<1>, dim3((unsigned)N, (unsigned)nb), dim3(256), (size_t)N * 8 * sizeof(float),
                       as_stream(stream), x, (int)N, (int)D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_dev, gscale, dx_accum);
  CREID_LAUNCH_RET();
}

int creid_triplet_bwd_batched(const float* x, int64_t nb, int64_t N, int64_t D, const float* dist_ap, const float* dist_an,
                              const int32_t* p_idx, const int32_t* n_idx, const float* coef, const float* gscale_dev,
                              float gscale, float* dx_accum, void* stream) {
  return triplet_bwd_impl(0, x, nb, N, D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_dev, gscale, dx_accum, stream);
}

int creid_triplet_cosine_bwd(const float* x_unit, int64_t N, int64_t D, const float* dist_ap, const float* dist_an,
                             const int32_t* p_idx, const int32_t* n_idx, const float* coef, const float* gscale_dev,
                             float gscale, float* dx_accum, void* stream) {
  return triplet_bwd_impl(1, x_unit, 1, N, D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_dev, gscale, dx_accum, stream);
}

int creid_rownorm_fwd(const float* x, int64_t N, int64_t D, int mode, float eps, float* y, float* norm, void* stream) {
  CREID_CHECK_ARG(x && y && norm && N > 0 && D > 0 && (mode == 0 || mode == 1));
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), x, (int)D, mode, eps, y, norm);
  CREID_LAUNCH_RET();
}

int creid_rownorm_bwd(const float* x, const float* norm, const float* dy, int64_t N, int64_t D, int mode, float eps,
                      float* dx, void* stream) {
  CREID_CHECK_ARG(x && norm && dy && dx && N > 0 && D > 0 && (mode == 0 || mode == 1));
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), x, norm, dy, (int)D, mode, eps,
                     dx);
  CREID_LAUNCH_RET();
}

int creid_hard_mine_from_dist(const float* dist_mat, const int64_t* labels, int64_t N, float* dist_ap, float* dist_an,
                              int32_t* p_idx, int32_t* n_idx, void* stream) {
  CREID_CHECK_ARG(dist_mat && labels && dist_ap && dist_an && p_idx && n_idx && N > 0);
  hipLaunchKernelGGL(mine_from_dist_kernel, dim3((unsigned)N), dim3(64), 0, as_stream(stream), dist_mat, labels, (int)N,
                     dist_ap, dist_an, p_idx, n_idx);
  CREID_LAUNCH_RET();
}

int creid_clamp_sqrt_inplace(float* d, int64_t n, float lo, void* stream) {
  CREID_CHECK_ARG(d && n >= 0);
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(clamp_sqrt_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d, n, lo);
  CREID_LAUNCH_RET();
}

int creid_triplet_bwd(const float* x, int64_t N, int64_t D, const float* dist_ap, const float* dist_an,
                      const int32_t* p_idx, const int32_t* n_idx, const float* coef, const float* gscale_dev,
                      float gscale, float* dx_accum, void* stream) {
  return creid_triplet_bwd_batched(x, 1, N, D, dist_ap, dist_an, p_idx, n_idx, coef, gscale_dev, gscale, dx_accum, stream);
}

int creid_center_loss_fwd(const float* x, const int64_t* labels, const float* centers, int64_t B, int64_t C,
                          int64_t D, float* row_sq, float* loss, void* stream) {
  CREID_CHECK_ARG(x && labels && centers && row_sq && loss && B > 0 && C > 0 && D > 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(center_row_kernel, dim3((unsigned)B), dim3(256), 0, s, x, labels, centers, (int)D, row_sq);
  hipLaunchKernelGGL(center_reduce_kernel, dim3(1), dim3(256), 0, s, row_sq, (int)B, (int)C, loss, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_center_loss_fwd_masked(const float* x, const int64_t* labels, const float* centers, const uint8_t* row_mask, int64_t B,
                                 int64_t C, int64_t D, float* row_sq, float* loss, void* stream) {
  CREID_CHECK_ARG(x && labels && centers && row_mask && row_sq && loss && B > 0 && C > 0 && D > 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(center_row_kernel, dim3((unsigned)B), dim3(256), 0, s, x, labels, centers, (int)D, row_sq);
  hipLaunchKernelGGL(center_reduce_kernel, dim3(1), dim3(256), 0, s, row_sq, (int)B, (int)C, loss, row_mask);
  CREID_LAUNCH_RET();
}

int creid_center_loss_bwd(const float* x, const int64_t* labels, const float* centers, const float* row_sq,
                          int64_t B, int64_t D, const float* gscale_dev, float gscale, float* dx_accum,
                          float* dcenters_accum, void* stream) {
  CREID_CHECK_ARG(x && labels && centers && row_sq && B > 0 && D > 0 && (dx_accum || dcenters_accum));
  hipLaunchKernelGGL(center_bwd_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), x, labels, centers,
                     row_sq, (int)B, (int)D, gscale_dev, gscale, dx_accum, dcenters_accum, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_center_loss_bwd_masked(const float* x, const int64_t* labels, const float* centers, const float* row_sq,
                                 const uint8_t* row_mask, int64_t B, int64_t D, const float* gscale_dev, float gscale,
                                 float* dx_accum, float* dcenters_accum, void* stream) {
  CREID_CHECK_ARG(x && labels && centers && row_sq && row_mask && B > 0 && D > 0 && (dx_accum || dcenters_accum));
  hipLaunchKernelGGL(center_bwd_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), x, labels, centers,
                     row_sq, (int)B, (int)D, gscale_dev, gscale, dx_accum, dcenters_accum, row_mask);
  CREID_LAUNCH_RET();
}

int creid_xent_ls(const float* logits, const int64_t* targets, int64_t B, int64_t C, float eps, float gscale,
                  float* row_loss, float* loss, float* dlogits, void* stream) {
  CREID_CHECK_ARG(logits && targets && row_loss && loss && B > 0 && C > 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(xent_ls_kernel, dim3((unsigned)B), dim3(256), 0, s, logits, targets, (int)B, (int)C, eps,
                     row_loss, dlogits, gscale, (const uint8_t*)nullptr);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, s, row_loss, (int)B, 1.0f / (float)B, loss, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_xent_ls_masked(const float* logits, const int64_t* targets, const uint8_t* row_mask, int64_t B, int64_t C, float eps,
                         float gscale, float* row_loss, float* loss, float* dlogits, void* stream) {
  CREID_CHECK_ARG(logits && targets && row_mask && row_loss && loss && B > 0 && C > 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(xent_ls_kernel, dim3((unsigned)B), dim3(256), 0, s, logits, targets, (int)B, (int)C, eps,
                     row_loss, dlogits, gscale, row_mask);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, s, row_loss, (int)B, 0.f, loss, row_mask);
  CREID_LAUNCH_RET();
}

int creid_bn1d_fwd(const float* x, int64_t B, int64_t D, const float* weight, const float* bias, float* running_mean,
                   float* running_var, int training, float momentum, float eps, float* y, float* save_mean,
                   float* save_invstd, void* stream) {
  CREID_CHECK_ARG(x && y && B > 0 && D > 0);
  CREID_CHECK_ARG(training || (running_mean && running_var));
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((unsigned)((D + 31) / 32)), dim3(256), 0, as_stream(stream), x, (int)B,
                     (int)D, weight, bias, running_mean, running_var, training, momentum, eps, y, save_mean,
                     save_invstd, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_bn1d_fwd_masked(const float* x, const uint8_t* row_mask, int64_t B, int64_t D, const float* weight, const float* bias,
                          float* running_mean, float* running_var, float momentum, float eps, float* y, float* save_mean,
                          float* save_invstd, void* stream) {
  CREID_CHECK_ARG(x && y && row_mask && B > 0 && D > 0);
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((unsigned)((D + 31) / 32)), dim3(256), 0, as_stream(stream), x, (int)B,
                     (int)D, weight, bias, running_mean, running_var, 1, momentum, eps, y, save_mean, save_invstd, row_mask);
  CREID_LAUNCH_RET();
}

int creid_bn1d_bwd(const float* x, const float* dy, int64_t B, int64_t D, const float* weight, const float* save_mean,
                   const float* save_invstd, float* dx_accum, float* dweight_accum, float* dbias_accum,
                   void* stream) {
  CREID_CHECK_ARG(x && dy && save_mean && save_invstd && dx_accum && B > 0 && D > 0);
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((unsigned)((D + 31) / 32)), dim3(256), 0, as_stream(stream), x, dy,
                     (int)B, (int)D, weight, save_mean, save_invstd, dx_accum, dweight_accum, dbias_accum, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_bn1d_bwd_masked(const float* x, const float* dy, const uint8_t* row_mask, int64_t B, int64_t D, const float* weight,
                          const float* save_mean, const float* save_invstd, float* dx_accum, float* dweight_accum,
                          float* dbias_accum, void* stream) {
  CREID_CHECK_ARG(x && dy && row_mask && save_mean && save_invstd && dx_accum && B > 0 && D > 0);
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((unsigned)((D + 31) / 32)), dim3(256), 0, as_stream(stream), x, dy,
                     (int)B, (int)D, weight, save_mean, save_invstd, dx_accum, dweight_accum, dbias_accum, row_mask);
  CREID_LAUNCH_RET();
}

int creid_gather_mean_rows(const float* emb, const int64_t* order, const int64_t* offsets, int64_t n_seg, int64_t D,
                           float* out, void* stream) {
  CREID_CHECK_ARG(emb && order && offsets && out && n_seg > 0 && D > 0);
  hipLaunchKernelGGL(gather_mean_rows_kernel, dim3((unsigned)n_seg), dim3(256), 0, as_stream(stream), emb, order,
                     offsets, (int)D, out);
  CREID_LAUNCH_RET();
}

int creid_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  CREID_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1);
  if (n == 0) return 0;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  CREID_LAUNCH_RET();
}

int creid_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  CREID_CHECK_ARG(p && g && m && v && hyper_dev && n >= 0);
  if (n % 4 != 0) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  if (n == 0) { hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, hyper_dev, beta1, beta2); CREID_LAUNCH_RET(); }
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, hyper_dev, beta1, beta2,
                     eps, weight_decay, grad_scale);
  CREID_LAUNCH_RET();
}

int creid_amp_scale(const float* x, int64_t n, const float* amp_state, float* y, void* stream) {
  CREID_CHECK_ARG(x && y && amp_state && n > 0);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(amp_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, n, amp_state, y);
  CREID_LAUNCH_RET();
}

int creid_amp_unscale_check(float* g, int64_t n, const float* amp_state, int32_t* amp_flags, void* stream) {
  CREID_CHECK_ARG(g && amp_state && amp_flags && n >= 0);
  if (n % 4 != 0) return CREID_E_SHAPE;
  if (n == 0) return 0;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(amp_unscale_check_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, n, amp_state, amp_flags);
  CREID_LAUNCH_RET();
}

int creid_amp_update(float* amp_state, int32_t* amp_flags, float growth_factor, float backoff_factor, int32_t growth_interval,
                     void* stream) {
  CREID_CHECK_ARG(amp_state && amp_flags && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0);
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, as_stream(stream), amp_state, amp_flags, growth_factor, backoff_factor,
                     (int)growth_interval);
  CREID_LAUNCH_RET();
}

int creid_adam_step_dev_amp(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, float beta1, float beta2,
                            float eps, float weight_decay, float grad_scale, const int32_t* skip_flag, void* stream) {
  CREID_CHECK_ARG(p && g && m && v && hyper_dev && skip_flag && n >= 0);
  if (n % 4 != 0) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  if (n == 0) { hipLaunchKernelGGL(adam_advance_amp_kernel, dim3(1), dim3(1), 0, s, hyper_dev, beta1, beta2, skip_flag); CREID_LAUNCH_RET(); }
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, hyper_dev, beta1, beta2, eps,
                     weight_decay, grad_scale, skip_flag);
  CREID_LAUNCH_RET();
}

int creid_sgd_scaled_step_amp(float* p, float* g, int64_t n, float lr, float grad_mul, const int32_t* skip_flag, void* stream) {
  CREID_CHECK_ARG(p && g && skip_flag && n >= 0);
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, n, lr, grad_mul, skip_flag);
  CREID_LAUNCH_RET();
}

int creid_sgd_scaled_step(float* p, float* g, int64_t n, float lr, float grad_mul, void* stream) {
  CREID_CHECK_ARG(p && g && n >= 0);
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, n, lr,
                     grad_mul);
  CREID_LAUNCH_RET();
}

size_t creid_ctl_heads_workspace_bytes(int64_t B, int64_t P, int64_t K, int64_t D, int64_t num_classes) {
  if (B <= 0 || P <= 0 || K <= 0 || D <= 0 || num_classes <= 0) return 0;
  return heads_carve(B, P, K, D, num_classes, nullptr, nullptr);
}

int creid_ctl_heads_fused(const creid_ctl_heads* a, void* stream) {
  CREID_CHECK_ARG(a && a->feat && a->labels && a->is_real && a->centers && a->bn_weight && a->bn_running_mean && a->bn_running_var &&
                  a->fc_weight && a->loss_weights && a->stats && a->g && a->workspace);
  CREID_CHECK_ARG(a->B > 0 && a->P >= 2 && a->K >= 2 && a->B == a->P * a->K && a->D > 0 && a->num_classes > 0 && a->num_centers > 0 &&
                  a->HW > 0 && a->split_logits >= 1 && a->split_dbnf >= 1 && a->margin >= 0.f);
  CREID_CHECK_ARG(!a->masked || a->lonely);
  if (a->K > 16 || a->D % 8 != 0 || a->B > 256 || 2 * a->P > 256) return CREID_E_SHAPE;   // (loo backward slots; triplet backward: one trip)
  if (a->g_dtype != CREID_F32 && a->g_dtype != CREID_BF16 && a->g_dtype != CREID_F16) return CREID_E_DTYPE;
  HeadsCtx c;
  c.a = *a;
  if (heads_carve(a->B, a->P, a->K, a->D, a->num_classes, (char*)a->workspace, &c.w) > a->workspace_bytes) return CREID_E_WS;
  if (((uintptr_t)a->workspace & 255) != 0) return CREID_E_ARG;
  c.mask = a->masked ? a->is_real : nullptr;
  const int B = (int)a->B, P = (int)a->P, K = (int)a->K, D = (int)a->D, C = (int)a->num_classes, R = 2 * P;
  c.gl_tn = (C + 63) / 64; c.gl_tm = (B + 63) / 64;
  c.gd_tn = (D + 63) / 64; c.gd_tm = (B + 63) / 64;
  c.gw_tn = (D + 63) / 64; c.gw_tm = (C + 63) / 64;
  const size_t smem_mine = (size_t)(D + (B > R ? B : R)) * sizeof(float);
  if (smem_mine > 64 * 1024) return CREID_E_SHAPE;
  const size_t smem_bwd = (size_t)(B > R ? B : R) * 8 * sizeof(float);
  hipStream_t s = as_stream(stream);
  const int nbn = (D + 31) / 32;
  hipLaunchKernelGGL(heads_stage1_kernel, dim3((unsigned)(HEADS_ZERO_WGS + 2 * B + nbn + P * K)), dim3(1024), smem_mine, s, c);
  hipLaunchKernelGGL(heads_stage2_kernel, dim3((unsigned)(K * R + c.gl_tn * c.gl_tm * a->split_logits + 2)), dim3(1024), smem_mine, s, c);
  hipLaunchKernelGGL(heads_stage3_kernel, dim3((unsigned)(2 * B + K)), dim3(256), smem_bwd, s, c);
  const int n4 = c.gd_tn * c.gd_tm * a->split_dbnf + (a->d_fc_weight ? c.gw_tn * c.gw_tm : 0) + B + 1 + (a->masked ? 1 : K * R);
  hipLaunchKernelGGL(heads_stage4_kernel, dim3((unsigned)n4), dim3(256), smem_bwd, s, c);
  hipLaunchKernelGGL(heads_stage5_kernel, dim3((unsigned)(nbn + 1 + (a->masked ? K * R : 0))), dim3(256), smem_bwd, s, c);
  if (a->bn_partial) {
    // the last bottleneck's bn3 column sums ride in the launch that writes g (16-bit compute types, 256-channel column groups)
    CREID_CHECK_ARG(a->bn_x && a->bn_mask && a->bn_mean && a->bn_invstd);
    if (a->g_dtype == CREID_F32 || D % 256 != 0) return CREID_E_SHAPE;
    const dim3 g6r((unsigned)(D / 256), (unsigned)((a->B * a->HW + 127) / 128));
    if (a->g_dtype == CREID_BF16) hipLaunchKernelGGL(heads_stage6r_kernel<Bf16T>, g6r, dim3(256), 0, s, c);
    else hipLaunchKernelGGL(heads_stage6r_kernel<F16T>, g6r, dim3(256), 0, s, c);
    CREID_LAUNCH_RET();
  }
  int hw_parts = (int)a->HW < 8 ? (int)a->HW : 8;         // 512 workgroups at B = 64: each re-derives its dfeat row once per 16 copies
  const dim3 g6((unsigned)(B * hw_parts));
  if (a->g_dtype == CREID_F32) hipLaunchKernelGGL(heads_stage6_kernel<CREID_F32>, g6, dim3(256), 0, s, c, hw_parts);
  else if (a->g_dtype == CREID_BF16) hipLaunchKernelGGL(heads_stage6_kernel<CREID_BF16>, g6, dim3(256), 0, s, c, hw_parts);
  else hipLaunchKernelGGL(heads_stage6_kernel<CREID_F16>, g6, dim3(256), 0, s, c, hw_parts);
  CREID_LAUNCH_RET();
}

}  // extern "C"
