// BatchNorm-backward finalize as a DEVICE routine that another kernel can run in a few extra workgroups.
// In the backward chain   dgrad_k (+ column sums of BN k-1)  ->  finalize_{k-1}  ->  apply_{k-1}  ->  dgrad_{k-1} ...
// the finalize is a 4-128-workgroup, latency-bound launch (un-profiled 4.4 us each, 0.235 ms per step for the 53 of them,
// CREID_BN_FIN_DRY) during which the GPU idles.  The weight gradient of convolution k is independent of that chain, so
// its launch is issued between dgrad_k and apply_{k-1} and carries finalize_{k-1} in its FIRST workgroups: they finish
// within ~3 us while the weight-gradient tiles fill the rest of the chip.  Same arithmetic as bn2d_bwd_finalize_kernel
// (elementwise.hip): fp64 sums over the partial rows; only the grouping of the fp64 additions differs.
#pragma once
#include "common.hpp"

struct BnBwdFinJob {
  const float* partial;      // [rows][2][C] (sum dy, sum dy*xhat); null: no job
  int rows, C;
  float inv_count;           // 1 / (rows the statistics were taken over)
  const float* mean;
  const float* invstd;
  const float* gamma;        // nullable
  float* sums;               // [3][C] coefficients of dx = A*dy + B*x + C
  float* dgamma;             // nullable, accumulated
  float* dbeta;              // nullable, accumulated
  int nblocks;               // workgroups that carry the job: ceil(C / cw)
  int cw;                    // channels per workgroup: 16, or 4 when C is small and the partial rows are many
};

// One workgroup of NT threads finalizes channels [block*CW, block*CW + CW).  lds: >= NT*4 bytes.
template <int NT, int CW>
__device__ __forceinline__ void bn_bwd_finalize_block_cw(const BnBwdFinJob& j, int block, void* lds) {
  constexpr int RG = NT / CW, NW = NT / 64;
  double* red = reinterpret_cast<double*>(lds);              // [NW][2][CW]
  const int tid = threadIdx.x, cl = tid % CW, rg = tid / CW;
  const int c = block * CW + cl, C = j.C;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int r = rg;
    for (; r + 3 * RG < j.rows; r += 4 * RG) {              // 8 independent loads per trip
      const float a0 = j.partial[((int64_t)r * 2) * C + c], b0 = j.partial[((int64_t)r * 2 + 1) * C + c];
      const float a1 = j.partial[((int64_t)(r + RG) * 2) * C + c], b1 = j.partial[((int64_t)(r + RG) * 2 + 1) * C + c];
      const float a2 = j.partial[((int64_t)(r + 2 * RG) * 2) * C + c], b2 = j.partial[((int64_t)(r + 2 * RG) * 2 + 1) * C + c];
      const float a3 = j.partial[((int64_t)(r + 3 * RG) * 2) * C + c], b3 = j.partial[((int64_t)(r + 3 * RG) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < j.rows; r += RG) {
      s1 += (double)j.partial[((int64_t)r * 2) * C + c];
      s2 += (double)j.partial[((int64_t)r * 2 + 1) * C + c];
    }
  }
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((tid & 63) < CW) { red[((tid >> 6) * 2 + 0) * CW + cl] = s1; red[((tid >> 6) * 2 + 1) * CW + cl] = s2; }
  float p_mu = 0.f, p_is = 0.f, p_g = 1.f, p_db = 0.f, p_dg = 0.f;
  if (rg == 0 && c < C) {
    p_mu = j.mean[c]; p_is = j.invstd[c];
    if (j.gamma) p_g = j.gamma[c];
    if (j.dbeta) p_db = j.dbeta[c];
    if (j.dgamma) p_dg = j.dgamma[c];
  }
  __syncthreads();
  if (rg == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) { s1 += red[(q * 2 + 0) * CW + cl]; s2 += red[(q * 2 + 1) * CW + cl]; }
    const float invM = j.inv_count;
    const float mu = p_mu, is = p_is, k1 = p_g * is;
    const float a1 = (float)s1 * invM, a2 = (float)s2 * invM;
    j.sums[c] = k1;
    j.sums[C + c] = -k1 * is * a2;
    j.sums[2 * C + c] = -k1 * a1 + k1 * is * a2 * mu;
    if (j.dbeta) j.dbeta[c] = p_db + (float)s1;
    if (j.dgamma) j.dgamma[c] = p_dg + (float)s2;
  }
}

template <int NT>
__device__ __forceinline__ void bn_bwd_finalize_block(const BnBwdFinJob& j, int block, void* lds) {
  if (j.cw == 4) bn_bwd_finalize_block_cw<NT, 4>(j, block, lds);     // uniform over the launch
  else bn_bwd_finalize_block_cw<NT, 16>(j, block, lds);
}

static inline void bn_bwd_fin_shape(BnBwdFinJob& j) {       // host: channels per workgroup and workgroup count
  j.cw = (j.C <= 256 && j.rows >= 512) ? 4 : 16;
  j.nblocks = (j.C + j.cw - 1) / j.cw;
}
