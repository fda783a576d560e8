// Body of the small fp32 MFMA GEMM (csrc/gemm_f32.hip) as a device function of the block index, shared by the stand-alone
// kernel and by the fused training-step heads (csrc/heads.hip: classifier forward / data gradient / weight gradient ride in
// multi-role launches there).  256 threads; see gemm_f32.hip for the description.
#pragma once
#include "common.hpp"

namespace {
constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = 65;
}

__device__ __forceinline__ void gemm_f32_body(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       float* __restrict__ Cm, int64_t ldc, int M, int N, int K,
                                                       float alpha, float beta, int split_k, const int bx_, const int by_, const int bz_) {
  __shared__ float As[GBK][GLD];
  __shared__ float Bs[GBK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = by_ * GBM, col0 = bx_ * GBN;
  const int kz = bz_;
  const int kchunk = ((K + split_k - 1) / split_k + GBK - 1) / GBK * GBK;
  const int kbeg = kz * kchunk, kend = min(K, kbeg + kchunk);
  // per-operand element->thread map: make the unit-stride direction the fast one
  const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int l31 = lane & 31, kh = lane >> 5;
  // register prefetch: the loads of k-tile t+1 fly while tile t is multiplied (these GEMMs are pure latency: a few
  // k-tiles per workgroup, every one of which used to expose a full global-load round trip)
  float ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      int kk, rr;
      if (a_kfast) { kk = e & 15; rr = e >> 4; } else { rr = e & 63; kk = e >> 6; }
      const int gr = row0 + rr, gk = k0 + kk;
      ra[i] = (gr < M && gk < kend) ? A[(int64_t)gr * sam + (int64_t)gk * sak] : 0.f;
      if (b_kfast) { kk = e & 15; rr = e >> 4; } else { rr = e & 63; kk = e >> 6; }
      const int gc = col0 + rr, gk2 = k0 + kk;
      rb[i] = (gc < N && gk2 < kend) ? B[(int64_t)gk2 * sbk + (int64_t)gc * sbn] : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      int kk, rr;
      if (a_kfast) { kk = e & 15; rr = e >> 4; } else { rr = e & 63; kk = e >> 6; }
      As[kk][rr] = ra[i];
      if (b_kfast) { kk = e & 15; rr = e >> 4; } else { rr = e & 63; kk = e >> 6; }
      Bs[kk][rr] = rb[i];
    }
  };
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    lstore();
    __syncthreads();
    if (k0 + GBK < kend) gload(k0 + GBK);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const float a = As[kk + kh][wm * 32 + l31];
      const float b = Bs[kk + kh][wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int c = col0 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (rr < M && c < N) {
      float* dst = Cm + (int64_t)rr * ldc + c;
      if (split_k == 1) *dst = alpha * acc[r] + (beta != 0.f ? beta * *dst : 0.f);
      else atomicAdd(dst, alpha * acc[r]);   // caller pre-scaled / zeroed C
    }
  }
}

