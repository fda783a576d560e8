// Small generic fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32), arbitrary operand strides:
//   C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + beta * C[m*ldc + n]
// Used for the classifier of the BNNeck head (modelling/bases.py:86-87 fc_query: Linear(2048->C,
// bias=False)) forward / dgrad / wgrad, where M or K is only the batch (64..256 rows): these
// GEMMs are latency-bound, so the kernel is a plain 64x64x16 LDS tile (K-major LDS, one 32x32
// MFMA tile per wave) with an optional split over K (partials combined by fp32 atomics only
// when split_k > 1; split_k == 1 is deterministic).
#include "common.hpp"

namespace {
constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = 65;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       float* __restrict__ Cm, int64_t ldc, int M, int N, int K,
                                                       float alpha, float beta, int split_k) {
  __shared__ float As[GBK][GLD];
  __shared__ float Bs[GBK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = blockIdx.y * GBM, col0 = blockIdx.x * GBN;
  const int kz = blockIdx.z;
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

__global__ __launch_bounds__(256) void scale_matrix_kernel(float* __restrict__ Cm, int64_t ldc, int M, int N,
                                                           float beta) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float* p = Cm + (i / N) * ldc + (i % N);
    *p = beta == 0.f ? 0.f : *p * beta;
  }
}

extern "C" int creid_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                              float* Cmat, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                              int32_t split_k, void* stream) {
  CREID_CHECK_ARG(A && B && Cmat && M > 0 && N > 0 && K > 0 && ldc >= N && split_k >= 1);
  if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  if (split_k > 1) {
    int64_t blocks = (M * N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_matrix_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Cmat, ldc, (int)M, (int)N, beta);
  }
  dim3 grid((unsigned)((N + GBN - 1) / GBN), (unsigned)((M + GBM - 1) / GBM), (unsigned)split_k);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, A, sam, sak, B, sbk, sbn, Cmat, ldc, (int)M, (int)N,
                     (int)K, alpha, beta, (int)split_k);
  CREID_LAUNCH_RET();
}
