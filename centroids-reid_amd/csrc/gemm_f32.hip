// Small generic fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32), arbitrary operand strides:
//   C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + beta * C[m*ldc + n]
// Used for the classifier of the BNNeck head (modelling/bases.py:86-87 fc_query: Linear(2048->C,
// bias=False)) forward / dgrad / wgrad, where M or K is only the batch (64..256 rows): these
// GEMMs are latency-bound, so the kernel is a plain 64x64x16 LDS tile (K-major LDS, one 32x32
// MFMA tile per wave) with an optional split over K (partials combined by fp32 atomics only
// when split_k > 1; split_k == 1 is deterministic).
#include "gemm_f32_body.hpp"

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       float* __restrict__ Cm, int64_t ldc, int M, int N, int K,
                                                       float alpha, float beta, int split_k) {
  gemm_f32_body(A, sam, sak, B, sbk, sbn, Cm, ldc, M, N, K, alpha, beta, split_k, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
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
