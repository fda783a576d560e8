// Stage A: the HBM-bound layers around the convolutions, NHWC, 16-byte accesses:
// training/eval BatchNorm2d (statistics finalize, apply + ReLU + residual, backward reduce/apply),
// MaxPool 3x3 s2, global average pool, input/weight layout transforms.
// Reference: nn.BatchNorm2d / ReLU / residual add of modelling/backbones/resnet.py:67-87,
// MaxPool2d(3,2,1) :98, AdaptiveAvgPool2d(1) modelling/baseline.py:89,93.
#include "conv_common.hpp"

template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Vec16<unsigned short> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x2_to_bf16x2_bits(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }   // value as stored
};

template <> struct Vec16<_Float16> {                            // f16 storage (the reference's own mixed precision)
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const _Float16* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = F16T::lo(w[i]); v[2 * i + 1] = F16T::hi(w[i]); }
  }
  static __device__ __forceinline__ void store(_Float16* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = F16T::pack2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

// ------------------------------------------------------------------ BN statistics finalize
// partial: [rows][2][C] (sum, sumsq per 128-row conv tile).  training: mean/invstd from the batch and
// running-stat update (momentum, unbiased variance); eval: mean = running_mean, invstd = rsqrt(rv+eps).
template <int CW>
__global__ __launch_bounds__(1024) void bn2d_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                            double count, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, int training, float momentum,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out,
                                                            float* __restrict__ scale_shift) {
  // 16 channels x 64 row-groups per workgroup: 64-B coalesced reads, 4 independent loads in flight per
  // thread, fp64 accumulation, LDS tree (the kernel is pure latency: keep the dependent chain short)
  constexpr int RG = 1024 / CW;                      // row groups: CW = 16 for wide layers, 4 when C is small and rows many
  __shared__ double red[16][2][CW];
  const int cl = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int c = blockIdx.x * CW + cl;
  if (!training) {
    if (rg == 0 && c < C) {
      const float mu = rmean[c], is = 1.0f / sqrtf(rvar[c] + eps);
      mean_out[c] = mu; invstd_out[c] = is;
      const float sc = is * (gamma ? gamma[c] : 1.f);
      scale_shift[c] = sc; scale_shift[C + c] = (beta ? beta[c] : 0.f) - mu * sc;
    }
    return;
  }
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
  {
    int r = rg;
    for (; r + 3 * RG < rows; r += 4 * RG) {          // 8 independent loads per trip
      const float a0 = partial[((int64_t)r * 2) * C + c], b0 = partial[((int64_t)r * 2 + 1) * C + c];
      const float a1 = partial[((int64_t)(r + RG) * 2) * C + c], b1 = partial[((int64_t)(r + RG) * 2 + 1) * C + c];
      const float a2 = partial[((int64_t)(r + 2 * RG) * 2) * C + c], b2 = partial[((int64_t)(r + 2 * RG) * 2 + 1) * C + c];
      const float a3 = partial[((int64_t)(r + 3 * RG) * 2) * C + c], b3 = partial[((int64_t)(r + 3 * RG) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < rows; r += RG) {
      s1 += (double)partial[((int64_t)r * 2) * C + c];
      s2 += (double)partial[((int64_t)r * 2 + 1) * C + c];
    }
  }
  // a wave holds 4 row groups x 16 channels: fold them with two shuffles, then 16 per-wave partials meet in LDS
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((threadIdx.x & 63) < CW) { red[threadIdx.x >> 6][0][cl] = s1; red[threadIdx.x >> 6][1][cl] = s2; }
  // per-channel parameters are fetched before the barrier (off the dependent tail of this latency-bound kernel)
  float p_gamma = 1.f, p_beta = 0.f, p_rm = 0.f, p_rv = 0.f;
  if (rg == 0 && c < C) {
    if (gamma) p_gamma = gamma[c];
    if (beta) p_beta = beta[c];
    if (rmean) p_rm = rmean[c];
    if (rvar) p_rv = rvar[c];
  }
  __syncthreads();
  if (rg == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mu = (float)mean, is = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[c] = mu;
    invstd_out[c] = is;
    const float sc = is * p_gamma;                           // y = x * scale + shift
    scale_shift[c] = sc; scale_shift[C + c] = p_beta - mu * sc;
    if (rmean) rmean[c] = (1.f - momentum) * p_rm + momentum * (float)mean;
    if (rvar) rvar[c] = (1.f - momentum) * p_rv + momentum * (float)(count > 1.0 ? var * count / (count - 1.0) : var);
  }
}

// stand-alone statistics (used when the producer is not a conv epilogue): partial[(blockIdx.y)][2][C]
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ x, int64_t M, int C, int rows_per_block,
                                                        float* __restrict__ partial) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256 * V];        // [which][row-lane][chunk-lane * V]
  const int cpr = C / V, cw = cpr < 32 ? cpr : 32, nrl = 256 / cw;     // cw chunk lanes x nrl row lanes
  const int cch = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c0 = (blockIdx.x * cw + cch) * V;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (c0 < C)
    for (int64_t r = r0 + rl; r < r1; r += nrl) {
      float v[V];
      Vec16<T>::load(x + r * C + c0, v);
#pragma unroll
      for (int k = 0; k < V; ++k) { s1[k] += v[k]; s2[k] = fmaf(v[k], v[k], s2[k]); }
    }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][(rl * cw + cch) * V + k] = s1[k]; red[1][(rl * cw + cch) * V + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cw * V; i += 256) {
    const int which = i / (cw * V), cl = i - which * cw * V;
    float a = 0.f;
    for (int q = 0; q < nrl; ++q) a += red[which][q * cw * V + cl];
    const int c = blockIdx.x * cw * V + cl;
    if (c < C) partial[((int64_t)blockIdx.y * 2 + which) * C + c] = a;
  }
}

// ------------------------------------------------------------------ BN apply (+residual)(+ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift,
                                                         const T* __restrict__ res, int relu, int64_t M, int C,
                                                         T* __restrict__ y, uint8_t* __restrict__ mask_out,
                                                         const float* __restrict__ res_scale_shift) {
  // res_scale_shift (nullable): `res` is the RAW output of the block's downsample convolution and its BatchNorm
  // (scale, shift) is applied here, r = res * s2 + t2 -- bn3 + downsample-BN + add + ReLU in one pass
  // (resnet.py:80-85), the normalised downsample tensor is never written
  // mask_out (bf16 only, V = 8): bit k of byte i = (y[8 i + k] > 0) -- the ReLU mask the backward needs, at 1/16 of the
  // bytes of re-reading y there
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V;
  const int64_t total = M * cpr, nthreads = (int64_t)gridDim.x * 256;
  const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(gtid % cpr) * V;     // nthreads % cpr == 0 -> this thread's channels never change
  float sc[V], sh[V];
#pragma unroll
  for (int k = 0; k < V; k += 4) {        // per-channel coefficients: 16-B loads, precomputed by the finalize
    const float4 a = *reinterpret_cast<const float4*>(scale_shift + c0 + k);
    const float4 b = *reinterpret_cast<const float4*>(scale_shift + C + c0 + k);
    sc[k] = a.x; sc[k + 1] = a.y; sc[k + 2] = a.z; sc[k + 3] = a.w;
    sh[k] = b.x; sh[k + 1] = b.y; sh[k + 2] = b.z; sh[k + 3] = b.w;
  }
  float sc2[V], sh2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { sc2[k] = 1.f; sh2[k] = 0.f; }
  if (res_scale_shift) {
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(res_scale_shift + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(res_scale_shift + C + c0 + k);
      sc2[k] = a.x; sc2[k + 1] = a.y; sc2[k + 2] = a.z; sc2[k + 3] = a.w;
      sh2[k] = b.x; sh2[k + 1] = b.y; sh2[k + 2] = b.z; sh2[k + 3] = b.w;
    }
  }
  for (int64_t i = gtid; i < total; i += 2 * nthreads) {      // two independent 16-B streams in flight per thread
    const int64_t i2 = i + nthreads;
    const bool two = i2 < total;
    float v[V], w[V], rv[V], rw[V];
    Vec16<T>::load(x + i * V, v);
    if (two) Vec16<T>::load(x + i2 * V, w);
    if (res) {
      Vec16<T>::load(res + i * V, rv);
      if (two) Vec16<T>::load(res + i2 * V, rw);
      if (res_scale_shift) {
#pragma unroll
        for (int k = 0; k < V; ++k) { rv[k] = fmaf(rv[k], sc2[k], sh2[k]); if (two) rw[k] = fmaf(rw[k], sc2[k], sh2[k]); }
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      v[k] = fmaf(v[k], sc[k], sh[k]) + (res ? rv[k] : 0.f);
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    Vec16<T>::store(y + i * V, v);
    if (mask_out) {
      unsigned m = 0;
#pragma unroll
      for (int k = 0; k < V; ++k) m |= (v[k] > 0.f ? 1u : 0u) << k;
      mask_out[i] = (uint8_t)m;
    }
    if (two) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        w[k] = fmaf(w[k], sc[k], sh[k]) + (res ? rw[k] : 0.f);
        if (relu) w[k] = fmaxf(w[k], 0.f);
      }
      Vec16<T>::store(y + i2 * V, w);
      if (mask_out) {
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) m |= (w[k] > 0.f ? 1u : 0u) << k;
        mask_out[i2] = (uint8_t)m;
      }
    }
  }
}

// ------------------------------------------------------------------ training-mode BN: finalize + apply in ONE launch
// (round 5).  For layers with few statistic rows (M <= 32768: <= 256 partial rows) every apply workgroup
// derives the coefficients of ITS OWN channel strip itself -- no finalize launch in front of it, no grid-wide wait, no atomics:
//   grid (C / (8 V) strips, row blocks), 256 threads = 8 chunk lanes (16 bytes = V channels each) x 32 row lanes;
//   head: the strip's (sum, sum of squares) partial rows summed in fp64 (32 row lanes -> three wave shuffles -> the four waves
//         meet in LDS), then mean / invstd / (scale, shift) exactly as bn2d_finalize_kernel evaluates them (same fp64 expressions,
//         same conversions; a sum of <= 256 fp32 addends is exact in fp64, hence independent of the association);
//   body: y = max(x * scale + shift (+ residual), 0) over the workgroup's row range of the strip, ReLU bits as bn2d_apply_kernel.
// The row-block-0 workgroups also publish mean / invstd / scale_shift (the backward reads them) and update the running
// statistics.  What is redundant is the head: row_blocks x the partial rows of a strip, served by L2 -- (row blocks / 64) of the
// tensor's own traffic, which is why the launcher keeps the row blocks few.
template <typename T>
__global__ __launch_bounds__(256) void bn2d_fin_apply_kernel(const T* __restrict__ x, const float* __restrict__ partial, int rows,
                                                             int C, int64_t M, double count, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float momentum, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                             float* __restrict__ scale_shift, const T* __restrict__ res, int relu,
                                                             T* __restrict__ y, uint8_t* __restrict__ mask_out,
                                                             int rows_per_block) {
  constexpr int V = Vec16<T>::N, SW = 8 * V;                     // strip width in channels (64 / 32)
  constexpr int RL = 256 / SW;                                   // row lanes of the head (4 / 8)
  __shared__ double red[RL][2][SW];
  __shared__ float coef[2][SW];
  const int tc = threadIdx.x & 7, tr = threadIdx.x >> 3;
  const int cbase = (int)blockIdx.x * SW, c0 = cbase + tc * V;
  // head, one channel per lane: lane (ch, rl) sums rows rl, rl + RL, ... of its channel's two statistic columns (coalesced SW-float
  // rows, eight independent loads in flight), the RL row lanes meet in LDS, lane (ch, 0) evaluates ONE channel's coefficients
  const int ch = threadIdx.x % SW, rl = threadIdx.x / SW;
  double s1 = 0.0, s2 = 0.0;
  {
    const float* p = partial + cbase + ch;
    int r = rl;
    for (; r + 3 * RL < rows; r += 4 * RL) {
      const float a0 = p[((int64_t)r * 2) * C], b0 = p[((int64_t)r * 2 + 1) * C];
      const float a1 = p[((int64_t)(r + RL) * 2) * C], b1 = p[((int64_t)(r + RL) * 2 + 1) * C];
      const float a2 = p[((int64_t)(r + 2 * RL) * 2) * C], b2 = p[((int64_t)(r + 2 * RL) * 2 + 1) * C];
      const float a3 = p[((int64_t)(r + 3 * RL) * 2) * C], b3 = p[((int64_t)(r + 3 * RL) * 2 + 1) * C];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < rows; r += RL) { s1 += (double)p[((int64_t)r * 2) * C]; s2 += (double)p[((int64_t)r * 2 + 1) * C]; }
  }
  red[rl][0][ch] = s1; red[rl][1][ch] = s2;
  float p_gamma = 1.f, p_beta = 0.f, p_rm = 0.f, p_rv = 0.f;
  const bool publish = blockIdx.y == 0;
  if (rl == 0) {
    if (gamma) p_gamma = gamma[cbase + ch];
    if (beta) p_beta = beta[cbase + ch];
    if (publish && rmean) p_rm = rmean[cbase + ch];
    if (publish && rvar) p_rv = rvar[cbase + ch];
  }
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  __syncthreads();
  if (rl == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < RL; ++q) { t1 += red[q][0][ch]; t2 += red[q][1][ch]; }
    const double mean = t1 / count;
    double var = t2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mu = (float)mean, is = (float)(1.0 / sqrt(var + (double)eps));
    const float scv = is * p_gamma, shv = p_beta - mu * scv;
    coef[0][ch] = scv; coef[1][ch] = shv;
    if (publish) {
      const int c = cbase + ch;
      mean_out[c] = mu; invstd_out[c] = is;
      scale_shift[c] = scv; scale_shift[C + c] = shv;
      if (rmean) rmean[c] = (1.f - momentum) * p_rm + momentum * (float)mean;
      if (rvar) rvar[c] = (1.f - momentum) * p_rv + momentum * (float)(count > 1.0 ? var * count / (count - 1.0) : var);
    }
  }
  __syncthreads();
  float sc[V], sh[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { sc[k] = coef[0][tc * V + k]; sh[k] = coef[1][tc * V + k]; }
  // body: four independent rows in flight per thread
  for (int64_t r = r0 + tr; r < r1; r += 128) {
    float v[4][V], rv[4][V];
    bool on[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = r + 32 * u;
      on[u] = rr < r1;
      if (on[u]) {
        Vec16<T>::load(x + rr * C + c0, v[u]);
        if (res) Vec16<T>::load(res + rr * C + c0, rv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!on[u]) continue;
      const int64_t rr = r + 32 * u;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        v[u][k] = fmaf(v[u][k], sc[k], sh[k]) + (res ? rv[u][k] : 0.f);
        if (relu) v[u][k] = fmaxf(v[u][k], 0.f);
      }
      Vec16<T>::store(y + rr * C + c0, v[u]);
      if (mask_out) {
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) m |= (v[u][k] > 0.f ? 1u : 0u) << k;
        mask_out[(rr * C + c0) / V] = (uint8_t)m;
      }
    }
  }
}

// Gradient of the max-pool at input pixel (b, iy, ix), channels [cc*V, cc*V+V): the (up to four) windows that contain the
// pixel, each contributing where its saved argmax tap points here -- maxpool_bwd_kernel's gather as a device function, so
// that the stem's BatchNorm backward can consume the pooled gradient without the full-resolution copy being written.
template <typename T>
__device__ __forceinline__ void pool_grad_gather(const T* __restrict__ dy, const uint8_t* __restrict__ idx, int b, int iy,
                                                 int ix, int cc, int OH, int OW, int cpr, float (&acc)[Vec16<T>::N]) {
  constexpr int V = Vec16<T>::N;
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int t = iy + 1 - r;
    if (t < 0 || (t & 1)) continue;
    const int oy = t >> 1;
    if (oy >= OH) continue;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int u = ix + 1 - s;
      if (u < 0 || (u & 1)) continue;
      const int ox = u >> 1;
      if (ox >= OW) continue;
      const int64_t o = ((((int64_t)b * OH + oy) * OW + ox) * cpr + cc) * V;
      float g[V];
      Vec16<T>::load(dy + o, g);
      uint8_t tap[V];
      if constexpr (V == 8) *reinterpret_cast<uint2*>(tap) = *reinterpret_cast<const uint2*>(idx + o);
      else *reinterpret_cast<unsigned*>(tap) = *reinterpret_cast<const unsigned*>(idx + o);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += (tap[k] == (uint8_t)(r * 3 + s)) ? g[k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = Vec16<T>::rnd(acc[k]);     // the value maxpool_bwd_kernel would have stored
}

struct PoolGrad {              // non-null dy: the BatchNorm backward's incoming gradient is max-pool-backward(dy, idx)
  const void* dy;
  const uint8_t* idx;
  int H, W;                    // the BatchNorm tensor's spatial size (pool input)
};

// ------------------------------------------------------------------ BN backward
// dy = g * (act > 0 if act) ; partial[(blockIdx.y)][2][C] = (sum dy, sum dy * xhat)
template <typename T>
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                              const T* __restrict__ act,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int64_t M, int C,
                                                              int rows_per_block, float* __restrict__ partial,
                                                              const uint8_t* __restrict__ mask, PoolGrad pg) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256 * V];
  const int cpr = C / V, cw = cpr < 32 ? cpr : 32, nrl = 256 / cw;
  const int cch = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c0 = (blockIdx.x * cw + cch) * V;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s1[V], s2[V], mu[V], is[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = 0.f; is[k] = 0.f; }
  if (c0 < C) {
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(mean + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(invstd + c0 + k);
      mu[k] = a.x; mu[k + 1] = a.y; mu[k + 2] = a.z; mu[k + 3] = a.w;
      is[k] = b.x; is[k + 1] = b.y; is[k + 2] = b.z; is[k + 3] = b.w;
    }
    for (int64_t r = r0 + rl; r < r1; r += nrl) {
      float xv[V], gv[V];
      Vec16<T>::load(x + r * C + c0, xv);
      if (pg.dy) {                                 // g = max-pool backward of the pooled gradient, gathered here
        const int hw = pg.H * pg.W, bb = (int)(r / hw), rem = (int)(r - (int64_t)bb * hw);
        pool_grad_gather<T>(reinterpret_cast<const T*>(pg.dy), pg.idx, bb, rem / pg.W, rem % pg.W, c0 / V, pg.H / 2, pg.W / 2, cpr, gv);
      } else {
        Vec16<T>::load(g + r * C + c0, gv);
      }
      if (mask) {                                  // ReLU mask as bits (bn2d_apply_kernel's mask_out), 8 channels per byte
        const unsigned m = mask[(r * C + c0) / V];
#pragma unroll
        for (int k = 0; k < V; ++k) gv[k] = ((m >> k) & 1u) ? gv[k] : 0.f;
      } else if (act) {
        float av[V];
        Vec16<T>::load(act + r * C + c0, av);
#pragma unroll
        for (int k = 0; k < V; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < V; ++k) { s1[k] += gv[k]; s2[k] = fmaf(gv[k], (xv[k] - mu[k]) * is[k], s2[k]); }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][(rl * cw + cch) * V + k] = s1[k]; red[1][(rl * cw + cch) * V + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cw * V; i += 256) {
    const int which = i / (cw * V), cl = i - which * cw * V;
    float a = 0.f;
    for (int q = 0; q < nrl; ++q) a += red[which][q * cw * V + cl];
    const int c = blockIdx.x * cw * V + cl;
    if (c < C) partial[((int64_t)blockIdx.y * 2 + which) * C + c] = a;
  }
}

// Downsample blocks: the block's incoming gradient g (through the output ReLU's bits) feeds TWO BatchNorm backward passes -- bn3
// (dx = A g' + B x + C, coefficients final) and the downsample branch's BatchNorm, which first needs its column sums
// (sum g', sum g' * xhat2).  This kernel does both in one pass over g and the bits: thread map, summation order and partial-row
// layout are bn2d_bwd_reduce_kernel's, the dx expression is bn2d_bwd_apply_kernel's -- results bit-identical to the two launches,
// one read of g (+ bits) and one launch less per downsample block.
template <typename T>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_reduce2_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                                     const uint8_t* __restrict__ mask,
                                                                     const float* __restrict__ coef, int64_t M, int C,
                                                                     int rows_per_block, T* __restrict__ dx,
                                                                     const T* __restrict__ x2, const float* __restrict__ mean2,
                                                                     const float* __restrict__ invstd2,
                                                                     float* __restrict__ partial2) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256 * V];
  const int cpr = C / V, cw = cpr < 32 ? cpr : 32, nrl = 256 / cw;
  const int cch = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c0 = (blockIdx.x * cw + cch) * V;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s1[V], s2[V], mu[V], is[V], ca[V], cb[V], cc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = 0.f; is[k] = 0.f; }
  if (c0 < C) {
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(mean2 + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(invstd2 + c0 + k);
      mu[k] = a.x; mu[k + 1] = a.y; mu[k + 2] = a.z; mu[k + 3] = a.w;
      is[k] = b.x; is[k + 1] = b.y; is[k + 2] = b.z; is[k + 3] = b.w;
      const float4 pa = *reinterpret_cast<const float4*>(coef + c0 + k);
      const float4 pb = *reinterpret_cast<const float4*>(coef + C + c0 + k);
      const float4 pc = *reinterpret_cast<const float4*>(coef + 2 * C + c0 + k);
      ca[k] = pa.x; ca[k + 1] = pa.y; ca[k + 2] = pa.z; ca[k + 3] = pa.w;
      cb[k] = pb.x; cb[k + 1] = pb.y; cb[k + 2] = pb.z; cb[k + 3] = pb.w;
      cc[k] = pc.x; cc[k + 1] = pc.y; cc[k + 2] = pc.z; cc[k + 3] = pc.w;
    }
    for (int64_t r = r0 + rl; r < r1; r += nrl) {
      float xv[V], gv[V], x2v[V], o[V];
      Vec16<T>::load(x + r * C + c0, xv);
      Vec16<T>::load(g + r * C + c0, gv);
      Vec16<T>::load(x2 + r * C + c0, x2v);
      const unsigned m = mask ? mask[(r * C + c0) / V] : 0xffu;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        gv[k] = ((m >> k) & 1u) ? gv[k] : 0.f;
        o[k] = fmaf(ca[k], gv[k], fmaf(cb[k], xv[k], cc[k]));
        s1[k] += gv[k]; s2[k] = fmaf(gv[k], (x2v[k] - mu[k]) * is[k], s2[k]);
      }
      Vec16<T>::store(dx + r * C + c0, o);
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][(rl * cw + cch) * V + k] = s1[k]; red[1][(rl * cw + cch) * V + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cw * V; i += 256) {
    const int which = i / (cw * V), cl = i - which * cw * V;
    float a = 0.f;
    for (int q = 0; q < nrl; ++q) a += red[which][q * cw * V + cl];
    const int c = blockIdx.x * cw * V + cl;
    if (c < C) partial2[((int64_t)blockIdx.y * 2 + which) * C + c] = a;
  }
}

// sums[2][C] = sum over row blocks; dgamma += sum dy*xhat ; dbeta += sum dy
template <int CW>
__global__ __launch_bounds__(1024) void bn2d_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                                double count, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                float* __restrict__ sums, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  constexpr int RG = 1024 / CW;                      // row groups: CW = 16 for wide layers, 4 when C is small and rows many
  __shared__ double red[16][2][CW];
  const int cl = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int c = blockIdx.x * CW + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
  {
    int r = rg;
    for (; r + 3 * RG < rows; r += 4 * RG) {          // 8 independent loads per trip
      const float a0 = partial[((int64_t)r * 2) * C + c], b0 = partial[((int64_t)r * 2 + 1) * C + c];
      const float a1 = partial[((int64_t)(r + RG) * 2) * C + c], b1 = partial[((int64_t)(r + RG) * 2 + 1) * C + c];
      const float a2 = partial[((int64_t)(r + 2 * RG) * 2) * C + c], b2 = partial[((int64_t)(r + 2 * RG) * 2 + 1) * C + c];
      const float a3 = partial[((int64_t)(r + 3 * RG) * 2) * C + c], b3 = partial[((int64_t)(r + 3 * RG) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < rows; r += RG) {
      s1 += (double)partial[((int64_t)r * 2) * C + c];
      s2 += (double)partial[((int64_t)r * 2 + 1) * C + c];
    }
  }
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((threadIdx.x & 63) < CW) { red[threadIdx.x >> 6][0][cl] = s1; red[threadIdx.x >> 6][1][cl] = s2; }
  float p_mu = 0.f, p_is = 0.f, p_g = 1.f, p_db = 0.f, p_dg = 0.f;
  if (rg == 0 && c < C) {
    p_mu = mean[c]; p_is = invstd[c];
    if (gamma) p_g = gamma[c];
    if (dbeta) p_db = dbeta[c];
    if (dgamma) p_dg = dgamma[c];
  }
  __syncthreads();
  if (rg == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
    // dx = k1*(dy - a1 - xhat*a2) = A*dy + B*x + Cc  with per-channel A, B, Cc
    const float invM = (float)(1.0 / count);
    const float mu = p_mu, is = p_is, k1 = p_g * is;
    const float a1 = (float)s1 * invM, a2 = (float)s2 * invM;
    sums[c] = k1;
    sums[C + c] = -k1 * is * a2;
    sums[2 * C + c] = -k1 * a1 + k1 * is * a2 * mu;
    if (dbeta) dbeta[c] = p_db + (float)s1;
    if (dgamma) dgamma[c] = p_dg + (float)s2;
  }
}

// dx = A*dy + B*x + Cc (per-channel coefficients from the finalize); optional gm_out = dy (ReLU-masked g)
template <typename T>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                             const T* __restrict__ act,
                                                             const float* __restrict__ coef, int64_t M, int C,
                                                             T* __restrict__ dx, T* __restrict__ gm_out,
                                                             const uint8_t* __restrict__ mask, PoolGrad pg) {
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V;
  const int64_t total = M * cpr, nthreads = (int64_t)gridDim.x * 256;
  const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(gtid % cpr) * V;
  float ca[V], cb[V], cc[V];
#pragma unroll
  for (int k = 0; k < V; k += 4) {
    const float4 a = *reinterpret_cast<const float4*>(coef + c0 + k);
    const float4 b = *reinterpret_cast<const float4*>(coef + C + c0 + k);
    const float4 c = *reinterpret_cast<const float4*>(coef + 2 * C + c0 + k);
    ca[k] = a.x; ca[k + 1] = a.y; ca[k + 2] = a.z; ca[k + 3] = a.w;
    cb[k] = b.x; cb[k + 1] = b.y; cb[k + 2] = b.z; cb[k + 3] = b.w;
    cc[k] = c.x; cc[k + 1] = c.y; cc[k + 2] = c.z; cc[k + 3] = c.w;
  }
  if (!pg.dy && !act) {
    // common path (mask bits or no ReLU): two independent chunks per trip, every load issued before the first use
    for (int64_t i = gtid; i < total; i += 2 * nthreads) {
      const int64_t i2 = i + nthreads;
      const bool two = i2 < total;
      float xv[V], gv[V], xw[V], gw[V];
      unsigned m1 = 0xffu, m2 = 0xffu;
      Vec16<T>::load(x + i * V, xv);
      Vec16<T>::load(g + i * V, gv);
      if (mask) m1 = mask[i];
      if (two) {
        Vec16<T>::load(x + i2 * V, xw);
        Vec16<T>::load(g + i2 * V, gw);
        if (mask) m2 = mask[i2];
      }
      float o[V];
#pragma unroll
      for (int k = 0; k < V; ++k) { gv[k] = ((m1 >> k) & 1u) ? gv[k] : 0.f; o[k] = fmaf(ca[k], gv[k], fmaf(cb[k], xv[k], cc[k])); }
      if (gm_out) Vec16<T>::store(gm_out + i * V, gv);
      Vec16<T>::store(dx + i * V, o);
      if (two) {
#pragma unroll
        for (int k = 0; k < V; ++k) { gw[k] = ((m2 >> k) & 1u) ? gw[k] : 0.f; o[k] = fmaf(ca[k], gw[k], fmaf(cb[k], xw[k], cc[k])); }
        if (gm_out) Vec16<T>::store(gm_out + i2 * V, gw);
        Vec16<T>::store(dx + i2 * V, o);
      }
    }
    return;
  }
  for (int64_t i = gtid; i < total; i += nthreads) {
    float xv[V], gv[V];
    Vec16<T>::load(x + i * V, xv);
    if (pg.dy) {
      const int64_t r = i / cpr;
      const int hw = pg.H * pg.W, bb = (int)(r / hw), rem = (int)(r - (int64_t)bb * hw);
      pool_grad_gather<T>(reinterpret_cast<const T*>(pg.dy), pg.idx, bb, rem / pg.W, rem % pg.W, (int)(i - r * cpr), pg.H / 2,
                          pg.W / 2, cpr, gv);
    } else {
      Vec16<T>::load(g + i * V, gv);
    }
    if (mask) {
      const unsigned m = mask[i];
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = ((m >> k) & 1u) ? gv[k] : 0.f;
    } else if (act) {
      float av[V];
      Vec16<T>::load(act + i * V, av);
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
    }
    if (gm_out) Vec16<T>::store(gm_out + i * V, gv);
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = fmaf(ca[k], gv[k], fmaf(cb[k], xv[k], cc[k]));
    Vec16<T>::store(dx + i * V, o);
  }
}

// ------------------------------------------------------------------ max-pool 3x3 s2 p1
// the V argmax taps of one thread as ONE 4- / 8-byte store (byte stores cost a full write transaction each)
template <int V>
__device__ __forceinline__ void store_taps(uint8_t* p, const uint8_t (&bi)[V]) {
  static_assert(V == 4 || V == 8, "16-byte vectors of 4- or 2-byte elements");
  uint32_t lo = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  if constexpr (V == 4) {
    *reinterpret_cast<uint32_t*>(p) = lo;
  } else {
    uint32_t hi = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
    *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int B, int H, int W, int C,
                                                          T* __restrict__ y, uint8_t* __restrict__ idx) {
  constexpr int V = Vec16<T>::N;
  const int OH = H / 2, OW = W / 2, cpr = C / V;
  const int64_t total = (int64_t)B * OH * OW * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cc, ox, oy, b;
    if (total < (int64_t)1 << 31) {                      // 32-bit index arithmetic (a 64-bit division is ~40 VALU instructions)
      const unsigned u = (unsigned)i, q1 = u / (unsigned)cpr, q2 = q1 / (unsigned)OW;
      cc = (int)(u - q1 * cpr); ox = (int)(q1 - q2 * OW); b = (int)(q2 / (unsigned)OH); oy = (int)(q2 - (unsigned)b * OH);
    } else {
      cc = (int)(i % cpr);
      int64_t p = i / cpr;
      ox = (int)(p % OW); p /= OW;
      oy = (int)(p % OH);
      b = (int)(p / OH);
    }
    float best[V];
    uint8_t bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iy = oy * 2 + r - 1, ix = ox * 2 + s - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          float v[V];
          Vec16<T>::load(x + (((int64_t)b * H + iy) * W + ix) * C + cc * V, v);
#pragma unroll
          for (int k = 0; k < V; ++k)
            if (v[k] > best[k]) { best[k] = v[k]; bi[k] = (uint8_t)(r * 3 + s); }   // first max wins
        }
      }
    Vec16<T>::store(y + i * V, best);
    if (idx) store_taps<V>(idx + i * V, bi);
  }
}

// BatchNorm apply (+ReLU) fused into the 3x3 s2 p1 max-pool of the stem (modelling/backbones/resnet.py:123-126: conv1 -> bn1
// -> [relu] -> maxpool): the normalised [B, H, W, C] tensor is only ever read by the pool, so it is never written.  Every
// tap value is rounded to the storage type before the comparison -- the pooled values and argmax taps are bit-identical to
// bn2d_apply_kernel followed by maxpool_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift,
                                                               int relu, int B, int H, int W, int C, T* __restrict__ y,
                                                               uint8_t* __restrict__ idx) {
  constexpr int V = Vec16<T>::N;
  const int OH = H / 2, OW = W / 2, cpr = C / V;
  const int64_t total = (int64_t)B * OH * OW * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cc, ox, oy, b;
    if (total < (int64_t)1 << 31) {                      // 32-bit index arithmetic (a 64-bit division is ~40 VALU instructions)
      const unsigned u = (unsigned)i, q1 = u / (unsigned)cpr, q2 = q1 / (unsigned)OW;
      cc = (int)(u - q1 * cpr); ox = (int)(q1 - q2 * OW); b = (int)(q2 / (unsigned)OH); oy = (int)(q2 - (unsigned)b * OH);
    } else {
      cc = (int)(i % cpr);
      int64_t p = i / cpr;
      ox = (int)(p % OW); p /= OW;
      oy = (int)(p % OH);
      b = (int)(p / OH);
    }
    float sc[V], sh[V], best[V];
    uint8_t bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = scale_shift[cc * V + k]; sh[k] = scale_shift[C + cc * V + k]; best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iy = oy * 2 + r - 1, ix = ox * 2 + s - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          float v[V];
          Vec16<T>::load(x + (((int64_t)b * H + iy) * W + ix) * C + cc * V, v);
#pragma unroll
          for (int k = 0; k < V; ++k) {
            float yv = fmaf(v[k], sc[k], sh[k]);
            if (relu) yv = fmaxf(yv, 0.f);
            yv = Vec16<T>::rnd(yv);
            if (yv > best[k]) { best[k] = yv; bi[k] = (uint8_t)(r * 3 + s); }   // first max wins
          }
        }
      }
    Vec16<T>::store(y + i * V, best);
    if (idx) store_taps<V>(idx + i * V, bi);
  }
}

// one workgroup per input row (b, iy): no 64-bit index divisions on the per-element path (they were ~3/4 of the
// kernel's time), the 8 argmax bytes of a window come in one 8-byte load
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C, T* __restrict__ dx) {
  // A pixel lies in 1 (even coordinate) or 2 (odd) windows per axis.  The row parity is uniform over the workgroup; the
  // column parity is made uniform per pass (even columns, then odd ones), so no wave executes window taps that only
  // some of its lanes need -- with mixed parities every wave walked all three column taps under partial masks.
  constexpr int V = Vec16<T>::N;
  const int OH = H / 2, OW = W / 2, cpr = C / V;
  const int row = blockIdx.x;                       // b * H + iy
  const int b = row / H, iy = row - b * H;
  T* drow = dx + (int64_t)row * W * C;
  const int nr = (iy & 1) ? 2 : 1;
  const int rr0 = (iy & 1) ? 0 : 1, rstep = 2;      // kernel rows r = rr0, rr0 + 2 (odd iy) or r = 1 (even iy)
  for (int par = 0; par < 2; ++par) {
    const int ns = par ? 2 : 1, ss0 = par ? 0 : 1;
    for (int v = threadIdx.x; v < (W / 2) * cpr; v += 256) {
      const int ix = 2 * (v / cpr) + par, cc = v % cpr;
      float acc[V];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = 0.f;
      for (int ri = 0; ri < nr; ++ri) {
        const int r = rr0 + ri * rstep, oy = (iy + 1 - r) >> 1;
        if (oy >= OH) continue;
        for (int si = 0; si < ns; ++si) {
          const int sx = ss0 + si * 2, ox = (ix + 1 - sx) >> 1;
          if (ox >= OW) continue;
          const int64_t o = ((((int64_t)b * OH + oy) * OW + ox) * cpr + cc) * V;
          float g[V];
          Vec16<T>::load(dy + o, g);
          uint8_t tap[V];
          if constexpr (V == 8) *reinterpret_cast<uint2*>(tap) = *reinterpret_cast<const uint2*>(idx + o);
          else *reinterpret_cast<unsigned*>(tap) = *reinterpret_cast<const unsigned*>(idx + o);
          const uint8_t want = (uint8_t)(r * 3 + sx);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += (tap[k] == want) ? g[k] : 0.f;
        }
      }
      Vec16<T>::store(drow + ((int64_t)ix * cpr + cc) * V, acc);
    }
  }
}

// ------------------------------------------------------------------ global average pool
// 32 channel vectors x 8 row lanes per workgroup; grid (C/V/32, B): HW/8 dependent loads per thread and
// B*C/(32V) workgroups instead of HW loads and B workgroups; lane partials meet in LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, int HW, int C, float* __restrict__ out,
                                                      int64_t* __restrict__ counter) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[8][32][V + 1];
  if (counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *counter += 1;   // creid_gap_fwd_count
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int b = blockIdx.y, cc = blockIdx.x * 32 + cl;
  const bool live = cc * V < C;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (live) {
    const T* p = x + (int64_t)b * HW * C + cc * V;
    for (int i = rl; i < HW; i += 8) {
      float v[V];
      Vec16<T>::load(p + (int64_t)i * C, v);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[rl][cl][k] = acc[k];
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * V; e += 256) {
    const int c2 = e / V, k = e - c2 * V;
    const int ch = (blockIdx.x * 32 + c2) * V + k;
    if (ch < C) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) a += red[q][c2][k];
      out[(int64_t)b * C + ch] = a / (float)HW;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dfeat, int HW, int C, int64_t total_chunks,
                                                      T* __restrict__ dx) {
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * 256) {
    int cc;
    int64_t b;
    if (total_chunks < (int64_t)1 << 31) { const unsigned u = (unsigned)i; cc = (int)(u % (unsigned)cpr); b = u / (unsigned)(cpr * HW); }
    else { cc = (int)(i % cpr); b = i / ((int64_t)cpr * HW); }
    float v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = dfeat[b * C + cc * V + k] * inv;
    Vec16<T>::store(dx + i * V, v);
  }
}

// ------------------------------------------------------------------ layout transforms
// NCHW fp32 image -> zero-padded NHWC4 [B, H+8, W+6, 4] (image at rows 3.., cols 3..).  Eight padded rows per workgroup trip (no
// 64-bit index divisions on the per-pixel path), a pixel's four values as ONE 8- / 16-byte store.
__device__ __forceinline__ void store_px4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store_px4(unsigned short* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(Bf16T::pack2(v[0], v[1]), Bf16T::pack2(v[2], v[3]));
}
__device__ __forceinline__ void store_px4(_Float16* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(F16T::pack2(v[0], v[1]), F16T::pack2(v[2], v[3]));
}
template <typename T>
__global__ __launch_bounds__(256) void image_pad_kernel(const float* __restrict__ x, int B, int H, int W, T* __restrict__ y) {
  const int PH = H + 8, PW = W + 6;
  const int rows = B * PH;
  constexpr int RB = 8;                                 // padded rows per workgroup trip (32-bit index arithmetic inside)
  if ((W & 3) == 0) {
    // a unit = four image pixels of one row (three 16-byte plane loads -> 32 contiguous output bytes) or one of the row's two
    // 3-pixel zero borders; rows above / below the image are all zeros
    const int G = W / 4 + 2;
    for (int row0 = blockIdx.x * RB; row0 < rows; row0 += gridDim.x * RB) {
      const int n = min(RB, rows - row0) * G;
      for (int e = threadIdx.x; e < n; e += 256) {
        const int rl = e / G, gq = e - rl * G, row = row0 + rl;
        const int b = row / PH, py = row - b * PH, iy = py - 3;
        T* dst = y + (int64_t)row * PW * 4;
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (gq == 0 || gq == G - 1) {
          T* d = dst + (gq == 0 ? 0 : (W + 3) * 4);
          store_px4(d, z); store_px4(d + 4, z); store_px4(d + 8, z);
          continue;
        }
        const int ix = (gq - 1) * 4;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
        if ((unsigned)iy < (unsigned)H) {
          const float* src = x + (((int64_t)b * 3) * H + iy) * W + ix;
          p0 = *reinterpret_cast<const float4*>(src);
          p1 = *reinterpret_cast<const float4*>(src + (int64_t)H * W);
          p2 = *reinterpret_cast<const float4*>(src + (int64_t)2 * H * W);
        }
        T* d = dst + (ix + 3) * 4;
        const float v0[4] = {p0.x, p1.x, p2.x, 0.f}, v1[4] = {p0.y, p1.y, p2.y, 0.f};
        const float v2[4] = {p0.z, p1.z, p2.z, 0.f}, v3[4] = {p0.w, p1.w, p2.w, 0.f};
        store_px4(d, v0); store_px4(d + 4, v1); store_px4(d + 8, v2); store_px4(d + 12, v3);
      }
    }
    return;
  }
  for (int row0 = blockIdx.x * RB; row0 < rows; row0 += gridDim.x * RB) {
    const int n = min(RB, rows - row0) * PW;
    for (int e = threadIdx.x; e < n; e += 256) {
      const int rl = e / PW, px = e - rl * PW, row = row0 + rl;
      const int b = row / PH, py = row - b * PH, iy = py - 3, ix = px - 3;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const float* src = x + (((int64_t)b * 3) * H + iy) * W + ix;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = src[(int64_t)c * H * W];
      }
      store_px4(y + ((int64_t)row * PW + px) * 4, v);
    }
  }
}

// OIHW fp32 -> [O][r][s][I] (forward) and [I][r][s][O] (data gradient), in the compute dtype
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, int O, int I, int kh, int kw,
                                                          T* __restrict__ w_krsc, T* __restrict__ w_crsk) {
  const int64_t total = (int64_t)O * I * kh * kw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i enumerates the KRSC layout (coalesced writes)
    const int c = (int)(i % I);
    int64_t t = i / I;
    const int s = (int)(t % kw); t /= kw;
    const int r = (int)(t % kh);
    const int o = (int)(t / kh);
    const float v = w[(((int64_t)o * I + c) * kh + r) * kw + s];
    ElemIO<T>::st(w_krsc + i, v);
    if (w_crsk) ElemIO<T>::st(w_crsk + (((int64_t)c * kh + r) * kw + s) * O + o, v);
  }
}

// Eval-mode BatchNorm as a per-channel affine, every layer of the network in ONE launch: a device table of
// {gamma, beta, running_mean, running_var, out[2][C], C, eps}; one workgroup per entry.  scale = gamma / sqrt(var + eps),
// shift = beta - mean * scale -- the arithmetic of bn2d_finalize_kernel's eval branch (and of torch's CPU eval transform),
// so that the folded convolution epilogue (conv_igemm.hip, g.epi_scale) equals finalize + apply.
struct BnFoldEntry { const float* gamma; const float* beta; const float* mean; const float* var; float* out; int C; float eps; };
__global__ __launch_bounds__(256) void bn_fold_multi_kernel(const BnFoldEntry* __restrict__ tab) {
  const BnFoldEntry e = tab[blockIdx.x];
  for (int c = blockIdx.y * 256 + threadIdx.x; c < e.C; c += gridDim.y * 256) {     // (grid.y = 8: one channel per thread up to C = 2048)
    const float mu = e.mean[c], is = 1.0f / sqrtf(e.var[c] + e.eps);
    const float sc = is * (e.gamma ? e.gamma[c] : 1.f);
    e.out[c] = sc;
    e.out[e.C + c] = (e.beta ? e.beta[c] : 0.f) - mu * sc;
  }
}

// All convolutions in ONE launch: a device table of {src, krsc, crsk, O, I, kh, kw, first element}; each
// thread finds its tensor by binary search over the cumulative element counts.
struct WPrepEntry { const float* w; void* krsc; void* crsk; int O, I, kh, kw; int64_t start; };
// One workgroup per (entry, 32x32 (o, c) tile): for every tap the tile is read coalesced along c, written to
// krsc coalesced along c, transposed through LDS and written to crsk coalesced along o.
template <typename T> struct Pack16 { static constexpr bool ok = false; static __device__ unsigned pack2(float, float) { return 0u; } };
template <> struct Pack16<unsigned short> { static constexpr bool ok = true; static __device__ __forceinline__ unsigned pack2(float a, float b) { return Bf16T::pack2(a, b); } };
template <> struct Pack16<_Float16> { static constexpr bool ok = true; static __device__ __forceinline__ unsigned pack2(float a, float b) { return F16T::pack2(a, b); } };

template <typename T>
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const WPrepEntry* __restrict__ tab, int n_ent,
                                                                const int* __restrict__ tile_start) {
  __shared__ float tl[32][33];
  int lo = 0, hi = n_ent - 1;
  const int bid = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile_start[mid] <= bid) lo = mid; else hi = mid - 1; }
  const WPrepEntry e = tab[lo];
  const int tiles_c = (e.I + 31) / 32;
  const int tix = bid - tile_start[lo];
  const int o0 = (tix / tiles_c) * 32, c0 = (tix % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int taps = e.kh * e.kw;
  if constexpr (Pack16<T>::ok) {
    // 16-bit copies of a full tile with <= 9 taps (every convolution of the network but the stem): the tile's 32 rows are 32
    // CONTIGUOUS runs of 32 * taps floats in the OIHW source -- 16-byte loads, converted once, parked in LDS as [o][c * taps + tap];
    // both copies leave as 16-byte stores (eight consecutive c of one (o, tap) / eight consecutive o of one (c, tap)).  The
    // per-tap form below reads the source strided by taps and writes 2-byte elements: 76 us per step against ~45.
    constexpr int MAXT = 9, P = 32 * MAXT + 8;
    __shared__ __attribute__((aligned(16))) unsigned short wb[32 * P];
    if (taps <= MAXT && e.crsk && (e.I & 31) == 0 && (e.O & 31) == 0) {
      const int rp = 32 * taps + 8, per_row = 8 * taps;          // row pitch (elements), float4 per row
      for (int idx = threadIdx.x; idx < 32 * per_row; idx += 256) {
        const int row = idx / per_row, j = idx - row * per_row;
        const float4 v = *reinterpret_cast<const float4*>(e.w + ((int64_t)(o0 + row) * e.I + c0) * taps + 4 * j);
        *reinterpret_cast<uint2*>(wb + row * rp + 4 * j) = make_uint2(Pack16<T>::pack2(v.x, v.y), Pack16<T>::pack2(v.z, v.w));
      }
      __syncthreads();
      const int half = 128 * taps;                               // items per copy: 32 rows x taps x four 8-element chunks
      for (int item = threadIdx.x; item < 2 * half; item += 256) {
        const bool to_crsk = item >= half;
        const int it = to_crsk ? item - half : item;
        const int a = it / (4 * taps), rem = it - a * 4 * taps, tap = rem >> 2, ch = rem & 3;
        unsigned short v[8];
        if (!to_crsk) {                                          // a = o: eight consecutive c
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = wb[a * rp + (8 * ch + k) * taps + tap];
        } else {                                                 // a = c: eight consecutive o
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = wb[(8 * ch + k) * rp + a * taps + tap];
        }
        const uint4 o4 = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
                                    v[6] | ((unsigned)v[7] << 16));
        T* dst = to_crsk ? reinterpret_cast<T*>(e.crsk) + ((int64_t)(c0 + a) * taps + tap) * e.O + o0 + 8 * ch
                         : reinterpret_cast<T*>(e.krsc) + ((int64_t)(o0 + a) * taps + tap) * e.I + c0 + 8 * ch;
        *reinterpret_cast<uint4*>(dst) = o4;
      }
      return;
    }
  }
  for (int tap = 0; tap < taps; ++tap) {
    // OIHW source: element (o, c, tap) at ((o*I + c)*taps + tap): gather (strided by taps) -- fp32 reads hit L2
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = o0 + ty + 8 * k, c = c0 + tx;
      tl[ty + 8 * k][tx] = (o < e.O && c < e.I) ? e.w[((int64_t)o * e.I + c) * taps + tap] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = o0 + ty + 8 * k, c = c0 + tx;
      if (o < e.O && c < e.I) ElemIO<T>::st(reinterpret_cast<T*>(e.krsc) + ((int64_t)o * taps + tap) * e.I + c, tl[ty + 8 * k][tx]);
      const int c2 = c0 + ty + 8 * k, o2 = o0 + tx;
      if (e.crsk && c2 < e.I && o2 < e.O)
        ElemIO<T>::st(reinterpret_cast<T*>(e.crsk) + ((int64_t)c2 * taps + tap) * e.O + o2, tl[tx][ty + 8 * k]);
    }
    __syncthreads();
  }
}

// stem: OIHW [64,3,7,7] -> [64][8][32] with k = r*32 + s*4 + c (zero padded)
template <typename T>
__global__ __launch_bounds__(256) void stem_weight_prep_kernel(const float* __restrict__ w, T* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 256) return;
  const int o = i >> 8, k = i & 255, r = k >> 5, s = (k & 31) >> 2, c = k & 3;
  const float v = (r < 7 && s < 7 && c < 3) ? w[((o * 3 + c) * 7 + r) * 7 + s] : 0.f;
  ElemIO<T>::st(out + i, v);
}

// NHWC (compute dtype) -> NCHW fp32 (to hand `base_out` back in the reference's layout)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, int B, int HW, int C, float* __restrict__ y) {
  const int64_t total = (int64_t)B * HW * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int64_t b = i / ((int64_t)HW * C);
    y[i] = ElemIO<T>::ld(x + (b * HW + p) * C + c);
  }
}

// ------------------------------------------------------------------ host
static inline unsigned ew_blocks_cap(int64_t total_threads_needed, int64_t mult, int64_t cap) {
  int64_t b = (total_threads_needed + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  // make blocks*256 a multiple of `mult` (mult is a power of two <= 512)
  const int64_t q = mult > 256 ? mult / 256 : 1;
  b = (b + q - 1) / q * q;
  return (unsigned)b;
}
static inline int64_t ew_env_cap() {
  static const int64_t cap = [] { const char* e = getenv("CREID_EW_BLOCKS"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 0); }();
  return cap;
}
static inline unsigned ew_blocks(int64_t total_threads_needed, int64_t mult) {
  const int64_t e = ew_env_cap();
  return ew_blocks_cap(total_threads_needed, mult, e > 0 ? e : 2048);
}
// Grid cap of a grid-stride kernel = the workgroups the chip can hold AT ONCE for THAT kernel (occupancy x compute units), asked of
// the runtime once per kernel.  A fixed 2048 (8 per CU) is right for kernels that fit 8 waves per SIMD, but the 16-bit BatchNorm
// apply / backward-apply kernels need 76-80 registers (6 waves per SIMD = 1536 workgroups): with 2048 the last 512 workgroups ran
// as a second round on a quarter-full chip -- 1.5 % of the whole B = 64 training step (round 5, profiles/r05_ew_grid.md).
template <auto KERNEL>
static inline unsigned ew_blocks_k(int64_t total_threads_needed, int64_t mult) {
  static const int64_t resident = [] {
    int per_cu = 0, dev = 0, ncu = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    return (int64_t)per_cu * ncu;
  }();
  const int64_t e = ew_env_cap();
  return ew_blocks_cap(total_threads_needed, mult, e > 0 ? e : resident);
}

#define DISPATCH_T(dtype, EXPR_F32, EXPR_BF16, EXPR_F16) \
  do {                                                  \
    if ((dtype) == CREID_F32) { EXPR_F32; }             \
    else if ((dtype) == CREID_BF16) { EXPR_BF16; }      \
    else if ((dtype) == CREID_F16) { EXPR_F16; }        \
    else return CREID_E_DTYPE;                          \
  } while (0)

// timing experiments only (results are then WRONG): CREID_BN_FIN_DRY bit 0 skips the forward finalize launches, bit 1 the
// backward ones -- measures what the 106 tiny launches cost inside a captured step
static bool fin_dry(int bit) {
  static const int v = creid_ablation_env("CREID_BN_FIN_DRY");
  return (v & bit) != 0;
}
extern "C" {

int creid_bn2d_finalize(const float* partial, int64_t rows, int64_t C, int64_t count, float* running_mean,
                        float* running_var, int training, float momentum, float eps, const float* gamma,
                        const float* beta, float* mean_out, float* invstd_out, float* scale_shift, void* stream) {
  CREID_CHECK_ARG(C > 0 && mean_out && invstd_out && scale_shift &&
                  (training ? (partial && rows > 0 && count > 0) : (running_mean && running_var)));
  // few channels, many partial rows (stem, layer1): 4 channels per workgroup -> 4x the workgroups, 1/4 of the dependent trips
  if (!fin_dry(1)) {
    if (C <= 256 && rows >= 512)
      hipLaunchKernelGGL(bn2d_finalize_kernel<4>, dim3((unsigned)((C + 3) / 4)), dim3(1024), 0, as_stream(stream), partial,
                         (int)rows, (int)C, (double)count, running_mean, running_var, training, momentum, eps, gamma, beta,
                         mean_out, invstd_out, scale_shift);
    else
      hipLaunchKernelGGL(bn2d_finalize_kernel<16>, dim3((unsigned)((C + 15) / 16)), dim3(1024), 0, as_stream(stream), partial,
                         (int)rows, (int)C, (double)count, running_mean, running_var, training, momentum, eps, gamma, beta,
                         mean_out, invstd_out, scale_shift);
  }
  CREID_LAUNCH_RET();
}

int64_t creid_col_stats_rows(int64_t M) { int64_t r = (M + 127) / 128; return r < 1 ? 1 : r; }

int creid_col_stats(const void* x, int64_t M, int64_t C, int dtype, float* partial, void* stream) {
  CREID_CHECK_ARG(x && partial && M > 0 && C > 0 && C % 8 == 0);
  const int rows = (int)creid_col_stats_rows(M);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(col_stats_kernel<float>, dim3((unsigned)((C / 4 + 31) / 32), rows), dim3(256), 0, s,
                                (const float*)x, M, (int)C, 128, partial),
             hipLaunchKernelGGL(col_stats_kernel<unsigned short>, dim3((unsigned)((C / 8 + 31) / 32), rows), dim3(256), 0,
                                s, (const unsigned short*)x, M, (int)C, 128, partial),
             hipLaunchKernelGGL(col_stats_kernel<_Float16>, dim3((unsigned)((C / 8 + 31) / 32), rows), dim3(256), 0,
                                s, (const _Float16*)x, M, (int)C, 128, partial));
  CREID_LAUNCH_RET();
}

int creid_bn2d_apply_mask(const void* x, const float* scale_shift, const void* residual, int relu, int64_t M, int64_t C,
                          int dtype, void* y, uint8_t* mask_out, void* stream) {
  CREID_CHECK_ARG(x && scale_shift && y && M > 0 && C > 0 && C % 8 == 0);
  if (mask_out && dtype == CREID_F32) return CREID_E_DTYPE;    // the bit mask is defined per 8-channel (16-byte bf16/f16) chunk
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn2d_apply_kernel<float>, dim3(ew_blocks_k<bn2d_apply_kernel<float>>(M * C / 4, C / 4)), dim3(256), 0, s,
                                (const float*)x, scale_shift, (const float*)residual, relu, M, (int)C, (float*)y,
                                (uint8_t*)nullptr, (const float*)nullptr),
             hipLaunchKernelGGL(bn2d_apply_kernel<unsigned short>, dim3(ew_blocks_k<bn2d_apply_kernel<unsigned short>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const unsigned short*)x, scale_shift, (const unsigned short*)residual, relu, M,
                                (int)C, (unsigned short*)y, mask_out, (const float*)nullptr),
             hipLaunchKernelGGL(bn2d_apply_kernel<_Float16>, dim3(ew_blocks_k<bn2d_apply_kernel<_Float16>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const _Float16*)x, scale_shift, (const _Float16*)residual, relu, M,
                                (int)C, (_Float16*)y, mask_out, (const float*)nullptr));
  CREID_LAUNCH_RET();
}

int creid_bn2d_finalize_apply_mask(const float* partial, int64_t rows, int64_t C, int64_t M, float* running_mean,
                                   float* running_var, float momentum, float eps, const float* gamma, const float* beta,
                                   float* mean_out, float* invstd_out, float* scale_shift, const void* x, const void* residual,
                                   int relu, int dtype, void* y, uint8_t* mask_out, int row_blocks, void* stream) {
  CREID_CHECK_ARG(partial && rows > 0 && C > 0 && M > 0 && mean_out && invstd_out && scale_shift && x && y);
  if (mask_out && dtype == CREID_F32) return CREID_E_DTYPE;
  const int64_t strip = dtype == CREID_F32 ? 32 : 64;            // 8 chunk lanes x 16 bytes
  if (C % strip != 0 || rows > 1024) return CREID_E_SHAPE;
  const int strips = (int)(C / strip);
  // few row blocks: the head (the strip's partial rows) is re-read by every row block; ~512 workgroups where that is cheap
  int rb = row_blocks > 0 ? row_blocks : (512 + strips - 1) / strips;
  if (row_blocks <= 0) { if (rb > 32) rb = 32; if (rb < 4) rb = 4; }
  int64_t rpb = ((M + rb - 1) / rb + 31) / 32 * 32;
  rb = (int)((M + rpb - 1) / rpb);
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)strips, (unsigned)rb);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn2d_fin_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, partial, (int)rows, (int)C, M,
                                (double)M, running_mean, running_var, momentum, eps, gamma, beta, mean_out, invstd_out, scale_shift,
                                (const float*)residual, relu, (float*)y, (uint8_t*)nullptr, (int)rpb),
             hipLaunchKernelGGL(bn2d_fin_apply_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)x, partial,
                                (int)rows, (int)C, M, (double)M, running_mean, running_var, momentum, eps, gamma, beta, mean_out,
                                invstd_out, scale_shift, (const unsigned short*)residual, relu, (unsigned short*)y, mask_out,
                                (int)rpb),
             hipLaunchKernelGGL(bn2d_fin_apply_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)x, partial, (int)rows,
                                (int)C, M, (double)M, running_mean, running_var, momentum, eps, gamma, beta, mean_out, invstd_out,
                                scale_shift, (const _Float16*)residual, relu, (_Float16*)y, mask_out, (int)rpb));
  CREID_LAUNCH_RET();
}

int creid_bn2d_apply_dual_mask(const void* x, const float* scale_shift, const void* x_res, const float* scale_shift_res, int relu,
                               int64_t M, int64_t C, int dtype, void* y, uint8_t* mask_out, void* stream) {
  CREID_CHECK_ARG(x && scale_shift && x_res && scale_shift_res && y && M > 0 && C > 0 && C % 8 == 0);
  if (mask_out && dtype == CREID_F32) return CREID_E_DTYPE;
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn2d_apply_kernel<float>, dim3(ew_blocks_k<bn2d_apply_kernel<float>>(M * C / 4, C / 4)), dim3(256), 0, s,
                                (const float*)x, scale_shift, (const float*)x_res, relu, M, (int)C, (float*)y,
                                (uint8_t*)nullptr, scale_shift_res),
             hipLaunchKernelGGL(bn2d_apply_kernel<unsigned short>, dim3(ew_blocks_k<bn2d_apply_kernel<unsigned short>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const unsigned short*)x, scale_shift, (const unsigned short*)x_res, relu, M,
                                (int)C, (unsigned short*)y, mask_out, scale_shift_res),
             hipLaunchKernelGGL(bn2d_apply_kernel<_Float16>, dim3(ew_blocks_k<bn2d_apply_kernel<_Float16>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const _Float16*)x, scale_shift, (const _Float16*)x_res, relu, M,
                                (int)C, (_Float16*)y, mask_out, scale_shift_res));
  CREID_LAUNCH_RET();
}

int creid_bn2d_apply(const void* x, const float* scale_shift, const void* residual, int relu, int64_t M, int64_t C,
                     int dtype, void* y, void* stream) {
  return creid_bn2d_apply_mask(x, scale_shift, residual, relu, M, C, dtype, y, nullptr, stream);
}

int64_t creid_bn2d_bwd_rows(int64_t M) { int64_t r = (M + 127) / 128; return r < 1 ? 1 : r; }

struct Reduce2 { const void* x2; const float* mean2; const float* invstd2; float* partial2; };

static int bn2d_bwd_impl(const void* x, const void* g, const void* act, const uint8_t* mask, PoolGrad pg, const float* mean,
                         const float* invstd, const float* gamma, int64_t M, int64_t C, int dtype, float* partial,
                         int partial_ready, float* sums, float* dgamma_accum, float* dbeta_accum, void* dx, void* gm_out,
                         void* stream, Reduce2 r2 = Reduce2{nullptr, nullptr, nullptr, nullptr}) {
  CREID_CHECK_ARG(x && (g || pg.dy) && mean && invstd && (partial || partial_ready == 2) && sums && dx && M > 0 && C > 0 && C % 8 == 0);
  if (mask && dtype == CREID_F32) return CREID_E_DTYPE;
  const int rows = (int)creid_bn2d_bwd_rows(M);
  hipStream_t s = as_stream(stream);
  // partial_ready: 0 = run the column pass here; 1 = `partial` already filled (fused dgrad epilogue); 2 = `sums` already
  // final (the finalize rode in a weight-gradient launch, creid_conv2d_wgrad_partials_bnfin): apply only
  if (!partial_ready)
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<float>, dim3((unsigned)((C / 4 + 31) / 32), rows), dim3(256), 0, s,
                                (const float*)x, (const float*)g, (const float*)act, mean, invstd, M, (int)C, 128, partial,
                                (const uint8_t*)nullptr, pg),
             hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<unsigned short>, dim3((unsigned)((C / 8 + 31) / 32), rows), dim3(256),
                                0, s, (const unsigned short*)x, (const unsigned short*)g, (const unsigned short*)act, mean,
                                invstd, M, (int)C, 128, partial, mask, pg),
             hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<_Float16>, dim3((unsigned)((C / 8 + 31) / 32), rows), dim3(256),
                                0, s, (const _Float16*)x, (const _Float16*)g, (const _Float16*)act, mean,
                                invstd, M, (int)C, 128, partial, mask, pg));
  if (partial_ready != 2 && !fin_dry(2)) {
    if (C <= 256 && rows >= 512)
      hipLaunchKernelGGL(bn2d_bwd_finalize_kernel<4>, dim3((unsigned)((C + 3) / 4)), dim3(1024), 0, s, partial, rows, (int)C,
                         (double)M, mean, invstd, gamma, sums, dgamma_accum, dbeta_accum);
    else
      hipLaunchKernelGGL(bn2d_bwd_finalize_kernel<16>, dim3((unsigned)((C + 15) / 16)), dim3(1024), 0, s, partial, rows, (int)C,
                         (double)M, mean, invstd, gamma, sums, dgamma_accum, dbeta_accum);
  }
  if (r2.x2) {
    // apply + the column sums of a second BatchNorm over the same masked gradient (downsample blocks), one pass
    if (act || gm_out || pg.dy || dtype == CREID_F32) return CREID_E_ARG;
    const dim3 grid2((unsigned)((C / 8 + 31) / 32), (unsigned)rows);
    if (dtype == CREID_BF16)
      hipLaunchKernelGGL(bn2d_bwd_apply_reduce2_kernel<unsigned short>, grid2, dim3(256), 0, s, (const unsigned short*)x,
                         (const unsigned short*)g, mask, sums, M, (int)C, 128, (unsigned short*)dx, (const unsigned short*)r2.x2,
                         r2.mean2, r2.invstd2, r2.partial2);
    else
      hipLaunchKernelGGL(bn2d_bwd_apply_reduce2_kernel<_Float16>, grid2, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)g,
                         mask, sums, M, (int)C, 128, (_Float16*)dx, (const _Float16*)r2.x2, r2.mean2, r2.invstd2, r2.partial2);
    CREID_LAUNCH_RET();
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn2d_bwd_apply_kernel<float>, dim3(ew_blocks_k<bn2d_bwd_apply_kernel<float>>(M * C / 4, C / 4)), dim3(256), 0, s,
                                (const float*)x, (const float*)g, (const float*)act, sums, M, (int)C,
                                (float*)dx, (float*)gm_out, (const uint8_t*)nullptr, pg),
             hipLaunchKernelGGL(bn2d_bwd_apply_kernel<unsigned short>, dim3(ew_blocks_k<bn2d_bwd_apply_kernel<unsigned short>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const unsigned short*)x, (const unsigned short*)g, (const unsigned short*)act, sums, M,
                                (int)C, (unsigned short*)dx, (unsigned short*)gm_out, mask, pg),
             hipLaunchKernelGGL(bn2d_bwd_apply_kernel<_Float16>, dim3(ew_blocks_k<bn2d_bwd_apply_kernel<_Float16>>(M * C / 8, C / 8)), dim3(256), 0, s,
                                (const _Float16*)x, (const _Float16*)g, (const _Float16*)act, sums, M,
                                (int)C, (_Float16*)dx, (_Float16*)gm_out, mask, pg));
  CREID_LAUNCH_RET();
}

int creid_bn2d_bwd_mask(const void* x, const void* g, const void* act, const uint8_t* mask, const float* mean,
                        const float* invstd, const float* gamma, int64_t M, int64_t C, int dtype, float* partial,
                        int partial_ready, float* sums, float* dgamma_accum, float* dbeta_accum, void* dx, void* gm_out,
                        void* stream) {
  CREID_CHECK_ARG(g);
  return bn2d_bwd_impl(x, g, act, mask, PoolGrad{nullptr, nullptr, 0, 0}, mean, invstd, gamma, M, C, dtype, partial, partial_ready,
                       sums, dgamma_accum, dbeta_accum, dx, gm_out, stream);
}

int creid_bn2d_bwd_mask_reduce2(const void* x, const void* g, const uint8_t* mask, const float* mean, const float* invstd,
                                const float* gamma, int64_t M, int64_t C, int dtype, float* partial, int partial_ready, float* sums,
                                float* dgamma_accum, float* dbeta_accum, void* dx, const void* x2, const float* mean2,
                                const float* invstd2, float* partial2, void* stream) {
  CREID_CHECK_ARG(g && x2 && mean2 && invstd2 && partial2);
  if (!creid_is16(dtype)) return CREID_E_DTYPE;
  return bn2d_bwd_impl(x, g, nullptr, mask, PoolGrad{nullptr, nullptr, 0, 0}, mean, invstd, gamma, M, C, dtype, partial, partial_ready,
                       sums, dgamma_accum, dbeta_accum, dx, nullptr, stream, Reduce2{x2, mean2, invstd2, partial2});
}

int creid_bn2d_bwd_pooled(const void* x, const void* dy_pooled, const uint8_t* idx, int64_t B, int64_t H, int64_t W,
                          const void* act, const float* mean, const float* invstd, const float* gamma, int64_t C, int dtype,
                          float* partial, float* sums, float* dgamma_accum, float* dbeta_accum, void* dx, void* stream) {
  CREID_CHECK_ARG(dy_pooled && idx && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
  return bn2d_bwd_impl(x, nullptr, act, nullptr, PoolGrad{dy_pooled, idx, (int)H, (int)W}, mean, invstd, gamma, B * H * W, C, dtype,
                       partial, 0, sums, dgamma_accum, dbeta_accum, dx, nullptr, stream);
}

int creid_bn2d_apply_maxpool3x3s2(const void* x, const float* scale_shift, int relu, int64_t B, int64_t H, int64_t W, int64_t C,
                                  int dtype, void* y, uint8_t* idx, void* stream) {
  CREID_CHECK_ARG(x && scale_shift && y && idx && B > 0 && H > 0 && W > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn_apply_maxpool_kernel<float>, dim3(ew_blocks(B * H * W * C / 16, 1)), dim3(256), 0, s,
                                (const float*)x, scale_shift, relu, (int)B, (int)H, (int)W, (int)C, (float*)y, idx),
             hipLaunchKernelGGL(bn_apply_maxpool_kernel<unsigned short>, dim3(ew_blocks(B * H * W * C / 32, 1)), dim3(256), 0, s,
                                (const unsigned short*)x, scale_shift, relu, (int)B, (int)H, (int)W, (int)C, (unsigned short*)y,
                                idx),
             hipLaunchKernelGGL(bn_apply_maxpool_kernel<_Float16>, dim3(ew_blocks(B * H * W * C / 32, 1)), dim3(256), 0, s,
                                (const _Float16*)x, scale_shift, relu, (int)B, (int)H, (int)W, (int)C, (_Float16*)y,
                                idx));
  CREID_LAUNCH_RET();
}

int creid_bn2d_bwd(const void* x, const void* g, const void* act, const float* mean, const float* invstd,
                   const float* gamma, int64_t M, int64_t C, int dtype, float* partial, int partial_ready, float* sums,
                   float* dgamma_accum, float* dbeta_accum, void* dx, void* gm_out, void* stream) {
  return creid_bn2d_bwd_mask(x, g, act, nullptr, mean, invstd, gamma, M, C, dtype, partial, partial_ready, sums, dgamma_accum,
                             dbeta_accum, dx, gm_out, stream);
}

int creid_maxpool3x3s2_fwd(const void* x, int64_t B, int64_t H, int64_t W, int64_t C, int dtype, void* y, uint8_t* idx,
                           void* stream) {
  CREID_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0);   // idx may be NULL (inference)
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(ew_blocks(B * H * W * C / 16, 1)), dim3(256), 0, s,
                                (const float*)x, (int)B, (int)H, (int)W, (int)C, (float*)y, idx),
             hipLaunchKernelGGL(maxpool_fwd_kernel<unsigned short>, dim3(ew_blocks(B * H * W * C / 32, 1)), dim3(256), 0, s,
                                (const unsigned short*)x, (int)B, (int)H, (int)W, (int)C, (unsigned short*)y, idx),
             hipLaunchKernelGGL(maxpool_fwd_kernel<_Float16>, dim3(ew_blocks(B * H * W * C / 32, 1)), dim3(256), 0, s,
                                (const _Float16*)x, (int)B, (int)H, (int)W, (int)C, (_Float16*)y, idx));
  CREID_LAUNCH_RET();
}

int creid_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C, int dtype,
                           void* dx, void* stream) {
  CREID_CHECK_ARG(dy && dx && idx && B > 0 && H > 0 && W > 0 && C % 8 == 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3((unsigned)(B * H)), dim3(256), 0, s,
                                (const float*)dy, idx, (int)B, (int)H, (int)W, (int)C, (float*)dx),
             hipLaunchKernelGGL(maxpool_bwd_kernel<unsigned short>, dim3((unsigned)(B * H)), dim3(256), 0, s,
                                (const unsigned short*)dy, idx, (int)B, (int)H, (int)W, (int)C, (unsigned short*)dx),
             hipLaunchKernelGGL(maxpool_bwd_kernel<_Float16>, dim3((unsigned)(B * H)), dim3(256), 0, s,
                                (const _Float16*)dy, idx, (int)B, (int)H, (int)W, (int)C, (_Float16*)dx));
  CREID_LAUNCH_RET();
}

static int gap_fwd_impl(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* feat, int64_t* counter, void* stream) {
  CREID_CHECK_ARG(x && feat && B > 0 && HW > 0 && C % 8 == 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3((unsigned)((C / 4 + 31) / 32), (unsigned)B), dim3(256), 0, s,
                                (const float*)x, (int)HW, (int)C, feat, counter),
             hipLaunchKernelGGL(gap_fwd_kernel<unsigned short>, dim3((unsigned)((C / 8 + 31) / 32), (unsigned)B),
                                dim3(256), 0, s, (const unsigned short*)x, (int)HW, (int)C, feat, counter),
             hipLaunchKernelGGL(gap_fwd_kernel<_Float16>, dim3((unsigned)((C / 8 + 31) / 32), (unsigned)B),
                                dim3(256), 0, s, (const _Float16*)x, (int)HW, (int)C, feat, counter));
  CREID_LAUNCH_RET();
}

int creid_gap_fwd(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* feat, void* stream) {
  return gap_fwd_impl(x, B, HW, C, dtype, feat, nullptr, stream);
}

int creid_gap_fwd_count(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* feat, int64_t* forward_counter,
                        void* stream) {
  CREID_CHECK_ARG(forward_counter);
  return gap_fwd_impl(x, B, HW, C, dtype, feat, forward_counter, stream);
}

int creid_gap_bwd(const float* dfeat, int64_t B, int64_t HW, int64_t C, int dtype, void* dx, void* stream) {
  CREID_CHECK_ARG(dfeat && dx && B > 0 && HW > 0 && C % 8 == 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(ew_blocks(B * HW * C / 4, 1)), dim3(256), 0, s, dfeat, (int)HW,
                                (int)C, B * HW * C / 4, (float*)dx),
             hipLaunchKernelGGL(gap_bwd_kernel<unsigned short>, dim3(ew_blocks(B * HW * C / 8, 1)), dim3(256), 0, s, dfeat,
                                (int)HW, (int)C, B * HW * C / 8, (unsigned short*)dx),
             hipLaunchKernelGGL(gap_bwd_kernel<_Float16>, dim3(ew_blocks(B * HW * C / 8, 1)), dim3(256), 0, s, dfeat,
                                (int)HW, (int)C, B * HW * C / 8, (_Float16*)dx));
  CREID_LAUNCH_RET();
}

int creid_image_to_nhwc4_pad(const float* x_nchw, int64_t B, int64_t H, int64_t W, int dtype, void* xpad, void* stream) {
  CREID_CHECK_ARG(x_nchw && xpad && B > 0 && H > 0 && W > 0);
  hipStream_t s = as_stream(stream);
  const int64_t total = B * (H + 8) * (W + 6);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(image_pad_kernel<float>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, x_nchw, (int)B, (int)H,
                                (int)W, (float*)xpad),
             hipLaunchKernelGGL(image_pad_kernel<unsigned short>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, x_nchw, (int)B,
                                (int)H, (int)W, (unsigned short*)xpad),
             hipLaunchKernelGGL(image_pad_kernel<_Float16>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, x_nchw, (int)B,
                                (int)H, (int)W, (_Float16*)xpad));
  CREID_LAUNCH_RET();
}

int creid_weight_prep(const float* w_oihw, int64_t O, int64_t I, int64_t kh, int64_t kw, int dtype, void* w_krsc,
                      void* w_crsk, void* stream) {
  CREID_CHECK_ARG(w_oihw && w_krsc && O > 0 && I > 0 && kh > 0 && kw > 0);
  hipStream_t s = as_stream(stream);
  const int64_t total = O * I * kh * kw;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, w_oihw, (int)O, (int)I,
                                (int)kh, (int)kw, (float*)w_krsc, (float*)w_crsk),
             hipLaunchKernelGGL(weight_prep_kernel<unsigned short>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, w_oihw,
                                (int)O, (int)I, (int)kh, (int)kw, (unsigned short*)w_krsc, (unsigned short*)w_crsk),
             hipLaunchKernelGGL(weight_prep_kernel<_Float16>, dim3(ew_blocks(total, 1)), dim3(256), 0, s, w_oihw,
                                (int)O, (int)I, (int)kh, (int)kw, (_Float16*)w_krsc, (_Float16*)w_crsk));
  CREID_LAUNCH_RET();
}

int64_t creid_weight_prep_entry_bytes(void) { return (int64_t)sizeof(WPrepEntry); }

int creid_weight_prep_multi(const void* table_dev, const int32_t* tile_start_dev, int64_t n_entries, int64_t total_tiles,
                            int dtype, void* stream) {
  CREID_CHECK_ARG(table_dev && tile_start_dev && n_entries > 0 && total_tiles > 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(weight_prep_multi_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, s,
                                (const WPrepEntry*)table_dev, (int)n_entries, tile_start_dev),
             hipLaunchKernelGGL(weight_prep_multi_kernel<unsigned short>, dim3((unsigned)total_tiles), dim3(256), 0, s,
                                (const WPrepEntry*)table_dev, (int)n_entries, tile_start_dev),
             hipLaunchKernelGGL(weight_prep_multi_kernel<_Float16>, dim3((unsigned)total_tiles), dim3(256), 0, s,
                                (const WPrepEntry*)table_dev, (int)n_entries, tile_start_dev));
  CREID_LAUNCH_RET();
}

int64_t creid_bn2d_fold_entry_bytes(void) { return (int64_t)sizeof(BnFoldEntry); }

int creid_bn2d_fold_multi(const void* table_dev, int64_t n_entries, void* stream) {
  CREID_CHECK_ARG(table_dev && n_entries > 0 && n_entries < (1 << 20));
  hipLaunchKernelGGL(bn_fold_multi_kernel, dim3((unsigned)n_entries, 8), dim3(256), 0, as_stream(stream),
                     (const BnFoldEntry*)table_dev);
  CREID_LAUNCH_RET();
}

int creid_stem_weight_prep(const float* w_oihw, int dtype, void* w_stem, void* stream) {
  CREID_CHECK_ARG(w_oihw && w_stem);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(stem_weight_prep_kernel<float>, dim3(64), dim3(256), 0, s, w_oihw, (float*)w_stem),
             hipLaunchKernelGGL(stem_weight_prep_kernel<unsigned short>, dim3(64), dim3(256), 0, s, w_oihw,
                                (unsigned short*)w_stem),
             hipLaunchKernelGGL(stem_weight_prep_kernel<_Float16>, dim3(64), dim3(256), 0, s, w_oihw,
                                (_Float16*)w_stem));
  CREID_LAUNCH_RET();
}

int creid_nhwc_to_nchw_f32(const void* x, int64_t B, int64_t HW, int64_t C, int dtype, float* y, void* stream) {
  CREID_CHECK_ARG(x && y && B > 0 && HW > 0 && C > 0);
  hipStream_t s = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_blocks(B * HW * C, 1)), dim3(256), 0, s, (const float*)x,
                                (int)B, (int)HW, (int)C, y),
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<unsigned short>, dim3(ew_blocks(B * HW * C, 1)), dim3(256), 0, s,
                                (const unsigned short*)x, (int)B, (int)HW, (int)C, y),
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, dim3(ew_blocks(B * HW * C, 1)), dim3(256), 0, s,
                                (const _Float16*)x, (int)B, (int)HW, (int)C, y));
  CREID_LAUNCH_RET();
}

}  // extern "C"

// ======================================================================================
// IBN (modelling/backbones/resnet_ibn_a.py:18-32): channels [0, c_in) use InstanceNorm2d(affine) --
// per-(image, channel) statistics over H*W, fp32, no running stats -- channels [c_in, C) use
// BatchNorm2d.  All kernels index statistics / coefficients per (image, channel); the BN half simply
// holds the same value for every image.  Statistics come from a per-image column pass
// (partial[(n*rpi + rb)][2][C], rpi row-blocks per image).
// ======================================================================================
template <typename T>
__global__ __launch_bounds__(256) void ibn_col_stats_kernel(const T* __restrict__ x, int HW, int C, int rows_per_block,
                                                            int rpi, float* __restrict__ partial) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256 * V];
  const int cpr = C / V, cw = cpr < 32 ? cpr : 32, nrl = 256 / cw;
  const int cch = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c0 = (blockIdx.x * cw + cch) * V;
  const int n = blockIdx.z, rb = blockIdx.y;
  const int r0 = rb * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  const T* xi = x + (int64_t)n * HW * C;
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (c0 < C)
    for (int r = r0 + rl; r < r1; r += nrl) {
      float v[V];
      Vec16<T>::load(xi + (int64_t)r * C + c0, v);
#pragma unroll
      for (int k = 0; k < V; ++k) { s1[k] += v[k]; s2[k] = fmaf(v[k], v[k], s2[k]); }
    }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][(rl * cw + cch) * V + k] = s1[k]; red[1][(rl * cw + cch) * V + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cw * V; i += 256) {
    const int which = i / (cw * V), cl = i - which * cw * V;
    float a = 0.f;
    for (int q = 0; q < nrl; ++q) a += red[which][q * cw * V + cl];
    const int c = blockIdx.x * cw * V + cl;
    if (c < C) partial[(((int64_t)n * rpi + rb) * 2 + which) * C + c] = a;
  }
}

// Workgroup (blockIdx.x, blockIdx.y) = 16 channels x 16 images (images 16 y .. 16 y + 15); 1024 threads = 16 channels x 64
// lanes.  InstanceNorm channels: lane (image i = rg & 15, quarter q = rg >> 4) sums row blocks q, q + 4, .. of image i, the four
// quarters meet in LDS in a fixed order (round 4: one lane per image summed all rpi row blocks one dependent load after the
// other, four images in a row at B = 256 -- 42 us per launch on 4-16 workgroups, 13 launches per IBN-a forward).  BatchNorm
// channels: the 64 lanes share the B * rpi row blocks, meet in LDS (fp64, fixed order) -- every image group repeats that
// reduction (a few hundred KB from L2) and writes the same statistics for its own images; group 0 updates the running statistics.
// Outputs: mean/invstd [B][C], scale_shift [B][2][C].
__global__ __launch_bounds__(1024) void ibn_finalize_kernel(const float* __restrict__ partial, int B, int rpi, int HW,
                                                            int C, int c_in, const float* __restrict__ in_w,
                                                            const float* __restrict__ in_b,
                                                            const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, int training, float momentum,
                                                            float eps, float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out,
                                                            float* __restrict__ scale_shift) {
  __shared__ double red[64][2][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int n0 = blockIdx.y * 16;
  const bool live = c < C, is_in = live && c < c_in, is_bn = live && c >= c_in;
  double s1 = 0.0, s2 = 0.0;
  if (is_bn && training) {
    int r = rg;
    for (; r + 192 < B * rpi; r += 256) {               // 8 independent loads per trip (the loop is pure load latency)
      const float a0 = partial[((int64_t)r * 2) * C + c], b0 = partial[((int64_t)r * 2 + 1) * C + c];
      const float a1 = partial[((int64_t)(r + 64) * 2) * C + c], b1 = partial[((int64_t)(r + 64) * 2 + 1) * C + c];
      const float a2 = partial[((int64_t)(r + 128) * 2) * C + c], b2 = partial[((int64_t)(r + 128) * 2 + 1) * C + c];
      const float a3 = partial[((int64_t)(r + 192) * 2) * C + c], b3 = partial[((int64_t)(r + 192) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < B * rpi; r += 64) {
      s1 += (double)partial[((int64_t)r * 2) * C + c];
      s2 += (double)partial[((int64_t)r * 2 + 1) * C + c];
    }
  } else if (is_in) {
    const int n = n0 + (rg & 15);
    if (n < B)
      for (int r = rg >> 4; r < rpi; r += 4) {
        s1 += (double)partial[(((int64_t)n * rpi + r) * 2) * C + c];
        s2 += (double)partial[(((int64_t)n * rpi + r) * 2 + 1) * C + c];
      }
  }
  red[rg][0][cl] = s1; red[rg][1][cl] = s2;
  __syncthreads();
  if (is_in) {
    const int il = rg & 15, n = n0 + il;
    if ((rg >> 4) == 0 && n < B) {
      const double t1 = ((red[il][0][cl] + red[16 + il][0][cl]) + red[32 + il][0][cl]) + red[48 + il][0][cl];
      const double t2 = ((red[il][1][cl] + red[16 + il][1][cl]) + red[32 + il][1][cl]) + red[48 + il][1][cl];
      const float g = in_w[c], b = in_b[c];
      const double mean = t1 / (double)HW;
      double var = t2 / (double)HW - mean * mean;
      if (var < 0.0) var = 0.0;
      const float mu = (float)mean, is = (float)(1.0 / sqrt(var + (double)eps));
      mean_out[(int64_t)n * C + c] = mu; invstd_out[(int64_t)n * C + c] = is;
      const float sc = is * g;
      scale_shift[((int64_t)n * 2) * C + c] = sc;
      scale_shift[((int64_t)n * 2 + 1) * C + c] = b - mu * sc;
    }
  } else if (is_bn) {
    const float g = bn_w[c - c_in], b = bn_b[c - c_in];
    float mu, is;
    if (training) {
      s1 = 0.0; s2 = 0.0;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
      const double cnt = (double)B * HW;
      const double mean = s1 / cnt;
      double var = s2 / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      mu = (float)mean; is = (float)(1.0 / sqrt(var + (double)eps));
      if (rg == 0 && blockIdx.y == 0) {
        rmean[c - c_in] = (1.f - momentum) * rmean[c - c_in] + momentum * mu;
        rvar[c - c_in] = (1.f - momentum) * rvar[c - c_in] + momentum * (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
      }
    } else {
      mu = rmean[c - c_in]; is = 1.0f / sqrtf(rvar[c - c_in] + eps);
    }
    const float sc = is * g, sh = b - mu * sc;
    const int n = n0 + rg;
    if (rg < 16 && n < B) {
      mean_out[(int64_t)n * C + c] = mu; invstd_out[(int64_t)n * C + c] = is;
      scale_shift[((int64_t)n * 2) * C + c] = sc;
      scale_shift[((int64_t)n * 2 + 1) * C + c] = sh;
    }
  }
}

// grid (blocks, B): one image per blockIdx.y; a thread keeps ONE channel vector for its whole loop (stride a
// multiple of C/V), so its per-(image, channel) coefficients are loaded once and no index division remains
template <typename T>
__global__ __launch_bounds__(256) void ibn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift,
                                                        int relu, int64_t M, int HW, int C, T* __restrict__ y,
                                                        uint8_t* __restrict__ mask_out) {
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V, n = blockIdx.y;
  const int total = HW * cpr, nthreads = gridDim.x * 256;       // 256 % cpr == 0 (host-checked)
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int c0 = (gtid % cpr) * V;
  float sc[V], sh[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    sc[k] = scale_shift[((int64_t)n * 2) * C + c0 + k];
    sh[k] = scale_shift[((int64_t)n * 2 + 1) * C + c0 + k];
  }
  const int64_t base = (int64_t)n * total;
  for (int i = gtid; i < total; i += nthreads) {
    float v[V];
    Vec16<T>::load(x + (base + i) * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      v[k] = fmaf(v[k], sc[k], sh[k]);
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    Vec16<T>::store(y + (base + i) * V, v);
    if (mask_out) {                                  // ReLU mask as bits, as bn2d_apply_kernel
      unsigned m = 0;
#pragma unroll
      for (int k = 0; k < V; ++k) m |= (v[k] > 0.f ? 1u : 0u) << k;
      mask_out[base + i] = (uint8_t)m;
    }
  }
}

// per-image partial sums of dy and dy*xhat (dy = g * [act > 0])
template <typename T>
__global__ __launch_bounds__(256) void ibn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                             const T* __restrict__ act,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int HW, int C,
                                                             int rows_per_block, int rpi,
                                                             float* __restrict__ partial, const uint8_t* __restrict__ mask) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][256 * V];
  const int cpr = C / V, cw = cpr < 32 ? cpr : 32, nrl = 256 / cw;
  const int cch = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c0 = (blockIdx.x * cw + cch) * V;
  const int n = blockIdx.z, rb = blockIdx.y;
  const int r0 = rb * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  const int64_t base = (int64_t)n * HW * C;
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (c0 < C)
    for (int r = r0 + rl; r < r1; r += nrl) {
      float xv[V], gv[V];
      Vec16<T>::load(x + base + (int64_t)r * C + c0, xv);
      Vec16<T>::load(g + base + (int64_t)r * C + c0, gv);
      if (mask) {
        const unsigned m = mask[(base + (int64_t)r * C + c0) / V];
#pragma unroll
        for (int k = 0; k < V; ++k) gv[k] = ((m >> k) & 1u) ? gv[k] : 0.f;
      } else if (act) {
        float av[V];
        Vec16<T>::load(act + base + (int64_t)r * C + c0, av);
#pragma unroll
        for (int k = 0; k < V; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        s1[k] += gv[k];
        s2[k] = fmaf(gv[k], (xv[k] - mean[(int64_t)n * C + c0 + k]) * invstd[(int64_t)n * C + c0 + k], s2[k]);
      }
    }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][(rl * cw + cch) * V + k] = s1[k]; red[1][(rl * cw + cch) * V + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cw * V; i += 256) {
    const int which = i / (cw * V), cl = i - which * cw * V;
    float a = 0.f;
    for (int q = 0; q < nrl; ++q) a += red[which][q * cw * V + cl];
    const int c = blockIdx.x * cw * V + cl;
    if (c < C) partial[(((int64_t)n * rpi + rb) * 2 + which) * C + c] = a;
  }
}

// coef [B][3][C]; d(in_w, in_b) += sum over images of per-image sums (ibn_in_grad_kernel); d(bn_w, bn_b) += batch
// sums.  Same 16-channel x 64-lane layout as ibn_finalize_kernel.
__global__ __launch_bounds__(1024) void ibn_bwd_finalize_kernel(const float* __restrict__ partial, int B, int rpi, int HW,
                                                                int C, int c_in, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ in_w,
                                                                const float* __restrict__ bn_w,
                                                                float* __restrict__ coef, float* __restrict__ per_img,
                                                                float* __restrict__ d_bn_w, float* __restrict__ d_bn_b) {
  __shared__ double red[64][2][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int n0 = blockIdx.y * 16;                       // 16 images per workgroup; lane (image rg & 15, quarter rg >> 4)
  const bool live = c < C, is_in = live && c < c_in, is_bn = live && c >= c_in;
  double s1 = 0.0, s2 = 0.0;
  if (is_bn) {
    int r = rg;
    for (; r + 192 < B * rpi; r += 256) {               // 8 independent loads per trip (the loop is pure load latency)
      const float a0 = partial[((int64_t)r * 2) * C + c], b0 = partial[((int64_t)r * 2 + 1) * C + c];
      const float a1 = partial[((int64_t)(r + 64) * 2) * C + c], b1 = partial[((int64_t)(r + 64) * 2 + 1) * C + c];
      const float a2 = partial[((int64_t)(r + 128) * 2) * C + c], b2 = partial[((int64_t)(r + 128) * 2 + 1) * C + c];
      const float a3 = partial[((int64_t)(r + 192) * 2) * C + c], b3 = partial[((int64_t)(r + 192) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < B * rpi; r += 64) {
      s1 += (double)partial[((int64_t)r * 2) * C + c];
      s2 += (double)partial[((int64_t)r * 2 + 1) * C + c];
    }
  } else if (is_in) {
    const int n = n0 + (rg & 15);
    if (n < B)
      for (int r = rg >> 4; r < rpi; r += 4) {
        s1 += (double)partial[(((int64_t)n * rpi + r) * 2) * C + c];
        s2 += (double)partial[(((int64_t)n * rpi + r) * 2 + 1) * C + c];
      }
  }
  red[rg][0][cl] = s1; red[rg][1][cl] = s2;
  __syncthreads();
  if (is_in) {
    const int il = rg & 15, n = n0 + il;
    if ((rg >> 4) == 0 && n < B) {
      const double t1 = ((red[il][0][cl] + red[16 + il][0][cl]) + red[32 + il][0][cl]) + red[48 + il][0][cl];
      const double t2 = ((red[il][1][cl] + red[16 + il][1][cl]) + red[32 + il][1][cl]) + red[48 + il][1][cl];
      const float g = in_w[c];
      const float invM = 1.0f / (float)HW;
      per_img[((int64_t)n * 2) * c_in + c] = (float)t1;         // reduced over images by ibn_in_grad_kernel
      per_img[((int64_t)n * 2 + 1) * c_in + c] = (float)t2;
      const float mu = mean[(int64_t)n * C + c], is = invstd[(int64_t)n * C + c], k1 = g * is;
      const float a1 = (float)t1 * invM, a2 = (float)t2 * invM;
      coef[((int64_t)n * 3) * C + c] = k1;
      coef[((int64_t)n * 3 + 1) * C + c] = -k1 * is * a2;
      coef[((int64_t)n * 3 + 2) * C + c] = -k1 * a1 + k1 * is * a2 * mu;
    }
  } else if (is_bn) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
    if (rg == 0 && blockIdx.y == 0) { if (d_bn_b) d_bn_b[c - c_in] += (float)s1; if (d_bn_w) d_bn_w[c - c_in] += (float)s2; }
    const float invM = (float)(1.0 / ((double)B * HW));
    const float mu = mean[c], is = invstd[c], k1 = bn_w[c - c_in] * is;      // same for every image: read image 0
    const float a1 = (float)s1 * invM, a2 = (float)s2 * invM;
    const float cA = k1, cB = -k1 * is * a2, cC = -k1 * a1 + k1 * is * a2 * mu;
    const int n = n0 + rg;
    if (rg < 16 && n < B) {
      coef[((int64_t)n * 3) * C + c] = cA;
      coef[((int64_t)n * 3 + 1) * C + c] = cB;
      coef[((int64_t)n * 3 + 2) * C + c] = cC;
    }
  }
}

// d(in_w, in_b) += sum over images of the per-image sums: 16 channels x 64 image lanes per workgroup, fixed-order LDS tree
// (round 4: one thread per channel walked all B images, 16 us of load latency per launch)
__global__ __launch_bounds__(1024) void ibn_in_grad_kernel(const float* __restrict__ per_img, int B, int c_in,
                                                           float* __restrict__ d_in_w, float* __restrict__ d_in_b) {
  __shared__ float red[64][2][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < c_in)
    for (int n = rg; n < B; n += 64) { s1 += per_img[((int64_t)n * 2) * c_in + c]; s2 += per_img[((int64_t)n * 2 + 1) * c_in + c]; }
  red[rg][0][cl] = s1; red[rg][1][cl] = s2;
  __syncthreads();
  if (rg == 0 && c < c_in) {
    s1 = 0.f; s2 = 0.f;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
    if (d_in_b) d_in_b[c] += s1;
    if (d_in_w) d_in_w[c] += s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ibn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                            const T* __restrict__ act, const float* __restrict__ coef,
                                                            int64_t M, int HW, int C, T* __restrict__ dx,
                                                            const uint8_t* __restrict__ mask) {
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V, n = blockIdx.y;
  const int total = HW * cpr, nthreads = gridDim.x * 256;       // 256 % cpr == 0 (host-checked)
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int c0 = (gtid % cpr) * V;
  float ca[V], cb[V], cc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    ca[k] = coef[((int64_t)n * 3) * C + c0 + k];
    cb[k] = coef[((int64_t)n * 3 + 1) * C + c0 + k];
    cc[k] = coef[((int64_t)n * 3 + 2) * C + c0 + k];
  }
  const int64_t base = (int64_t)n * total;
  for (int i = gtid; i < total; i += nthreads) {
    float xv[V], gv[V], o[V];
    Vec16<T>::load(x + (base + i) * V, xv);
    Vec16<T>::load(g + (base + i) * V, gv);
    if (mask) {
      const unsigned m = mask[base + i];
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = ((m >> k) & 1u) ? gv[k] : 0.f;
    } else if (act) {
      float av[V];
      Vec16<T>::load(act + (base + i) * V, av);
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = fmaf(ca[k], gv[k], fmaf(cb[k], xv[k], cc[k]));
    Vec16<T>::store(dx + (base + i) * V, o);
  }
}

extern "C" {

// workgroups per image for the IBN apply passes: ~4 vectors per thread, at most 64 per image
// blocks per image of the per-image grid-stride IBN kernels (grid = (blocks, B)).  Round 5: a whole number of rounds -- with B images
// the grid is trimmed to the largest multiple of the ~2048 workgroups the chip holds that does not exceed what was asked for
// (configs[3] training, B = 56: 50 blocks per image = 2800 workgroups = 1.37 rounds -> 36 = 2016, one round; profiles/r05_ew_grid.md)
static unsigned ibn_img_blocks(int64_t vecs_per_image, int64_t B = 1) {
  int64_t b = (vecs_per_image + 1023) / 1024;
  b = b < 1 ? 1 : (b > 64 ? 64 : b);
  const int64_t resident = 2048, total = b * B;
  if (total > resident) {
    const int64_t trimmed = (total / resident) * resident / B;
    if (trimmed >= 1 && trimmed < b) b = trimmed;
  }
  return (unsigned)b;
}

int64_t creid_ibn_rows_per_image(int64_t HW) { int64_t r = (HW + 127) / 128; return r < 1 ? 1 : r; }

int creid_ibn_fwd_mask(const void* x, int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w, const float* in_b,
                  const float* bn_w, const float* bn_b, float* running_mean, float* running_var, int training,
                  float momentum, float eps, int relu, int dtype, float* partial, int partial_ready, float* mean_out,
                  float* invstd_out, float* scale_shift, void* y, uint8_t* mask_out, void* stream) {
  CREID_CHECK_ARG(x && y && in_w && in_b && bn_w && bn_b && running_mean && running_var && partial && mean_out &&
                  invstd_out && scale_shift && B > 0 && HW > 0 && C % 8 == 0 && c_in > 0 && c_in < C);
  if (256 % (C / 8) != 0 || 256 % (C / 4) != 0) return CREID_E_SHAPE;      // a thread keeps one channel vector
  if (mask_out && dtype == CREID_F32) return CREID_E_DTYPE;
  if (partial_ready && HW % 128 != 0) return CREID_E_SHAPE;                 // conv tiles must not straddle images
  const int rpi = (int)creid_ibn_rows_per_image(HW);
  hipStream_t s = as_stream(stream);
  if (!partial_ready)
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(ibn_col_stats_kernel<float>, dim3((unsigned)((C / 4 + 31) / 32), rpi, (unsigned)B), dim3(256),
                                0, s, (const float*)x, (int)HW, (int)C, 128, rpi, partial),
             hipLaunchKernelGGL(ibn_col_stats_kernel<unsigned short>, dim3((unsigned)((C / 8 + 31) / 32), rpi, (unsigned)B),
                                dim3(256), 0, s, (const unsigned short*)x, (int)HW, (int)C, 128, rpi, partial),
             hipLaunchKernelGGL(ibn_col_stats_kernel<_Float16>, dim3((unsigned)((C / 8 + 31) / 32), rpi, (unsigned)B),
                                dim3(256), 0, s, (const _Float16*)x, (int)HW, (int)C, 128, rpi, partial));
  hipLaunchKernelGGL(ibn_finalize_kernel, dim3((unsigned)((C + 15) / 16), (unsigned)((B + 15) / 16)), dim3(1024), 0, s, partial, (int)B,
                     rpi, (int)HW, (int)C, (int)c_in, in_w, in_b, bn_w, bn_b, running_mean, running_var, training, momentum,
                     eps, mean_out, invstd_out, scale_shift);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(ibn_apply_kernel<float>, dim3(ibn_img_blocks(HW * C / 4, B), (unsigned)B), dim3(256), 0, s, (const float*)x,
                                scale_shift, relu, B * HW, (int)HW, (int)C, (float*)y, (uint8_t*)nullptr),
             hipLaunchKernelGGL(ibn_apply_kernel<unsigned short>, dim3(ibn_img_blocks(HW * C / 8, B), (unsigned)B), dim3(256), 0, s,
                                (const unsigned short*)x, scale_shift, relu, B * HW, (int)HW, (int)C, (unsigned short*)y, mask_out),
             hipLaunchKernelGGL(ibn_apply_kernel<_Float16>, dim3(ibn_img_blocks(HW * C / 8, B), (unsigned)B), dim3(256), 0, s,
                                (const _Float16*)x, scale_shift, relu, B * HW, (int)HW, (int)C, (_Float16*)y, mask_out));
  CREID_LAUNCH_RET();
}

int creid_ibn_fwd(const void* x, int64_t B, int64_t HW, int64_t C, int64_t c_in, const float* in_w, const float* in_b,
                  const float* bn_w, const float* bn_b, float* running_mean, float* running_var, int training,
                  float momentum, float eps, int relu, int dtype, float* partial, int partial_ready, float* mean_out,
                  float* invstd_out, float* scale_shift, void* y, void* stream) {
  return creid_ibn_fwd_mask(x, B, HW, C, c_in, in_w, in_b, bn_w, bn_b, running_mean, running_var, training, momentum, eps, relu, dtype, partial, partial_ready, mean_out, invstd_out, scale_shift, y, nullptr, stream);
}

int creid_ibn_bwd_mask(const void* x, const void* g, const void* act, const uint8_t* mask, const float* mean, const float* invstd, int64_t B,
                  int64_t HW, int64_t C, int64_t c_in, const float* in_w, const float* bn_w, int dtype, float* partial,
                  int partial_ready, float* coef, float* per_img, float* d_in_w, float* d_in_b, float* d_bn_w,
                  float* d_bn_b, void* dx, void* stream) {
  CREID_CHECK_ARG(x && g && mean && invstd && in_w && bn_w && partial && coef && per_img && dx && B > 0 && HW > 0 &&
                  C % 8 == 0 && c_in > 0 && c_in < C);
  if (256 % (C / 8) != 0 || 256 % (C / 4) != 0) return CREID_E_SHAPE;
  if (mask && dtype == CREID_F32) return CREID_E_DTYPE;
  if (partial_ready && HW % 128 != 0) return CREID_E_SHAPE;
  const int rpi = (int)creid_ibn_rows_per_image(HW);
  hipStream_t s = as_stream(stream);
  if (!partial_ready)
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(ibn_bwd_reduce_kernel<float>, dim3((unsigned)((C / 4 + 31) / 32), rpi, (unsigned)B), dim3(256),
                                0, s, (const float*)x, (const float*)g, (const float*)act, mean, invstd, (int)HW, (int)C, 128,
                                rpi, partial, (const uint8_t*)nullptr),
             hipLaunchKernelGGL(ibn_bwd_reduce_kernel<unsigned short>, dim3((unsigned)((C / 8 + 31) / 32), rpi, (unsigned)B),
                                dim3(256), 0, s, (const unsigned short*)x, (const unsigned short*)g,
                                (const unsigned short*)act, mean, invstd, (int)HW, (int)C, 128, rpi, partial, mask),
             hipLaunchKernelGGL(ibn_bwd_reduce_kernel<_Float16>, dim3((unsigned)((C / 8 + 31) / 32), rpi, (unsigned)B),
                                dim3(256), 0, s, (const _Float16*)x, (const _Float16*)g,
                                (const _Float16*)act, mean, invstd, (int)HW, (int)C, 128, rpi, partial, mask));
  hipLaunchKernelGGL(ibn_bwd_finalize_kernel, dim3((unsigned)((C + 15) / 16), (unsigned)((B + 15) / 16)), dim3(1024), 0, s, partial,
                     (int)B, rpi, (int)HW, (int)C, (int)c_in, mean, invstd, in_w, bn_w, coef, per_img, d_bn_w, d_bn_b);
  hipLaunchKernelGGL(ibn_in_grad_kernel, dim3((unsigned)((c_in + 15) / 16)), dim3(1024), 0, s, per_img, (int)B, (int)c_in,
                     d_in_w, d_in_b);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(ibn_bwd_apply_kernel<float>, dim3(ibn_img_blocks(HW * C / 4, B), (unsigned)B), dim3(256), 0, s,
                                (const float*)x, (const float*)g, (const float*)act, coef, B * HW, (int)HW, (int)C, (float*)dx, (const uint8_t*)nullptr),
             hipLaunchKernelGGL(ibn_bwd_apply_kernel<unsigned short>, dim3(ibn_img_blocks(HW * C / 8, B), (unsigned)B), dim3(256), 0, s,
                                (const unsigned short*)x, (const unsigned short*)g, (const unsigned short*)act, coef, B * HW,
                                (int)HW, (int)C, (unsigned short*)dx, mask),
             hipLaunchKernelGGL(ibn_bwd_apply_kernel<_Float16>, dim3(ibn_img_blocks(HW * C / 8, B), (unsigned)B), dim3(256), 0, s,
                                (const _Float16*)x, (const _Float16*)g, (const _Float16*)act, coef, B * HW,
                                (int)HW, (int)C, (_Float16*)dx, mask));
  CREID_LAUNCH_RET();
}

int creid_ibn_bwd(const void* x, const void* g, const void* act, const float* mean, const float* invstd, int64_t B,
                  int64_t HW, int64_t C, int64_t c_in, const float* in_w, const float* bn_w, int dtype, float* partial,
                  int partial_ready, float* coef, float* per_img, float* d_in_w, float* d_in_b, float* d_bn_w,
                  float* d_bn_b, void* dx, void* stream) {
  return creid_ibn_bwd_mask(x, g, act, nullptr, mean, invstd, B, HW, C, c_in, in_w, bn_w, dtype, partial, partial_ready, coef, per_img, d_in_w, d_in_b, d_bn_w, d_bn_b, dx, stream);
}

}  // extern "C"
