// Eval-mode stem in ONE launch (round 4): 7 x 7 stride-2 convolution 3 -> 64 + folded BatchNorm (+ ReLU for the IBN-a variant)
// + 3 x 3 stride-2 max-pool (modelling/backbones/resnet.py:95-98,123-126: conv1 -> bn1 -> [relu] -> maxpool).  As two launches
// (the tile kernel igemm_bf16_dma_kernel<64, 2> + maxpool_fwd_kernel) the stem costs 80 + 52 us of the 2.15 ms embedding forward at
// B = 128: the tile kernel runs a K = 256, N = 64 GEMM at 250 TF/s (the padded weight matrix re-fetched by every 128-row tile, the
// 7 x 7 patches re-gathered 8 px at a time through the 64 B/clk operand path) and the full-resolution tensor [B, H/2, W/2, 64] --
// 134 MB -- is written only to be read once by the pool.  Here
//   * the INPUT of a step (RS convolution rows of one image = 2 RS + 5 rows of the zero-padded NHWC4 image, one CONTIGUOUS block
//     of xpad) is staged in LDS once, prefetched into registers a step ahead; all 14 multiply steps of an output pixel read
//     their operand fragment from it at shifted addresses: lane (pixel x, half kh) of step j reads the 16 bytes of input pixels
//     2x + 4 (j & 1) + 2 kh, + 1 in row 2y + (j >> 1) -- a contiguous 512 bytes per half wave, no bounds logic (xpad carries the
//     padding);
//   * the WEIGHTS live in REGISTERS for the whole launch (64 channels x 224 = 7 rows x 32: 28 fragments per lane);
//   * the WEIGHTS are the MFMA's first operand and the pixels its second, so that a lane of the 32 x 32 result holds ONE pixel
//     and four consecutive channels per register quad: after the affine (+ ReLU) the rounded values go straight into an LDS
//     image of the convolution rows, [row slot][pixel slot][64 channels], without a transposing read;
//   * the pool reads that image: a ring of RS + 1 rows, the extra one being the last row of the previous step (pool row p needs
//     convolution rows 2p - 1 .. 2p + 1), so nothing is computed twice inside a workgroup's contiguous run of steps; a run that
//     starts inside an image computes the one row above it first.  Even and odd columns are stored in separate halves of a row
//     and the 16-byte channel chunks are XOR-swizzled by the column, which makes the epilogue's 8-byte writes (stride one pixel
//     across lanes) and the pool's 16-byte reads (stride two pixels) both conflict-free;
//   * only the pooled tensor [B, H/4, W/4, 64] is written.
// Same k order as the tile kernel (k = r * 32 + s * 4 + c; the all-zero eighth kernel row is skipped, which adds exact zeros) and
// the same epilogue arithmetic: the pooled output is bit-identical to creid_stem_conv_fwd_affine + creid_maxpool3x3s2_fwd
// (tests/test_eval_fold_gpu.py).
#include "conv_common.hpp"
#include <stdlib.h>

namespace creid_stem {

// running maximum of 16-byte chunks of eight 16-bit values.  bf16: the upper element of a word is compared as the float the whole
// word spells (the lower element's bits only extend its mantissa: the order of two different upper elements is unchanged, equal
// ones stay equal after the final mask), the lower one after a shift; f16: the packed maximum.  Equal to the compare-and-keep of
// maxpool_fwd_kernel for every value but a NaN or a tie of +0 with -0.
// lane I of the caller's quad
template <int I> __device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true));
}

template <typename ET> struct PoolMax;
// (inline asm: written as fmaxf / elementwise_max the compiler canonicalises every loaded operand first -- one extra v_max per value)
__device__ __forceinline__ float pool_vmax(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ unsigned pool_pkmax_f16(unsigned a, unsigned b) {
  unsigned d;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <> struct PoolMax<Bf16T> {
  float lo[4], hi[4];
  __device__ __forceinline__ explicit PoolMax(const uint4& v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { hi[k] = __uint_as_float(w[k]); lo[k] = __uint_as_float(w[k] << 16); }
  }
  __device__ __forceinline__ void update(const uint4& v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = pool_vmax(hi[k], __uint_as_float(w[k]));
      lo[k] = pool_vmax(lo[k], __uint_as_float(w[k] << 16));
    }
  }
  __device__ __forceinline__ void merge(const PoolMax& p) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { hi[k] = pool_vmax(hi[k], p.hi[k]); lo[k] = pool_vmax(lo[k], p.lo[k]); }
  }
  __device__ __forceinline__ uint4 result() const {
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (__float_as_uint(hi[k]) & 0xffff0000u) | (__float_as_uint(lo[k]) >> 16);
    return make_uint4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct PoolMax<F16T> {
  unsigned m[4];
  __device__ __forceinline__ explicit PoolMax(const uint4& v) { m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w; }
  __device__ __forceinline__ void update(const uint4& v) {
    m[0] = pool_pkmax_f16(m[0], v.x); m[1] = pool_pkmax_f16(m[1], v.y);
    m[2] = pool_pkmax_f16(m[2], v.z); m[3] = pool_pkmax_f16(m[3], v.w);
  }
  __device__ __forceinline__ void merge(const PoolMax& p) {
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = pool_pkmax_f16(m[k], p.m[k]);
  }
  __device__ __forceinline__ uint4 result() const { return make_uint4(m[0], m[1], m[2], m[3]); }
};

// W1: convolution columns (W / 2); RS: convolution rows per step; NW: waves per workgroup
template <int W1, int RS, int NW, typename ET>
__global__ __launch_bounds__(NW * 64, 8 / NW) void stem_pool_kernel(const unsigned short* __restrict__ xpad, int H1,
                                                                     const unsigned short* __restrict__ w_stem,
                                                                     const float* __restrict__ scale_shift, int relu,
                                                                     unsigned short* __restrict__ out, int n_tiles, int abl) {
  constexpr int NT = NW * 64;
  constexpr int SPR = W1 / 32;                          // 32-pixel blocks per convolution row
  constexpr int PITCH = (2 * W1 + 6) * 4;               // elements per input row (NHWC4)
  constexpr int CPRW = W1 + 3;                          // 16-byte chunks per input row
  constexpr int IN_ROWS = 2 * RS + 5;
  constexpr int IN_ELEMS = IN_ROWS * PITCH;
  constexpr int NLD = (IN_ROWS * CPRW + NT - 1) / NT;   // chunks per thread and step
  constexpr int NSLOT = RS + 1, HALF = W1 / 2;
  constexpr int CO_ELEMS = NSLOT * W1 * 64;
  constexpr int PR = ((RS / 2) % 2 == 0 && ((RS / 4) * HALF * 8) % NT == 0) ? 2 : 1;   // pool rows per item
  constexpr int WP = 256 + 8;                           // weight row pitch of the one-time LDS image (conflict-free fragment reads)
  static_assert(W1 % 32 == 0 && RS % 2 == 0 && PITCH % 8 == 0 && 64 * WP <= CO_ELEMS, "stem tile");
  static_assert((IN_ELEMS + CO_ELEMS) * 2 <= (NW == 8 ? 160 : 80) * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[IN_ELEMS + CO_ELEMS];
  unsigned short* in = smem;
  unsigned short* co = smem + IN_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int TPI = H1 / RS, H2 = H1 / 2;
  const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(n_tiles, t_begin + per);
  const int n_iter = t_end - t_begin;
  if (n_iter <= 0) return;
  const int it0 = (t_begin % TPI) != 0 ? -1 : 0;        // a run that starts inside an image first computes the row above it

  // folded BatchNorm: a lane of the result holds channels 32 n + 8 q + 4 kh + 0..3 in register quad q -- the same four channels for
  // all lanes of its half wave.  Sixteen registers hold ONE (scale, shift) per (n, q), that of channel .. + (lane & 3); the epilogue
  // fetches the other three from the quad's lanes (DPP quad_perm broadcast).  Read as float4's from an LDS table instead, the
  // 16 broadcast reads per 32 x 64 block were as much LDS traffic as all 14 operand fragments of the block.
  float my_sc[2][4], my_sh[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      my_sc[n][q] = scale_shift[32 * n + 8 * q + 4 * kh + (lane & 3)];
      my_sh[n][q] = scale_shift[64 + 32 * n + 8 * q + 4 * kh + (lane & 3)];
    }
  // weights: coalesced into an LDS image (aliasing the row ring, which is first written behind the loop's first barrier), then
  // this lane's 28 fragments: channel 32 n + l31, k = 16 j + 8 kh ..
  s16x8 bw[2][14];
  {
    unsigned short* wl = co;
    for (int i = tid; i < 64 * 28; i += NT) {
      const int o = i / 28, c = i - o * 28;
      *reinterpret_cast<uint4*>(wl + o * WP + c * 8) = *reinterpret_cast<const uint4*>(w_stem + o * 256 + c * 8);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 14; ++j) bw[n][j] = *reinterpret_cast<const s16x8*>(wl + (32 * n + l31) * WP + 16 * j + 8 * kh);
  }

  auto geom = [&](int it, int& b, int& r_first, int& nrows) {
    const int tile = t_begin + (it < 0 ? 0 : it);
    b = tile / TPI;
    r_first = (tile - b * TPI) * RS;
    nrows = RS;
    if (it < 0) { r_first -= 1; nrows = 1; }
  };
  const float floor_ = relu ? 0.f : -INFINITY;          // ReLU without a branch: max(v, -inf) = v
  uint4 areg[NLD];
  auto load_a = [&](int it) {
    int b, r_first, nrows;
    geom(it, b, r_first, nrows);
    const uint4* src = reinterpret_cast<const uint4*>(xpad + ((int64_t)b * (2 * H1 + 8) + 2 * r_first) * PITCH);
    const int nch = (2 * nrows + 5) * CPRW;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + NT * u;                       // (branch-free: threads beyond the block re-read its last chunk)
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (!CREID_ABL_ON(abl, 2)) v = src[min(idx, nch - 1)];
      areg[u] = v;
    }
  };
  auto put_a = [&](int it) {
    int b, r_first, nrows;
    geom(it, b, r_first, nrows);
    const int nch = (2 * nrows + 5) * CPRW;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + NT * u;
      if (idx < nch) reinterpret_cast<uint4*>(in)[idx] = areg[u];
    }
  };

  load_a(it0);
  __syncthreads();                                                // weight fragments are in registers: the image may go
  put_a(it0);
  if (it0 + 1 < n_iter) load_a(it0 + 1);
  for (int it = it0; it < n_iter; ++it) {
    int b, r_first, nrows;
    geom(it, b, r_first, nrows);
    __syncthreads();                                              // step `it`'s input is in LDS; the pool of step it - 1 has read the ring
    const int n_sub = nrows * SPR;
    for (int sub = wave; sub < n_sub; sub += NW) {
      const int row_l = sub / SPR, col = (sub - row_l * SPR) * 32 + l31;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      if (!CREID_ABL_ON(abl, 1)) {
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        u32x4v pf[4];
        const unsigned a0 = (unsigned)(uintptr_t)(in + (2 * row_l) * PITCH + (2 * col + 2 * kh) * 4);
        // the second half of a kernel row sits 4 pixels = 32 bytes to the right
        auto rdj = [&](int j, u32x4v& d) {
          const unsigned addr = a0 + (unsigned)((j >> 1) * PITCH * 2 + (j & 1) * 32);
          asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
        };
        rdj(0, pf[0]); rdj(1, pf[1]); rdj(2, pf[2]);
#pragma unroll
        for (int j = 0; j < 14; ++j) {
          if (j + 3 < 14) rdj(j + 3, pf[(j + 3) & 3]);
          const int young = (j + 3 < 14) ? 3 : 13 - j;             // reads younger than step j's
          u32x4v& p = pf[j & 3];
          if (young == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p) :: "memory");
          else if (young == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(p) :: "memory");
          else if (young == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(p) :: "memory");
          else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(p) :: "memory");
          acc0 = ET::mfma(bw[0][j], __builtin_bit_cast(s16x8, p), acc0);
          acc1 = ET::mfma(bw[1][j], __builtin_bit_cast(s16x8, p), acc1);
        }
      }
      if (CREID_ABL_ON(abl, 4)) continue;
      // affine (+ ReLU), round, into the row ring: even columns in the first half of a row, odd ones in the second
      const int R = r_first + row_l;
      const int slot = R % NSLOT;
      unsigned short* px = co + ((slot * W1) + (col >> 1) + (col & 1) * HALF) * 64;
      const int sw = col & 7;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // channel 32 n + 8 q + 4 kh + i: its (scale, shift) sit in lane i of this lane's quad (see my_sc above)
          const f32x16& acc = n ? acc1 : acc0;
          float v0 = fmaf(acc[4 * q], quad_bcast<0>(my_sc[n][q]), quad_bcast<0>(my_sh[n][q]));
          float v1 = fmaf(acc[4 * q + 1], quad_bcast<1>(my_sc[n][q]), quad_bcast<1>(my_sh[n][q]));
          float v2 = fmaf(acc[4 * q + 2], quad_bcast<2>(my_sc[n][q]), quad_bcast<2>(my_sh[n][q]));
          float v3 = fmaf(acc[4 * q + 3], quad_bcast<3>(my_sc[n][q]), quad_bcast<3>(my_sh[n][q]));
          v0 = fmaxf(v0, floor_); v1 = fmaxf(v1, floor_); v2 = fmaxf(v2, floor_); v3 = fmaxf(v3, floor_);
          *reinterpret_cast<uint2*>(px + (((4 * n + q) ^ sw) << 3) + 4 * kh) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
    }
    __syncthreads();                                              // the rows are in the ring; the input tile is free
    if (it + 1 < n_iter) put_a(it + 1);
    if (it + 2 < n_iter) load_a(it + 2);
    if (it < 0 || CREID_ABL_ON(abl, 8)) continue;
    // 3 x 3 / 2 max-pool over the ring: item = (pool row, pool column, 16-byte channel chunk).  Branch-free: a tap above the image
    // or left of it is replaced by its neighbour inside the window (a duplicate does not change a maximum), all nine reads are
    // requested before the first comparison.
    const int p_first = r_first >> 1;
    // lane -> (pixel of a group of four, 16-byte channel chunk): ds_read_b128 is serviced in four NON-contiguous 16-lane groups
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32); with this map every group reads all eight chunks of two ADJACENT pool
    // pixels, whose taps lie in different halves of the 256-byte bank row -- the plain map (chunk = lane & 7, pixel = lane >> 3)
    // put pixels p and p + 2 (same half, complementary chunk halves XOR-ed onto each other by the column swizzle) into one group
    const int l5 = lane & 31;
    const int dpx = l5 < 4 ? 0 : l5 < 12 ? 2 : l5 < 16 ? 0 : l5 < 20 ? 3 : l5 < 28 ? 1 : 3;
    const int c_lane = l5 < 4 ? l5 : l5 < 12 ? l5 - 4 : l5 < 16 ? l5 - 8 : l5 < 20 ? l5 - 16 : l5 < 28 ? l5 - 20 : l5 - 24;
    if constexpr (PR == 2) {
      // two vertically adjacent pool rows per item: five ring rows x three columns, the middle row's maximum serves both
      for (int item = tid; item < (RS / 4) * HALF * 8; item += NT) {
        const int pq = (item >> 5) * 4 + dpx, c = c_lane, pxl = pq % HALF, pp = pq / HALF;
        const int py = p_first + 2 * pp;
        uint4 tap[15];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const int R = max(2 * py + r - 1, 0);
          const unsigned short* rowp = co + (R % NSLOT) * W1 * 64;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int cc = max(2 * pxl + s - 1, 0);
            tap[r * 3 + s] = *reinterpret_cast<const uint4*>(rowp + ((cc >> 1) + (cc & 1) * HALF) * 64 + ((c ^ (cc & 7)) << 3));
          }
        }
        PoolMax<ET> mid(tap[6]), top(tap[0]), bot(tap[9]);
        mid.update(tap[7]); mid.update(tap[8]);
#pragma unroll
        for (int t = 1; t < 6; ++t) top.update(tap[t]);
#pragma unroll
        for (int t = 10; t < 15; ++t) bot.update(tap[t]);
        top.merge(mid); bot.merge(mid);
        unsigned short* o = out + (((int64_t)b * H2 + py) * HALF + pxl) * 64 + c * 8;
        *reinterpret_cast<uint4*>(o) = top.result();
        *reinterpret_cast<uint4*>(o + HALF * 64) = bot.result();
      }
    } else {
      for (int item = tid; item < (RS / 2) * HALF * 8; item += NT) {
        const int pq = (item >> 5) * 4 + dpx, c = c_lane, pxl = pq % HALF, pyl = pq / HALF;
        const int py = p_first + pyl;
        uint4 tap[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int R = max(2 * py + r - 1, 0);
          const unsigned short* rowp = co + (R % NSLOT) * W1 * 64;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int cc = max(2 * pxl + s - 1, 0);
            tap[r * 3 + s] = *reinterpret_cast<const uint4*>(rowp + ((cc >> 1) + (cc & 1) * HALF) * 64 + ((c ^ (cc & 7)) << 3));
          }
        }
        PoolMax<ET> pm(tap[0]);
#pragma unroll
        for (int t = 1; t < 9; ++t) pm.update(tap[t]);
        *reinterpret_cast<uint4*>(out + (((int64_t)b * H2 + py) * HALF + pxl) * 64 + c * 8) = pm.result();
      }
    }
  }
}

}  // namespace creid_stem

extern "C" {

/* Eval-mode stem in one launch: see include/creid.h.  CREID_E_SHAPE: outside the kernel's tiles (the caller then runs
 * creid_stem_conv_fwd_affine + creid_maxpool3x3s2_fwd). */
int creid_stem_conv_pool_fwd_affine(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                                    const float* scale_shift, int relu, int dtype, void* stream) {
  CREID_CHECK_ARG(xpad && w_stem && y && scale_shift && batch > 0 && H > 0 && W > 0);
  if (!creid_is16(dtype)) return CREID_E_DTYPE;
  if (H % 8 || W % 4) return CREID_E_SHAPE;
  const int H1 = (int)(H / 2), W1 = (int)(W / 2);
  hipStream_t s = as_stream(stream);
#ifdef CREID_ABL_BUILD
  const char* ae = CREID_KNOB_ENV("CREID_STEM_ABL");            // 1 no multiplies, 2 no input loads, 4 no epilogue, 8 no pool
  const int abl = ae ? atoi(ae) : 0;
#else
  const int abl = 0;
#endif
  int wgs_env = 0;
  { const char* e = CREID_KNOB_ENV("CREID_STEM_WGS"); if (e) wgs_env = atoi(e); }   // read per call (tests)
#define CREID_STEM_LAUNCH(W1_, RS_, NW_, ET_)                                                                          \
  do {                                                                                                                 \
    if (H1 % RS_) return CREID_E_SHAPE;                                                                                \
    const int n_tiles = (int)batch * (H1 / RS_);                                                                       \
    int wgs = 256 * (8 / NW_);                                                                                         \
    if (wgs_env > 0) wgs = wgs_env;                                                                                    \
    if (wgs > n_tiles) wgs = n_tiles;                                                                                  \
    hipLaunchKernelGGL((creid_stem::stem_pool_kernel<W1_, RS_, NW_, ET_>), dim3((unsigned)wgs), dim3(NW_ * 64), 0, s,  \
                       (const unsigned short*)xpad, H1, (const unsigned short*)w_stem, scale_shift, relu ? 1 : 0,      \
                       (unsigned short*)y, n_tiles, abl);                                                              \
  } while (0)
  // W = 128: four-wave workgroups, two per CU (one multiplies while the other pools: 43 vs 50 us at B = 128) unless
  // CREID_STEM_FORM=0; W = 320 only fits the eight-wave form (137 KB of LDS)
  int form = 1;
  { const char* e = CREID_KNOB_ENV("CREID_STEM_FORM"); if (e) form = atoi(e); }      // read per call (tests)
  if (W1 == 64) {
    if (form == 1 && H1 % 4 == 0) { if (dtype == CREID_F16) CREID_STEM_LAUNCH(64, 4, 4, F16T); else CREID_STEM_LAUNCH(64, 4, 4, Bf16T); }
    else { if (dtype == CREID_F16) CREID_STEM_LAUNCH(64, 8, 8, F16T); else CREID_STEM_LAUNCH(64, 8, 8, Bf16T); }
  } else if (W1 == 160) {
    if (dtype == CREID_F16) CREID_STEM_LAUNCH(160, 4, 8, F16T); else CREID_STEM_LAUNCH(160, 4, 8, Bf16T);
  } else {
    return CREID_E_SHAPE;
  }
#undef CREID_STEM_LAUNCH
  CREID_LAUNCH_RET();
}

}  // extern "C"
