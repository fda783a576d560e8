// Geometry shared by the implicit-GEMM convolution kernels (stage A).
//
// Every convolution of the backbone (modelling/backbones/resnet.py:56-61,94,109) is lowered to
// an im2col-FREE GEMM  C[M, N] = sum_k A[m, k] * W[n, k]  over NHWC activations:
//   m = (b, oy, ox) output pixel, n = output channel, k = (tap, c) with tap = (r, s);
// the A row for a tap is a CONTIGUOUS run of `span` channels of one source pixel, so the A tile
// is gathered straight from the activation tensor with 16-byte loads (zero-filled outside the
// image) -- no column buffer is ever materialised.  The same kernel computes the data gradient
// (`transposed` = 1: source = dY, weights pre-transposed to [Cin][r][s][Cout]) and the stem
// (7x7 s2 on a pre-padded NHWC4 image: span = 32 = 8 pixels x 4 channels per kernel row).
#pragma once
#include "common.hpp"

struct IGemmGeom {
  int M;              // rows of the GEMM = batch * OH * OW
  int OH, OW;         // spatial dims that index the rows
  int SH, SW;         // source spatial dims (tensor the A operand is gathered from)
  int pitch;          // elements per source pixel (channels; 4 for the padded stem image)
  int log2span;       // log2(elements of K per tap)
  int kw;             // taps per kernel row: tap -> (r = tap / kw, s = tap % kw)
  int stride, pad;
  int transposed;     // 0: iy = oy*stride + r - pad ; 1: ty = oy + pad - r, iy = ty/stride if divisible
  int K;              // reduction length = taps << log2span
  int N;              // output channels
  int check_bounds;   // 0: source is pre-padded (stem)
  float inv_ohow, inv_ow;   // 1/(OH*OW), 1/OW for fast_divmod (M < 2^24)
  int add_compact;          // epilogue add_src is a stride-2 COMPACT tensor [B, OH/2, OW/2, N]: added at even (y, x) only
  int parity;               // transposed stride-2 (data gradient): GEMM rows are enumerated parity-class major --
                            // row m = class*(M/4) + (b, y/2, x/2), class = (y&1)*2 + (x&1) -- so a 128-row tile holds
                            // one class and only that class's taps (1, 2, 2 or 4 of 9 for 3x3) are visited
  const unsigned char* add_mask;   // nullable (full-resolution add_src only): bit mask applied to add_src before the add,
                                   // one byte per 8 channels -- the residual-branch gradient is then the UNMASKED incoming
                                   // gradient plus the ReLU bits, and no masked copy has to be written and re-read
  const float* epi_scale;   // nullable: per-output-channel affine y = acc * scale + shift applied to the fp32 accumulators --
  const float* epi_shift;   // eval-mode BatchNorm (running statistics are constants) folded into the convolution
  int epi_relu;             // ReLU after the affine and after "+ add_src" (the block's residual) when that is given
  int abl;                  // timing-only ablation bits (CREID_IGEMM_ABL, producer/consumer kernel): 1 no MFMA, 2 no DMA after the
                            // first k-tile, 4 no fragment reads, 8 no copy-out stores -- results are WRONG when set.  Only the
                            // ablation build (build.py --ablation -> libcreid_hip_abl.so) looks at them
};

// The switches cost 0.07 ms per step when compiled into the k-loops (same-box A/B), so the shipped library folds them away.
#ifdef CREID_ABL_BUILD
#define CREID_ABL_ON(word, bits) ((word) & (bits))
#else
#define CREID_ABL_ON(word, bits) 0
#endif

static inline void igemm_finish_geom(IGemmGeom& g) {
  g.inv_ohow = 1.0f / (float)(g.OH * g.OW);
  g.inv_ow = 1.0f / (float)g.OW;
  g.add_compact = 0;
  g.parity = 0;
  g.add_mask = nullptr;
  g.epi_scale = nullptr;
  g.epi_shift = nullptr;
  g.epi_relu = 0;
  g.abl = 0;
}

// Source pixel of output row (oy, ox) under tap (r, s); returns false when it falls outside.
// stride is 1 or 2 (check_desc / the stem), so the lattice test and the division are a mask and a shift --
// a runtime `%` and `/` here expand to ~60 VALU instructions per row on the k-loop path of every conv kernel.
__device__ __forceinline__ bool igemm_src_pixel(const IGemmGeom& g, int oy, int ox, int r, int s, int& iy, int& ix) {
  const int sh = g.stride - 1;                       // log2(stride); for stride 2 also the parity mask
  if (!g.transposed) {
    iy = (oy << sh) + r - g.pad;
    ix = (ox << sh) + s - g.pad;
  } else {
    const int ty = oy + g.pad - r, tx = ox + g.pad - s;
    if (((ty | tx) & sh) != 0) return false;         // not on the stride lattice
    iy = ty >> sh; ix = tx >> sh;                    // arithmetic shift: negatives stay negative, rejected below
  }
  if (!g.check_bounds) return true;
  return (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
}

// m / d and m % d for 0 <= m < 2^24 (exact in fp32) with a precomputed reciprocal: ~8 VALU ops instead
// of the ~25-op integer division sequence (the row -> (b, oy, ox) decomposition is on the k-loop path
// of the weight-gradient kernel).
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
  q = (int)((float)m * inv_d);
  r = m - q * d;
  if (r >= d) { ++q; r -= d; }
  if (r < 0) { --q; r += d; }
}

// Optional fusion of the NEXT BatchNorm-backward's column reduction into a data-gradient epilogue: the tile
// just produced is g = dL/da for the layer whose raw conv output is `x` and post-ReLU activation is `act`;
// the epilogue accumulates (sum dy, sum dy*xhat), dy = g*[act>0], per channel into
// partial[tile_m][2][N] (the layout creid_bn2d_bwd expects, one row pair per 128 rows).
struct BnRedArgs {
  const void* x;
  const void* act;       // nullable: no ReLU mask
  const float* mean;
  const float* invstd;
  float* partial;
  int prefetch;          // fetch x / act before the C tile is staged (A/B knob CREID_BNRED_PREFETCH, default 1)
  int tiles_per_image;   // 0: mean/invstd are per channel [N]; > 0: per (image, channel) [B][N] (IBN), this many
                         // 128-row tiles per image
  const unsigned char* mask;   // nullable: ReLU mask as bits (creid_bn2d_apply_mask), one byte per 8 channels; replaces `act`
};

// XCD-aware bijective remap of the linear workgroup id (consecutive ids land on different XCDs;
// give every XCD a contiguous run of tiles so neighbouring tiles share operand panels in its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  return base + (bid >> 3);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static constexpr int DT = CREID_F32;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<_Float16> {
  static constexpr int DT = CREID_F16;
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};
template <> struct ElemIO<unsigned short> {   // bf16 bits
  static constexpr int DT = CREID_BF16;
  static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};
