// Stage A: implicit-GEMM convolution forward / data-gradient for NHWC activations.
// (replaces nn.Conv2d fwd + dgrad of modelling/backbones/resnet.py:56-61,94,109 and
//  resnet_ibn_a.py:40-48,83,110 -- bias-free 7x7 s2, 3x3 (s1/s2), 1x1 (s1/s2) convolutions.)
//
// bf16: 128 x BN x 64 tile (2x2 waves, 64 x BN/2 per wave) on v_mfma_f32_32x32x16_bf16, fp32 accumulate; LDS
//       row-major [row][64] with the 16-B chunk index XOR-swizzled by (row>>1)&7 (conflict-free ds_read_b128).
//       Three kernels share that tiling and the epilogue:
//         igemm_bf16_dma_kernel  global->LDS DMA gather (the A rows come from different image rows / taps),
//                                NS-stage ring, 256 threads -- the default for every 1x1 / 3x3 layer;
//         igemm_bf16_ws_kernel   512 threads, producer waves issue the DMA, consumer waves multiply -- long-k
//                                128 x 64 tiles;
//         igemm_bf16_kernel      register-staged gather -- the stem (32-element taps).
// f32 : 128 x BN x 16 tile on v_mfma_f32_32x32x2_f32 (exact f32; parity mode), K-major LDS.
// Epilogue (both): optional "+ add_src" (residual-gradient accumulation in dgrad), store in the
// activation dtype, and optional per-tile per-channel (sum, sum of squares) partials of the fp32
// accumulators for the training-mode BatchNorm that follows every convolution.
#include "conv_common.hpp"
#include <type_traits>
#include "wgrad_reduce.hpp"
#include "tune.hpp"
#include <stdlib.h>

// defined in conv_wgrad.hip
bool wgrad_make_reduce_job(const creid_conv_desc* d, int dtype, const void* ws, size_t ws_bytes, float* dw, int accumulate,
                           WRedJob& j);
int wgrad_reduce_job_launch(const WRedJob& j, hipStream_t s);
// defined in conv_stream.hip
int launch_stream1x1(int M, int K, int N, const void* src, const void* wgt, void* out, float* bn_part, hipStream_t s);
int launch_stream2(int M, int K, int N, const void* src, const void* wgt, void* out, float* bn_part, const void* add_src,
                   const float* epi_scale, const float* epi_shift, int epi_relu, int bn_cap, int dtype, hipStream_t s);
int launch_conv3x3_c64(int M, int H, int Wd, const void* src, const void* wgt, void* out, float* bn_part, const float* epi_scale,
                       const float* epi_shift, int epi_relu, int dtype, hipStream_t s);
// defined in conv_pipe.hip
int launch_igemm_pp(const IGemmGeom& g, const void* src, const void* wgt, void* out, const void* add_src, float* bn_part,
                    int variant, int dtype, hipStream_t s);

// Out-of-image taps read this 128-byte page of zeros instead of selecting zeros per dword (saves 3 VALU
// per load in the gather path).
__device__ __attribute__((aligned(128))) unsigned g_zero_page[32];

// ------------------------------------------------------------------------------------ bf16
template <int BN>
__global__ __launch_bounds__(256, BN == 128 ? 1 : 2) void igemm_bf16_kernel(IGemmGeom g, const unsigned short* __restrict__ src,
                                                         const unsigned short* __restrict__ wgt,
                                                         unsigned short* __restrict__ out,
                                                         const unsigned short* __restrict__ add_src,
                                                         float* __restrict__ bn_part, int tiles_n) {
  constexpr int BK = 64, TNW = BN / 64, NB = BN / 32;
  constexpr int CP = BN + 8;                                   // C-tile staging pitch (elements)
  constexpr int LDS_ELEMS = (128 * BK + BN * BK) > (128 * CP) ? (128 * BK + BN * BK) : (128 * CP);
  __shared__ __attribute__((aligned(16))) unsigned short smem[LDS_ELEMS];
  unsigned short* As = smem;                 // single-buffered tiles + register prefetch: half the LDS of a
  unsigned short* Bs = smem + 128 * BK;      // double buffer -> twice the resident workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * 128, col0 = tile_n * BN;
  const int span_mask = (1 << g.log2span) - 1;
  const bool tap_uniform = g.log2span >= 6;        // a 64-wide k-tile never straddles two taps (all but the stem)

  const int lrow = tid >> 3, lch = tid & 7;
  int oy[4], ox[4], bpix[4], soff[4];
  bool vm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = lrow + 32 * i, m = row0 + r;
    vm[i] = m < g.M;
    const int mm = vm[i] ? m : 0;
    int b, rem;
    fast_divmod(mm, g.OH * g.OW, g.inv_ohow, b, rem);
    fast_divmod(rem, g.OW, g.inv_ow, oy[i], ox[i]);
    bpix[i] = b * g.SH * g.SW;
    soff[i] = r * BK + ((lch ^ ((r >> 1) & 7)) << 3);
  }
  const unsigned short* wp[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wp[i] = wgt + (int64_t)(col0 + lrow + 32 * i) * g.K + 8 * lch;

  // A-row source pointers are recomputed only when the k-loop enters a new tap (tap-major K order);
  // inside a tap consecutive k-tiles just advance by 64 channels.
  const unsigned short* aptr[4];
  int amul[4];                 // 1 for a real source row, 0 for the zero page (kills the channel offset)
  int cur_tap = -1;
  const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_zero_page) + 8 * lch;
  auto set_tap = [&](int tap) {
    const int r = tap / g.kw, s = tap - r * g.kw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy, ix;
      const bool ok = vm[i] && igemm_src_pixel(g, oy[i], ox[i], r, s, iy, ix);
      aptr[i] = ok ? src + (int64_t)(bpix[i] + iy * g.SW + ix) * g.pitch + 8 * lch : zpage;
      amul[i] = ok ? 1 : 0;
    }
  };
  // 3-deep register prefetch ring: the loads of k-tile t+3 are issued while tile t is multiplied, so a
  // load has ~3 compute phases to land (HBM/L2 latency >> one 16-MFMA phase at 2 workgroups per CU).
  // The ring lives in NAMED scalars (arrays / lambdas taking array references kept ending up in scratch).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector: always SROA-able
  struct Stage { u32x4 a0, a1, a2, a3, b0, b1, b2, b3; };
  Stage st0, st1, st2;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define IGEMM_LDA(I_, DST, C_) DST = *reinterpret_cast<const u32x4*>(aptr[I_] + (C_) * amul[I_])
#define IGEMM_LDA_G(I_, DST, R_, S_, C_)                                                                \
  do {                                                                                                  \
    int iy, ix;                                                                                         \
    if (vm[I_] && igemm_src_pixel(g, oy[I_], ox[I_], R_, S_, iy, ix))                                   \
      DST = *reinterpret_cast<const u32x4*>(src + (int64_t)(bpix[I_] + iy * g.SW + ix) * g.pitch + (C_)); \
    else                                                                                                \
      DST = zero4;                                                                                      \
  } while (0)
#define IGEMM_GLOAD(T_, ST)                                                                             \
  do {                                                                                                  \
    const int t_ = (T_);                                                                                \
    if (tap_uniform) {                                                                                  \
      const int tap_ = (t_ * BK) >> g.log2span;                                                         \
      if (tap_ != cur_tap) { set_tap(tap_); cur_tap = tap_; }                                           \
      const int c_ = (t_ * BK) & span_mask;                                                             \
      IGEMM_LDA(0, ST.a0, c_); IGEMM_LDA(1, ST.a1, c_); IGEMM_LDA(2, ST.a2, c_); IGEMM_LDA(3, ST.a3, c_); \
    } else {                                                                                            \
      const int kk_ = t_ * BK + 8 * lch;                                                                \
      const int tap_ = kk_ >> g.log2span, c_ = kk_ & span_mask;                                         \
      const int r_ = tap_ / g.kw, s_ = tap_ - r_ * g.kw;                                                \
      IGEMM_LDA_G(0, ST.a0, r_, s_, c_); IGEMM_LDA_G(1, ST.a1, r_, s_, c_);                             \
      IGEMM_LDA_G(2, ST.a2, r_, s_, c_); IGEMM_LDA_G(3, ST.a3, r_, s_, c_);                             \
    }                                                                                                   \
    ST.b0 = *reinterpret_cast<const u32x4*>(wp[0] + t_ * BK);                                           \
    ST.b1 = *reinterpret_cast<const u32x4*>(wp[1] + t_ * BK);                                           \
    if constexpr (NB == 4) {                                                                            \
      ST.b2 = *reinterpret_cast<const u32x4*>(wp[2] + t_ * BK);                                         \
      ST.b3 = *reinterpret_cast<const u32x4*>(wp[3] + t_ * BK);                                         \
    }                                                                                                   \
  } while (0)
#define IGEMM_LSTORE(ST)                                                                                \
  do {                                                                                                  \
    *reinterpret_cast<u32x4*>(&As[soff[0]]) = ST.a0; *reinterpret_cast<u32x4*>(&As[soff[1]]) = ST.a1;   \
    *reinterpret_cast<u32x4*>(&As[soff[2]]) = ST.a2; *reinterpret_cast<u32x4*>(&As[soff[3]]) = ST.a3;   \
    *reinterpret_cast<u32x4*>(&Bs[soff[0]]) = ST.b0; *reinterpret_cast<u32x4*>(&Bs[soff[1]]) = ST.b1;   \
    if constexpr (NB == 4) {                                                                            \
      *reinterpret_cast<u32x4*>(&Bs[soff[2]]) = ST.b2; *reinterpret_cast<u32x4*>(&Bs[soff[3]]) = ST.b3; \
    }                                                                                                   \
  } while (0)

  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  const int l31 = lane & 31, kh = lane >> 5;
  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = 2 * kk + kh;
      s16x8 a[2], b[TNW];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        a[i] = *reinterpret_cast<const s16x8*>(&As[r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int c = wn * (BN / 2) + j * 32 + l31;
        b[j] = *reinterpret_cast<const s16x8*>(&Bs[c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                              __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
  };
  // Every prefetch is unconditional (the tile index is clamped; re-loading the last tile is harmless).
  // Main loop: whole triples of k-tiles with the ring stages named statically (no register rotation);
  // the 0-2 leftover tiles run through a rotating tail.
  const int last = nk - 1;
  IGEMM_GLOAD(0, st0);
  IGEMM_GLOAD(min(1, last), st1);
  IGEMM_GLOAD(min(2, last), st2);
  const int nk3 = nk - nk % 3;
  int t = 0;
  for (; t < nk3; t += 3) {
    __syncthreads();               // fragment reads of the previous tile are done
    IGEMM_LSTORE(st0);
    __syncthreads();
    IGEMM_GLOAD(min(t + 3, last), st0);
    compute();
    __syncthreads();
    IGEMM_LSTORE(st1);
    __syncthreads();
    IGEMM_GLOAD(min(t + 4, last), st1);
    compute();
    __syncthreads();
    IGEMM_LSTORE(st2);
    __syncthreads();
    IGEMM_GLOAD(min(t + 5, last), st2);
    compute();
  }
  for (; t < nk; ++t) {            // at most two iterations
    __syncthreads();
    IGEMM_LSTORE(st0);
    __syncthreads();
    st0 = st1;
    compute();
  }
#undef IGEMM_GLOAD
#undef IGEMM_LSTORE
#undef IGEMM_LDA
#undef IGEMM_LDA_G
  __syncthreads();

  // ---- epilogue: per-channel (sum, sumsq) partials from the fp32 accumulators, then the C tile is
  // staged through LDS (bf16) so that global stores (and the add_src reads) are 16-B coalesced rows.
  float s1v[TNW], s2v[TNW];
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    const int cl = wn * (BN / 2) + j * 32 + l31;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float v = acc[i][j][r];               // rows >= M were zero-filled -> contribute 0
        s1 += v; s2 = fmaf(v, v, s2);
        smem[rl * CP + cl] = f32_to_bf16_bits(v);
      }
    }
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    s1v[j] = s1; s2v[j] = s2;
  }
  __syncthreads();
  constexpr int CPR = BN / 8;                                  // 16-B chunks per tile row
#pragma unroll
  for (int i = 0; i < (128 * CPR) / 256; ++i) {
    const int id = tid + 256 * i, rl = id / CPR, ch = id - rl * CPR;
    const int rr = row0 + rl;
    if (rr < g.M) {
      uint4 v = *reinterpret_cast<const uint4*>(&smem[rl * CP + ch * 8]);
      const int64_t off = (int64_t)rr * g.N + col0 + ch * 8;
      if (add_src) {
        const uint4 a = *reinterpret_cast<const uint4*>(add_src + off);
        unsigned* vw = &v.x; const unsigned* aw = &a.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(vw[q] << 16) + __uint_as_float(aw[q] << 16);
          const float hi = __uint_as_float(vw[q] & 0xffff0000u) + __uint_as_float(aw[q] & 0xffff0000u);
          vw[q] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
        }
      }
      *reinterpret_cast<uint4*>(out + off) = v;
    }
  }
  if (bn_part) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [2 (wm)][2 (s1,s2)][BN]
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + l31;
      if (kh == 0) { red[(wm * 2 + 0) * BN + cl] = s1v[j]; red[(wm * 2 + 1) * BN + cl] = s2v[j]; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, cl = i - which * BN;
      bn_part[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
    }
  }
}

// ------------------------------------------------------------------------------------ bf16, LDS-DMA
// Same tiling / epilogue as igemm_bf16_kernel, but both operand tiles are fetched with gfx950's
// global->LDS DMA (`global_load_lds_dwordx4`: LDS[wave base + lane*16] <- that lane's 16 global bytes): no
// staging VGPRs, no ds_write pass, ~16 address VALU per k-tile.  The LDS image is linear (8 rows x 128 B per
// wave-instruction); the conflict-free XOR swizzle is applied on the SOURCE side (lane holding LDS chunk c'
// of row r fetches global chunk c' ^ ((r>>1)&7)) and again on the fragment reads.  Out-of-image taps and
// rows >= M fetch from a page of zeros.  NS-stage LDS ring: the DMAs of k-tiles t+1 .. t+NS-2 fly while tile t
// is multiplied; `s_waitcnt vmcnt(n)` with n = the DMA instructions of the younger tiles still in flight; one
// bare s_barrier per k-tile.  Measured (r01, profiles/r01_igemm_stage_sweep.md): NS = 2 wins on every ResNet50
// layer -- the loop is not DMA-latency bound, and 3+ stages cost a workgroup per CU.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BN, int NS, typename ET = Bf16T>
__global__ __launch_bounds__(256, (NS * (128 + BN) * 128 <= 80 * 1024) ? 2 : 1) void igemm_bf16_dma_kernel(IGemmGeom g, const unsigned short* __restrict__ src,
                                                                 const unsigned short* __restrict__ wgt,
                                                                 unsigned short* __restrict__ out,
                                                                 const unsigned short* __restrict__ add_src,
                                                                 float* __restrict__ bn_part, int tiles_n,
                                                                 BnRedArgs bnred) {
  constexpr int BK = 64, TNW = BN / 64, NBI = BN / 32;          // NBI = B-tile DMA instructions per wave
  constexpr int CPT = 128 + 4;                                   // transposed C staging (see igemm_bf16_ws_kernel)
  constexpr int TILE_A = 128 * BK, TILE_B = BN * BK, STAGE = TILE_A + TILE_B;
  constexpr int LDS_ELEMS = (NS * STAGE) > (BN * CPT) ? (NS * STAGE) : (BN * CPT);
  constexpr int LPT = 4 + NBI;                                   // DMA instructions per wave per k-tile
  static_assert((NS - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[LDS_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * 128, col0 = tile_n * BN;
  const int span_mask = (1 << g.log2span) - 1;

  // DMA lane map: instruction i of this wave covers tile rows ((i*4 + wave)*8 .. +7); lane -> (row, LDS chunk)
  const int lr8 = lane >> 3, lcp = lane & 7;
  int oy[4], ox[4], bpix[4], gch[4];
  bool vm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 8 + lr8, m = row0 + r;
    vm[i] = m < g.M;
    const int mm = vm[i] ? m : 0;
    int b, rem;
    fast_divmod(mm, g.OH * g.OW, g.inv_ohow, b, rem);
    fast_divmod(rem, g.OW, g.inv_ow, oy[i], ox[i]);
    bpix[i] = b * g.SH * g.SW;
    gch[i] = (lcp ^ ((r >> 1) & 7)) << 3;                     // swizzled SOURCE chunk (elements)
  }
  const unsigned short* wp[NBI];
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int r = (i * 4 + wave) * 8 + lr8;
    wp[i] = wgt + (int64_t)(col0 + r) * g.K + ((lcp ^ ((r >> 1) & 7)) << 3);
  }
  const unsigned short* aptr[4];
  int amul[4];
  int cur_tap = -1, cur_r = 0, cur_s = -1;                     // taps are visited in order: (r, s) advance incrementally
  const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_zero_page);
  auto set_tap = [&]() {
    if (++cur_s == g.kw) { cur_s = 0; ++cur_r; }
    const int r = cur_r, s = cur_s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy, ix;
      const bool ok = vm[i] && igemm_src_pixel(g, oy[i], ox[i], r, s, iy, ix);
      aptr[i] = ok ? src + (int64_t)(bpix[i] + iy * g.SW + ix) * g.pitch + gch[i] : zpage;
      amul[i] = ok ? 1 : 0;
    }
  };
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // The stem (7x7 s2 on the pre-padded NHWC4 image): a tap is one kernel row = 32 elements = 64 bytes, so a
  // 64-wide k-tile holds TWO taps; source chunk sc of the 128-byte tile row comes from kernel row 2t + (sc >> 2)
  // at byte offset (sc & 3) * 16.  No bounds checks (padding), the pointer just advances two image rows per k-tile.
  const bool stem = g.log2span == 5;
  const int stem_step = 2 * g.SW * g.pitch;
  if (stem) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sc = gch[i] >> 3;
      aptr[i] = vm[i] ? src + (int64_t)(bpix[i] + (oy[i] * 2 + (sc >> 2)) * g.SW + ox[i] * 2) * g.pitch + (sc & 3) * 8 : zpage;
      amul[i] = vm[i] ? 1 : 0;
    }
  }
  auto issue = [&](int t, int buf) {
    // running pointers: inside a tap every k-tile advances the A rows by 64 channels (0 for the zero page), the
    // stem by two image rows, the weight rows by 64 columns -- no per-tile multiplies
    if (!stem) {
      const int tap = (t * BK) >> g.log2span;
      if (tap != cur_tap) { set_tap(); cur_tap = tap; }
    }
    unsigned short* la = smem + buf * STAGE + wave * 512;      // + i * 2048 elements (4 KiB) per instruction
    unsigned short* lb = smem + buf * STAGE + TILE_A + wave * 512;
    const int a_step = stem ? stem_step : BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[i], (lptr_t)(la + i * 2048), 16, 0, 0);
      aptr[i] += amul[i] ? a_step : 0;
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)wp[i], (lptr_t)(lb + i * 2048), 16, 0, 0);
      wp[i] += BK;
    }
  };

  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < nk) issue(p, p);
  int buf = 0;                                                 // t % NS
  for (int t = 0; t < nk; ++t) {
    // this wave's DMA pieces of tile t have landed once at most the younger tiles' instructions are pending
    const int younger = min(nk - 1 - t, NS - 2);
    if constexpr (NS >= 5) { if (younger == 3) wait_vm<3 * LPT>(); }
    if constexpr (NS >= 4) { if (younger == 2) wait_vm<2 * LPT>(); }
    if constexpr (NS >= 3) { if (younger == 1) wait_vm<1 * LPT>(); }
    if (younger == 0) wait_vm<0>();
    // bare s_barrier: __syncthreads() carries a release fence that drains vmcnt to 0, i.e. the whole ring.
    // LDS-DMA data is visible once vmcnt has counted it; this wave's ds_reads of tile t-1 were consumed by MFMAs.
    asm volatile("s_barrier" ::: "memory");                    // everyone's pieces landed; buffer (t-1)%NS is free
    if (t + NS - 1 < nk) issue(t + NS - 1, buf == 0 ? NS - 1 : buf - 1);
    const unsigned short* As = smem + buf * STAGE;
    buf = (buf + 1 == NS) ? 0 : buf + 1;
    const unsigned short* Bs = As + TILE_A;
    // All fragment reads of the k-tile are issued before the first MFMA (the compiler then waits with
    // decreasing lgkmcnt): with a read->wait->MFMA chain per 16-wide k slice the LDS latency was exposed four
    // times per k-tile and a wave kept its MFMA pipe ~16% busy.
    s16x8 a[4][2], b[4][TNW];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = 2 * kk + kh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        a[kk][i] = *reinterpret_cast<const s16x8*>(&As[r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int c = wn * (BN / 2) + j * 32 + l31;
        b[kk][j] = *reinterpret_cast<const s16x8*>(&Bs[c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
          acc[i][j] = ET::mfma(a[kk][i], b[kk][j], acc[i][j]);
  }
  __syncthreads();

  // ---- epilogue (identical to igemm_bf16_kernel)
  // fused BN-backward reduction: its two operand tiles (x, act) are fetched NOW, so the loads fly while the
  // accumulators are staged through LDS (they used to start only after the C tile had been stored)
  constexpr int NPRE = (128 * (BN / 8)) / 256;
  constexpr int CPR = BN / 8;
  static_assert(16 % CPR == 0, "copy-out map: 8 or 16 column octets per tile row");
  const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
  auto unit_of = [&](int i, int& rl, int& ch) {                  // copy-out map of igemm_bf16_ws_kernel, 4 waves
    const int Q = (wave + 4 * i) * 16 + g4 * 4 + q4;
    ch = Q % CPR;
    rl = 4 * (Q / CPR) + t4;
  };
  uint4 pre_x[NPRE], pre_a[NPRE];
  unsigned pre_m[NPRE];                                          // ReLU mask bits of the chunk (bnred.mask) instead of pre_a
  if (bnred.x && bnred.prefetch) {
    const unsigned short* bx0 = reinterpret_cast<const unsigned short*>(bnred.x);
    const unsigned short* ba0 = reinterpret_cast<const unsigned short*>(bnred.act);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      int rl, ch;
      unit_of(i, rl, ch);
      const int rr = row0 + rl;
      pre_x[i] = make_uint4(0u, 0u, 0u, 0u);
      pre_a[i] = make_uint4(ET::ONE2, ET::ONE2, ET::ONE2, ET::ONE2);
      pre_m[i] = 0xffu;
      if (rr < g.M) {
        const int64_t off = (int64_t)rr * g.N + col0 + ch * 8;
        pre_x[i] = *reinterpret_cast<const uint4*>(bx0 + off);
        if (bnred.mask) pre_m[i] = bnred.mask[off >> 3];
        else if (ba0) pre_a[i] = *reinterpret_cast<const uint4*>(ba0 + off);
      }
    }
  }
  float s1v[TNW], s2v[TNW];
  if (g.epi_scale) {
    // eval-mode BatchNorm folded into the convolution: y = acc * scale[c] + shift[c] on the fp32 accumulators (a lane's
    // column is fixed per j), ReLU here unless the block's residual is still to be added in the copy-out
    const bool relu_now = g.epi_relu && !add_src;
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + l31;
      const float sc = g.epi_scale[col0 + cl], sh = g.epi_shift[col0 + cl];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wm * 64 + i * 32 + 8 * q + 4 * kh;
          float v0 = fmaf(acc[i][j][4 * q], sc, sh), v1 = fmaf(acc[i][j][4 * q + 1], sc, sh);
          float v2 = fmaf(acc[i][j][4 * q + 2], sc, sh), v3 = fmaf(acc[i][j][4 * q + 3], sc, sh);
          if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          *reinterpret_cast<uint2*>(&smem[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
      }
      s1v[j] = 0.f; s2v[j] = 0.f;
    }
  } else {
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    const int cl = wn * (BN / 2) + j * 32 + l31;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = wm * 64 + i * 32 + 8 * q + 4 * kh;
        const float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
        s1 += v0; s2 = fmaf(v0, v0, s2);
        s1 += v1; s2 = fmaf(v1, v1, s2);
        s1 += v2; s2 = fmaf(v2, v2, s2);
        s1 += v3; s2 = fmaf(v3, v3, s2);
        *reinterpret_cast<uint2*>(&smem[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
      }
    }
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    s1v[j] = s1; s2v[j] = s2;
  }
  }
  __syncthreads();
  constexpr int NIT = (128 * CPR) / 256;
  u32x2 trlo[NIT], trhi[NIT];
  {
    const int sq = lane & 3, sj = (lane >> 2) & 3;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int Qs = (wave + 4 * i) * 16 + g4 * 4 + sq;
      const unsigned addr = (unsigned)(uintptr_t)&smem[((Qs % CPR) * 8 + sj) * CPT + 4 * (Qs / CPR)];
      asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                   : "=&v"(trlo[i]), "=&v"(trhi[i]) : "v"(addr), "i"(4 * CPT * 2) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(trlo[i]), "+v"(trhi[i]));
  }
  constexpr int NRG = 256 / CPR;                               // row groups among the threads sharing a chunk column
  const unsigned short* bx = reinterpret_cast<const unsigned short*>(bnred.x);
  const unsigned short* bact = reinterpret_cast<const unsigned short*>(bnred.act);
  float rs1[8], rs2[8], rmu[8], ris[8];
  if (bx) {
    const int c0 = col0 + ((g4 * 4 + q4) % CPR) * 8;
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      const int64_t so = bnred.tiles_per_image ? (int64_t)(tile_m / bnred.tiles_per_image) * g.N : 0;   // per-image statistics (IBN)
      const float4 a = *reinterpret_cast<const float4*>(bnred.mean + so + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(bnred.invstd + so + c0 + k);
      rmu[k] = a.x; rmu[k + 1] = a.y; rmu[k + 2] = a.z; rmu[k + 3] = a.w;
      ris[k] = b.x; ris[k + 1] = b.y; ris[k + 2] = b.z; ris[k + 3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { rs1[k] = 0.f; rs2[k] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    int rl, ch;
    unit_of(i, rl, ch);
    const int rr = row0 + rl;
    if (rr < g.M) {
      uint4 v = make_uint4(trlo[i].x, trlo[i].y, trhi[i].x, trhi[i].y);
      const int64_t off = (int64_t)rr * g.N + col0 + ch * 8;
      if (add_src) {
        int64_t aoff = off;
        bool has = true;
        if (g.add_compact) {                                     // the stride-2 downsample branch's gradient, compact
          int ab, arem, ay, ax;
          fast_divmod(rr, g.OH * g.OW, g.inv_ohow, ab, arem);
          fast_divmod(arem, g.OW, g.inv_ow, ay, ax);
          has = ((ay | ax) & 1) == 0;
          aoff = (int64_t)((ab * (g.OH >> 1) + (ay >> 1)) * (g.OW >> 1) + (ax >> 1)) * g.N + col0 + ch * 8;
        }
        if (has) {
          const uint4 a = *reinterpret_cast<const uint4*>(add_src + aoff);
          const unsigned am = g.add_mask ? (unsigned)g.add_mask[aoff >> 3] : 0xffu;
          unsigned* vw = &v.x; const unsigned* aw = &a.x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float alo = ((am >> (2 * q)) & 1u) ? ET::lo(aw[q]) : 0.f;
            const float ahi = ((am >> (2 * q + 1)) & 1u) ? ET::hi(aw[q]) : 0.f;
            float lo = ET::lo(vw[q]) + alo;
            float hi = ET::hi(vw[q]) + ahi;
            if (g.epi_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }     // folded BatchNorm + residual: ReLU after the add
            vw[q] = ET::pack2(lo, hi);
          }
        }
      }
      *reinterpret_cast<uint4*>(out + off) = v;
      if (bx) {                                                // fused BN-backward column reduction
        uint4 xv = pre_x[i], av = pre_a[i];
        unsigned mb = pre_m[i];
        if (!bnred.prefetch) {
          xv = *reinterpret_cast<const uint4*>(bx + off);
          av = make_uint4(ET::ONE2, ET::ONE2, ET::ONE2, ET::ONE2);
          if (bnred.mask) mb = bnred.mask[off >> 3];
          else if (bact) av = *reinterpret_cast<const uint4*>(bact + off);
        }
        const unsigned* vw = &v.x; const unsigned* xw = &xv.x; const unsigned* aw = &av.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float g0 = ET::lo(vw[q]), g1 = ET::hi(vw[q]);
          const float a0 = ET::lo(aw[q]), a1 = ET::hi(aw[q]);
          const float x0 = ET::lo(xw[q]), x1 = ET::hi(xw[q]);
          g0 = (a0 > 0.f && ((mb >> (2 * q)) & 1u)) ? g0 : 0.f; g1 = (a1 > 0.f && ((mb >> (2 * q + 1)) & 1u)) ? g1 : 0.f;
          rs1[2 * q] += g0; rs1[2 * q + 1] += g1;
          rs2[2 * q] = fmaf(g0, (x0 - rmu[2 * q]) * ris[2 * q], rs2[2 * q]);
          rs2[2 * q + 1] = fmaf(g1, (x1 - rmu[2 * q + 1]) * ris[2 * q + 1], rs2[2 * q + 1]);
        }
      }
    }
  }
  if (bx) {
    __syncthreads();                                           // staged C tile no longer needed
    float* red2 = reinterpret_cast<float*>(smem);              // [NRG][2][BN]
    const int cidx = g4 * 4 + q4;
    const int rg = (wave * 4 + t4) * (16 / CPR) + cidx / CPR, cb = (cidx % CPR) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { red2[(rg * 2 + 0) * BN + cb + k] = rs1[k]; red2[(rg * 2 + 1) * BN + cb + k] = rs2[k]; }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, cl = i - which * BN;
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < NRG; ++q) a += red2[(q * 2 + which) * BN + cl];
      bnred.partial[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = a;
    }
  }
  if (bn_part) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + l31;
      if (kh == 0) { red[(wm * 2 + 0) * BN + cl] = s1v[j]; red[(wm * 2 + 1) * BN + cl] = s2v[j]; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, cl = i - which * BN;
      bn_part[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
    }
  }
}

// ------------------------------------------------------------------------------------ bf16, warp-specialised
// Measured on MI355X (r01, tools/probes/lds_fill_bw_probe.hip + loop ablations): in igemm_bf16_dma_kernel a
// k-tile costs ~0.5 us of DMA *issue* time (each global_load_lds_dwordx4 holds the issuing wave for 60-180
// cycles) plus ~0.4 us of ds_read + MFMA time, and because the same four waves do both, the two mostly add up.
// Here a 512-thread workgroup splits the roles: waves 0-3 (consumers, 2x2 over the 128 x BN tile) only read
// fragments and issue MFMAs; waves 4-7 (producers) only compute gather addresses and issue the LDS-DMA pieces
// into an NS-deep LDS ring.  One s_barrier per k-tile couples them:
//   barrier B_t  <=  producers: their pieces of tile t have landed (vmcnt);  consumers: done reading tile t-1
//   after B_t    :   producers issue tile t+NS-1 into slot (t-1)%NS, consumers multiply tile t.
// Tiling, swizzle, zero page and the epilogue maths are those of igemm_bf16_dma_kernel (512 threads copy out).
// BM = 128: the consumer waves hold 64 x BN/2 sub-tiles (two workgroups per CU where the ring allows it; THREE for the
// 128 x 64 tile with a 2-deep ring: 48 KB of LDS, and the register allocator is held to 80 VGPRs (it needs 76) -- measured
// -0.06 ms per step on the rule-selected schedule, round 3).  BM = 256: 128 x BN/2
// sub-tiles -- 0.75 instead of 1 fragment read per MFMA and 1.125 instead of 1.5 KB of LDS traffic per MFMA; one workgroup
// per CU (the ring is 48 KB per stage), so only for grids that still fill the chip with 256-row tiles.
template <int BN, int NS, int BM = 128, typename ET = Bf16T>
__global__ __launch_bounds__(512, (BN == 64 && NS == 2 && BM == 128) ? 6 : ((NS * (BM + BN) * 128 <= 80 * 1024) ? 2 : 1)) void igemm_bf16_ws_kernel(IGemmGeom g, const unsigned short* __restrict__ src,
                                                                const unsigned short* __restrict__ wgt,
                                                                unsigned short* __restrict__ out,
                                                                const unsigned short* __restrict__ add_src,
                                                                float* __restrict__ bn_part, int tiles_n,
                                                                BnRedArgs bnred, WRedJob wred) {
  constexpr int NT = 512;
  constexpr int BK = 64, TNW = BN / 64, NBI = BN / 32;
  constexpr int MI = BM / 64, NAI = BM / 32;                     // 32-row blocks per consumer wave; A-tile DMA instructions per producer wave
  constexpr int CPT = BM + 4;                                    // transposed staging: [BN columns][BM rows + 4]
  constexpr int TILE_A = BM * BK, TILE_B = BN * BK, STAGE = TILE_A + TILE_B;
  constexpr int CPR = BN / 8, NRG = NT / CPR;
  constexpr int RED_ELEMS = NRG * 2 * BN * 2;                    // fp32 reduction scratch, in 2-byte units
  constexpr int LDS0 = (NS * STAGE) > (BN * CPT) ? (NS * STAGE) : (BN * CPT);
  constexpr int LDS_ELEMS = LDS0 > RED_ELEMS ? LDS0 : RED_ELEMS;
  constexpr int LPT = NAI + NBI;
  static_assert(NS >= 2 && (NS - 2) * LPT <= 63, "ring depth");
  static_assert(BM == 128 || BM == 256, "row tile");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[LDS_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // piggybacked job: the LAST wred.nblocks workgroups sum the split partials of the PREVIOUS weight-gradient launch
  // instead of computing a tile.  Workgroups are dispatched in order, so these start when the final wave of tiles
  // is running and fill the CUs that wave leaves idle (placed first they held every workgroup slot for ~10 us and
  // delayed the tiles: measured, no gain over the stand-alone launch).
  const int nred = wred.ws ? wred.nblocks : 0;
  const int ntile_wgs = (int)gridDim.x - nred;
  if ((int)blockIdx.x >= ntile_wgs) {
    wgrad_reduce_block<NT>(wred, (int)blockIdx.x - ntile_wgs, reinterpret_cast<float*>(smem));
    return;
  }
  const bool producer = wave >= 4;
  const int cw = wave & 3;                                       // role-local wave index
  const int wm = cw >> 1, wn = cw & 1;
  const int bid = xcd_remap((int)blockIdx.x, ntile_wgs);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int l31 = lane & 31, kh = lane >> 5;
  // parity-class row order (g.parity): the tile's class fixes which taps exist; r, s step by 2 from (r0, s0)
  const int msub = g.M >> 2;
  const int cls = g.parity ? row0 / msub : 0, py = cls >> 1, px = cls & 1;
  const int tstep = g.parity ? 2 : 1;
  const int r0 = g.parity ? ((py + g.pad) & 1) : 0, s0 = g.parity ? ((px + g.pad) & 1) : 0;
  const int kh_taps = (g.K >> g.log2span) / g.kw;                 // kernel rows
  const int nk = g.parity ? ((kh_taps - r0 + 1) / 2) * ((g.kw - s0 + 1) / 2) * ((1 << g.log2span) / BK) : g.K / BK;
  // GEMM row -> raster pixel index of the output tensor (identity unless parity-class order)
  auto pixel_of = [&](int rr) -> int {
    if (!g.parity) return rr;
    const int rem2 = rr - cls * msub;
    int b, rem, a, c;
    fast_divmod(rem2, (g.OH >> 1) * (g.OW >> 1), g.inv_ohow * 4.f, b, rem);
    fast_divmod(rem, g.OW >> 1, g.inv_ow * 2.f, a, c);
    return (b * g.OH + 2 * a + py) * g.OW + 2 * c + px;
  };

  f32x16 acc[MI][TNW];
  if (producer) {
    const int span_mask = (1 << g.log2span) - 1;
    const int lr8 = lane >> 3, lcp = lane & 7;
    int oy[NAI], ox[NAI], bpix[NAI], gch[NAI];
    bool vm[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int r = (i * 4 + cw) * 8 + lr8, m = row0 + r;
      vm[i] = m < g.M;
      const int mm = vm[i] ? m : row0;
      const int pix = pixel_of(mm);
      int b, rem;
      fast_divmod(pix, g.OH * g.OW, g.inv_ohow, b, rem);
      fast_divmod(rem, g.OW, g.inv_ow, oy[i], ox[i]);
      bpix[i] = b * g.SH * g.SW;
      gch[i] = (lcp ^ ((r >> 1) & 7)) << 3;
    }
    const unsigned short* wp[NBI];
    const unsigned short* wbase[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int r = (i * 4 + cw) * 8 + lr8;
      wbase[i] = wgt + (int64_t)(col0 + r) * g.K + ((lcp ^ ((r >> 1) & 7)) << 3);
      wp[i] = wbase[i];
    }
    const unsigned short* aptr[NAI];
    int amul[NAI];
    int cur_tap = -1, cur_r = r0, cur_s = s0 - tstep;
    const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_zero_page);
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto issue = [&](int t, int buf) {
      if (CREID_ABL_ON(g.abl, 2) && t > 0) return;
      const int tap = (t * BK) >> g.log2span;
      if (tap != cur_tap) {
        cur_tap = tap;
        cur_s += tstep;
        if (cur_s >= g.kw) { cur_s = s0; cur_r += tstep; }
        if (g.parity) {
#pragma unroll
          for (int i = 0; i < NBI; ++i) wp[i] = wbase[i] + ((cur_r * g.kw + cur_s) << g.log2span);
        }
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          int iy, ix;
          const bool ok = vm[i] && igemm_src_pixel(g, oy[i], ox[i], cur_r, cur_s, iy, ix);
          aptr[i] = ok ? src + (int64_t)(bpix[i] + iy * g.SW + ix) * g.pitch + gch[i] : zpage;
          amul[i] = ok ? 1 : 0;
        }
      }
      unsigned short* la = smem + buf * STAGE + cw * 512;
      unsigned short* lb = smem + buf * STAGE + TILE_A + cw * 512;
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)aptr[i], (lptr_t)(la + i * 2048), 16, 0, 0);
        aptr[i] += amul[i] ? BK : 0;                             // running pointers (see igemm_bf16_dma_kernel)
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)wp[i], (lptr_t)(lb + i * 2048), 16, 0, 0);
        wp[i] += BK;
      }
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < nk) issue(p, p);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      const int younger = min(nk - 1 - t, NS - 2);
      if constexpr (NS >= 5) { if (younger == 3) wait_vm<3 * LPT>(); }
      if constexpr (NS >= 4) { if (younger == 2) wait_vm<2 * LPT>(); }
      if constexpr (NS >= 3) { if (younger == 1) wait_vm<1 * LPT>(); }
      if (younger == 0) wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");                    // B_t
      if (t + NS - 1 < nk) issue(t + NS - 1, buf == 0 ? NS - 1 : buf - 1);
      buf = (buf + 1 == NS) ? 0 : buf + 1;
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_barrier" ::: "memory");                    // B_t: tile t is in LDS
      const unsigned short* As = smem + buf * STAGE;
      const unsigned short* Bs = As + TILE_A;
      buf = (buf + 1 == NS) ? 0 : buf + 1;
      // fragment reads run one 16-wide k slice ahead of the MFMAs (register double buffer): the LDS latency of
      // slice kk+1 hides behind the MFMAs of slice kk instead of being exposed four times per k-tile
      s16x8 a[2][MI], b[2][TNW];
      auto load_frags = [&](int kk, int sl) {
        const int ch = 2 * kk + kh;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int r = wm * (BM / 2) + i * 32 + l31;
          a[sl][i] = *reinterpret_cast<const s16x8*>(&As[r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int c = wn * (BN / 2) + j * 32 + l31;
          b[sl][j] = *reinterpret_cast<const s16x8*>(&Bs[c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
        }
      };
      auto mma = [&](int sl) {
        if (CREID_ABL_ON(g.abl, 1)) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j)
            acc[i][j] = ET::mfma(a[sl][i], b[sl][j], acc[i][j]);
      };
      if (CREID_ABL_ON(g.abl, 4)) continue;
      load_frags(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(1, 1); mma(0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(2, 0); mma(1);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(3, 1); mma(0);
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
    }
  }
  __syncthreads();

  // ---- epilogue: consumers stage the C tile (bf16) in LDS, all 512 threads copy it out
  // fused BN-backward reduction: its two operand tiles (x, act) are fetched NOW, so the loads fly while the
  // accumulators are staged through LDS (they used to start only after the C tile had been stored)
  // copy-out map (see below): a 16-lane group owns 4 rows x 4 column octets; lane = 16*g4 + 4*q4 + t4 stores row t4 of the
  // row quad, octet g4*4 + q4 of the wave's 16-octet strip
  constexpr int NPRE = BM == 128 ? (128 * (BN / 8)) / 512 : 1;   // (the fused BN reduction exists for 128-row tiles only)
  const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
  auto unit_of = [&](int i, int& rl, int& ch) {
    const int Q = (wave + 8 * i) * 16 + g4 * 4 + q4;
    ch = Q % CPR;
    rl = 4 * (Q / CPR) + t4;
  };
  uint4 pre_x[NPRE], pre_a[NPRE];
  unsigned pre_m[NPRE];                                          // ReLU mask bits of the chunk (bnred.mask) instead of pre_a
  if (BM == 128 && bnred.x && bnred.prefetch) {
    const unsigned short* bx0 = reinterpret_cast<const unsigned short*>(bnred.x);
    const unsigned short* ba0 = reinterpret_cast<const unsigned short*>(bnred.act);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      int rl, ch;
      unit_of(i, rl, ch);
      const int rr = row0 + rl;
      pre_x[i] = make_uint4(0u, 0u, 0u, 0u);
      pre_a[i] = make_uint4(ET::ONE2, ET::ONE2, ET::ONE2, ET::ONE2);
      pre_m[i] = 0xffu;
      if (rr < g.M) {
        const int64_t off = (int64_t)pixel_of(rr) * g.N + col0 + ch * 8;
        pre_x[i] = *reinterpret_cast<const uint4*>(bx0 + off);
        if (bnred.mask) pre_m[i] = bnred.mask[off >> 3];
        else if (ba0) pre_a[i] = *reinterpret_cast<const uint4*>(ba0 + off);
      }
    }
  }
  // The C tile is staged COLUMN-major: a lane's four consecutive accumulator rows of one column are one packed 8-byte
  // LDS store (2 v_cvt_pk_bf16_f32 + 1 ds_write_b64 per 4 values; the row-major image took a 2-byte store per value),
  // and the copy-out gets its row-major 16-byte chunks back through the transposing LDS read.  Column pitch 264 B:
  // 16 consecutive columns start 2 banks apart (conflict-free b64 stores), the 4 x 4 units of a transposing read too.
  float s1v[TNW], s2v[TNW];
  if (!producer && g.epi_scale) {
    // eval-mode BatchNorm folded into the convolution (see igemm_bf16_dma_kernel)
    const bool relu_now = g.epi_relu && !add_src;
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + l31;
      const float sc = g.epi_scale[col0 + cl], sh = g.epi_shift[col0 + cl];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wm * (BM / 2) + i * 32 + 8 * q + 4 * kh;
          float v0 = fmaf(acc[i][j][4 * q], sc, sh), v1 = fmaf(acc[i][j][4 * q + 1], sc, sh);
          float v2 = fmaf(acc[i][j][4 * q + 2], sc, sh), v3 = fmaf(acc[i][j][4 * q + 3], sc, sh);
          if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          *reinterpret_cast<uint2*>(&smem[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
      }
      s1v[j] = 0.f; s2v[j] = 0.f;
    }
  } else if (!producer) {
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + l31;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wm * (BM / 2) + i * 32 + 8 * q + 4 * kh;
          const float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
          s1 += v0; s2 = fmaf(v0, v0, s2);
          s1 += v1; s2 = fmaf(v1, v1, s2);
          s1 += v2; s2 = fmaf(v2, v2, s2);
          s1 += v3; s2 = fmaf(v3, v3, s2);
          *reinterpret_cast<uint2*>(&smem[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
      }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      s1v[j] = s1; s2v[j] = s2;
    }
  }
  __syncthreads();
  // transposing reads: in a 16-lane group lane s supplies the 8-byte unit (column 8 * octet(s & 3) + (s >> 2), the row quad)
  // and lane l receives (row l & 3, columns 8 * octet(l >> 2) + 0..3); a second read 4 columns on completes the 16-byte chunk.
  // Every lane takes part (the data crosses lanes), only the global store is predicated.
  constexpr int NIT = (BM * CPR) / NT;
  static_assert(16 % CPR == 0, "copy-out map: 8 or 16 column octets per tile row");
  u32x2 trlo[NIT], trhi[NIT];
  {
    const int sq = lane & 3, sj = (lane >> 2) & 3;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int Qs = (wave + 8 * i) * 16 + g4 * 4 + sq;
      const unsigned addr = (unsigned)(uintptr_t)&smem[((Qs % CPR) * 8 + sj) * CPT + 4 * (Qs / CPR)];
      asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                   : "=&v"(trlo[i]), "=&v"(trhi[i]) : "v"(addr), "i"(4 * CPT * 2) : "memory");
    }
  }
  const unsigned short* bx = BM == 128 ? reinterpret_cast<const unsigned short*>(bnred.x) : nullptr;
  const unsigned short* bact = reinterpret_cast<const unsigned short*>(bnred.act);
  float rs1[8], rs2[8], rmu[8], ris[8];
  if (bx) {
    const int c0 = col0 + ((g4 * 4 + q4) % CPR) * 8;            // this thread's column octet is the same in every pass
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      const int64_t so = bnred.tiles_per_image ? (int64_t)(tile_m / bnred.tiles_per_image) * g.N : 0;   // per-image statistics (IBN)
      const float4 a = *reinterpret_cast<const float4*>(bnred.mean + so + c0 + k);
      const float4 b = *reinterpret_cast<const float4*>(bnred.invstd + so + c0 + k);
      rmu[k] = a.x; rmu[k + 1] = a.y; rmu[k + 2] = a.z; rmu[k + 3] = a.w;
      ris[k] = b.x; ris[k + 1] = b.y; ris[k + 2] = b.z; ris[k + 3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { rs1[k] = 0.f; rs2[k] = 0.f; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(trlo[i]), "+v"(trhi[i]));
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    int rl, ch;
    unit_of(i, rl, ch);
    const int rr = row0 + rl;
    if (rr < g.M) {
      uint4 v = make_uint4(trlo[i].x, trlo[i].y, trhi[i].x, trhi[i].y);
      const int prow = pixel_of(rr);
      const int64_t off = (int64_t)prow * g.N + col0 + ch * 8;
      if (add_src) {
        int64_t aoff = off;
        bool has = true;
        if (g.add_compact) {                                     // the stride-2 downsample branch's gradient, compact
          int ab, arem, ay, ax;
          fast_divmod(prow, g.OH * g.OW, g.inv_ohow, ab, arem);
          fast_divmod(arem, g.OW, g.inv_ow, ay, ax);
          has = ((ay | ax) & 1) == 0;
          aoff = (int64_t)((ab * (g.OH >> 1) + (ay >> 1)) * (g.OW >> 1) + (ax >> 1)) * g.N + col0 + ch * 8;
        }
        if (has) {
          const uint4 a = *reinterpret_cast<const uint4*>(add_src + aoff);
          const unsigned am = g.add_mask ? (unsigned)g.add_mask[aoff >> 3] : 0xffu;
          unsigned* vw = &v.x; const unsigned* aw = &a.x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float alo = ((am >> (2 * q)) & 1u) ? ET::lo(aw[q]) : 0.f;
            const float ahi = ((am >> (2 * q + 1)) & 1u) ? ET::hi(aw[q]) : 0.f;
            float lo = ET::lo(vw[q]) + alo;
            float hi = ET::hi(vw[q]) + ahi;
            if (g.epi_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }     // folded BatchNorm + residual: ReLU after the add
            vw[q] = ET::pack2(lo, hi);
          }
        }
      }
      if (!CREID_ABL_ON(g.abl, 8)) *reinterpret_cast<uint4*>(out + off) = v;
      if (bx) {
        uint4 xv = pre_x[i % NPRE], av = pre_a[i % NPRE];        // (NPRE == NIT whenever bx can be non-null)
        unsigned mb = pre_m[i % NPRE];
        if (!bnred.prefetch) {
          xv = *reinterpret_cast<const uint4*>(bx + off);
          av = make_uint4(ET::ONE2, ET::ONE2, ET::ONE2, ET::ONE2);
          if (bnred.mask) mb = bnred.mask[off >> 3];
          else if (bact) av = *reinterpret_cast<const uint4*>(bact + off);
        }
        const unsigned* vw = &v.x; const unsigned* xw = &xv.x; const unsigned* aw = &av.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float g0 = ET::lo(vw[q]), g1 = ET::hi(vw[q]);
          const float a0 = ET::lo(aw[q]), a1 = ET::hi(aw[q]);
          const float x0 = ET::lo(xw[q]), x1 = ET::hi(xw[q]);
          g0 = (a0 > 0.f && ((mb >> (2 * q)) & 1u)) ? g0 : 0.f; g1 = (a1 > 0.f && ((mb >> (2 * q + 1)) & 1u)) ? g1 : 0.f;
          rs1[2 * q] += g0; rs1[2 * q + 1] += g1;
          rs2[2 * q] = fmaf(g0, (x0 - rmu[2 * q]) * ris[2 * q], rs2[2 * q]);
          rs2[2 * q + 1] = fmaf(g1, (x1 - rmu[2 * q + 1]) * ris[2 * q + 1], rs2[2 * q + 1]);
        }
      }
    }
  }
  if (bx) {
    __syncthreads();
    float* red2 = reinterpret_cast<float*>(smem);                // [NRG][2][BN]
    const int cidx = g4 * 4 + q4;
    const int rg = (wave * 4 + t4) * (16 / CPR) + cidx / CPR, cb = (cidx % CPR) * 8;   // threads that share an octet
#pragma unroll
    for (int k = 0; k < 8; ++k) { red2[(rg * 2 + 0) * BN + cb + k] = rs1[k]; red2[(rg * 2 + 1) * BN + cb + k] = rs2[k]; }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, cl = i - which * BN;
      float a = 0.f;
#pragma unroll 8
      for (int q = 0; q < NRG; ++q) a += red2[(q * 2 + which) * BN + cl];
      bnred.partial[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = a;
    }
  }
  if (bn_part) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (!producer) {
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int cl = wn * (BN / 2) + j * 32 + l31;
        if (kh == 0) { red[(wm * 2 + 0) * BN + cl] = s1v[j]; red[(wm * 2 + 1) * BN + cl] = s2v[j]; }
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, cl = i - which * BN;
      if constexpr (BM == 128) {
        bn_part[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
      } else {                                                   // partial rows stay per 128 rows: one per consumer-wave row half
        bn_part[((int64_t)(tile_m * 2 + 0) * 2 + which) * g.N + col0 + cl] = red[(0 * 2 + which) * BN + cl];
        if (row0 + 128 < g.M) bn_part[((int64_t)(tile_m * 2 + 1) * 2 + which) * g.N + col0 + cl] = red[(1 * 2 + which) * BN + cl];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ f32
template <int BN>
__global__ __launch_bounds__(256) void igemm_f32_kernel(IGemmGeom g, const float* __restrict__ src,
                                                        const float* __restrict__ wgt, float* __restrict__ out,
                                                        const float* __restrict__ add_src,
                                                        float* __restrict__ bn_part, int tiles_n) {
  // LDS operand image of a 16-deep k-tile (the same as dist.hip's fp32 distance kernel): [kh = k & 1][half = k >> 3][row][4] -- the
  // per-lane MFMA operand is A[i = lane & 31][k = 2 step + (lane >> 5)], so four consecutive steps' values are one 16-byte unit
  // (ds_read_b128) and a staged float4 is two 8-byte writes ((x, z) -> kh 0, (y, w) -> kh 1).  The k order of the accumulation is
  // the sequential one of rounds 1-5: results unchanged.
  constexpr int BK = 16, PA = 128 * 4 + 16, PB = BN * 4 + 16, TNW = BN / 64, NB = BN / 64;
  __shared__ __attribute__((aligned(16))) float As[2][2][2][PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][2][2][PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * 128, col0 = tile_n * BN;
  const int span_mask = (1 << g.log2span) - 1;

  const int lrow = tid >> 2, lkc = tid & 3;
  int oy[2], ox[2], bpix[2];
  bool vm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = row0 + lrow + 64 * i;
    vm[i] = m < g.M;
    const int mm = vm[i] ? m : 0;
    const int b = mm / (g.OH * g.OW), rem = mm - b * (g.OH * g.OW);
    oy[i] = rem / g.OW; ox[i] = rem - oy[i] * g.OW;
    bpix[i] = b * g.SH * g.SW;
  }
  const float* wp[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) wp[i] = wgt + (int64_t)(col0 + lrow + 64 * i) * g.K + 4 * lkc;
  float4 ra[2], rb[NB];
  unsigned amask[2] = {0u, 0u};                    // all ones where the staged tap is a real pixel (else the tile gets zeros)
  auto gload = [&](int t) {                        // branch-free: an absent tap reads the tensor's first element and is masked in lstore
    const int kk = t * BK + 4 * lkc;
    const int tap = kk >> g.log2span, c = kk & span_mask;
    const int r = tap / g.kw, s = tap - r * g.kw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int iy, ix;
      const bool ok = vm[i] && igemm_src_pixel(g, oy[i], ox[i], r, s, iy, ix);
      amask[i] = ok ? 0xffffffffu : 0u;
      const int64_t off = ok ? (int64_t)(bpix[i] + iy * g.SW + ix) * g.pitch + c : 0;
      ra[i] = *reinterpret_cast<const float4*>(src + off);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4*>(wp[i] + t * BK);
  };
  const int sh = lkc >> 1, so = lrow * 4 + 2 * (lkc & 1);          // global k = 4 lkc + {0..3}: steps 2 lkc, 2 lkc + 1
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & amask[i]); };
      *reinterpret_cast<float2*>(&As[buf][0][sh][so + 256 * i]) = make_float2(mk(ra[i].x), mk(ra[i].z));
      *reinterpret_cast<float2*>(&As[buf][1][sh][so + 256 * i]) = make_float2(mk(ra[i].y), mk(ra[i].w));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      *reinterpret_cast<float2*>(&Bs[buf][0][sh][so + 256 * i]) = make_float2(rb[i].x, rb[i].z);
      *reinterpret_cast<float2*>(&Bs[buf][1][sh][so + 256 * i]) = make_float2(rb[i].y, rb[i].w);
    }
  };
  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = g.K / BK;
  const int l31 = lane & 31, kh = lane >> 5;
  const int fa = (wm * 64 + l31) * 4, fb = (wn * (BN / 2) + l31) * 4;
  // Two-phase k-loop (round 6, from the evaluation's distance kernels): 8 TNW MFMAs (64 cycles each) per phase with everything
  // else issued in their shadow -- phase 1: the second half's fragment reads + the LDS writes of the next k-tile, barrier,
  // phase 2: the global loads of the k-tile after that + the next k-tile's first-half fragment reads.  (Before: reads, MFMAs,
  // writes, barrier in sequence per k-tile; the two waves a SIMD holds fell into step and the matrix pipe idled with them.)
  float4 a0[2], b0[TNW], a1[2], b1[TNW];
  auto frag = [&](int buf, int half, float4 (&a)[2], float4 (&b)[TNW]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(&As[buf][kh][half][fa + 128 * i]);
#pragma unroll
    for (int j = 0; j < TNW; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[buf][kh][half][fb + 128 * j]);
  };
  auto mma = [&](const float4 (&a)[2], const float4 (&b)[TNW]) {   // steps in k order: 4 half + s4 multiplies k = 2 step + kh
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float av[2], bv[TNW];
#pragma unroll
      for (int i = 0; i < 2; ++i) { const float c4[4] = {a[i].x, a[i].y, a[i].z, a[i].w}; av[i] = c4[s4]; }
#pragma unroll
      for (int j = 0; j < TNW; ++j) { const float c4[4] = {b[j].x, b[j].y, b[j].z, b[j].w}; bv[j] = c4[s4]; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  gload(nk > 1 ? 1 : 0);
  frag(0, 0, a0, b0);
  for (int t = 0; t + 1 < nk; ++t) {
    const int buf = t & 1;
    __builtin_amdgcn_sched_barrier(0);
    frag(buf, 1, a1, b1);
    lstore(buf ^ 1);
    mma(a0, b0);
#pragma unroll
    for (int i = 0; i < 2 + TNW; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
    for (int i = 0; i < 2 + NB; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 2, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * TNW - (2 + TNW) - (2 + NB), 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    gload(t + 2 < nk ? t + 2 : nk - 1);
    frag(buf ^ 1, 0, a0, b0);
    mma(a1, b1);
#pragma unroll
    for (int i = 0; i < 2 + NB; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x020, 1, 1); }
#pragma unroll
    for (int i = 0; i < 2 + TNW; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * TNW - (2 + NB) - (2 + TNW), 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  frag((nk - 1) & 1, 1, a1, b1);                   // last k-tile: nothing left to stage
  mma(a0, b0);
  mma(a1, b1);
  __syncthreads();                                 // the reduction scratch below aliases the operand image
  float* red = &As[0][0][0][0];
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    const int cl = wn * (BN / 2) + j * 32 + l31;
    const int c = col0 + cl;
    float s1 = 0.f, s2 = 0.f;
    const float esc = g.epi_scale ? g.epi_scale[c] : 1.f, esh = g.epi_scale ? g.epi_shift[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (rr < g.M) {
          float v = acc[i][j][r];
          if (g.epi_scale) v = fmaf(v, esc, esh);                   // folded eval-mode BatchNorm: bn2d_apply_kernel's arithmetic
          if (add_src) v += add_src[(int64_t)rr * g.N + c];
          if (g.epi_relu) v = fmaxf(v, 0.f);
          s1 += v; s2 = fmaf(v, v, s2);
          out[(int64_t)rr * g.N + c] = v;
        }
      }
    }
    if (bn_part) {
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (kh == 0) { red[(wm * 2 + 0) * BN + cl] = s1; red[(wm * 2 + 1) * BN + cl] = s2; }
    }
  }
  if (bn_part) {
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, cl = i - which * BN;
      bn_part[((int64_t)tile_m * 2 + which) * g.N + col0 + cl] = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
    }
  }
}

// ------------------------------------------------------------------------------------ host
static int ilog2_exact(int64_t v) {
  int l = 0;
  while ((1LL << l) < v) ++l;
  return ((1LL << l) == v) ? l : -1;
}

static int ws_stages_env() {
  static const int v = [] { const char* e = getenv("CREID_IGEMM_WS_STAGES"); int x = e ? atoi(e) : 0; return (x == 2 || x == 4) ? x : 3; }();
  return v;
}

static int launch_igemm(const IGemmGeom& g_in, const void* src, const void* wgt, void* out, const void* add_src,
                        float* bn_part, int dtype, hipStream_t s, BnRedArgs bnred = BnRedArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0},
                        const WRedJob* wred_in = nullptr) {
#ifdef CREID_ABL_BUILD
  const char* abl_e = CREID_KNOB_ENV("CREID_IGEMM_ABL");                  // per call: the probes sweep it inside one process
  const int abl_env = abl_e ? atoi(abl_e) : 0;
  { static int warned = -1; if (abl_env && abl_env != warned) { warned = abl_env; creid_ablation_env("CREID_IGEMM_ABL"); } }
#else
  const int abl_env = 0;
#endif
  IGemmGeom g = g_in;
  g.abl = abl_env;
  WRedJob wred{};
  if (wred_in) wred = *wred_in;
  static_assert(sizeof(float) * (4 * 512 + 8192) <= 2 * (128 + 64) * 64 * 2, "reduce scratch must fit the smallest LDS ring");
  const int tiles_m = (g.M + 127) / 128;
  // pick the N tile: 128 unless that leaves the chip (256 CUs) under-filled or N is only 64
  int bn = 128;
  static const int min_wgs = [] { const char* e = getenv("CREID_IGEMM_BN128_MIN_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 384; }();
  if (g.N % 128 != 0 || (int64_t)tiles_m * (g.N / 128) < min_wgs) bn = 64;
  // measured plan for this GEMM shape (tune.hpp): N tile and ring depth of the producer/consumer kernel; the 4th key slot is
  // transposed | stride << 1 (a stride-2 and a stride-1 3x3 layer can share M, N, K but not their gather pattern)
  TunePlan tp;
  int tuned_stages = 0, tuned_dma = 0, tuned_stream = 0, tuned_bm256 = 0, tuned_stream2 = 0, tuned_pp = -1;
  // (folded eval-mode launches -- epi_scale set -- look for a plan measured with THAT epilogue first: key bit 3; plans recorded
  // before the bit existed carry no mode and serve both)
  const int plan_d = g.transposed | (g.stride << 1);
  const bool have_plan = creid_is16(dtype) && ((g.epi_scale && creid_tune_lookup(CREID_TUNE_IGEMM, g.M, g.N, g.K, plan_d | 8, tp)) ||
                                                  creid_tune_lookup(CREID_TUNE_IGEMM, g.M, g.N, g.K, plan_d, tp));
  if (have_plan && dtype == CREID_F16 && tp.p2 == 2) {
    // (the first persistent 1x1 kernel of conv_stream.hip is bf16 only: f16 launches of those shapes take the built-in rule)
  } else if (have_plan && tp.p2 == 5) {
    tuned_pp = tp.p0;                                              // plan kind 5: all-waves-multiply persistent kernel, p0 = its variant word
  } else if (have_plan && (tp.p0 == 64 || tp.p0 == 128) && g.N % tp.p0 == 0 && (tp.p1 == 2 || tp.p1 == 3 || tp.p1 == 4)) {
    bn = tp.p0; tuned_stages = tp.p1; tuned_dma = tp.p2 == 1; tuned_stream = tp.p2 == 2; tuned_bm256 = tp.p2 == 3; tuned_stream2 = tp.p2 == 4;
  }
  // layer1's 3 x 3, 64 -> 64 forward: halo tile in LDS, weights in registers (conv_stream.hip conv3x3_c64_kernel); CREID_C64_3X3=0:
  // the tile kernels
  {
    const char* ce = CREID_KNOB_ENV("CREID_C64_3X3");                       // read per call: tests toggle it
    const bool c64_on = !ce || atoi(ce) != 0;
    if (c64_on && creid_is16(dtype) && !g.transposed && g.N == 64 && g.K == 576 && g.log2span == 6 && g.kw == 3 && g.stride == 1 &&
        g.pad == 1 && g.pitch == 64 && g.check_bounds && !add_src && !bnred.x && !wred.ws && g.SH == g.OH && g.SW == g.OW &&
        !(bn_part && g.epi_scale)) {
      const int rc = launch_conv3x3_c64(g.M, g.OH, g.OW, src, wgt, out, bn_part, g.epi_scale, g.epi_shift, g.epi_relu, dtype, s);
      if (rc != CREID_E_SHAPE) return rc;
    }
  }
  // all-waves-multiply persistent kernel (conv_pipe.hip): plan kind 5, or CREID_IGEMM_PP = 0x1000 | variant for every launch it covers
  {
    const char* pe = CREID_KNOB_ENV("CREID_IGEMM_PP");                     // read per call: tests and the tuner toggle it
    const int force_pp = pe ? (int)strtol(pe, nullptr, 0) : 0;
    const bool pp_off = pe && force_pp == 0;                              // CREID_IGEMM_PP=0: never, not even where a plan selects it
    if (creid_is16(dtype) && !bnred.x && !wred.ws && g.log2span >= 6 && !pp_off && (force_pp || tuned_pp >= 0)) {
      const int rc = launch_igemm_pp(g, src, wgt, out, add_src, bn_part, force_pp ? (force_pp & 0xfff) : tuned_pp, dtype, s);
      if (rc != CREID_E_SHAPE) return rc;
    }
  }
  // persistent streaming kernel for the small-K 1x1 stride-1 forward convolutions (conv_stream.hip): plan kind 2, or
  // CREID_STREAM1X1=1 for every GEMM it covers
  {
    const char* fe = CREID_KNOB_ENV("CREID_STREAM1X1");                    // read per call: tests toggle it
    const int force_stream = fe ? atoi(fe) : 0;
    const bool plain_1x1 = dtype == CREID_BF16 && !g.transposed && g.K == (1 << g.log2span) && g.stride == 1 && g.pad == 0 &&
                           g.kw == 1 && g.check_bounds && !add_src && !bnred.x && !wred.ws && g.pitch == g.K && !g.epi_scale;
    if (plain_1x1 && (force_stream || tuned_stream)) {
      const int rc = launch_stream1x1(g.M, g.K, g.N, src, wgt, out, bn_part, s);
      if (rc != CREID_E_SHAPE) return rc;
    }
    // second form (plan kind 4 / CREID_STREAM2=1): also the folded eval-mode epilogue with the block's residual
    const char* f2 = CREID_KNOB_ENV("CREID_STREAM2");
    const int force2 = f2 ? atoi(f2) : 0;
    const bool fwd_1x1 = creid_is16(dtype) && !g.transposed && g.K == (1 << g.log2span) && g.stride == 1 && g.pad == 0 &&
                         g.kw == 1 && g.check_bounds && !bnred.x && !wred.ws && g.pitch == g.K && !g.add_compact && !g.add_mask &&
                         !(bn_part && (g.epi_scale || add_src));
    if (fwd_1x1 && (force2 || tuned_stream2)) {
      // (plan: p1 = 2 -> widest column slab the LDS allows, 3 -> at most 128 columns, 4 -> 64)
      const int cap = tuned_stream2 ? (tuned_stages == 3 ? 128 : (tuned_stages == 4 ? 64 : 0)) : 0;
      const int rc = launch_stream2(g.M, g.K, g.N, src, wgt, out, bn_part, add_src, g.epi_scale, g.epi_shift, g.epi_relu, cap, dtype, s);
      if (rc != CREID_E_SHAPE) return rc;
    }
  }
  if (g.N % bn != 0) return CREID_E_SHAPE;
  const int tiles_n = g.N / bn;
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(256);
  static const int use_dma = [] { const char* e = getenv("CREID_IGEMM_DMA"); return e ? atoi(e) : 1; }();
  const bool stem_geom = g.log2span == 5 && !g.transposed && !g.check_bounds && g.kw == 1 && g.stride == 2 && g.pad == 0;
  static const int stem_dma = [] { const char* e = getenv("CREID_STEM_DMA"); return e ? atoi(e) : 1; }();
  if (creid_is16(dtype) && use_dma && (g.log2span >= 6 || (stem_geom && stem_dma))) {
    if (g.K % 64 != 0) return CREID_E_SHAPE;
    static const int use_ws = [] { const char* e = getenv("CREID_IGEMM_WS"); return e ? atoi(e) : 1; }();

    // Producer / consumer split everywhere (single-stream step, r01: +4 % from the long-k 64-wide tiles with a
    // 3-deep ring, +2.4 % more from 2-deep rings -- two workgroups per CU -- on all other tiles; standalone sweeps in
    // profiles/r01_igemm_ws_sweep.md).  CREID_IGEMM_WS=0: the 4-wave DMA kernel; =2: force CREID_IGEMM_WS_STAGES.
    static const int ws_min_k = [] { const char* e = getenv("CREID_IGEMM_WS_MIN_K"); int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    if (g.log2span >= 6 && use_ws >= 1 && !(tuned_dma && !wred.ws && !bnred.x)) {
      // stride-2 3x3 data gradients: parity-class row order, 9 tap-tiles per 4 output pixels instead of 36
      static const int parity_on = [] { const char* e = getenv("CREID_DGRAD_PARITY"); return e ? atoi(e) : 1; }();
      IGemmGeom gp = g;
      if (parity_on && g.transposed && g.stride == 2 && g.kw == 3 && g.pad == 1 && (g.K >> g.log2span) == 9 &&
          g.OH % 2 == 0 && g.OW % 2 == 0 && (g.M / 4) % 128 == 0 && !g.add_compact && bnred.tiles_per_image == 0)
        gp.parity = 1;
      int ws_stages = use_ws == 2 ? ws_stages_env() : ((bn == 64 && g.K >= ws_min_k) ? 3 : 2);
      if (tuned_stages && use_ws != 2) ws_stages = tuned_stages;
      const dim3 block_ws(512);
      // 256-row tiles (128 x 64 consumer sub-tiles, one workgroup per CU): plan kind 3, or CREID_IGEMM_BM=256 wherever the grid
      // still has >= CREID_IGEMM_BM256_MIN_WGS (default 256) workgroups; not with the fused BN reduction / parity-class rows
      {
        const char* be = CREID_KNOB_ENV("CREID_IGEMM_BM");                 // read per call: tests and the tuner toggle it
        const int force_bm = be ? atoi(be) : 0;
        static const int bm256_min = [] { const char* e = getenv("CREID_IGEMM_BM256_MIN_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
        const bool can256 = g.N % 128 == 0 && !bnred.x && !gp.parity;
        const int64_t wgs256 = (int64_t)((g.M + 255) / 256) * (g.N / 128);
        if (can256 && force_bm != 128 && (tuned_bm256 || (force_bm == 256 && wgs256 >= bm256_min))) {
          const int st256 = (ws_stages >= 3 || g.K >= 256) ? 3 : 2;
          const dim3 grid256((unsigned)(wgs256 + (wred.ws ? wred.nblocks : 0)));
          const int tn256 = g.N / 128;
#define CREID_WS256_LAUNCH(NS_, ET_)                                                                                  \
  hipLaunchKernelGGL((igemm_bf16_ws_kernel<128, NS_, 256, ET_>), grid256, block_ws, 0, s, gp, (const unsigned short*)src, \
                     (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part, tn256, bnred, wred)
          if (dtype == CREID_F16) { if (st256 == 3) CREID_WS256_LAUNCH(3, F16T); else CREID_WS256_LAUNCH(2, F16T); }
          else { if (st256 == 3) CREID_WS256_LAUNCH(3, Bf16T); else CREID_WS256_LAUNCH(2, Bf16T); }
#undef CREID_WS256_LAUNCH
          return (int)hipGetLastError();
        }
      }
      // 4 x 32 KB at BN = 128 is one workgroup per CU (the C staging reuses the ring): only on request of a measured plan
      if (ws_stages == 4 && bn == 128 && !(tuned_stages == 4 && use_ws != 2)) ws_stages = 3;
      const dim3 grid_ws((unsigned)(tiles_m * tiles_n + (wred.ws ? wred.nblocks : 0)));
#define CREID_WS_LAUNCH1(BN_, NS_, ET_)                                                                               \
  hipLaunchKernelGGL((igemm_bf16_ws_kernel<BN_, NS_, 128, ET_>), grid_ws, block_ws, 0, s, gp, (const unsigned short*)src, \
                     (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part, tiles_n, \
                     bnred, wred)
#define CREID_WS_LAUNCH(BN_, NS_) do { if (dtype == CREID_F16) CREID_WS_LAUNCH1(BN_, NS_, F16T); else CREID_WS_LAUNCH1(BN_, NS_, Bf16T); } while (0)
      if (bn == 128) { if (ws_stages == 4) CREID_WS_LAUNCH(128, 4); else if (ws_stages == 2) CREID_WS_LAUNCH(128, 2); else CREID_WS_LAUNCH(128, 3); }
      else { if (ws_stages == 4) CREID_WS_LAUNCH(64, 4); else if (ws_stages == 2) CREID_WS_LAUNCH(64, 2); else CREID_WS_LAUNCH(64, 3); }
#undef CREID_WS_LAUNCH
#undef CREID_WS_LAUNCH1
      return (int)hipGetLastError();
    }
    if (wred.ws) { const int rc = wgrad_reduce_job_launch(wred, s); if (rc) return rc; wred.ws = nullptr; }
    // LDS ring depth (CREID_IGEMM_STAGES = 2..5, default 2 -- measured r01: deeper rings LOSE, the k-loop is bound
    // by the LDS->MFMA chain and by workgroups/CU, not by DMA latency; 3+ stages cost occupancy)
    static const int stages_env = [] { const char* e = getenv("CREID_IGEMM_STAGES"); int v = e ? atoi(e) : 0; return (v >= 2 && v <= 5) ? v : 2; }();
    const int stages = (tuned_dma && tuned_stages) ? tuned_stages : stages_env;
#define CREID_DMA_LAUNCH1(BN_, NS_, ET_)                                                                               \
  hipLaunchKernelGGL((igemm_bf16_dma_kernel<BN_, NS_, ET_>), grid, block, 0, s, g, (const unsigned short*)src,         \
                     (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part, tiles_n, \
                     bnred)
#define CREID_DMA_LAUNCH(BN_, NS_) do { if (dtype == CREID_F16) CREID_DMA_LAUNCH1(BN_, NS_, F16T); else CREID_DMA_LAUNCH1(BN_, NS_, Bf16T); } while (0)
    if (bn == 128) {
      switch (stages) { case 2: CREID_DMA_LAUNCH(128, 2); break; case 3: CREID_DMA_LAUNCH(128, 3); break;
                        case 4: case 5: CREID_DMA_LAUNCH(128, 4); break;   // 5 x 32 KB would not fit
                        default: CREID_DMA_LAUNCH(128, 2); break; }
    } else {
      switch (stages) { case 2: CREID_DMA_LAUNCH(64, 2); break; case 3: CREID_DMA_LAUNCH(64, 3); break;
                        case 4: CREID_DMA_LAUNCH(64, 4); break; case 5: CREID_DMA_LAUNCH(64, 5); break;
                        default: CREID_DMA_LAUNCH(64, 2); break; }
    }
#undef CREID_DMA_LAUNCH
#undef CREID_DMA_LAUNCH1
  } else if (bnred.x || g.add_mask || (g.epi_scale && creid_is16(dtype)) || dtype == CREID_F16) {
    return CREID_E_DTYPE;          // the fused reduction / masked add / folded BatchNorm (and f16) exist only in the 16-bit LDS-DMA kernels
  } else if (wred.ws && wgrad_reduce_job_launch(wred, s) != 0) {
    return (int)hipGetLastError();
  } else if (dtype == CREID_BF16) {
    if (g.K % 64 != 0) return CREID_E_SHAPE;
    if (bn == 128)
      hipLaunchKernelGGL(igemm_bf16_kernel<128>, grid, block, 0, s, g, (const unsigned short*)src,
                         (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part, tiles_n);
    else
      hipLaunchKernelGGL(igemm_bf16_kernel<64>, grid, block, 0, s, g, (const unsigned short*)src,
                         (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part, tiles_n);
  } else if (dtype == CREID_F32) {
    if (g.K % 16 != 0) return CREID_E_SHAPE;
    if (bn == 128)
      hipLaunchKernelGGL(igemm_f32_kernel<128>, grid, block, 0, s, g, (const float*)src, (const float*)wgt, (float*)out,
                         (const float*)add_src, bn_part, tiles_n);
    else
      hipLaunchKernelGGL(igemm_f32_kernel<64>, grid, block, 0, s, g, (const float*)src, (const float*)wgt, (float*)out,
                         (const float*)add_src, bn_part, tiles_n);
  } else {
    return CREID_E_DTYPE;
  }
  return (int)hipGetLastError();
}

static int check_desc(const creid_conv_desc* d) {
  if (!d || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->in_c <= 0 || d->out_c <= 0) return CREID_E_ARG;
  if (d->kh != d->kw || (d->kh != 1 && d->kh != 3) || d->stride < 1 || d->stride > 2) return CREID_E_SHAPE;
  if (ilog2_exact(d->in_c) < 0 || ilog2_exact(d->out_c) < 0 || d->in_c < 64 || d->out_c < 64) return CREID_E_SHAPE;
  if (d->out_h != (d->in_h + 2 * d->pad - d->kh) / d->stride + 1) return CREID_E_SHAPE;
  if (d->out_w != (d->in_w + 2 * d->pad - d->kw) / d->stride + 1) return CREID_E_SHAPE;
  if (d->batch * d->in_h * d->in_w >= (1LL << 24) || d->batch * d->out_h * d->out_w >= (1LL << 24) || d->batch * d->in_h * d->in_w * d->in_c > (1LL << 40)) return CREID_E_SHAPE;
  return 0;
}

extern "C" {

int64_t creid_conv2d_bn_partial_rows(const creid_conv_desc* d) {
  if (!d) return 0;
  return (d->batch * d->out_h * d->out_w + 127) / 128;   /* (sum, sumsq) row PAIRS */
}

int creid_conv2d_fwd_nhwc(const creid_conv_desc* d, const void* x, const void* w_krsc, void* y, float* bn_partial,
                          int dtype, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CREID_CHECK_ARG(x && w_krsc && y);
  IGemmGeom g;
  g.M = (int)(d->batch * d->out_h * d->out_w); g.OH = (int)d->out_h; g.OW = (int)d->out_w;
  g.SH = (int)d->in_h; g.SW = (int)d->in_w; g.pitch = (int)d->in_c; g.log2span = ilog2_exact(d->in_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 0;
  g.K = (int)(d->kh * d->kw * d->in_c); g.N = (int)d->out_c; g.check_bounds = 1;
  igemm_finish_geom(g);
  return launch_igemm(g, x, w_krsc, y, nullptr, bn_partial, dtype, as_stream(stream));
}

/* Forward convolution with an eval-mode BatchNorm folded into the epilogue: y = act(conv(x) * scale + shift (+ residual)),
 * scale_shift = [2][out_c] fp32 (creid_bn2d_fold_multi), act = ReLU when relu != 0.  No statistics, no second pass. */
int creid_conv2d_fwd_affine_nhwc(const creid_conv_desc* d, const void* x, const void* w_krsc, void* y, const float* scale_shift,
                                 const void* residual, int relu, int dtype, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CREID_CHECK_ARG(x && w_krsc && y && scale_shift);
  IGemmGeom g;
  g.M = (int)(d->batch * d->out_h * d->out_w); g.OH = (int)d->out_h; g.OW = (int)d->out_w;
  g.SH = (int)d->in_h; g.SW = (int)d->in_w; g.pitch = (int)d->in_c; g.log2span = ilog2_exact(d->in_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 0;
  g.K = (int)(d->kh * d->kw * d->in_c); g.N = (int)d->out_c; g.check_bounds = 1;
  igemm_finish_geom(g);
  g.epi_scale = scale_shift; g.epi_shift = scale_shift + d->out_c; g.epi_relu = relu ? 1 : 0;
  return launch_igemm(g, x, w_krsc, y, residual, nullptr, dtype, as_stream(stream));
}

int creid_conv2d_dgrad_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx, const void* add_src,
                            int dtype, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CREID_CHECK_ARG(dy && w_crsk && dx);
  IGemmGeom g;
  g.M = (int)(d->batch * d->in_h * d->in_w); g.OH = (int)d->in_h; g.OW = (int)d->in_w;
  g.SH = (int)d->out_h; g.SW = (int)d->out_w; g.pitch = (int)d->out_c; g.log2span = ilog2_exact(d->out_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 1;
  g.K = (int)(d->kh * d->kw * d->out_c); g.N = (int)d->in_c; g.check_bounds = 1;
  igemm_finish_geom(g);
  return launch_igemm(g, dy, w_crsk, dx, add_src, nullptr, dtype, as_stream(stream));
}

/* dgrad with the NEXT BatchNorm-backward's column reduction fused into the epilogue (bf16 only). */
int creid_conv2d_dgrad_bnred_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx,
                                  const void* add_src, const void* bn_x, const void* bn_act, const float* bn_mean,
                                  const float* bn_invstd, float* bn_partial, int64_t bn_stat_image_rows,
                                  int add_src_stride, int dtype, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CREID_CHECK_ARG(dy && w_crsk && dx && bn_x && bn_mean && bn_invstd && bn_partial && bn_stat_image_rows >= 0);
  if (bn_stat_image_rows % 128 != 0) return CREID_E_SHAPE;
  if (add_src_stride != 1 && add_src_stride != 2) return CREID_E_ARG;
  if (add_src_stride == 2 && (!add_src || d->in_h % 2 || d->in_w % 2)) return CREID_E_SHAPE;
  static const int use_dma = [] { const char* e = getenv("CREID_IGEMM_DMA"); return e ? atoi(e) : 1; }();
  if (!creid_is16(dtype) || !use_dma) return CREID_E_DTYPE;
  IGemmGeom g;
  g.M = (int)(d->batch * d->in_h * d->in_w); g.OH = (int)d->in_h; g.OW = (int)d->in_w;
  g.SH = (int)d->out_h; g.SW = (int)d->out_w; g.pitch = (int)d->out_c; g.log2span = ilog2_exact(d->out_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 1;
  g.K = (int)(d->kh * d->kw * d->out_c); g.N = (int)d->in_c; g.check_bounds = 1;
  igemm_finish_geom(g);
  static const int prefetch = [] { const char* e = getenv("CREID_BNRED_PREFETCH"); return e ? atoi(e) : 1; }();
  g.add_compact = add_src_stride == 2;
  BnRedArgs br{bn_x, bn_act, bn_mean, bn_invstd, bn_partial, prefetch, (int)(bn_stat_image_rows / 128)};
  return launch_igemm(g, dy, w_crsk, dx, add_src, nullptr, dtype, as_stream(stream), br);
}

/* Data gradient with everything that can ride in the same launch: "+ add_src" (full or stride-2 compact), the NEXT
 * BatchNorm-backward's column reduction (bn_x != NULL; bf16), and the split reduction of the PREVIOUS weight-gradient
 * (bn_mask != NULL: the ReLU mask comes as bits from creid_bn2d_apply_mask, one byte per 8 channels, and bn_act is not read)
 * launch (wred_desc != NULL: creid_conv2d_wgrad_partials of that convolution wrote wred_ws; the first workgroups of
 * this launch sum the partials into wred_dw).  Where the fused kernel does not apply (fp32 parity mode, stem) the
 * reduction runs as its own launch first -- same result. */
int creid_conv2d_dgrad_fused_nhwc(const creid_conv_desc* d, const void* dy, const void* w_crsk, void* dx,
                                  const void* add_src, int add_src_stride, const uint8_t* add_mask, const void* bn_x,
                                  const void* bn_act, const uint8_t* bn_mask, const float* bn_mean, const float* bn_invstd, float* bn_partial,
                                  int64_t bn_stat_image_rows, const creid_conv_desc* wred_desc, float* wred_dw,
                                  int wred_accumulate, const void* wred_ws, size_t wred_ws_bytes, int dtype, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CREID_CHECK_ARG(dy && w_crsk && dx && bn_stat_image_rows >= 0);
  if (bn_x) CREID_CHECK_ARG(bn_mean && bn_invstd && bn_partial);
  if (bn_stat_image_rows % 128 != 0) return CREID_E_SHAPE;
  if (add_src_stride != 1 && add_src_stride != 2) return CREID_E_ARG;
  if (add_src_stride == 2 && (!add_src || d->in_h % 2 || d->in_w % 2)) return CREID_E_SHAPE;
  static const int use_dma = [] { const char* e = getenv("CREID_IGEMM_DMA"); return e ? atoi(e) : 1; }();
  if (bn_x && (!creid_is16(dtype) || !use_dma)) return CREID_E_DTYPE;
  if (add_mask && (!add_src || add_src_stride != 1 || !creid_is16(dtype) || !use_dma)) return CREID_E_ARG;
  WRedJob job{};
  bool have_job = false;
  if (wred_desc) {
    CREID_CHECK_ARG(wred_dw && wred_ws);
    if (!wgrad_make_reduce_job(wred_desc, dtype, wred_ws, wred_ws_bytes, wred_dw, wred_accumulate, job)) return CREID_E_SHAPE;
    have_job = true;
  }
  IGemmGeom g;
  g.M = (int)(d->batch * d->in_h * d->in_w); g.OH = (int)d->in_h; g.OW = (int)d->in_w;
  g.SH = (int)d->out_h; g.SW = (int)d->out_w; g.pitch = (int)d->out_c; g.log2span = ilog2_exact(d->out_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 1;
  g.K = (int)(d->kh * d->kw * d->out_c); g.N = (int)d->in_c; g.check_bounds = 1;
  igemm_finish_geom(g);
  static const int prefetch = [] { const char* e = getenv("CREID_BNRED_PREFETCH"); return e ? atoi(e) : 1; }();
  g.add_compact = add_src_stride == 2;
  g.add_mask = add_mask;
  BnRedArgs br{bn_x, bn_mask ? nullptr : bn_act, bn_mean, bn_invstd, bn_partial, prefetch, (int)(bn_stat_image_rows / 128), bn_mask};
  if (!bn_x) br = BnRedArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
  return launch_igemm(g, dy, w_crsk, dx, add_src, nullptr, dtype, as_stream(stream), br, have_job ? &job : nullptr);
}

/* stem: 7x7 stride-2 pad-3 conv, 3 -> 64 channels, on the pre-padded NHWC4 image
 * xpad [B, H+8, W+6, 4] (image at rows 3..H+2, cols 3..W+2, zeros elsewhere); w_stem [64][8][32]
 * (k = r*32 + s*4 + c, zero for s = 7, c = 3 and r = 7); y [B, H/2, W/2, 64]. */
int creid_stem_conv_fwd(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                        float* bn_partial, int dtype, void* stream) {
  CREID_CHECK_ARG(xpad && w_stem && y && batch > 0 && H > 0 && W > 0);
  if (H % 2 || W % 2) return CREID_E_SHAPE;
  IGemmGeom g;
  g.M = (int)(batch * (H / 2) * (W / 2)); g.OH = (int)(H / 2); g.OW = (int)(W / 2);
  g.SH = (int)(H + 8); g.SW = (int)(W + 6); g.pitch = 4; g.log2span = 5;
  g.kw = 1; g.stride = 2; g.pad = 0; g.transposed = 0; g.K = 256; g.N = 64; g.check_bounds = 0;
  igemm_finish_geom(g);
  return launch_igemm(g, xpad, w_stem, y, nullptr, bn_partial, dtype, as_stream(stream));
}

/* the stem with its eval-mode BatchNorm (and, for the IBN-a variant, ReLU) folded into the epilogue */
int creid_stem_conv_fwd_affine(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* w_stem, void* y,
                               const float* scale_shift, int relu, int dtype, void* stream) {
  CREID_CHECK_ARG(xpad && w_stem && y && scale_shift && batch > 0 && H > 0 && W > 0);
  if (H % 2 || W % 2) return CREID_E_SHAPE;
  IGemmGeom g;
  g.M = (int)(batch * (H / 2) * (W / 2)); g.OH = (int)(H / 2); g.OW = (int)(W / 2);
  g.SH = (int)(H + 8); g.SW = (int)(W + 6); g.pitch = 4; g.log2span = 5;
  g.kw = 1; g.stride = 2; g.pad = 0; g.transposed = 0; g.K = 256; g.N = 64; g.check_bounds = 0;
  igemm_finish_geom(g);
  g.epi_scale = scale_shift; g.epi_shift = scale_shift + 64; g.epi_relu = relu ? 1 : 0;
  return launch_igemm(g, xpad, w_stem, y, nullptr, nullptr, dtype, as_stream(stream));
}

}  // extern "C"
