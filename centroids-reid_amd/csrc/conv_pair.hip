// Eval-mode block boundary of layer1 in ONE launch (round 4): conv3 (1 x 1, 64 -> 256) + folded bn3 + residual
// + ReLU of bottleneck i, and conv1 (1 x 1, 256 -> 64) + folded bn1 + ReLU of bottleneck i + 1 (modelling/backbones/resnet.py:77-87
// then :69-71 of the next block).  As two launches the 256-channel block output (134 MB at batch 128) is written by the first and
// read back by the second, which does nothing else of weight: 168 MB of traffic for 4.3 GFLOP.  Here a resident workgroup owns a
// 128-pixel row tile with ALL 256 output channels: the finished, rounded block output goes to memory (it is the next block's
// residual) AND, as the same 16-byte row-major chunks, into a second LDS operand image, against which the resident conv1 weights
// are multiplied right away -- the second convolution reads nothing from memory but its 32 KB of weights, once per workgroup.
// Only layer1 fits: a 128-pixel tile of the block output is 64 KB there, 128 KB in layer2.
// The first half of the kernel is igemm1x1_stream2_kernel<256, 1, 1> (conv_stream.hip) line for line -- operands through
// registers a tile ahead, stores a tile late, column-major packed staging read back through the transposing LDS read -- and the
// second multiply runs in the k order of the tile kernels (four 64-deep chunks, 16-wide slices inside): both outputs are
// bit-identical to the two launches (tests/test_eval_fold_gpu.py::test_block_boundary_one_launch_equals_two_launches).
#include "conv_common.hpp"
#include <stdlib.h>

namespace creid_pair {

// STATS (the next block's conv1 feeds an IBN layer, resnet_ibn_a.py:27-32): out1 is the RAW convolution output and bn_part1 gets
// the per-tile (sum, sum of squares) of the fp32 accumulators, [tiles][2][64] -- creid_conv2d_fwd_nhwc's statistics partials
// (rows of a tile belong to one image when H * W % 128 = 0; M % 128 = 0 required: no padded rows in the sums).
// 16-byte chunk swizzle of the second operand image: conflict-free for the 16-byte WRITES of the copy-out lanes (eight contiguous
// lanes = four consecutive rows x two adjacent chunks, banks mod 128 bytes) as well as for the fragment reads (ds_read_b128's lane
// groups over 32 consecutive rows, banks mod 256 bytes); the (r >> 1) & 7 swizzle of the DMA-written images serves only the reads
__device__ __forceinline__ int a2_key(int r) { return (((r & 3) << 1) ^ ((r >> 2) & 3)) & 7; }

template <typename ET, bool STATS>
__global__ __launch_bounds__(512, 1) void c3_c1_kernel(const unsigned short* __restrict__ src,      // [M][64]   conv2's output
                                                       const unsigned short* __restrict__ w3,       // [256][64]
                                                       const unsigned short* __restrict__ res,      // [M][256]  the block's input
                                                       const float* __restrict__ ss3,               // [2][256]  folded bn3
                                                       unsigned short* __restrict__ out3,           // [M][256]  block output
                                                       const unsigned short* __restrict__ w1,       // [64][256] next block's conv1
                                                       const float* __restrict__ ss1,               // [2][64]   folded bn1
                                                       unsigned short* __restrict__ out1,           // [M][64]
                                                       float* __restrict__ bn_part1, int M, int tiles_m, int abl) {
  constexpr int BN = 256, HB = 128, NH = 2, TW = 4;
  constexpr int CPT = 128 + 4;                         // staging pitch: [HB columns][128 rows + 4]
  constexpr int CPR = HB / 8, NIT = (128 * CPR) / 512; // 16-byte chunks per row of a half; chunks per thread and half (4)
  constexpr int CPR2 = 8, NIT2 = (128 * CPR2) / 512;   // the 64-column second output (2)
  constexpr int W3_ELEMS = BN * 64, TILE_ELEMS = 128 * 64, STAGE_ELEMS = HB * CPT, W1_ELEMS = 4 * 64 * 64, A2_ELEMS = 2 * TILE_ELEMS;
  constexpr int RED_ELEMS = STATS ? 4 * 2 * 64 * 2 : 0;  // fp32 [4 row waves][2][64] in 2-byte units
  static_assert(2 * (W3_ELEMS + TILE_ELEMS + STAGE_ELEMS + W1_ELEMS + A2_ELEMS + RED_ELEMS) <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[W3_ELEMS + TILE_ELEMS + STAGE_ELEMS + W1_ELEMS + A2_ELEMS + RED_ELEMS + 8];
  unsigned short* Ws = smem;                           // conv3 weights [256][64]
  unsigned short* slot = smem + W3_ELEMS;              // this tile's A operand [128][64]
  unsigned short* stage = slot + TILE_ELEMS;           // [HB][CPT]
  unsigned short* Ws1 = stage + STAGE_ELEMS;           // conv1 weights [4 k-chunks][64 columns][64]
  unsigned short* A2 = Ws1 + W1_ELEMS;                 // the finished half of the block output as an operand: [2 k-chunks][128][64]
  float* red = reinterpret_cast<float*>(A2 + A2_ELEMS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int l31 = lane & 31, kh = lane >> 5;
  const int lr8 = lane >> 3, lcp = lane & 7;
  const int groups = (int)gridDim.x, wg = (int)blockIdx.x;
  const int n_iter = wg < tiles_m ? (tiles_m - 1 - wg) / groups + 1 : 0;
  if (n_iter == 0) return;

  // both weight slabs -> LDS, once (row r of k-chunk kc at [kc][r][64], 16-byte chunks XOR-swizzled by (r >> 1) & 7)
  for (int i = wave; i < BN / 8; i += 8) {
    const int r = i * 8 + lr8;
    *reinterpret_cast<uint4*>(Ws + r * 64 + ((lcp ^ ((r >> 1) & 7)) << 3)) = *reinterpret_cast<const uint4*>(w3 + (int64_t)r * 64 + lcp * 8);
  }
  for (int i = wave; i < 4 * 8; i += 8) {
    const int kc = i >> 3, r = (i & 7) * 8 + lr8;
    *reinterpret_cast<uint4*>(Ws1 + (kc * 64 + r) * 64 + ((lcp ^ ((r >> 1) & 7)) << 3)) =
        *reinterpret_cast<const uint4*>(w1 + (int64_t)r * 256 + kc * 64 + lcp * 8);
  }
  uint4 areg[2];
  auto load_a = [&](int it) {
    const int row0 = (wg + it * groups) * 128;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = (wave + 8 * u) * 8 + lr8, m = row0 + r;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (!CREID_ABL_ON(abl, 4)) v = *reinterpret_cast<const uint4*>(src + (int64_t)min(m, M - 1) * 64 + lcp * 8);
      areg[u] = m < M ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto put_a = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = (wave + 8 * u) * 8 + lr8;
      *reinterpret_cast<uint4*>(slot + r * 64 + ((lcp ^ ((r >> 1) & 7)) << 3)) = areg[u];
    }
  };
  const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
  auto unit_of = [&](int i, int cpr, int& rl, int& ch) {
    const int Q = (wave + 8 * i) * 16 + g4 * 4 + q4;
    ch = Q % cpr;
    rl = 4 * (Q / cpr) + t4;
  };
  uint4 resn[NH][NIT];                                  // residual chunks of the NEXT tile
  uint4 outv[NH][NIT];                                  // finished block-output chunks of the PREVIOUS tile
  uint4 out1v[NIT2];                                    // finished conv1 chunks of the PREVIOUS tile
  auto load_res = [&](int it) {
    const int row0 = (wg + it * groups) * 128;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        int rl, ch;
        unit_of(i, CPR, rl, ch);
        resn[h][i] = make_uint4(0u, 0u, 0u, 0u);
        if (!CREID_ABL_ON(abl, 2)) resn[h][i] = *reinterpret_cast<const uint4*>(res + (int64_t)min(row0 + rl, M - 1) * BN + h * HB + ch * 8);
      }
  };
  auto store_out = [&](int it) {
    const int row0 = (wg + it * groups) * 128;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        int rl, ch;
        unit_of(i, CPR, rl, ch);
        if (row0 + rl < M && !CREID_ABL_ON(abl, 1)) *reinterpret_cast<uint4*>(out3 + (int64_t)(row0 + rl) * BN + h * HB + ch * 8) = outv[h][i];
      }
#pragma unroll
    for (int i = 0; i < NIT2; ++i) {
      int rl, ch;
      unit_of(i, CPR2, rl, ch);
      if (row0 + rl < M && !CREID_ABL_ON(abl, 1)) *reinterpret_cast<uint4*>(out1 + (int64_t)(row0 + rl) * 64 + ch * 8) = out1v[i];
    }
  };

  load_a(0);
  load_res(0);
  float sc[TW], sh[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) { sc[j] = ss3[(2 * j + wc) * 32 + l31]; sh[j] = ss3[BN + (2 * j + wc) * 32 + l31]; }
  float sc1 = 1.f, sh1 = 0.f;
  if constexpr (!STATS) { sc1 = ss1[wc * 32 + l31]; sh1 = ss1[64 + wc * 32 + l31]; }

  for (int it = 0; it < n_iter; ++it) {
    __syncthreads();                                              // everyone is done with tile it - 1 (slot, staging, A2)
    put_a();                                                      // this tile's rows, loaded a tile ago
    uint4 resc[NH][NIT];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) resc[h][i] = resn[h][i];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(resc[h][i].x), "+v"(resc[h][i].y), "+v"(resc[h][i].z), "+v"(resc[h][i].w));
    // this tile's requests: loads first, then the previous tile's stores (see igemm1x1_stream2_kernel)
    if (it + 1 < n_iter) { load_a(it + 1); load_res(it + 1); }
    if (it > 0) store_out(it - 1);
    __syncthreads();                                              // the slot holds tile `it`
    f32x16 acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int chk = 2 * kk + kh;
      const int r = wr * 32 + l31;
      const s16x8 a = *reinterpret_cast<const s16x8*>(&slot[r * 64 + ((chk ^ ((r >> 1) & 7)) << 3)]);
      s16x8 b[TW];
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const int cc = (2 * j + wc) * 32 + l31;
        b[j] = *reinterpret_cast<const s16x8*>(&Ws[cc * 64 + ((chk ^ ((cc >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[j] = ET::mfma(a, b[j], acc[j]);
    }
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (h > 0) asm volatile("s_barrier" ::: "memory");          // the previous half: staging read out, A2 multiplied
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        if (((2 * j + wc) * 32) / HB != h) continue;              // (wave-uniform)
        const int cl = (2 * j + wc) * 32 - h * HB + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wr * 32 + 8 * q + 4 * kh;
          float v0 = fmaf(acc[j][4 * q], sc[j], sh[j]), v1 = fmaf(acc[j][4 * q + 1], sc[j], sh[j]);
          float v2 = fmaf(acc[j][4 * q + 2], sc[j], sh[j]), v3 = fmaf(acc[j][4 * q + 3], sc[j], sh[j]);
          // the fp32 results must exist as such: left to itself the compiler folds fma + conversion into v_fma_mix{lo,hi}_f16,
          // which rounds the exact product-sum ONCE to f16 -- 2e-5 of the values then differ from the two-step rounding of
          // every other kernel in the last bit
          asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
          *reinterpret_cast<uint2*>(&stage[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
      }
      __syncthreads();                                            // the half is staged
      u32x2 trlo[NIT], trhi[NIT];
      {
        const int sq = lane & 3, sj = (lane >> 2) & 3;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int Qs = (wave + 8 * i) * 16 + g4 * 4 + sq;
          const unsigned addr = (unsigned)(uintptr_t)&stage[((Qs % CPR) * 8 + sj) * CPT + 4 * (Qs / CPR)];
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                       : "=&v"(trlo[i]), "=&v"(trhi[i]) : "v"(addr), "i"(4 * CPT * 2) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(trlo[i]), "+v"(trhi[i]));
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        uint4 v = make_uint4(trlo[i].x, trlo[i].y, trhi[i].x, trhi[i].y);
        const uint4 a = resc[h][i];
        unsigned* vw = &v.x; const unsigned* aw = &a.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float lo = ET::lo(vw[q]) + ET::lo(aw[q]);
          float hi = ET::hi(vw[q]) + ET::hi(aw[q]);
          lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f);
          vw[q] = ET::pack2(lo, hi);
        }
        outv[h][i] = v;
        // the same chunk as an operand of the second multiply: row rl, channels h * 128 + ch * 8 ..
        int rl, ch;
        unit_of(i, CPR, rl, ch);
        *reinterpret_cast<uint4*>(A2 + (ch >> 3) * TILE_ELEMS + rl * 64 + (((ch & 7) ^ a2_key(rl)) << 3)) = v;
      }
      __syncthreads();                                            // this half of the block output is an operand
      if (!CREID_ABL_ON(abl, 8)) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int chk = 2 * kk + kh;
            const int r = wr * 32 + l31, cc = wc * 32 + l31;
            const s16x8 a = *reinterpret_cast<const s16x8*>(&A2[kc * TILE_ELEMS + r * 64 + ((chk ^ a2_key(r)) << 3)]);
            const s16x8 b = *reinterpret_cast<const s16x8*>(&Ws1[((2 * h + kc) * 64 + cc) * 64 + ((chk ^ ((cc >> 1) & 7)) << 3)]);
            acc2 = ET::mfma(a, b, acc2);
          }
      }
    }
    // conv1's epilogue: folded bn1 + ReLU, staged column-major in the (free) staging area, read back as 16-byte row chunks
    {
      const int cl = wc * 32 + l31;
      if constexpr (STATS) {                                      // column sums of the fp32 accumulators (igemm1x1_stream2_kernel's)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc2[r]; s1 += v; s2 = fmaf(v, v, s2); }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (kh == 0) { red[(wr * 2 + 0) * 64 + cl] = s1; red[(wr * 2 + 1) * 64 + cl] = s2; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = wr * 32 + 8 * q + 4 * kh;
        float v0 = acc2[4 * q], v1 = acc2[4 * q + 1], v2 = acc2[4 * q + 2], v3 = acc2[4 * q + 3];
        if constexpr (!STATS) {
          v0 = fmaf(v0, sc1, sh1); v1 = fmaf(v1, sc1, sh1); v2 = fmaf(v2, sc1, sh1); v3 = fmaf(v3, sc1, sh1);
          asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));    // (as above)
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        *reinterpret_cast<uint2*>(&stage[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
      }
    }
    __syncthreads();
    if constexpr (STATS) {
      if (tid < 2 * 64) {
        const int which = tid >> 6, cl = tid & 63;
        bn_part1[((int64_t)(wg + it * groups) * 2 + which) * 64 + cl] =
            (red[(0 * 2 + which) * 64 + cl] + red[(1 * 2 + which) * 64 + cl]) + (red[(2 * 2 + which) * 64 + cl] + red[(3 * 2 + which) * 64 + cl]);
      }
    }
    {
      u32x2 trlo[NIT2], trhi[NIT2];
      const int sq = lane & 3, sj = (lane >> 2) & 3;
#pragma unroll
      for (int i = 0; i < NIT2; ++i) {
        const int Qs = (wave + 8 * i) * 16 + g4 * 4 + sq;
        const unsigned addr = (unsigned)(uintptr_t)&stage[((Qs % CPR2) * 8 + sj) * CPT + 4 * (Qs / CPR2)];
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                     : "=&v"(trlo[i]), "=&v"(trhi[i]) : "v"(addr), "i"(4 * CPT * 2) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NIT2; ++i) asm volatile("" : "+v"(trlo[i]), "+v"(trhi[i]));
#pragma unroll
      for (int i = 0; i < NIT2; ++i) out1v[i] = make_uint4(trlo[i].x, trlo[i].y, trhi[i].x, trhi[i].y);
    }
  }
  store_out(n_iter - 1);
}

}  // namespace creid_pair

extern "C" {

/* see include/creid.h */
static int launch_pair(int64_t M, int64_t c_mid, int64_t c_out, int64_t c_next, const void* a2, const void* w3_krsc, const float* fold3,
                       const void* residual, void* out3, const void* w1_krsc, const float* fold1, void* out1, float* bn_part1,
                       int dtype, hipStream_t s) {
  if (!creid_is16(dtype)) return CREID_E_DTYPE;
  if (c_mid != 64 || c_out != 256 || c_next != 64 || M >= ((int64_t)1 << 31) - 128) return CREID_E_SHAPE;
  if (bn_part1 && M % 128 != 0) return CREID_E_SHAPE;
  const int tiles_m = (int)((M + 127) / 128);
  int wgs = 256;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) wgs = v; }   // read per call (tests)
  if (wgs > tiles_m) wgs = tiles_m;
#ifdef CREID_ABL_BUILD
  const char* ae = CREID_KNOB_ENV("CREID_PAIR_ABL");            // 1 no stores, 2 no residual loads, 4 no A loads, 8 no second multiply
  const int abl = ae ? atoi(ae) : 0;
#else
  const int abl = 0;
#endif
#define CREID_PAIR_LAUNCH(ET_, ST_)                                                                                             \
  hipLaunchKernelGGL((creid_pair::c3_c1_kernel<ET_, ST_>), dim3((unsigned)wgs), dim3(512), 0, s, (const unsigned short*)a2,     \
                     (const unsigned short*)w3_krsc, (const unsigned short*)residual, fold3, (unsigned short*)out3,             \
                     (const unsigned short*)w1_krsc, fold1, (unsigned short*)out1, bn_part1, (int)M, tiles_m, abl)
  if (bn_part1) { if (dtype == CREID_F16) CREID_PAIR_LAUNCH(F16T, true); else CREID_PAIR_LAUNCH(Bf16T, true); }
  else { if (dtype == CREID_F16) CREID_PAIR_LAUNCH(F16T, false); else CREID_PAIR_LAUNCH(Bf16T, false); }
#undef CREID_PAIR_LAUNCH
  return (int)hipGetLastError();
}

int creid_bottleneck_c3_c1_fwd_affine(int64_t M, int64_t c_mid, int64_t c_out, int64_t c_next, const void* a2, const void* w3_krsc,
                                      const float* fold3, const void* residual, void* out3, const void* w1_krsc, const float* fold1,
                                      void* out1, int dtype, void* stream) {
  CREID_CHECK_ARG(a2 && w3_krsc && fold3 && residual && out3 && w1_krsc && fold1 && out1 && M > 0);
  return launch_pair(M, c_mid, c_out, c_next, a2, w3_krsc, fold3, residual, out3, w1_krsc, fold1, out1, nullptr, dtype, as_stream(stream));
}

int creid_bottleneck_c3_c1_fwd_stats(int64_t M, int64_t c_mid, int64_t c_out, int64_t c_next, const void* a2, const void* w3_krsc,
                                     const float* fold3, const void* residual, void* out3, const void* w1_krsc, void* out1_raw,
                                     float* bn_partial1, int dtype, void* stream) {
  CREID_CHECK_ARG(a2 && w3_krsc && fold3 && residual && out3 && w1_krsc && out1_raw && bn_partial1 && M > 0);
  return launch_pair(M, c_mid, c_out, c_next, a2, w3_krsc, fold3, residual, out3, w1_krsc, nullptr, out1_raw, bn_partial1, dtype,
                     as_stream(stream));
}

}  // extern "C"
