// Stage A: PERSISTENT streaming kernel for the small-K 1x1 stride-1 convolutions of layer1 / layer2
// (modelling/backbones/resnet.py:56,60 conv1 / conv3 of the Bottleneck, :109 the stride-1 downsample) and their data
// gradients.  These GEMMs -- C[M, N] = A[M, K] . W[N, K]^T with K = 64..256, M = 32 768..131 072 rows at B = 64 -- are
// HBM-streaming work (64 -> 256 moves 84 MB for 4.3 GFLOP), and the tile-per-workgroup kernels of conv_igemm.hip run
// them at 2-3 TB/s: a workgroup lives for one 128-row tile, i.e. prologue -> DMA -> wait a full load round trip -> 16
// MFMAs -> epilogue, with nothing to overlap the round trip except a second short-lived workgroup, and it re-fetches
// the whole weight matrix (a third of its LDS fill traffic at K = 64) for every tile.  Here:
//   * a workgroup is resident for the whole launch and walks a strided sequence of row tiles;
//   * the weight slab W[BN x K] (<= 64 KB) is loaded into LDS ONCE per workgroup;
//   * four producer waves stream the A tiles of FUTURE row tiles into an LDS ring with the global->LDS DMA while four
//     consumer waves multiply the current one and write it out, so the load round trip is off the critical path.
// Same LDS image as conv_igemm.hip (row-major [row][64] bf16, 16-B chunks XOR-swizzled by (row >> 1) & 7 on the DMA
// source side), same 32x32x16 MFMA fragment map, same epilogue arithmetic (bf16 C tile staged through LDS for 16-B
// stores, per-tile (sum, sum of squares) partials of the fp32 accumulators for the BatchNorm that follows).
// Every wave executes the same sequence of s_barrier's by construction: one per k-chunk, then a fixed number per
// epilogue; all loop bounds are workgroup-uniform.
#include "conv_common.hpp"
#include <stdlib.h>

__device__ __attribute__((aligned(128))) unsigned g_stream_zero_page[32];   // rows >= M fetch this page of zeros

namespace {
template <int N> __device__ __forceinline__ void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void st_barrier() { asm volatile("s_barrier" ::: "memory"); }
}  // namespace

// BN: output columns per workgroup (64 / 128 / 256); KCH: K / 64 (1, 2, 4); NSA: A-ring depth in 16 KB chunks.
template <int BN, int KCH, int NSA>
__global__ __launch_bounds__(512, 1) void igemm1x1_stream_kernel(const unsigned short* __restrict__ src, int M, int K, int N,
                                                                  const unsigned short* __restrict__ wgt,
                                                                  unsigned short* __restrict__ out,
                                                                  float* __restrict__ bn_part, int tiles_m, int tiles_n, int dbg) {
  constexpr int TNW = BN / 64;                         // 32-wide column blocks per consumer wave (x 2 row blocks)
  constexpr int HB = BN > 128 ? 128 : BN;              // columns staged / copied out at a time
  constexpr int NH = BN / HB;                          // halves per tile
  constexpr int CP = HB + 8;                           // staging pitch (elements)
  constexpr int W_ELEMS = KCH * BN * 64, RING_ELEMS = NSA * 128 * 64, STAGE_ELEMS = 128 * CP;
  static_assert(2 * (W_ELEMS + RING_ELEMS + STAGE_ELEMS) <= 160 * 1024, "LDS budget");
  static_assert(NSA >= 2 && (NSA - 1) * 4 <= 63, "ring depth");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[W_ELEMS + RING_ELEMS + STAGE_ELEMS];
  unsigned short* Ws = smem;                           // [KCH][BN][64]
  unsigned short* ring = smem + W_ELEMS;               // [NSA][128][64]
  unsigned short* stage = smem + W_ELEMS + RING_ELEMS; // [128][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int cw = wave & 3, wm = cw >> 1, wn = cw & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int lr8 = lane >> 3, lcp = lane & 7;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;

  // this workgroup: a fixed column slab, row tiles tile_m0, tile_m0 + stride, ...
  const int groups = (int)gridDim.x / tiles_n;         // workgroups per column slab (grid is a multiple of tiles_n)
  const int tile_n = (int)blockIdx.x % tiles_n, wg_in_group = (int)blockIdx.x / tiles_n;
  const int col0 = tile_n * BN;
  const int n_iter = wg_in_group < tiles_m ? (tiles_m - 1 - wg_in_group) / groups + 1 : 0;
  const int n_chunks = n_iter * KCH;

  // ---- weight slab -> LDS, once (all 8 waves; KCH * BN / 8 wave-instructions of 8 rows x 128 B)
  {
    constexpr int NW = KCH * (BN / 8);
    for (int i = wave; i < NW; i += 8) {
      const int kc = i / (BN / 8), r = (i - kc * (BN / 8)) * 8 + lr8;
      const unsigned short* p = wgt + (int64_t)(col0 + r) * K + kc * 64 + ((lcp ^ ((r >> 1) & 7)) << 3);
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Ws + (kc * BN + (i - kc * (BN / 8)) * 8) * 64), 16, 0, 0);
    }
  }

  f32x16 acc[2][TNW];
  if (producer) {
    // chunk c = (iteration c / KCH, k-chunk c % KCH) -> ring slot c % NSA; 4 DMA instructions per wave per chunk
    const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_stream_zero_page);
    int gch[4], rloc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rloc[i] = (i * 4 + cw) * 8 + lr8;
      gch[i] = (lcp ^ ((rloc[i] >> 1) & 7)) << 3;
    }
    auto issue = [&](int c) {
      const int it = c / KCH, kc = c - it * KCH;
      const int row0 = (wg_in_group + it * groups) * 128;
      unsigned short* la = ring + (c % NSA) * (128 * 64) + cw * 512;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = row0 + rloc[i];
        const unsigned short* p = m < M ? src + (int64_t)m * K + kc * 64 + gch[i] : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(la + i * 2048), 16, 0, 0);
      }
    };
    int issued = 0;
#pragma unroll
    for (int p = 0; p < NSA - 1; ++p)
      if (p < n_chunks) { issue(p); ++issued; }
    int c = 0;
    for (int it = 0; it < n_iter; ++it) {
      for (int kc = 0; kc < KCH; ++kc, ++c) {
        // chunk c (and the weight slab, issued before it) has landed once at most the younger chunks are pending
        const int younger = issued - c - 1;
        if (younger >= 2) st_wait_vm<8>(); else if (younger == 1) st_wait_vm<4>(); else st_wait_vm<0>();
        st_barrier();                                              // B1(c)
        if (issued < n_chunks && issued - c < NSA) { issue(issued); ++issued; }   // into the slot chunk c-1 just left
      }
      // epilogue barriers (the consumers' copy-out); producers only keep step
#pragma unroll
      for (int h = 0; h < NH; ++h) { st_barrier(); st_barrier(); }
      if (bn_part) { st_barrier(); st_barrier(); }
    }
    return;
  }

  // ---- consumers
  st_wait_vm<0>();                                                 // this wave's share of the weight slab
  int c = 0;
  for (int it = 0; it < n_iter; ++it) {
    const int tile_m = wg_in_group + it * groups, row0 = tile_m * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kc = 0; kc < KCH; ++kc, ++c) {
      st_barrier();                                                // B1(c): chunk c is in LDS
      const unsigned short* As = ring + (c % NSA) * (128 * 64);
      const unsigned short* Bs = Ws + kc * BN * 64;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int ch = 2 * kk + kh;
        s16x8 a[2], b[TNW];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wm * 64 + i * 32 + l31;
          a[i] = *reinterpret_cast<const s16x8*>(&As[r * 64 + ((ch ^ ((r >> 1) & 7)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int cc = wn * (BN / 2) + j * 32 + l31;
          b[j] = *reinterpret_cast<const s16x8*>(&Bs[cc * 64 + ((ch ^ ((cc >> 1) & 7)) << 3)]);
        }
        if (dbg & 4) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue: column statistics from the fp32 accumulators, then the C tile through LDS in HB-wide halves
    float s1v[TNW], s2v[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 = fmaf(v, v, s2); }   // rows >= M are zero
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      s1v[j] = s1; s2v[j] = s2;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      // this wave's columns wn*(BN/2) + j*32 .. : half h holds columns [h*HB, (h+1)*HB)
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int cglob = wn * (BN / 2) + j * 32;                  // first column of the 32-wide block
        if (cglob / HB != h) continue;                             // compile-time after unrolling
        const int cl = cglob - h * HB + l31;
        if (dbg & 2) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            stage[rl * CP + cl] = f32_to_bf16_bits(acc[i][j][r]);
          }
      }
      st_barrier();                                                // E1: the half is staged
      constexpr int CPR = HB / 8;                                  // 16-B chunks per staged row
#pragma unroll
      for (int i = 0; i < (128 * CPR) / 256; ++i) {
        const int id = tid + 256 * i, rl = id / CPR, ch = id - rl * CPR;
        const int rr = row0 + rl;
        if (rr < M && !(dbg & 1)) {
          const uint4 v = *reinterpret_cast<const uint4*>(&stage[rl * CP + ch * 8]);
          *reinterpret_cast<uint4*>(out + (int64_t)rr * N + col0 + h * HB + ch * 8) = v;
        }
      }
      st_barrier();                                                // E2: the staging area is free again
    }
    if (bn_part) {
      float* red = reinterpret_cast<float*>(stage);                // [2 (wm)][2 (s1, s2)][BN]
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int cl = wn * (BN / 2) + j * 32 + l31;
        if (kh == 0) { red[(wm * 2 + 0) * BN + cl] = s1v[j]; red[(wm * 2 + 1) * BN + cl] = s2v[j]; }
      }
      st_barrier();
      for (int i = tid; i < 2 * BN; i += 256) {
        const int which = i / BN, cl = i - which * BN;
        bn_part[((int64_t)tile_m * 2 + which) * N + col0 + cl] = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
      }
      st_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------ host
// Returns CREID_E_SHAPE when the GEMM is outside the kernel's scope (the caller then uses conv_igemm.hip's kernels).
int launch_stream1x1(int M, int K, int N, const void* src, const void* wgt, void* out, float* bn_part, hipStream_t s) {
  if (K != 64 && K != 128 && K != 256) return CREID_E_SHAPE;
  int bn = N >= 256 ? 256 : N;
  if (bn != 64 && bn != 128 && bn != 256) return CREID_E_SHAPE;
  if (N % bn != 0) return CREID_E_SHAPE;
  if (K == 256 && bn == 256) bn = 128;                 // the weight slab must fit: BN * K <= 32 K elements
  if ((int64_t)bn * K > 32768) return CREID_E_SHAPE;
  const int tiles_m = (M + 127) / 128, tiles_n = N / bn;
  // persistent grid: one workgroup per CU (the LDS budget allows one), split evenly over the column slabs
  int wgs = 256;                                       // read per call (tests shrink it to force many tiles per workgroup)
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) wgs = v; }
  int dbg = 0;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_DBG"); if (e) dbg = atoi(e); }
  int groups = wgs / tiles_n;
  if (groups < 1) groups = 1;
  if (groups > tiles_m) groups = tiles_m;
  const dim3 grid((unsigned)(groups * tiles_n)), block(512);
#define CREID_ST_LAUNCH(BN_, KCH_, NSA_)                                                                              \
  hipLaunchKernelGGL((igemm1x1_stream_kernel<BN_, KCH_, NSA_>), grid, block, 0, s, (const unsigned short*)src, M, K, N, \
                     (const unsigned short*)wgt, (unsigned short*)out, bn_part, tiles_m, tiles_n, dbg)
  if (K == 64) {
    if (bn == 256) CREID_ST_LAUNCH(256, 1, 3); else if (bn == 128) CREID_ST_LAUNCH(128, 1, 3); else CREID_ST_LAUNCH(64, 1, 3);
  } else if (K == 128) {
    if (bn == 256) CREID_ST_LAUNCH(256, 2, 3); else if (bn == 128) CREID_ST_LAUNCH(128, 2, 3); else CREID_ST_LAUNCH(64, 2, 3);
  } else {
    if (bn == 128) CREID_ST_LAUNCH(128, 4, 3); else CREID_ST_LAUNCH(64, 4, 4);
  }
#undef CREID_ST_LAUNCH
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------ second form (round 3)
// Same idea -- resident workgroups, weight slab in LDS once, A tiles streamed into a ring -- with the two things the first form
// lacked (profiles/r02_stream1x1_experiment.md: it lost on wide outputs because 256 threads stored alone and the epilogue ran
// serially after the MFMAs), and with the epilogues of the eval-mode forward:
//   * all eight waves are alike: wave (wr, wc) multiplies rows wr * 32 .. + 31 of the 128-row tile against the 32-column blocks
//     2 j + wc, stages them column-major (packed 8-byte LDS stores, read back through the transposing LDS read -- the copy-out of
//     igemm_bf16_ws_kernel) and all 512 threads copy out;
//   * everything a tile needs from memory is requested ONE TILE EARLIER and nothing is waited for inside the tile that issued
//     it: the A rows of tile t + 2 and the residual chunks of tile t + 1 are loaded into registers at the top of tile t, the A
//     registers of tile t + 1 (loaded a tile ago) go to the LDS ring there, and the finished 16-byte chunks of tile t - 1 are
//     stored there.  The operands come through REGISTERS and every load is branch-free: in the first version (LDS-DMA for
//     A, predicated residual loads and stores in the same waves) the compiler's counted waits did not survive the control
//     flow and it placed full vmcnt(0) waits right after the requests of the SAME tile -- every load and store sat on the
//     critical path (ablation: compute 32 + stores 27 + residual 28 of 77 us).  Now the only full waits stand where
//     everything outstanding is a tile old (ISA checked: put_a / the residual hand-over, then loads, then stores, then MFMAs);
//   * epilogues: BatchNorm column sums of the fp32 accumulators (training forward), or the folded eval-mode affine (+ ReLU)
//     with the block's residual added in the copy-out (the arithmetic of igemm_bf16_ws_kernel: identical bits).
// K = 64 / 128 / 256 (KCH = 1, 2, 4); BN = 64 / 128 / 256 columns per workgroup; RT = 2 ring slots, or 1 where LDS allows no more
// (K = 256 with 64-column slabs, K = 128 with 256-column slabs): the tile is then written into the slot between two barriers.
// AXF (round 6, creid_conv1x1_bnrelu_fwd): the A operand is the RAW output of the previous convolution and its BatchNorm + ReLU
// (y = max(x * scale[c] + shift[c], 0), the arithmetic of bn2d_apply_kernel) is applied on the way from the load registers to
// the LDS ring -- conv3 of a Bottleneck consuming conv2's raw output (modelling/backbones/resnet.py:73-78): the stand-alone
// apply pass (one read + one write of the tensor and a dependent launch) disappears.  The normalised tensor and its ReLU bits
// are still written once (the weight gradient and the BatchNorm backward read them), by the column-slab-0 workgroups, from the
// registers that feed the ring: same values, same bits as the apply kernel's.  A thread's channels are fixed (k-chunk kc,
// 16-byte chunk lcp): their (scale, shift) live in registers for the whole launch.
struct AXform {
  const float* scale_shift;            // [2][K]
  unsigned short* a_out;               // [M][K] normalised + ReLU, 16-bit
  uint8_t* mask_out;                   // [M * K / 8] ReLU bits
};

template <typename ET>
__device__ __forceinline__ uint4 axf_chunk(uint4 v, const float (&sc)[8], const float (&sh)[8], unsigned& bits) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned o[4];
  bits = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lo = fmaf(ET::lo(w[q]), sc[2 * q], sh[2 * q]) + 0.f;          // (+ 0.f: bn2d_apply_kernel's "+ residual" slot)
    float hi = fmaf(ET::hi(w[q]), sc[2 * q + 1], sh[2 * q + 1]) + 0.f;
    lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f);
    bits |= (lo > 0.f ? 1u : 0u) << (2 * q);
    bits |= (hi > 0.f ? 1u : 0u) << (2 * q + 1);
    o[q] = ET::pack2(lo, hi);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int BN, int KCH, int RT, typename ET = Bf16T, bool AXF = false>
__global__ __launch_bounds__(512, 2) void igemm1x1_stream2_kernel(const unsigned short* __restrict__ src, int M, int K, int N,
                                                                   const unsigned short* __restrict__ wgt,
                                                                   unsigned short* __restrict__ out, float* __restrict__ bn_part,
                                                                   const unsigned short* __restrict__ add_src,
                                                                   const float* __restrict__ epi_scale,
                                                                   const float* __restrict__ epi_shift, int epi_relu,
                                                                   int tiles_m, int tiles_n, int abl, AXform xf = AXform{nullptr, nullptr, nullptr}) {
  constexpr int TW = BN / 64;                          // 32-column blocks per wave
  constexpr int HB = BN > 128 ? 128 : BN;              // columns staged / copied out at a time
  constexpr int NH = BN / HB;
  constexpr int CPT = 128 + 4;                         // staging pitch: [HB columns][128 rows + 4]
  constexpr int CPR = HB / 8, NIT = (128 * CPR) / 512; // 16-byte chunks per row; chunks per thread and half
  constexpr int W_ELEMS = KCH * BN * 64, TILE_ELEMS = KCH * 128 * 64, RING_ELEMS = RT * TILE_ELEMS, STAGE_ELEMS = HB * CPT;
  constexpr int RED_ELEMS = 4 * 2 * BN * 2;            // fp32 [4 row waves][2][BN] in 2-byte units
  static_assert(2 * (W_ELEMS + RING_ELEMS + STAGE_ELEMS + RED_ELEMS) <= 160 * 1024, "LDS budget");
  static_assert(16 % CPR == 0 && (RT == 1 || RT == 2), "copy-out map / ring");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[W_ELEMS + RING_ELEMS + STAGE_ELEMS + RED_ELEMS];
  unsigned short* Ws = smem;                           // [KCH][BN][64]
  unsigned short* ring = smem + W_ELEMS;               // [RT][KCH][128][64]
  unsigned short* stage = ring + RING_ELEMS;           // [HB][CPT]
  float* red = reinterpret_cast<float*>(stage + STAGE_ELEMS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int l31 = lane & 31, kh = lane >> 5;
  const int lr8 = lane >> 3, lcp = lane & 7;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_stream_zero_page);

  const int groups = (int)gridDim.x / tiles_n;
  const int tile_n = (int)blockIdx.x % tiles_n, wg_in_group = (int)blockIdx.x / tiles_n;
  const int col0 = tile_n * BN;
  const int n_iter = wg_in_group < tiles_m ? (tiles_m - 1 - wg_in_group) / groups + 1 : 0;
  if (n_iter == 0) return;

  // weight slab -> LDS, once (row r of k-chunk kc at [kc][r][64], 16-byte chunks XOR-swizzled by (r >> 1) & 7)
  {
    constexpr int NW = KCH * (BN / 8);
    for (int i = wave; i < NW; i += 8) {
      const int kc = i / (BN / 8), r = (i - kc * (BN / 8)) * 8 + lr8;
      const uint4 v = *reinterpret_cast<const uint4*>(wgt + (int64_t)(col0 + r) * K + kc * 64 + lcp * 8);
      *reinterpret_cast<uint4*>(Ws + (kc * BN + r) * 64 + ((lcp ^ ((r >> 1) & 7)) << 3)) = v;
    }
  }
  // A tile of iteration `it`: KCH * 16 wave-loads of 8 rows x 128 B, two per wave and k-chunk, into registers ...
  uint4 areg[KCH][2];
  auto load_a = [&](int it) {
    const int row0 = (wg_in_group + it * groups) * 128;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = (wave + 8 * u) * 8 + lr8, m = row0 + r;
        // (branch-free: a row past M reads row M - 1 and is zeroed -- predicated loads would break the compiler's counted waits)
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (!CREID_ABL_ON(abl, 4)) v = *reinterpret_cast<const uint4*>(src + (int64_t)min(m, M - 1) * K + kc * 64 + lcp * 8);
        areg[kc][u] = m < M ? v : make_uint4(0u, 0u, 0u, 0u);
      }
  };
  // AXF: this thread's channels are kc * 64 + lcp * 8 .. + 7 for every tile
  float xsc[AXF ? KCH : 1][8], xsh[AXF ? KCH : 1][8];
  uint4 sreg[AXF ? KCH : 1][2];                                   // transformed chunks waiting for their global store
  unsigned smk[AXF ? KCH : 1][2];
  const bool side = AXF && xf.a_out != nullptr && tile_n == 0;
  if constexpr (AXF) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int k = 0; k < 8; k += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(xf.scale_shift + kc * 64 + lcp * 8 + k);
        const float4 b4 = *reinterpret_cast<const float4*>(xf.scale_shift + K + kc * 64 + lcp * 8 + k);
        xsc[kc][k] = a4.x; xsc[kc][k + 1] = a4.y; xsc[kc][k + 2] = a4.z; xsc[kc][k + 3] = a4.w;
        xsh[kc][k] = b4.x; xsh[kc][k + 1] = b4.y; xsh[kc][k + 2] = b4.z; xsh[kc][k + 3] = b4.w;
      }
  }
  // ... and from there into ring slot it % RT (same LDS image as the DMA kernels)
  auto put_a = [&](int it) {
    unsigned short* slot = ring + (it % RT) * TILE_ELEMS;
    const int row0 = (wg_in_group + it * groups) * 128;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = (wave + 8 * u) * 8 + lr8;
        uint4 v = areg[kc][u];
        if constexpr (AXF) {
          unsigned bits;
          const uint4 t = axf_chunk<ET>(v, xsc[kc], xsh[kc], bits);
          v = row0 + r < M ? t : make_uint4(0u, 0u, 0u, 0u);       // rows past M stay zero rows (relu(shift) is not zero)
          sreg[kc][u] = v; smk[kc][u] = bits;
        }
        *reinterpret_cast<uint4*>(slot + kc * (128 * 64) + r * 64 + ((lcp ^ ((r >> 1) & 7)) << 3)) = v;
      }
  };
  // the normalised tile (and its ReLU bits) leave from the same registers, issued where the finished output chunks are stored
  auto side_store = [&](int it) {
    if constexpr (AXF) {
      if (side) {
        const int row0 = (wg_in_group + it * groups) * 128;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int m = row0 + (wave + 8 * u) * 8 + lr8;
            if (m < M) {
              const int64_t e = (int64_t)m * K + kc * 64 + lcp * 8;
              *reinterpret_cast<uint4*>(xf.a_out + e) = sreg[kc][u];
              xf.mask_out[e >> 3] = (uint8_t)smk[kc][u];
            }
          }
      }
    }
  };
  // copy-out map of one half (see igemm_bf16_ws_kernel): lane = 16 g4 + 4 q4 + t4 owns row t4 of a row quad, one column octet
  const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
  auto unit_of = [&](int i, int& rl, int& ch) {
    const int Q = (wave + 8 * i) * 16 + g4 * 4 + q4;
    ch = Q % CPR;
    rl = 4 * (Q / CPR) + t4;
  };
  uint4 resn[NH][NIT];                                  // residual chunks of the NEXT tile (loaded one tile ahead)
  uint4 outv[NH][NIT];                                  // finished chunks of the PREVIOUS tile (stored one tile late)
  auto load_res = [&](int it) {
    const int row0 = (wg_in_group + it * groups) * 128;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        int rl, ch;
        unit_of(i, rl, ch);
        const int rr = row0 + rl;
        resn[h][i] = make_uint4(0u, 0u, 0u, 0u);
        if (!CREID_ABL_ON(abl, 2)) resn[h][i] = *reinterpret_cast<const uint4*>(add_src + (int64_t)min(rr, M - 1) * N + col0 + h * HB + ch * 8);
      }
  };
  auto store_out = [&](int it) {
    const int row0 = (wg_in_group + it * groups) * 128;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        int rl, ch;
        unit_of(i, rl, ch);
        const int rr = row0 + rl;
        if (rr < M && !CREID_ABL_ON(abl, 1)) *reinterpret_cast<uint4*>(out + (int64_t)rr * N + col0 + h * HB + ch * 8) = outv[h][i];
      }
  };

  load_a(0);
  if constexpr (RT == 2) {
    put_a(0);                                                     // tile 0 is in the ring before the first barrier
    side_store(0);
    if (n_iter > 1) load_a(1);
  }
  if (add_src) load_res(0);
  float sc[TW], sh[TW];
  if (epi_scale) {
#pragma unroll
    for (int j = 0; j < TW; ++j) { sc[j] = epi_scale[col0 + (2 * j + wc) * 32 + l31]; sh[j] = epi_shift[col0 + (2 * j + wc) * 32 + l31]; }
  }
  const bool relu_now = epi_relu && !add_src;

  for (int it = 0; it < n_iter; ++it) {
    const int tile_m = wg_in_group + it * groups;
    __syncthreads();                                              // RT = 2: tile `it` is in the ring; everyone is done with tile it - 1
    // first everything that CONSUMES loads issued a tile ago (whatever the wait before it, it is short) ...
    if constexpr (RT == 2) {
      if (it + 1 < n_iter) put_a(it + 1);                         // into the slot tile it - 1 just left
    } else {
      put_a(it);                                                  // one slot (K = 256 / wide slabs: LDS): this tile, loaded a tile ago
    }
    uint4 resc[NH][NIT];
    if (add_src) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < NIT; ++i) resc[h][i] = resn[h][i];
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(resc[h][i].x), "+v"(resc[h][i].y), "+v"(resc[h][i].z), "+v"(resc[h][i].w));
    // ... then this tile's requests: nothing below waits for them before the next trip (loads first: the compiler guards the
    // re-use of their destination registers with a full wait, which must not see the stores)
    if (it + RT < n_iter) load_a(it + RT);
    if (add_src && it + 1 < n_iter) load_res(it + 1);
    if (it > 0) store_out(it - 1);
    if constexpr (RT == 2) { if (it + 1 < n_iter) side_store(it + 1); } else side_store(it);
    if constexpr (RT == 1) __syncthreads();                       // the slot holds tile `it`
    f32x16 acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned short* At = ring + (it % RT) * TILE_ELEMS;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      const unsigned short* As = At + kc * (128 * 64);
      const unsigned short* Bs = Ws + kc * BN * 64;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int chk = 2 * kk + kh;
        const int r = wr * 32 + l31;
        const s16x8 a = *reinterpret_cast<const s16x8*>(&As[r * 64 + ((chk ^ ((r >> 1) & 7)) << 3)]);
        s16x8 b[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) {
          const int cc = (2 * j + wc) * 32 + l31;
          b[j] = *reinterpret_cast<const s16x8*>(&Bs[cc * 64 + ((chk ^ ((cc >> 1) & 7)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < TW; ++j)
          acc[j] = ET::mfma(a, b[j], acc[j]);
      }
    }
    if (bn_part) {                                                // column sums of the fp32 accumulators (rows >= M are zero)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[j][r]; s1 += v; s2 = fmaf(v, v, s2); }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const int cl = (2 * j + wc) * 32 + l31;
        if (kh == 0) { red[(wr * 2 + 0) * BN + cl] = s1; red[(wr * 2 + 1) * BN + cl] = s2; }
      }
    }
    if (CREID_ABL_ON(abl, 8)) continue;                           // (timing ablation: no staging / copy-out at all)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (h > 0) asm volatile("s_barrier" ::: "memory");          // the previous half has been read out of the staging area
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        if (((2 * j + wc) * 32) / HB != h) continue;              // (wave-uniform)
        const int cl = (2 * j + wc) * 32 - h * HB + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wr * 32 + 8 * q + 4 * kh;
          float v0 = acc[j][4 * q], v1 = acc[j][4 * q + 1], v2 = acc[j][4 * q + 2], v3 = acc[j][4 * q + 3];
          if (epi_scale) {
            v0 = fmaf(v0, sc[j], sh[j]); v1 = fmaf(v1, sc[j], sh[j]); v2 = fmaf(v2, sc[j], sh[j]); v3 = fmaf(v3, sc[j], sh[j]);
            if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          }
          *reinterpret_cast<uint2*>(&stage[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
      }
      __syncthreads();                                            // the half is staged (and `red` is complete)
      if (h == 0 && bn_part) {
        for (int i = tid; i < 2 * BN; i += 512) {
          const int which = i / BN, cl = i - which * BN;
          bn_part[((int64_t)tile_m * 2 + which) * N + col0 + cl] =
              (red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl]) + (red[(2 * 2 + which) * BN + cl] + red[(3 * 2 + which) * BN + cl]);
        }
      }
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
        if (add_src) {
          const uint4 a = resc[h][i];
          unsigned* vw = &v.x; const unsigned* aw = &a.x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float lo = ET::lo(vw[q]) + ET::lo(aw[q]);
            float hi = ET::hi(vw[q]) + ET::hi(aw[q]);
            if (epi_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            vw[q] = ET::pack2(lo, hi);
          }
        }
        outv[h][i] = v;
      }
    }
  }
  store_out(n_iter - 1);
}

// Returns CREID_E_SHAPE when the GEMM is outside the kernel's scope (the caller then uses conv_igemm.hip's kernels).
int launch_stream2(int M, int K, int N, const void* src, const void* wgt, void* out, float* bn_part, const void* add_src,
                   const float* epi_scale, const float* epi_shift, int epi_relu, int bn_cap, int dtype, hipStream_t s) {
  if (K != 64 && K != 128 && K != 256) return CREID_E_SHAPE;
  int bn = N >= 256 ? 256 : N;
  if (K == 256 && bn > 64) bn = 64;                    // LDS: weight slab + one 64 KB tile slot + staging
  if (bn != 64 && bn != 128 && bn != 256) return CREID_E_SHAPE;
  if (N % bn != 0) return CREID_E_SHAPE;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM2_BN"); const int v = e ? atoi(e) : 0; if (v == 64 || v == 128) bn_cap = v; }   // (experiments)
  if ((bn_cap == 64 || bn_cap == 128) && bn_cap < bn && N % bn_cap == 0) bn = bn_cap;
  const int tiles_m = (M + 127) / 128, tiles_n = N / bn;
  int wgs = 256;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) wgs = v; }   // read per call (tests)
  int groups = wgs / tiles_n;
  if (groups < 1) groups = 1;
  if (groups > tiles_m) groups = tiles_m;
  const dim3 grid((unsigned)(groups * tiles_n)), block(512);
#ifdef CREID_ABL_BUILD
  static const int abl = creid_ablation_env("CREID_STREAM2_ABL");   // 1 no stores, 2 no residual loads, 4 no A loads, 8 no copy-out
#else
  const int abl = 0;
#endif
#define CREID_ST2_LAUNCH1(BN_, KCH_, RT_, ET_)                                                                          \
  hipLaunchKernelGGL((igemm1x1_stream2_kernel<BN_, KCH_, RT_, ET_>), grid, block, 0, s, (const unsigned short*)src, M, K, N, \
                     (const unsigned short*)wgt, (unsigned short*)out, bn_part, (const unsigned short*)add_src, epi_scale, \
                     epi_shift, epi_relu, tiles_m, tiles_n, abl)
#define CREID_ST2_LAUNCH(BN_, KCH_, RT_) \
  do { if (dtype == CREID_F16) CREID_ST2_LAUNCH1(BN_, KCH_, RT_, F16T); else CREID_ST2_LAUNCH1(BN_, KCH_, RT_, Bf16T); } while (0)
  if (K == 64) {
    if (bn == 256) CREID_ST2_LAUNCH(256, 1, 2); else if (bn == 128) CREID_ST2_LAUNCH(128, 1, 2); else CREID_ST2_LAUNCH(64, 1, 2);
  } else if (K == 128) {
    if (bn == 256) CREID_ST2_LAUNCH(256, 2, 1); else if (bn == 128) CREID_ST2_LAUNCH(128, 2, 2); else CREID_ST2_LAUNCH(64, 2, 2);
  } else {
    CREID_ST2_LAUNCH(64, 4, 1);
  }
#undef CREID_ST2_LAUNCH
#undef CREID_ST2_LAUNCH1
  return (int)hipGetLastError();
}

// creid_conv1x1_bnrelu_fwd: the training forward of a 1 x 1 stride-1 convolution whose input is still RAW (BatchNorm + ReLU applied
// on the operand path, the normalised tensor and its ReLU bits written as side outputs) -- see AXform above.
static int launch_stream2_axf(int M, int K, int N, const void* src, const void* wgt, void* out, float* bn_part, AXform xf, int dtype,
                              hipStream_t s) {
  // K = 64 / 128 (layer1 / layer2 bottlenecks).  The transform's registers (16 coefficients and 10 pending-store registers per
  // k-chunk) do not fit beside the 256-column accumulators at K = 128, nor beside four k-chunks at K = 256: those shapes take
  // 128-column slabs / are refused (the caller keeps the stand-alone apply pass) rather than spill.
  if (K != 64 && K != 128) return CREID_E_SHAPE;
  int bn = N >= 256 ? 256 : N;
  if (K == 128 && bn > 128) bn = 128;
  if (bn != 64 && bn != 128 && bn != 256) return CREID_E_SHAPE;
  if (N % bn != 0) return CREID_E_SHAPE;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM2_BN"); const int v = e ? atoi(e) : 0; if ((v == 64 || v == 128) && v < bn && N % v == 0) bn = v; }
  const int tiles_m = (M + 127) / 128, tiles_n = N / bn;
  int wgs = 256;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) wgs = v; }
  int groups = wgs / tiles_n;
  if (groups < 1) groups = 1;
  if (groups > tiles_m) groups = tiles_m;
  const dim3 grid((unsigned)(groups * tiles_n)), block(512);
#define CREID_AXF_LAUNCH1(BN_, KCH_, RT_, ET_)                                                                          \
  hipLaunchKernelGGL((igemm1x1_stream2_kernel<BN_, KCH_, RT_, ET_, true>), grid, block, 0, s, (const unsigned short*)src, M, K, N, \
                     (const unsigned short*)wgt, (unsigned short*)out, bn_part, (const unsigned short*)nullptr, (const float*)nullptr, \
                     (const float*)nullptr, 0, tiles_m, tiles_n, 0, xf)
#define CREID_AXF_LAUNCH(BN_, KCH_, RT_) \
  do { if (dtype == CREID_F16) CREID_AXF_LAUNCH1(BN_, KCH_, RT_, F16T); else CREID_AXF_LAUNCH1(BN_, KCH_, RT_, Bf16T); } while (0)
  if (K == 64) {
    if (bn == 256) CREID_AXF_LAUNCH(256, 1, 2); else if (bn == 128) CREID_AXF_LAUNCH(128, 1, 2); else CREID_AXF_LAUNCH(64, 1, 2);
  } else {
    if (bn == 128) CREID_AXF_LAUNCH(128, 2, 2); else CREID_AXF_LAUNCH(64, 2, 2);
  }
#undef CREID_AXF_LAUNCH
#undef CREID_AXF_LAUNCH1
  return (int)hipGetLastError();
}

extern "C" int creid_conv1x1_bnrelu_fwd(const void* x_raw, const float* scale_shift, const void* w_krsc, int64_t M, int64_t K,
                                        int64_t N, void* y, float* bn_partial, void* a_out, uint8_t* mask_out, int dtype,
                                        void* stream) {
  CREID_CHECK_ARG(x_raw && scale_shift && w_krsc && y && M > 0 && K > 0 && N > 0 && ((a_out == nullptr) == (mask_out == nullptr)));
  if (!creid_is16(dtype)) return CREID_E_DTYPE;
  if (M > 0x7fffffff) return CREID_E_SHAPE;
  return launch_stream2_axf((int)M, (int)K, (int)N, x_raw, w_krsc, y, bn_partial, AXform{scale_shift, (unsigned short*)a_out, mask_out},
                            dtype, as_stream(stream));
}

// ------------------------------------------------------------------------------------ 3 x 3, 64 -> 64 (round 4)
// layer1's conv2 (modelling/backbones/resnet.py:58, Bottleneck.conv2 with planes = 64): M = B * 64 * 32 pixels, N = 64, K = 576.
// The tile kernels run it 4x above its HBM floor (B = 128: 44 us for 67 MB): every one of the nine taps re-fetches its 128-row
// A tile through the 64 B/clk operand path (9 x 16 KB per tile for 16 KB of distinct input), the 72 KB of weights are re-read by
// every workgroup-tile, and a 128 x 64 tile gives a wave 8 MFMAs per barrier.  Here
//   * the INPUT HALO TILE is staged in LDS once per tile: the 128 / W + 2 image rows a tile of 128 consecutive output pixels
//     touches, W + 2 pixel slots per row (zero pixels left and right; rows outside the image are zero rows), and all nine taps
//     read their A fragments from it at shifted addresses (slot = (oy + r) * (W + 2) + ox + s) -- no per-tap fetch, no bounds logic
//     in the multiply loop;
//   * the WEIGHTS live in REGISTERS: wave (wr, wc) always multiplies against output channels wc * 32 .. + 31, i.e. the same 36
//     B fragments (9 taps x 4 k-slices x 16 B per lane = 144 registers) for every tile of the launch -- LDS carries only A;
//   * workgroups are persistent over a CONTIGUOUS run of tiles (vertically adjacent tiles share two input rows: the re-read
//     hits the workgroup's own L2), the next tile's rows are loaded into registers a tile ahead and finished tiles are stored a
//     tile late (the scheme of igemm1x1_stream2_kernel).
// Same k order as the tile kernels (tap-major, 16-wide slices inside a tap) and the same epilogue arithmetic: identical output
// bits; statistics partials equal up to the grouping of the column sums.  Forward only (training statistics / folded eval-mode
// affine / plain); W = 32 or 64 with H * W % 128 = 0; anything else takes the tile kernels.
// T2D (image widths that are no divisor / multiple of 128 pixels, e.g. the 80 x 80 maps of the 320 x 320 configuration): the tile
// is TW = 16 columns x 8 rows of the image instead of 128 consecutive pixels, the halo tile carries its left and right columns
// too (zero outside the image), the image width is a run-time value Wrt; everything behind the staging is unchanged.
template <int TW, typename ET, bool T2D>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_kernel(const unsigned short* __restrict__ src, int M, int H, int Wrt,
                                                              const unsigned short* __restrict__ wgt,
                                                              unsigned short* __restrict__ out, float* __restrict__ bn_part,
                                                              const float* __restrict__ epi_scale, const float* __restrict__ epi_shift,
                                                              int epi_relu, int n_tiles, int abl) {
  constexpr int W = TW;                                 // (full-row tiles: the image width)
  constexpr int RW = 128 / TW;                          // output rows per tile
  constexpr int PW = TW + 2, PX = (RW + 2) * PW;        // pixel slots per halo row / per halo tile
  constexpr int SLOT = PX * 64;                         // elements per ring slot
  constexpr int NLD = T2D ? (PX * 8 + 511) / 512 : ((RW + 2) * W * 8) / 512;   // 16-byte chunks per thread and halo tile
  constexpr int CPT = 128 + 4, CPR = 8, NIT = 2;        // staging pitch; 16-byte chunks per output row; chunks per thread
  static_assert(T2D || ((RW + 2) * W * 8) % 512 == 0, "halo tile load map");
  constexpr int WP = 576 + 8;                           // weight row pitch in the one-time LDS image (conflict-free 16-byte reads)
  __shared__ __attribute__((aligned(1024))) unsigned short smem[2 * SLOT + 64 * CPT + 4 * 2 * 64 * 2 + 64 * WP];
  unsigned short* ring = smem;
  unsigned short* stage = smem + 2 * SLOT;
  float* red = reinterpret_cast<float*>(stage + 64 * CPT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int l31 = lane & 31, kh = lane >> 5;
  // contiguous run of tiles per workgroup
  const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(n_tiles, t_begin + per);
  const int n_iter = t_end - t_begin;
  if (n_iter <= 0) return;

  // zero both ring slots once: the pad pixels are never written again
  for (int i = tid; i < 2 * SLOT / 8; i += 512) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0u, 0u, 0u, 0u);

  // the wave's 36 weight fragments: output channel wc * 32 + l31, tap-major k, 8 consecutive channels (2 kk + kh) * 8 ..  The 72 KB
  // come in ONCE per workgroup with coalesced 16-byte loads and go through LDS (fragment-shaped global loads -- a different 1152-byte
  // row per lane -- cost more than four tiles of multiplies)
  constexpr int TREG = 5;                               // taps whose weight fragments live in registers (80); the rest are read from LDS
  s16x8 bw[TREG][4];
  unsigned short* wl = reinterpret_cast<unsigned short*>(red + 4 * 2 * 64);
  {
    for (int i = tid; i < 64 * 72; i += 512) {
      const int o = i / 72, c = i - o * 72;
      *reinterpret_cast<uint4*>(wl + o * WP + c * 8) = *reinterpret_cast<const uint4*>(wgt + (int64_t)o * 576 + c * 8);
    }
    __syncthreads();
    const unsigned short* wrow = wl + (wc * 32 + l31) * WP + kh * 8;
#pragma unroll
    for (int tap = 0; tap < TREG; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bw[tap][kk] = *reinterpret_cast<const s16x8*>(wrow + tap * 64 + kk * 16);
  }
  const unsigned wl_lane = (unsigned)(uintptr_t)(wl + (wc * 32 + l31) * WP + kh * 8);   // this lane's weight row in the LDS image
  const int Wimg = T2D ? Wrt : W;
  const int tiles_x = T2D ? Wimg / TW : 1;
  const int tiles_per_img = (H * Wimg) / 128;
  // tile -> image, first output row, first output column
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    b = tile / tiles_per_img;
    const int t = tile - b * tiles_per_img;
    if constexpr (T2D) { const int ty = t / tiles_x; y0 = ty * RW; x0 = (t - ty * tiles_x) * TW; }
    else { y0 = t * RW; x0 = 0; }
  };
  uint4 areg[NLD];
  auto load_a = [&](int it) {
    int b, y0, x0;
    tile_origin(t_begin + it, b, y0, x0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 512 * u, ch = idx & 7, pix = idx >> 3;
      if constexpr (T2D) {
        const int row = pix / PW, col = pix - row * PW;
        const int y = y0 - 1 + row, x = x0 - 1 + col;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)Wimg && pix < PX;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (!CREID_ABL_ON(abl, 2)) v = *reinterpret_cast<const uint4*>(src + ((int64_t)(b * H + (ok ? y : 0)) * Wimg + (ok ? x : 0)) * 64 + ch * 8);
        areg[u] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      } else {
        const int row = pix / W, col = pix - row * W;
        const int y = y0 - 1 + row;
        const bool ok = (unsigned)y < (unsigned)H;
        // (branch-free: a row outside the image reads row 0 of the image and is zeroed -- see igemm1x1_stream2_kernel)
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (!CREID_ABL_ON(abl, 2)) v = *reinterpret_cast<const uint4*>(src + ((int64_t)(b * H + (ok ? y : 0)) * W + col) * 64 + ch * 8);
        areg[u] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto put_a = [&](int it) {
    unsigned short* slot = ring + (it & 1) * SLOT;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + 512 * u, ch = idx & 7, pix = idx >> 3;
      int p;
      if constexpr (T2D) { p = pix; if (pix >= PX) continue; }
      else { const int row = pix / W, col = pix - row * W; p = row * PW + col + 1; }
      *reinterpret_cast<uint4*>(slot + p * 64 + ((ch ^ ((p >> 1) & 7)) << 3)) = areg[u];
    }
  };
  const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
  auto unit_of = [&](int i, int& rl, int& ch) {
    const int Q = (wave + 8 * i) * 16 + g4 * 4 + q4;
    ch = Q % CPR;
    rl = 4 * (Q / CPR) + t4;
  };
  uint4 outv[NIT];
  auto store_out = [&](int it) {
    const int64_t row0 = (int64_t)(t_begin + it) * 128;
    int b = 0, y0 = 0, x0 = 0;
    if constexpr (T2D) tile_origin(t_begin + it, b, y0, x0);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      int rl, ch;
      unit_of(i, rl, ch);
      int64_t pixel = row0 + rl;
      if constexpr (T2D) pixel = ((int64_t)b * H + y0 + rl / TW) * Wimg + x0 + (rl % TW);
      if (!CREID_ABL_ON(abl, 8)) *reinterpret_cast<uint4*>(out + pixel * 64 + ch * 8) = outv[i];
    }
  };
  float sc = 1.f, sh = 0.f;
  if (epi_scale) { sc = epi_scale[wc * 32 + l31]; sh = epi_shift[wc * 32 + l31]; }
  // this lane's output pixel inside the tile and its tap (0, 0) halo slot
  const int ml = wr * 32 + l31, oy = ml / W, ox = ml - oy * W;
  const int p00 = oy * PW + ox;

  __syncthreads();                                                // ring zeroed
  load_a(0);
  put_a(0);
  if (n_iter > 1) load_a(1);
  for (int it = 0; it < n_iter; ++it) {
    __syncthreads();                                              // tile `it` is in the ring; everyone is done with tile it - 1
    if (it + 1 < n_iter) put_a(it + 1);
    // stores BEFORE the next loads: the one full vmcnt wait of a tile (the compiler's, in front of put_a above) then sees only
    // requests that are a whole tile old; the barrier in the middle of the tile must not wait for memory at all
    if (it > 0) store_out(it - 1);
    if (it + 2 < n_iter) load_a(it + 2);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned short* At = ring + (it & 1) * SLOT;
    // 36 multiply steps; the A fragment of step f + 3 (and, from tap TREG on, its weight fragment from the LDS image) is requested
    // before step f multiplies: asm reads with counted waits -- left to the compiler, 144 registers of weights made it re-use one
    // register quad and wait a full LDS round trip per MFMA (6100 of a tile's 7400 cycles), and with all nine taps in registers
    // there is no room for a pipeline at all
    {
      typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
      u32x4v af[4], bf[4];
      auto rd = [&](int f, u32x4v& da, u32x4v& db) {
        const int tap = f >> 2, kk = f & 3;
        const int p = p00 + (tap / 3) * PW + (tap % 3);
        const unsigned addr = (unsigned)(uintptr_t)(At + p * 64 + (((2 * kk + kh) ^ ((p >> 1) & 7)) << 3));
        asm volatile("ds_read_b128 %0, %1" : "=v"(da) : "v"(addr) : "memory");
        if (tap >= TREG) asm volatile("ds_read_b128 %0, %1" : "=v"(db) : "v"(wl_lane + (unsigned)((tap * 64 + kk * 16) * 2)) : "memory");
      };
      auto nrd = [](int f) constexpr { return f < 36 ? 1 + ((f >> 2) >= TREG ? 1 : 0) : 0; };
      if (!CREID_ABL_ON(abl, 1)) {
      rd(0, af[0], bf[0]); rd(1, af[1], bf[1]); rd(2, af[2], bf[2]);
#pragma unroll
      for (int f = 0; f < 36; ++f) {
        if (f + 3 < 36) rd(f + 3, af[(f + 3) & 3], bf[(f + 3) & 3]);
        const int young = nrd(f + 1) + nrd(f + 2) + nrd(f + 3);      // reads younger than step f's
        u32x4v& a = af[f & 3]; u32x4v& b = bf[f & 3];
        if (young == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) :: "memory");
        else if (young == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a), "+v"(b) :: "memory");
        else if (young == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b) :: "memory");
        else if (young == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a), "+v"(b) :: "memory");
        else if (young == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b) :: "memory");
        else if (young == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a), "+v"(b) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a), "+v"(b) :: "memory");
        if ((f >> 2) < TREG) acc = ET::mfma(__builtin_bit_cast(s16x8, a), bw[(f >> 2) < TREG ? (f >> 2) : 0][f & 3], acc);
        else acc = ET::mfma(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc);
      }
      }
    }
    if (bn_part) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float v = acc[r]; s1 += v; s2 = fmaf(v, v, s2); }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (kh == 0) { red[(wr * 2 + 0) * 64 + wc * 32 + l31] = s1; red[(wr * 2 + 1) * 64 + wc * 32 + l31] = s2; }
    }
    if (CREID_ABL_ON(abl, 4)) continue;                           // (timing ablation: no staging / copy-out)
    {
      const int cl = wc * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = wr * 32 + 8 * q + 4 * kh;
        float v0 = acc[4 * q], v1 = acc[4 * q + 1], v2 = acc[4 * q + 2], v3 = acc[4 * q + 3];
        if (epi_scale) {
          v0 = fmaf(v0, sc, sh); v1 = fmaf(v1, sc, sh); v2 = fmaf(v2, sc, sh); v3 = fmaf(v3, sc, sh);
          if (epi_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        }
        *reinterpret_cast<uint2*>(&stage[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the tile is staged (and `red` is complete): LDS only --
                                                                  // __syncthreads() would also wait for this tile's loads / stores
    if (bn_part) {
      for (int i = tid; i < 2 * 64; i += 512) {
        const int which = i / 64, cl = i - which * 64;
        bn_part[((int64_t)(t_begin + it) * 2 + which) * 64 + cl] =
            (red[(0 * 2 + which) * 64 + cl] + red[(1 * 2 + which) * 64 + cl]) + (red[(2 * 2 + which) * 64 + cl] + red[(3 * 2 + which) * 64 + cl]);
      }
    }
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
    for (int i = 0; i < NIT; ++i) outv[i] = make_uint4(trlo[i].x, trlo[i].y, trhi[i].x, trhi[i].y);
  }
  store_out(n_iter - 1);
}

// Returns CREID_E_SHAPE when the convolution is outside the kernel's scope.
int launch_conv3x3_c64(int M, int H, int Wd, const void* src, const void* wgt, void* out, float* bn_part, const float* epi_scale,
                       const float* epi_shift, int epi_relu, int dtype, hipStream_t s) {
  const bool full_rows = (Wd == 32 || Wd == 64) && H > 0 && (H * Wd) % 128 == 0;
  const bool tile2d = !full_rows && Wd > 0 && Wd % 16 == 0 && H > 0 && H % 8 == 0;      // 16 x 8 tiles (80 x 80: 50 per image)
  if ((!full_rows && !tile2d) || M % (H * Wd) != 0 || !creid_is16(dtype)) return CREID_E_SHAPE;
  const int n_tiles = M / 128;
  int wgs = 256;
  { const char* e = CREID_KNOB_ENV("CREID_STREAM1X1_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) wgs = v; }   // read per call (tests)
  if (wgs > n_tiles) wgs = n_tiles;
  const dim3 grid((unsigned)wgs), block(512);
#ifdef CREID_ABL_BUILD
  const char* ae = CREID_KNOB_ENV("CREID_C64_ABL");             // 1 no multiplies, 2 no input loads, 4 no staging / copy-out, 8 no stores
  const int abl = ae ? atoi(ae) : 0;
#else
  const int abl = 0;
#endif
#define CREID_C64_LAUNCH(W_, ET_, T2D_)                                                                             \
  hipLaunchKernelGGL((conv3x3_c64_kernel<W_, ET_, T2D_>), grid, block, 0, s, (const unsigned short*)src, M, H, Wd,   \
                     (const unsigned short*)wgt, (unsigned short*)out, bn_part, epi_scale, epi_shift, epi_relu, n_tiles, abl)
  if (tile2d) { if (dtype == CREID_F16) CREID_C64_LAUNCH(16, F16T, true); else CREID_C64_LAUNCH(16, Bf16T, true); }
  else if (Wd == 32) { if (dtype == CREID_F16) CREID_C64_LAUNCH(32, F16T, false); else CREID_C64_LAUNCH(32, Bf16T, false); }
  else { if (dtype == CREID_F16) CREID_C64_LAUNCH(64, F16T, false); else CREID_C64_LAUNCH(64, Bf16T, false); }
#undef CREID_C64_LAUNCH
  return (int)hipGetLastError();
}
