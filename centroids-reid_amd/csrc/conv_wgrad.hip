// Stage A: convolution weight gradient  dW[n, (tap, c)] = sum_m dY[m, n] * Xg[m, (tap, c)]
// (replaces nn.Conv2d wgrad of modelling/backbones/resnet.py:56-61,94,109).
//
// Both operands are stored PIXEL-major (the reduction index m is the row index of dY and X), so
// this is a "TN" GEMM.  bf16: the tiles land in LDS exactly as they lie in memory ([pixel][channel])
// and the MFMA fragments (8 consecutive pixels of one channel per lane) come out of gfx950's
// transposing LDS read `ds_read_b64_tr_b16` -- no register or memory transpose.
//   wgrad_bf16_dma_kernel (all 1x1 / 3x3 layers): both tiles are fetched with the global->LDS DMA into linear
//     rows, bank conflicts avoided by an XOR swizzle of 64-byte units applied on the DMA source column; the
//     gathered operand's pixel coordinates advance incrementally (64 pixels per k-step); fragment reads run one
//     16-pixel slice ahead of the MFMAs.
//   wgrad_bf16_kernel (stem, span 32): register-staged 16-B stores, row pitch = tile width + 32 elements.
// f32 (parity mode): K-major LDS is the natural layout for v_mfma_f32_32x32x2_f32.
// The pixel range is split over gridDim.y workgroups; fp32 partial tiles go to the workspace and
// wgrad_reduce_kernel<SL> sums them in a fixed order (deterministic) and scatters into the OIHW fp32 gradient.
#include "conv_common.hpp"
#include "wgrad_reduce.hpp"
#include "bn_fin.hpp"
#include "tune.hpp"
#include <stdlib.h>

namespace {
constexpr int WKS = 64;   // pixels per k-step (bf16)
constexpr int WKF = 16;   // pixels per k-step (f32)
}

// Two MFMA fragments (channel offsets +0 / +32) x two 4-pixel halves from one base address.
// The loads are issued in one asm statement; the matching s_waitcnt statement below names every
// destination "+v" so no consumer can be scheduled before the data has landed (hipcc does not
// count inline-asm LDS reads itself).
template <int OFF_HI>
__device__ __forceinline__ void tr_load4(const unsigned short* p, s16x4& f0lo, s16x4& f0hi, s16x4& f1lo, s16x4& f1hi) {
  const unsigned addr = (unsigned)(uintptr_t)p;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\t"
      "ds_read_b64_tr_b16 %1, %4 offset:%5\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:64\n\t"
      "ds_read_b64_tr_b16 %3, %4 offset:%6"
      : "=&v"(f0lo), "=&v"(f0hi), "=&v"(f1lo), "=&v"(f1hi)
      : "v"(addr), "i"(OFF_HI), "i"(OFF_HI + 64)
      : "memory");
}
__device__ __forceinline__ void tr_wait8(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
}

template <int NW>
__device__ __forceinline__ void tr_wait8_n(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(NW) : "memory");
}

// s_waitcnt lgkmcnt(NW) naming the 2 * (IM + JN) destination registers of one fragment set (IM + JN transposing read pairs)
template <int NW, int IM, int JN>
__device__ __forceinline__ void tr_wait_set(s16x4 (&al)[IM], s16x4 (&ah)[IM], s16x4 (&bl)[JN], s16x4 (&bh)[JN]) {
  if constexpr (IM == 2 && JN == 2)
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]) : "n"(NW) : "memory");
  else if constexpr (IM == 2 && JN == 1)
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(bl[0]), "+v"(bh[0]) : "n"(NW) : "memory");
  else if constexpr (IM == 1 && JN == 2)
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(al[0]), "+v"(ah[0]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]) : "n"(NW) : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(al[0]), "+v"(ah[0]), "+v"(bl[0]), "+v"(bh[0]) : "n"(NW) : "memory");
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(IGemmGeom g, const unsigned short* __restrict__ dy,
                                                         const unsigned short* __restrict__ x, int NCO,
                                                         float* __restrict__ ws, int tiles_k, int m_per_split) {
  constexpr int PA = TM + 32, PB = TN + 32;                 // LDS row pitches (elements)
  constexpr int IM = TM / 64, JN = TN / 64;                 // MFMA tiles per wave
  constexpr int CA = TM / 8, CB = TN / 8;                   // 16-B chunks per row
  constexpr int RA = 256 / CA, RB = 256 / CB;               // rows per load pass
  constexpr int NA = WKS / RA, NBL = WKS / RB;              // load passes
  __shared__ __attribute__((aligned(16))) unsigned short Ad[WKS * PA];
  __shared__ __attribute__((aligned(16))) unsigned short Bx[WKS * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int tile_co = tile / tiles_k, tile_k = tile - tile_co * tiles_k;
  const int co0 = tile_co * TM, k0 = tile_k * TN;
  const int m_begin = split * m_per_split, m_end = min(g.M, m_begin + m_per_split);
  const int span_mask = (1 << g.log2span) - 1;

  // load maps
  const int a_ch = tid % CA, a_row = tid / CA;
  const int b_ch = tid % CB, b_row = tid / CB;
  const int kcol = k0 + 8 * b_ch;
  const int tap = kcol >> g.log2span, cc = kcol & span_mask;
  const int tr = tap / g.kw, ts = tap - tr * g.kw;
  // 3-deep register prefetch ring (first-class vector type so the ring stays in VGPRs)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  struct Stage { u32x4 a[NA]; u32x4 b[NBL]; };
  Stage st0, st1, st2;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define WG_GLOAD(MB_, ST)                                                                                \
  do {                                                                                                   \
    const int mb_ = (MB_);                                                                               \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                     \
      const int m = mb_ + a_row + RA * i;                                                                \
      ST.a[i] = (m < m_end) ? *reinterpret_cast<const u32x4*>(dy + (int64_t)m * NCO + co0 + 8 * a_ch) : zero4; \
    }                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NBL; ++i) {                                                    \
      const int m = mb_ + b_row + RB * i;                                                                \
      ST.b[i] = zero4;                                                                                   \
      if (m < m_end) {                                                                                   \
        int b, rem, oy, ox;                                                                              \
        fast_divmod(m, g.OH * g.OW, g.inv_ohow, b, rem);                                                 \
        fast_divmod(rem, g.OW, g.inv_ow, oy, ox);                                                        \
        int iy, ix;                                                                                      \
        if (igemm_src_pixel(g, oy, ox, tr, ts, iy, ix))                                                  \
          ST.b[i] = *reinterpret_cast<const u32x4*>(x + (int64_t)((b * g.SH + iy) * g.SW + ix) * g.pitch + cc); \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
#define WG_LSTORE(ST)                                                                                    \
  do {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                       \
      *reinterpret_cast<u32x4*>(&Ad[(a_row + RA * i) * PA + 8 * a_ch]) = ST.a[i];                        \
    _Pragma("unroll") for (int i = 0; i < NBL; ++i)                                                      \
      *reinterpret_cast<u32x4*>(&Bx[(b_row + RB * i) * PB + 8 * b_ch]) = ST.b[i];                        \
  } while (0)

  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposing-read lane map: this lane supplies (row = 8*(lane>>5) + (li>>2) [+4q] [+16kk], col = 16*((lane>>4)&1) + 4*(li&3))
  const int li = lane & 15;
  const int t_row = 8 * (lane >> 5) + (li >> 2), t_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);

  // loads past m_end are predicated to zero, so the ring can always run 3 steps ahead
  WG_GLOAD(m_begin, st0);
  WG_GLOAD(m_begin + WKS, st1);
  WG_GLOAD(m_begin + 2 * WKS, st2);
  for (int mb = m_begin; mb < m_end; mb += WKS) {
    __syncthreads();            // previous step's fragment reads are done
    WG_LSTORE(st0);
    __syncthreads();
    st0 = st1; st1 = st2;
    WG_GLOAD(mb + 3 * WKS, st2);
#pragma unroll
    for (int kk = 0; kk < WKS / 16; ++kk) {
      s16x4 al[2], ah[2], bl[2], bh[2];
      tr_load4<4 * PA * 2>(&Ad[(kk * 16 + t_row) * PA + wm * (TM / 2) + t_col], al[0], ah[0], al[1], ah[1]);
      tr_load4<4 * PB * 2>(&Bx[(kk * 16 + t_row) * PB + wn * (TN / 2) + t_col], bl[0], bh[0], bl[1], bh[1]);
      tr_wait8(al[0], ah[0], al[1], ah[1], bl[0], bh[0], bl[1], bh[1]);
      __builtin_amdgcn_sched_barrier(0);
      s16x8 a[IM], b[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) a[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < JN; ++j) b[j] = __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                              __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
  }
#undef WG_GLOAD
#undef WG_LSTORE
  // partial tile -> workspace [split][NCO][K]
  const int l31 = lane & 31, kh = lane >> 5;
  float* wsp = ws + (int64_t)split * NCO * g.K;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int kc = k0 + wn * (TN / 2) + j * 32 + l31;
        wsp[(int64_t)co * g.K + kc] = acc[i][j][r];
      }
}

// ------------------------------------------------------------------------------------ bf16, LDS-DMA
// Same math as wgrad_bf16_kernel; both pixel-major tiles are fetched with the global->LDS DMA into LINEAR rows
// (pitch = tile width), double-buffered.  Bank conflicts of the transposing reads are avoided by an XOR
// swizzle of the 64-byte units of a row (key = row&3 for 256-B rows, (row>>1)&1 for 128-B rows: the four
// pixel rows a 16-lane group touches land on four different 16-bank ranges), applied on the DMA SOURCE column
// and again in the per-lane read address.  Rows past the split / out-of-image taps read a page of zeros.
__device__ __attribute__((aligned(128))) unsigned g_wgrad_zero_page[32];

template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void tr_load2(unsigned addr, s16x4& lo, s16x4& hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(OFF_LO), "i"(OFF_HI) : "memory");
}

template <int N> __device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NS-stage LDS ring (see igemm_bf16_dma_kernel: with two stages the loop runs at global->LDS latency).
// WS = true: 512 threads, waves 4-7 issue the DMA (producers), waves 0-3 read fragments and multiply (consumers),
// one s_barrier per k-tile between them -- the split of igemm_bf16_ws_kernel.
// KG = 2 (non-WS): 512 threads = two k-groups of four waves.  Each group runs the whole pipeline (its own LDS ring, its own DMA
// and fragment reads) over HALF of the split's pixel range on the SAME output tile; at the end group 1 hands its accumulators to
// group 0 through LDS (fixed order: deterministic) and one partial tile leaves the CU.  Same waves per CU as two workgroups of
// a split twice as fine, half the fp32 partial-tile traffic (write here + re-read by the split reduction).
// ET = element type traits (common.hpp Bf16T / F16T): the operands move as raw 16-bit words (DMA, transposing LDS reads), so the
// element type only selects the MFMA instruction -- f16 is the reference's own mixed precision (utils/misc.py:111, precision=16)
template <int TM, int TN, int NS, bool WS = false, int KG = 1, typename ET = Bf16T>
__global__ __launch_bounds__(WS ? 512 : 256 * KG, (KG * NS * (TM + TN) * 128 <= 80 * 1024) ? ((WS || KG == 2) ? 4 : 2) : ((WS || KG == 2) ? 2 : 1)) void wgrad_bf16_dma_kernel(IGemmGeom g, const unsigned short* __restrict__ dy,
                                                                 const unsigned short* __restrict__ x, int NCO,
                                                                 float* __restrict__ ws, int tiles_k, int m_per_split,
                                                                 int xcd_tiles, int xcd_splits, BnBwdFinJob fin, int abl) {
  constexpr int IM = TM / 64, JN = TN / 64;
  constexpr int LPR_A = TM / 8, LPR_B = TN / 8;                // lanes (16-B chunks) per tile row
  constexpr int RPI_A = 64 / LPR_A, RPI_B = 64 / LPR_B;        // rows per wave-instruction
  constexpr int NIA = WKS / (4 * RPI_A), NIB = WKS / (4 * RPI_B);   // DMA instructions per wave per tile
  constexpr int TILE_A = WKS * TM, TILE_B = WKS * TN, STAGE = TILE_A + TILE_B;
  constexpr int LPT = NIA + NIB;
  static_assert((NS - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
  static_assert(KG == 1 || (!WS && KG == 2 && KG * NS * STAGE * 2 <= 160 * 1024 && TM * TN * 2 <= KG * NS * STAGE), "k-groups");
  __shared__ __attribute__((aligned(1024))) unsigned short smem_all[KG * NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;       // role-local wave index
  const int kg = KG > 1 ? (tid >> 8) : 0;                                    // k-group of this wave
  unsigned short* smem = smem_all + kg * NS * STAGE;                         // the group's own ring
  const bool producer = !WS || (tid >> 6) >= 4, consumer = !WS || (tid >> 6) < 4;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware split placement (xcd_tiles > 0, 1-D grid): workgroup b runs on XCD b % 8, and every XCD has its own
  // 4 MB L2.  A pixel range (split) is read by ALL tiles of the weight matrix, so all workgroups of a split go to
  // ONE XCD (split % 8 == XCD): the range's dY / X rows are fetched from HBM once and then re-read from that L2.
  // With the tile-major placement each XCD touched every pixel range and the operand re-reads (35x the unique bytes
  // for a 512 -> 512 3x3 layer) all missed to the fabric.
  // Fewer than 8 splits (xcd_splits in {1, 2, 4}): the 8 / splits XCDs that share a pixel range interleave its tiles.
  // carried job (bn_fin.hpp): the first workgroups (a multiple of 8, so the tiles keep their XCD) finalize the BatchNorm
  // backward that the NEXT launch in the stream needs; 1-D grid only
  int bid0 = (int)blockIdx.x;
  if (fin.partial) {
    const int nf8 = (fin.nblocks + 7) & ~7;
    if (bid0 < nf8) {
      if (bid0 < fin.nblocks) bn_bwd_finalize_block<WS ? 512 : 256 * KG>(fin, bid0, smem_all);
      return;
    }
    bid0 -= nf8;
  }
  int tile, split;
  if (xcd_tiles > 0 && xcd_splits >= 8) {
    const int bid = bid0, j = bid >> 3, ls = j / xcd_tiles;
    tile = j - ls * xcd_tiles;
    split = (bid & 7) + 8 * ls;
    if (split >= xcd_splits) return;
  } else if (xcd_tiles > 0) {
    const int bid = bid0, x8 = bid & 7, share = 8 / xcd_splits;
    split = x8 % xcd_splits;
    tile = x8 / xcd_splits + share * (bid >> 3);
    if (tile >= xcd_tiles) return;
  } else {
    tile = blockIdx.x; split = blockIdx.y;
  }
  const int tile_co = tile / tiles_k, tile_k = tile - tile_co * tiles_k;
  const int co0 = tile_co * TM, k0 = tile_k * TN;
  // the split's pixel range, cut into KG runs of nt k-steps (every group executes nt barriers; a run past the range reads zeros)
  const int m_begin0 = split * m_per_split, m_end0 = min(g.M, m_begin0 + m_per_split);
  const int nt = ((m_end0 - m_begin0 + WKS - 1) / WKS + KG - 1) / KG;
  const int m_begin = m_begin0 + kg * nt * WKS, m_end = min(m_end0, m_begin + nt * WKS);
  const int span_mask = (1 << g.log2span) - 1;
  const int tap = k0 >> g.log2span, cc = k0 & span_mask;
  const int tr = tap / g.kw, ts = tap - tr * g.kw;
  const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_wgrad_zero_page);

  // DMA lane maps (row within the instruction's row group, swizzled source column)
  const int ar = lane / LPR_A, aq = lane % LPR_A;
  const int br = lane / LPR_B, bq = lane % LPR_B;
  int arow[NIA], acol[NIA], brow[NIB], bcol[NIB];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    arow[i] = (i * 4 + wave) * RPI_A + ar;
    const int key = (TM == 128) ? (arow[i] & 3) : ((arow[i] >> 1) & 1);
    acol[i] = ((((aq >> 2) ^ key) << 2) + (aq & 3)) << 3;
  }
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    brow[i] = (i * 4 + wave) * RPI_B + br;
    const int key = (TN == 128) ? (brow[i] & 3) : ((brow[i] >> 1) & 1);
    bcol[i] = ((((bq >> 2) ^ key) << 2) + (bq & 3)) << 3;
  }
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // Pixel coordinates of this lane's B rows.  A k-step advances every row by WKS = 64 pixels: when 64 is a
  // multiple of OW (all ResNet shapes) that is "ox unchanged, oy += 64/OW, carry into the image index", a
  // handful of integer ops instead of two reciprocal divmods per DMA instruction per k-step.
  const bool inc_ok = (WKS % g.OW) == 0 && (WKS / g.OW) <= g.OH && g.check_bounds != 2;
  const int dy_rows = inc_ok ? WKS / g.OW : 0;
  int pb[NIB], poy[NIB], pox[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int m = min(m_begin + brow[i], g.M - 1);
    int rem;
    fast_divmod(m, g.OH * g.OW, g.inv_ohow, pb[i], rem);
    fast_divmod(rem, g.OW, g.inv_ow, poy[i], pox[i]);
  }
  // LINEAR fast path (every 1x1 / 3x3 layer of the backbone): with 64 | OW-steps and SH == OH*stride the gathered
  // source pixel of a row advances by a CONSTANT number of elements per k-step -- image wrap included -- so both
  // operands keep running pointers (one 64-bit add per DMA instruction) and only the row's validity
  // (split end, vertical bound; the horizontal bound never changes) is re-evaluated.  PMC before this change:
  // 150 VALU + 164 SALU per k-step and wave against 16 MFMAs.
  const bool linear = inc_ok && g.log2span >= 6 && !g.transposed && g.SH == g.OH * g.stride && g.SW == g.OW * g.stride;
  const unsigned short* pa[NIA];
  const unsigned short* pbx[NIB];
  int ma[NIA], mbr[NIB], iy0[NIB];
  bool xok[NIB];
  const int a_inc = WKS * NCO, b_inc = dy_rows * g.stride * g.SW * g.pitch;
  if (linear) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      ma[i] = m_begin + arow[i];
      pa[i] = dy + (int64_t)ma[i] * NCO + co0 + acol[i];
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      mbr[i] = m_begin + brow[i];
      int bb, rem, oy, ox;                                      // un-clamped coordinates of the (possibly virtual) row
      fast_divmod(mbr[i], g.OH * g.OW, g.inv_ohow, bb, rem);
      fast_divmod(rem, g.OW, g.inv_ow, oy, ox);
      poy[i] = oy;
      iy0[i] = tr - g.pad;                                      // iy = poy*stride + iy0
      const int ix = ox * g.stride + ts - g.pad;
      xok[i] = (unsigned)ix < (unsigned)g.SW;
      pbx[i] = x + ((int64_t)(bb * g.SH + oy * g.stride + iy0[i]) * g.SW + ix) * g.pitch + cc + bcol[i];
    }
  }
  auto issue = [&](int mb, int buf) {
    if (CREID_ABL_ON(abl, 2) && mb != m_begin) return;                      // timing ablation: only the first k-tile is fetched
    unsigned short* la = smem + buf * STAGE + wave * 512;
    unsigned short* lb = smem + buf * STAGE + TILE_A + wave * 512;
    if (linear) {
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const unsigned short* p = (ma[i] < m_end) ? pa[i] : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(la + i * 2048), 16, 0, 0);
        pa[i] += a_inc; ma[i] += WKS;
      }
#pragma unroll
      for (int i = 0; i < NIB; ++i) {
        const bool ok = mbr[i] < m_end && xok[i] && (unsigned)(poy[i] * g.stride + iy0[i]) < (unsigned)g.SH;
        const unsigned short* p = ok ? pbx[i] : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lb + i * 2048), 16, 0, 0);
        pbx[i] += b_inc; mbr[i] += WKS;
        poy[i] += dy_rows;
        if (poy[i] >= g.OH) poy[i] -= g.OH;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int m = mb + arow[i];
      const unsigned short* p = (m < m_end) ? dy + (int64_t)m * NCO + co0 + acol[i] : zpage;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(la + i * 2048), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int m = mb + brow[i];
      const unsigned short* p = zpage;
      int b, oy, ox;
      if (inc_ok) {
        b = pb[i]; oy = poy[i]; ox = pox[i];
        poy[i] += dy_rows;
        if (poy[i] >= g.OH) { poy[i] -= g.OH; ++pb[i]; }
      } else {
        int rem;
        fast_divmod(min(m, g.M - 1), g.OH * g.OW, g.inv_ohow, b, rem);
        fast_divmod(rem, g.OW, g.inv_ow, oy, ox);
      }
      if (m < m_end) {
        int iy, ix;
        if (g.log2span == 5) {
          // stem: a tap is one 32-element kernel row, so the TN-wide tile spans TN/32 taps: column col of the
          // tile row lies in kernel row tr + (col >> 5) at element (col & 31); pre-padded image, no bounds checks
          const int col = cc + bcol[i];
          p = x + (int64_t)((b * g.SH + oy * 2 + tr + (col >> 5)) * g.SW + ox * 2) * g.pitch + (col & 31);
        } else if (igemm_src_pixel(g, oy, ox, tr, ts, iy, ix))
          p = x + (int64_t)((b * g.SH + iy) * g.SW + ix) * g.pitch + cc + bcol[i];
      }
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lb + i * 2048), 16, 0, 0);
    }
  };

  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposing-read lane map + swizzled per-fragment base byte addresses (relative to the stage base)
  const int li = lane & 15;
  const int t_row = 8 * (lane >> 5) + (li >> 2), t_col = 16 * ((lane >> 4) & 1) + 4 * (li & 3);
  const int keyA = (TM == 128) ? (t_row & 3) : ((t_row >> 1) & 1);
  const int keyB = (TN == 128) ? (t_row & 3) : ((t_row >> 1) & 1);
  unsigned fa[IM], fb[JN];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int col = wm * (TM / 2) + i * 32 + t_col;
    fa[i] = lds0 + 2u * (unsigned)(t_row * TM + (((col >> 5) ^ keyA) << 5) + (col & 31));
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int col = wn * (TN / 2) + j * 32 + t_col;
    fb[j] = lds0 + 2u * (unsigned)(TILE_A + t_row * TN + (((col >> 5) ^ keyB) << 5) + (col & 31));
  }

  // fragment reads run one 16-pixel slice ahead of the MFMAs (two register sets): `s_waitcnt lgkmcnt(NRD)`
  // retires the older slice's NRD transposing reads while the younger slice's NRD stay in flight.  A slice reads exactly the
  // IM + JN fragments the wave multiplies (a 64-wide tile side used to be read twice: a third of the LDS read traffic of the
  // 128 x 64 tile nearly every plan picks)
  constexpr int NRD = 2 * (IM + JN);
  auto compute = [&](unsigned sb) {
    s16x4 al[2][IM], ah[2][IM], bl[2][JN], bh[2][JN];
#define WG_LOAD(KK, S)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < IM; ++i)                                                       \
      tr_load2<(KK) * 16 * TM * 2, ((KK) * 16 + 4) * TM * 2>(fa[i] + sb, al[S][i], ah[S][i]);            \
    _Pragma("unroll") for (int j = 0; j < JN; ++j)                                                       \
      tr_load2<(KK) * 16 * TN * 2, ((KK) * 16 + 4) * TN * 2>(fb[j] + sb, bl[S][j], bh[S][j]);
#define WG_MMA(S, NWAIT)                                                                                 \
    {                                                                                                    \
      tr_wait_set<NWAIT, IM, JN>(al[S], ah[S], bl[S], bh[S]);                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      s16x8 a[IM], b[JN];                                                                                \
      _Pragma("unroll") for (int i = 0; i < IM; ++i) a[i] = __builtin_shufflevector(al[S][i], ah[S][i], 0, 1, 2, 3, 4, 5, 6, 7); \
      _Pragma("unroll") for (int j = 0; j < JN; ++j) b[j] = __builtin_shufflevector(bl[S][j], bh[S][j], 0, 1, 2, 3, 4, 5, 6, 7); \
      if (!CREID_ABL_ON(abl, 1)) {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < IM; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < JN; ++j)                                                   \
          acc[i][j] = ET::mfma(a[i], b[j], acc[i][j]);                                                   \
      }                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
    if (CREID_ABL_ON(abl, 4)) return;                                                                                 
    WG_LOAD(0, 0)
    WG_LOAD(1, 1) WG_MMA(0, NRD)
    WG_LOAD(2, 0) WG_MMA(1, NRD)
    WG_LOAD(3, 1) WG_MMA(0, NRD)
    WG_MMA(1, 0)
#undef WG_LOAD
#undef WG_MMA
  };
  if constexpr (!WS) {
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < nt) issue(m_begin + p * WKS, p);
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      const int younger = min(nt - 1 - t, NS - 2);
      if constexpr (NS >= 5) { if (younger == 3) wg_wait_vm<3 * LPT>(); }
      if constexpr (NS >= 4) { if (younger == 2) wg_wait_vm<2 * LPT>(); }
      if constexpr (NS >= 3) { if (younger == 1) wg_wait_vm<1 * LPT>(); }
      if (younger == 0) wg_wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");     // bare barrier: a fence would drain the whole ring (vmcnt 0)
      if (t + NS - 1 < nt) issue(m_begin + (t + NS - 1) * WKS, buf == 0 ? NS - 1 : buf - 1);
      compute((unsigned)(buf * STAGE * 2));
      buf = (buf + 1 == NS) ? 0 : buf + 1;
    }
  } else if (producer) {                            // separate loops: the two roles' registers never coexist
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < nt) issue(m_begin + p * WKS, p);
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      const int younger = min(nt - 1 - t, NS - 2);
      if constexpr (NS >= 3) { if (younger == 1) wg_wait_vm<1 * LPT>(); }
      if (younger == 0) wg_wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");
      if (t + NS - 1 < nt) issue(m_begin + (t + NS - 1) * WKS, buf == 0 ? NS - 1 : buf - 1);
      buf = (buf + 1 == NS) ? 0 : buf + 1;
    }
    return;
  } else {
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_barrier" ::: "memory");
      compute((unsigned)(buf * STAGE * 2));
      buf = (buf + 1 == NS) ? 0 : buf + 1;
    }
  }
  if constexpr (KG > 1) {
    __syncthreads();                                             // both rings are dead
    float* ex = reinterpret_cast<float*>(smem_all);
    const int t256 = tid & 255;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[((i * JN + j) * 16 + r) * 256 + t256] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((i * JN + j) * 16 + r) * 256 + t256];
  }
  const int l31 = lane & 31, kh = lane >> 5;
  float* wsp = ws + (int64_t)split * NCO * g.K;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int kc = k0 + wn * (TN / 2) + j * 32 + l31;
        wsp[(int64_t)co * g.K + kc] = acc[i][j][r];
      }
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(IGemmGeom g, const float* __restrict__ dy,
                                                        const float* __restrict__ x, int NCO, float* __restrict__ ws,
                                                        int tiles_k, int m_per_split) {
  constexpr int IM = TM / 64, JN = TN / 64;
  constexpr int CA = TM / 4, CB = TN / 4;
  constexpr int RA = 256 / CA, RB = 256 / CB;
  constexpr int NA = (WKF + RA - 1) / RA, NBL = (WKF + RB - 1) / RB;
  __shared__ __attribute__((aligned(16))) float Ad[WKF][TM];
  __shared__ __attribute__((aligned(16))) float Bx[WKF][TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int tile_co = tile / tiles_k, tile_k = tile - tile_co * tiles_k;
  const int co0 = tile_co * TM, k0 = tile_k * TN;
  const int m_begin = split * m_per_split, m_end = min(g.M, m_begin + m_per_split);
  const int span_mask = (1 << g.log2span) - 1;
  const int a_ch = tid % CA, a_row = tid / CA;
  const int b_ch = tid % CB, b_row = tid / CB;
  const int kcol = k0 + 4 * b_ch;
  const int tap = kcol >> g.log2span, cc = kcol & span_mask;
  const int tr = tap / g.kw, ts = tap - tr * g.kw;
  float4 ra[NA], rb[NBL];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int pr = a_row + RA * i, m = mb + pr;
      ra[i] = (pr < WKF && m < m_end) ? *reinterpret_cast<const float4*>(dy + (int64_t)m * NCO + co0 + 4 * a_ch)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int pr = b_row + RB * i, m = mb + pr;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pr < WKF && m < m_end) {
        const int b = m / (g.OH * g.OW), rem = m - b * (g.OH * g.OW);
        const int oy = rem / g.OW, ox = rem - oy * g.OW;
        int iy, ix;
        if (igemm_src_pixel(g, oy, ox, tr, ts, iy, ix))
          rb[i] = *reinterpret_cast<const float4*>(x + (int64_t)((b * g.SH + iy) * g.SW + ix) * g.pitch + cc);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (a_row + RA * i < WKF) *reinterpret_cast<float4*>(&Ad[a_row + RA * i][4 * a_ch]) = ra[i];
#pragma unroll
    for (int i = 0; i < NBL; ++i)
      if (b_row + RB * i < WKF) *reinterpret_cast<float4*>(&Bx[b_row + RB * i][4 * b_ch]) = rb[i];
  };
  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, kh = lane >> 5;
  gload(m_begin);
  for (int mb = m_begin; mb < m_end; mb += WKF) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (mb + WKF < m_end) gload(mb + WKF);
#pragma unroll
    for (int kk = 0; kk < WKF; kk += 2) {
      float a[IM], b[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) a[i] = Ad[kk + kh][wm * (TM / 2) + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < JN; ++j) b[j] = Bx[kk + kh][wn * (TN / 2) + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float* wsp = ws + (int64_t)split * NCO * g.K;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int kc = k0 + wn * (TN / 2) + j * 32 + l31;
        wsp[(int64_t)co * g.K + kc] = acc[i][j][r];
      }
}

// sum over splits and scatter [NCO][K = (tap, within)] -> OIHW fp32 gradient (accumulating).
// r = tap / kw_taps; s = tap % kw_taps + within / cpitch; c = within % cpitch.
// Workgroup = (256/SL) float4 element groups x SL split lanes: every thread streams its 16-byte column slice
// through the splits it owns (4 loads in flight), the SL partial sums meet in LDS in a FIXED order
// (deterministic), and lane 0 of each group writes the four results.  The host picks SL from the split count
// so that the per-thread loop stays <= ~8 deep whether a layer has 4 splits of a 2.4M-element tensor or 512
// splits of a 4096-element one.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int NCO, int K,
                                                           int log2span, int kw_taps, int cpitch, int cin, int kh,
                                                           int kw, float* __restrict__ dw_oihw, int accumulate) {
  constexpr int EG = 256 / SL;
  __shared__ float4 red[SL > 1 ? SL : 1][EG];
  const int64_t total = (int64_t)NCO * K, total4 = total >> 2;      // K % 64 == 0
  const int eg = threadIdx.x % EG, sl = threadIdx.x / EG;
  const int64_t i4 = (int64_t)blockIdx.x * EG + eg;
  const bool live = i4 < total4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float4* p = reinterpret_cast<const float4*>(ws) + i4;
    int sp = sl;
    for (; sp + 3 * SL < splits; sp += 4 * SL) {
      const float4 a = p[(int64_t)sp * total4], b = p[(int64_t)(sp + SL) * total4];
      const float4 c = p[(int64_t)(sp + 2 * SL) * total4], d = p[(int64_t)(sp + 3 * SL) * total4];
      acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
      acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; sp < splits; sp += SL) {
      const float4 a = p[(int64_t)sp * total4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
  }
  if constexpr (SL > 1) {
    red[sl][eg] = acc;
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int q = 1; q < SL; ++q) { const float4 v = red[q][eg]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  if (!live) return;
  const int64_t i = i4 << 2;
  if (kh == 1 && kw == 1 && cpitch == cin && (1 << log2span) == cin &&
      (reinterpret_cast<uintptr_t>(dw_oihw) & 15) == 0) {                  // 1x1: OIHW == [co][c], same index
    float4* dst = reinterpret_cast<float4*>(dw_oihw + i);
    if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    *dst = acc;
    return;
  }
  const int span_mask = (1 << log2span) - 1;
  const int co = (int)(i / K), kc0 = (int)(i - (int64_t)co * K);
  const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kc = kc0 + e;
    const int tap = kc >> log2span, within = kc & span_mask;
    const int r = tap / kw_taps, s = tap % kw_taps + within / cpitch, c = within % cpitch;
    if (r < kh && s < kw && c < cin) {
      float* dst = dw_oihw + (((int64_t)co * cin + c) * kh + r) * kw + s;
      *dst = accumulate ? *dst + v[e] : v[e];
    }
  }
}

// 3x3 layers with few splits: one workgroup per output channel.  The partial row [tap][c] (K floats, contiguous) is
// summed over the splits with coalesced reads, transposed to OIHW's [c][tap] order in LDS, and leaves as ONE
// contiguous K-float run (the generic kernel writes those 4-byte values 36 bytes apart).
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ ws, int splits, int NCO, int K,
                                                                int cin, int taps, float* __restrict__ dw_oihw,
                                                                int accumulate) {
  extern __shared__ float tr_lds[];                              // [K] in (c, tap) order
  const int co = blockIdx.x;
  const int64_t total = (int64_t)NCO * K;
  const float* src = ws + (int64_t)co * K;
  for (int e = threadIdx.x; e < K; e += 256) {
    float a0 = 0.f, a1 = 0.f;
    int sp = 0;
    for (; sp + 1 < splits; sp += 2) { a0 += src[(int64_t)sp * total + e]; a1 += src[(int64_t)(sp + 1) * total + e]; }
    if (sp < splits) a0 += src[(int64_t)sp * total + e];
    const int tap = e / cin, c = e - tap * cin;
    tr_lds[c * taps + tap] = a0 + a1;
  }
  __syncthreads();
  float* dst = dw_oihw + (int64_t)co * K;
  for (int e = threadIdx.x; e < K; e += 256) dst[e] = accumulate ? dst[e] + tr_lds[e] : tr_lds[e];
}

// ------------------------------------------------------------------------------------ host
static int ilog2x(int64_t v) { int l = 0; while ((1LL << l) < v) ++l; return ((1LL << l) == v) ? l : -1; }

struct WgradPlan { int tm, tn, tiles, tiles_k, splits, m_per_split, xcd, stages, ws, kg; };   // stages / ws / kg: 0 = default

static WgradPlan plan_wgrad(int M, int NCO, int K, int dtype, int stride = 1) {
  // The fp32 partial tiles cost  workgroups x TM x TN x 8 bytes  of traffic per layer (write + re-read by the
  // reduce).  Measured (r01): shrinking the tile to cut that traffic LOSES (6147 vs 6370 img/s) -- the larger
  // tile's MFMA/LDS efficiency matters more and the partials mostly stay in the 256 MB Infinity Cache -- so the
  // largest tile wins by default; CREID_WGRAD_MAX_SPLITS re-enables the size/split trade-off for experiments.
  // Split target: ~512 workgroups per launch (r01 sweep with stand-alone reduce launches: 384-448 best; r02 with the
  // reduction riding on the next launch: 256 / 384 / 512 / 768 -> 7.20 / 7.10 / 7.08 / 7.18 ms per step).
  static const int target = [] { const char* e = getenv("CREID_WGRAD_TARGET_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  static const int max_splits_pref = [] { const char* e = getenv("CREID_WGRAD_MAX_SPLITS"); int v = e ? atoi(e) : 0; return v > 0 ? v : (1 << 30); }();
  const int ks = creid_is16(dtype) ? WKS : WKF;
  const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  WgradPlan p;
  p.stages = 0; p.ws = 0; p.kg = 0;
  TunePlan tp;
  if (creid_is16(dtype) && creid_tune_lookup(CREID_TUNE_WGRAD, M, NCO, K, stride << 1, tp) && (tp.p0 == 64 || tp.p0 == 128) &&
      (tp.p1 == 64 || tp.p1 == 128) && NCO % tp.p0 == 0 && K % tp.p1 == 0 && tp.p2 >= 1) {
    // measured plan for this shape: tile tp.p0 x tp.p1, tp.p2 = pixel splits | ring depth << 16 | producer/consumer << 20 |
    // two k-groups << 21
    p.tm = tp.p0; p.tn = tp.p1;
    p.stages = (tp.p2 >> 16) & 15; p.ws = (tp.p2 >> 20) & 1; p.kg = (tp.p2 >> 21) & 1;
    tp.p2 &= 0xffff;
    p.tiles_k = K / p.tn;
    p.tiles = (NCO / p.tm) * p.tiles_k;
    int splits = tp.p2;
    const int max_splits = (M + ks - 1) / ks;
    if (splits > max_splits) splits = max_splits;
    p.m_per_split = ((M + splits - 1) / splits + ks - 1) / ks * ks;
    p.splits = (M + p.m_per_split - 1) / p.m_per_split;
    static const int xcd_mode = [] { const char* e = getenv("CREID_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    p.xcd = (xcd_mode && p.splits >= 8) ? 1 : 0;
    return p;
  }
  for (int ci = 0; ci < 3; ++ci) {
    int tm = cand[ci][0], tn = cand[ci][1];
    if (NCO % tm != 0) tm = 64;
    if (K % tn != 0) tn = 64;
    p.tm = tm; p.tn = tn;
    p.tiles_k = K / tn;
    p.tiles = (NCO / tm) * p.tiles_k;
    int splits = (target + p.tiles - 1) / p.tiles;
    // XCD placement (see wgrad_bf16_dma_kernel): a multiple of 8 splits, at least 8, so that every XCD owns whole
    // pixel ranges; CREID_WGRAD_XCD=0 keeps the round-1 rule
    static const int xcd_mode = [] { const char* e = getenv("CREID_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    p.xcd = 0;
    if (xcd_mode && creid_is16(dtype)) {
      if (splits >= 8) { splits = (splits + 4) / 8 * 8; p.xcd = 1; }                     // nearest multiple of 8
      else if (xcd_mode >= 2) { splits = splits >= 6 ? 8 : (splits >= 3 ? 4 : splits); p.xcd = 1; }   // 1, 2, 4, 8
    }
    const int max_splits = (M + 4 * ks - 1) / (4 * ks);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.m_per_split = ((M + splits - 1) / splits + ks - 1) / ks * ks;
    p.splits = (M + p.m_per_split - 1) / p.m_per_split;
    if (p.xcd && p.splits < 8 && (8 % p.splits) != 0) p.xcd = 0;      // the shared-range map needs 1, 2, 4 (or >= 8) splits
    if (p.splits <= max_splits_pref) break;
  }
  return p;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_job_kernel(BnBwdFinJob j) {
  __shared__ __attribute__((aligned(16))) double fin_lds[4 * 2 * 16];
  bn_bwd_finalize_block<256>(j, (int)blockIdx.x, fin_lds);
}

template <int TM, int TN>
// returns true when the launch carried the BatchNorm-backward finalize job `fin_in` (bn_fin.hpp) in its first workgroups
static bool launch_wgrad_t(const IGemmGeom& g, const void* dy, const void* x, int NCO, float* ws, const WgradPlan& p,
                           int dtype, hipStream_t s, const BnBwdFinJob* fin_in = nullptr) {
  dim3 grid((unsigned)p.tiles, (unsigned)p.splits), block(256);
  const bool xcd_on = p.xcd != 0;
  const int xt = xcd_on ? p.tiles : 0;
  const int share = p.splits >= 8 ? 1 : 8 / p.splits;
  BnBwdFinJob fin{};
  if (fin_in && xcd_on) fin = *fin_in;                            // the carried job needs the 1-D grid
  const unsigned nf8 = fin.partial ? (unsigned)((fin.nblocks + 7) & ~7) : 0u;
  const dim3 grid_x((p.splits >= 8 ? (unsigned)(p.tiles * ((p.splits + 7) / 8) * 8) : (unsigned)(8 * ((p.tiles + share - 1) / share))) + nf8);
  const dim3 grid_dma = xcd_on ? grid_x : grid;
  static const int use_dma = [] { const char* e = getenv("CREID_WGRAD_DMA"); return e ? atoi(e) : 1; }();
#ifdef CREID_ABL_BUILD
  static const int wg_abl = creid_ablation_env("CREID_WGRAD_ABL");   // 1: no MFMA, 2: no DMA after the first k-tile, 4: no fragment reads either
#else
  const int wg_abl = 0;                                              // (the switches exist in the ablation build only: conv_common.hpp)
#endif
  static const int stages_env = [] { const char* e = getenv("CREID_WGRAD_STAGES"); int v = e ? atoi(e) : 0; return (v >= 2 && v <= 4) ? v : 2; }();
  static const int stem_dma = [] { const char* e = getenv("CREID_STEM_DMA"); return e ? atoi(e) : 1; }();
  const bool stem_geom = g.log2span == 5 && !g.check_bounds && g.kw == 1 && g.stride == 2 && g.pad == 0 && stem_dma;
  if (creid_is16(dtype) && use_dma && ((1 << g.log2span) >= TN || stem_geom)) {
    static const int use_ws_env = [] { const char* e = getenv("CREID_WGRAD_WS"); return e ? atoi(e) : 0; }();
    const int use_ws = p.ws ? 1 : use_ws_env;
    const int stages = (p.stages >= 2 && p.stages <= 4) ? p.stages : stages_env;
    static const int kg_env = [] { const char* e = getenv("CREID_WGRAD_KG"); return e ? atoi(e) : 0; }();
    const bool kg2 = (p.kg || kg_env == 2) && !use_ws && !stem_geom;
    constexpr bool kg3_fits = 2 * 3 * (TM + TN) * 128 <= 160 * 1024;
#define CREID_WG_LAUNCH(NS_, WS_, KG_, BLOCK_)                                                                                   \
    do {                                                                                                                          \
      if (dtype == CREID_F16)                                                                                                     \
        hipLaunchKernelGGL((wgrad_bf16_dma_kernel<TM, TN, NS_, WS_, KG_, F16T>), grid_dma, BLOCK_, 0, s, g, (const unsigned short*)dy, \
                           (const unsigned short*)x, NCO, ws, p.tiles_k, p.m_per_split, xt, p.splits, fin, wg_abl);               \
      else                                                                                                                        \
        hipLaunchKernelGGL((wgrad_bf16_dma_kernel<TM, TN, NS_, WS_, KG_, Bf16T>), grid_dma, BLOCK_, 0, s, g, (const unsigned short*)dy, \
                           (const unsigned short*)x, NCO, ws, p.tiles_k, p.m_per_split, xt, p.splits, fin, wg_abl);               \
    } while (0)
    if (kg2 && stages >= 3 && kg3_fits) {
      if constexpr (kg3_fits) CREID_WG_LAUNCH(3, false, 2, dim3(512));
    } else if (kg2) CREID_WG_LAUNCH(2, false, 2, dim3(512));
    else if (use_ws && !stem_geom) CREID_WG_LAUNCH(2, true, 1, dim3(512));
    else if (stages == 2) CREID_WG_LAUNCH(2, false, 1, block);
    else if (stages == 3) CREID_WG_LAUNCH(3, false, 1, block);
    else CREID_WG_LAUNCH(4, false, 1, block);
#undef CREID_WG_LAUNCH
    return fin.partial != nullptr;
  }
  else if (dtype == CREID_F16) return false;                     // (run_wgrad refuses f16 shapes the DMA kernels do not cover)
  else if (dtype == CREID_BF16)
    hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN>), grid, block, 0, s, g, (const unsigned short*)dy,
                       (const unsigned short*)x, NCO, ws, p.tiles_k, p.m_per_split);
  else
    hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN>), grid, block, 0, s, g, (const float*)dy, (const float*)x, NCO, ws,
                       p.tiles_k, p.m_per_split);
  return false;
}

static int run_wgrad(const IGemmGeom& g, const void* dy, const void* x, int NCO, float* dw, int kw_taps, int cpitch,
                     int cin, int kh, int kw, int accumulate, void* ws, size_t ws_bytes, int dtype, hipStream_t s,
                     int phases = 3, const BnBwdFinJob* fin = nullptr) {
  if (!creid_is16(dtype) && dtype != CREID_F32) return CREID_E_DTYPE;
  // f16 exists in the LDS-DMA kernels only (no register-staged fallback): shapes those do not cover are refused, not mis-run
  if (dtype == CREID_F16) {
    static const int dma_on = [] { const char* e = getenv("CREID_WGRAD_DMA"); return e ? atoi(e) : 1; }();
    if (!dma_on) return CREID_E_DTYPE;
  }
  const WgradPlan p = plan_wgrad(g.M, NCO, g.K, dtype, g.stride);
  if (dtype == CREID_F16 && (phases & 1)) {
    // (the same predicate launch_wgrad_t routes by, CREID_STEM_DMA included: with the stem's DMA path switched off an f16 stem has
    // no kernel at all -- refuse it here instead of summing an unwritten workspace into dw)
    static const int stem_dma = [] { const char* e = getenv("CREID_STEM_DMA"); return e ? atoi(e) : 1; }();
    const bool stem_geom = g.log2span == 5 && !g.check_bounds && g.kw == 1 && g.stride == 2 && g.pad == 0 && stem_dma;
    if (!((1 << g.log2span) >= p.tn || stem_geom)) return CREID_E_SHAPE;
  }
  const size_t need = (size_t)p.splits * NCO * g.K * sizeof(float);
  if (ws_bytes < need) return CREID_E_WS;
  if (phases & 1) {
    bool carried;
    if (p.tm == 128 && p.tn == 128) carried = launch_wgrad_t<128, 128>(g, dy, x, NCO, (float*)ws, p, dtype, s, fin);
    else if (p.tm == 128 && p.tn == 64) carried = launch_wgrad_t<128, 64>(g, dy, x, NCO, (float*)ws, p, dtype, s, fin);
    else if (p.tm == 64 && p.tn == 128) carried = launch_wgrad_t<64, 128>(g, dy, x, NCO, (float*)ws, p, dtype, s, fin);
    else carried = launch_wgrad_t<64, 64>(g, dy, x, NCO, (float*)ws, p, dtype, s, fin);
    if (fin && !carried)                                         // kernel variants without the carrier: stand-alone launch
      hipLaunchKernelGGL(bn_bwd_finalize_job_kernel, dim3((unsigned)fin->nblocks), dim3(256), 0, s, *fin);
  }
  if (!(phases & 2)) return (int)hipGetLastError();
  if (kh * kw == 9 && cpitch == cin && (1 << g.log2span) == cin && kw_taps == kw && p.splits <= 16 &&
      (size_t)g.K * sizeof(float) <= 48 * 1024) {
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)NCO), dim3(256), (size_t)g.K * sizeof(float), s,
                       (const float*)ws, p.splits, NCO, g.K, cin, 9, dw, accumulate);
    return (int)hipGetLastError();
  }
  const int64_t total4 = (int64_t)NCO * g.K / 4;
#define CREID_WRED(SL_)                                                                                              \
  hipLaunchKernelGGL(wgrad_reduce_kernel<SL_>, dim3((unsigned)((total4 + 256 / SL_ - 1) / (256 / SL_))), dim3(256), 0, s, \
                     (const float*)ws, p.splits, NCO, g.K, g.log2span, kw_taps, cpitch, cin, kh, kw, dw, accumulate)
  if (p.splits <= 8) CREID_WRED(1);
  else if (p.splits <= 32) CREID_WRED(4);
  else if (p.splits <= 128) CREID_WRED(16);
  else CREID_WRED(64);
#undef CREID_WRED
  return (int)hipGetLastError();
}

// ---- the split reduction as a job (wgrad_reduce.hpp): stand-alone launch and the glue used by the data-gradient
// launch that carries it
__global__ __launch_bounds__(512) void wgrad_reduce_job_kernel(WRedJob j) {
  extern __shared__ __attribute__((aligned(16))) float wred_lds[];
  wgrad_reduce_block<512>(j, (int)blockIdx.x, wred_lds);
}

int wgrad_reduce_job_launch(const WRedJob& j, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_reduce_job_kernel, dim3((unsigned)j.nblocks), dim3(512), (size_t)(4 * 512 + j.K) * sizeof(float), s, j);
  return (int)hipGetLastError();
}

bool wgrad_make_reduce_job(const creid_conv_desc* d, int dtype, const void* ws, size_t ws_bytes, float* dw, int accumulate,
                           WRedJob& j) {
  if (!d || !ws || !dw) return false;
  const int M = (int)(d->batch * d->out_h * d->out_w), K = (int)(d->kh * d->kw * d->in_c), NCO = (int)d->out_c;
  const WgradPlan p = plan_wgrad(M, NCO, K, dtype, d->stride);
  if (ws_bytes < (size_t)p.splits * NCO * K * sizeof(float)) return false;
  const bool ok = wred_make_job(j, (const float*)ws, dw, p.splits, NCO, K, (int)d->in_c, d->kh, d->kw, accumulate);
  // timing experiments only: the carrier workgroups are launched but do nothing (gradients are then WRONG)
  static const int dry = creid_ablation_env("CREID_WRED_DRY");
  if (ok && dry) { j.K = 0; j.splits = 0; }
  return ok;
}

extern "C" {

size_t creid_conv2d_wgrad_workspace_bytes(const creid_conv_desc* d, int dtype) {
  if (!d) return 0;
  const int M = (int)(d->batch * d->out_h * d->out_w), K = (int)(d->kh * d->kw * d->in_c);
  const WgradPlan p = plan_wgrad(M, (int)d->out_c, K, dtype, d->stride);
  return (size_t)p.splits * d->out_c * K * sizeof(float);
}

static int conv_wgrad_phases(const creid_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int accumulate,
                             void* ws, size_t ws_bytes, int dtype, void* stream, int phases, const BnBwdFinJob* fin = nullptr);

int creid_conv2d_wgrad_nhwc(const creid_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int accumulate,
                            void* ws, size_t ws_bytes, int dtype, void* stream) {
  CREID_CHECK_ARG(x && dy && dw_oihw);
  return conv_wgrad_phases(d, x, dy, dw_oihw, accumulate, ws, ws_bytes, dtype, stream, 3);
}

int creid_conv2d_wgrad_partials(const creid_conv_desc* d, const void* x, const void* dy, void* ws, size_t ws_bytes,
                                int dtype, void* stream) {
  CREID_CHECK_ARG(x && dy);
  return conv_wgrad_phases(d, x, dy, nullptr, 0, ws, ws_bytes, dtype, stream, 1);
}

int creid_conv2d_wgrad_partials_bnfin(const creid_conv_desc* d, const void* x, const void* dy, void* ws, size_t ws_bytes,
                                      int dtype, const float* bn_partial, int64_t bn_rows, int64_t bn_C, int64_t bn_count,
                                      const float* bn_mean, const float* bn_invstd, const float* bn_gamma, float* bn_sums,
                                      float* bn_dgamma, float* bn_dbeta, void* stream) {
  CREID_CHECK_ARG(x && dy && bn_partial && bn_mean && bn_invstd && bn_sums && bn_rows > 0 && bn_C > 0 && bn_count > 0);
  BnBwdFinJob fin{bn_partial, (int)bn_rows, (int)bn_C, (float)(1.0 / (double)bn_count), bn_mean, bn_invstd, bn_gamma, bn_sums,
                  bn_dgamma, bn_dbeta, 0, 16};
  bn_bwd_fin_shape(fin);
  return conv_wgrad_phases(d, x, dy, nullptr, 0, ws, ws_bytes, dtype, stream, 1, &fin);
}

int creid_conv2d_wgrad_reduce(const creid_conv_desc* d, float* dw_oihw, int accumulate, const void* ws, size_t ws_bytes,
                              int dtype, void* stream) {
  CREID_CHECK_ARG(dw_oihw);
  return conv_wgrad_phases(d, nullptr, nullptr, dw_oihw, accumulate, const_cast<void*>(ws), ws_bytes, dtype, stream, 2);
}

// BatchNorm-backward finalize + split reduction in ONE launch: the first workgroups (a multiple of 8) finalize, the rest sum
// the previous weight gradient's partial tiles.  Both are off each other's data; the reduction is bandwidth work that fills the
// ~4 us during which the 4-128 finalize workgroups wait on their dependent loads, and the data gradient before it keeps its
// full grid (no reduction tail).
__global__ __launch_bounds__(512) void wred_bnfin_kernel(WRedJob j, BnBwdFinJob fin) {
  extern __shared__ __attribute__((aligned(16))) float wredfin_lds[];
  const int nf8 = fin.partial ? ((fin.nblocks + 7) & ~7) : 0;
  if ((int)blockIdx.x < nf8) {
    if ((int)blockIdx.x < fin.nblocks) bn_bwd_finalize_block<512>(fin, (int)blockIdx.x, wredfin_lds);
    return;
  }
  wgrad_reduce_block<512>(j, (int)blockIdx.x - nf8, wredfin_lds);
}

int creid_bn2d_bwd_finalize_wred(const float* bn_partial, int64_t bn_rows, int64_t bn_C, int64_t bn_count, const float* bn_mean,
                                 const float* bn_invstd, const float* bn_gamma, float* bn_sums, float* bn_dgamma,
                                 float* bn_dbeta, const creid_conv_desc* wred_desc, float* wred_dw, int wred_accumulate,
                                 const void* wred_ws, size_t wred_ws_bytes, int dtype, void* stream) {
  CREID_CHECK_ARG(bn_partial && bn_mean && bn_invstd && bn_sums && bn_rows > 0 && bn_C > 0 && bn_count > 0 && wred_desc &&
                  wred_dw && wred_ws);
  BnBwdFinJob fin{bn_partial, (int)bn_rows, (int)bn_C, (float)(1.0 / (double)bn_count), bn_mean, bn_invstd, bn_gamma, bn_sums,
                  bn_dgamma, bn_dbeta, 0, 16};
  bn_bwd_fin_shape(fin);
  WRedJob j{};
  if (!wgrad_make_reduce_job(wred_desc, dtype, wred_ws, wred_ws_bytes, wred_dw, wred_accumulate, j)) return CREID_E_SHAPE;
  const int nf8 = (fin.nblocks + 7) & ~7;
  size_t lds = (size_t)(4 * 512 + j.K) * sizeof(float);
  if (lds < 4096) lds = 4096;
  hipLaunchKernelGGL(wred_bnfin_kernel, dim3((unsigned)(nf8 + j.nblocks)), dim3(512), lds, as_stream(stream), j, fin);
  CREID_LAUNCH_RET();
}

/* The split reduction as the SAME job a data-gradient launch would have carried (wgrad_reduce_block, identical summation
 * order), as its own launch: for the reductions that find no carrier (end of a layer group / of the backward pass), so that
 * a schedule with and without carriers gives bit-identical gradients. */
int creid_conv2d_wgrad_reduce_job(const creid_conv_desc* d, float* dw_oihw, int accumulate, const void* ws, size_t ws_bytes,
                                  int dtype, void* stream) {
  CREID_CHECK_ARG(d && dw_oihw && ws);
  WRedJob j{};
  if (!wgrad_make_reduce_job(d, dtype, ws, ws_bytes, dw_oihw, accumulate, j))
    return creid_conv2d_wgrad_reduce(d, dw_oihw, accumulate, ws, ws_bytes, dtype, stream);
  return wgrad_reduce_job_launch(j, as_stream(stream));
}

static int conv_wgrad_phases(const creid_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int accumulate,
                             void* ws, size_t ws_bytes, int dtype, void* stream, int phases, const BnBwdFinJob* fin) {
  CREID_CHECK_ARG(d && ws);
  if (ilog2x(d->in_c) < 0 || d->in_c < 64 || d->out_c % 64 != 0) return CREID_E_SHAPE;
  IGemmGeom g;
  g.M = (int)(d->batch * d->out_h * d->out_w); g.OH = (int)d->out_h; g.OW = (int)d->out_w;
  g.SH = (int)d->in_h; g.SW = (int)d->in_w; g.pitch = (int)d->in_c; g.log2span = ilog2x(d->in_c);
  g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.transposed = 0;
  g.K = (int)(d->kh * d->kw * d->in_c); g.N = (int)d->out_c; g.check_bounds = 1;
  { static const int noinc = [] { const char* e = getenv("CREID_WGRAD_NOINC"); return e ? atoi(e) : 0; }(); if (noinc) g.check_bounds = 2; }
  igemm_finish_geom(g);
  return run_wgrad(g, dy, x, (int)d->out_c, dw_oihw, d->kw, (int)d->in_c, (int)d->in_c, d->kh, d->kw, accumulate, ws,
                   ws_bytes, dtype, as_stream(stream), phases, fin);
}

size_t creid_stem_conv_wgrad_workspace_bytes(int64_t batch, int64_t H, int64_t W, int dtype) {
  // (stride 2: the launch looks its plan up with the stem geometry's stride -- with the default of 1 a measured plan of a 1 x 1
  // layer that happens to share (M, 64, 256) at another batch size answered here but not there: workspace too small)
  const WgradPlan p = plan_wgrad((int)(batch * (H / 2) * (W / 2)), 64, 256, dtype, 2);
  return (size_t)p.splits * 64 * 256 * sizeof(float);
}

/* stem weight gradient into the OIHW [64,3,7,7] fp32 tensor (see creid_stem_conv_fwd for layouts). */
int creid_stem_conv_wgrad(int64_t batch, int64_t H, int64_t W, const void* xpad, const void* dy, float* dw_oihw,
                          int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream) {
  CREID_CHECK_ARG(xpad && dy && dw_oihw && ws && batch > 0 && H > 0 && W > 0);
  if (H % 2 || W % 2) return CREID_E_SHAPE;
  IGemmGeom g;
  g.M = (int)(batch * (H / 2) * (W / 2)); g.OH = (int)(H / 2); g.OW = (int)(W / 2);
  g.SH = (int)(H + 8); g.SW = (int)(W + 6); g.pitch = 4; g.log2span = 5;
  g.kw = 1; g.stride = 2; g.pad = 0; g.transposed = 0; g.K = 256; g.N = 64; g.check_bounds = 0;
  igemm_finish_geom(g);
  return run_wgrad(g, dy, xpad, 64, dw_oihw, 1, 4, 3, 7, 7, accumulate, ws, ws_bytes, dtype, as_stream(stream));
}

}  // extern "C"
