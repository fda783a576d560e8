// Stage A, second generation: implicit-GEMM convolution forward / data gradient in which EVERY wave multiplies.
// (same operator as conv_igemm.hip: nn.Conv2d fwd + dgrad of modelling/backbones/resnet.py:56-61,94,109.)
//
// Why a second kernel (profiles/r03_conv_ablation.md, r03_bm256_table.md): in igemm_bf16_ws_kernel only four of the eight
// waves issue MFMAs, one barrier per 64-deep k-tile couples them to the four producer waves, and a 128 x 128 tile needs as
// many L1-path clocks for its operands (32 KB at 64 B/clk = 512) as matrix-pipe clocks for its MFMAs (512 per SIMD): MFMA,
// DMA, fragment reads and the epilogue ADD UP.  Here:
//   * BM x BN = 256 x 256 / 128 x 256 / 256 x 128 macro-tiles, BK = 64: 64 KB of operands (1024 L1-path clocks) per 2048
//     matrix-pipe clocks per SIMD for the 256 x 256 tile -- the operand path is half idle at full MFMA rate;
//   * all eight waves hold accumulators (256 x 256: 2 x 4 waves of 128 x 64 = 8 v_mfma_f32_32x32x16 blocks, 128 registers)
//     and every wave issues its own share of the global->LDS DMA for the NEXT k-tile (8 KB per wave and k-tile);
//   * the two waves that share a SIMD (w and w + 4) run half a phase apart ("ping-pong"): a k-tile is cut into 4 / KPH phases;
//     in each phase one wave group reads its fragments for the phase (ds_read_b128, swizzled image as in conv_igemm.hip) and
//     issues DMA pieces while the other group issues the phase's MFMAs under s_setprio 1; bare s_barrier between the halves,
//     counted / grouped vmcnt only once per k-tile;
//   * workgroups are PERSISTENT: a workgroup walks tiles id, id + grid, ... (XCD-contiguous order), the first k-tile of the
//     next tile is requested before the epilogue of the current one starts, and the epilogue stages the C tile in 32-row
//     pieces through a staging area OUTSIDE the operand ring, so its stores drain under the next tile's loads.
// Arithmetic per output element is that of the tile kernels (same k order inside v_mfma_f32_32x32x16_bf16, same epilogue
// operations): outputs are bit-identical (tests/test_conv_pipe_gpu.py).
#include "conv_common.hpp"
#include "tune.hpp"
#include <type_traits>
#include <stdlib.h>

namespace creid_pp {   // (a NAMED namespace: rocprofv3 prints `(anonymous namespace)::` in front of kernels of an unnamed one, which the
                      // profile tooling's name clean-up would cut to nothing)

__device__ __attribute__((aligned(128))) unsigned g_zero_page_pp[32];

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// workgroup barrier that orders LDS accesses only: __syncthreads() would also drain vmcnt, i.e. wait for the next tile's
// first k-tile (an LDS-DMA in flight) at every epilogue step
__device__ __forceinline__ void pp_lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// One LDS-DMA piece: LDS[lds_addr + lane * 16] <- the lane's 16 global bytes.  Written as inline asm ON PURPOSE: hipcc models a
// global_load_lds builtin as a FLAT access that may touch LDS, and from then on every wait it inserts for a fragment read is
// a full `s_waitcnt lgkmcnt(0)` (pending-flat state), which exposes the latency of the reads issued just before it.  The asm
// form is invisible to that bookkeeping; its completion is counted by hand (pp_wait_vm).  M0 is not used by anything else here.
__device__ __forceinline__ void pp_glds16(const unsigned short* gsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}

// DMA piece schedule: phases 0 .. NPH-2 of a k-tile carry the NP pieces of the NEXT k-tile (the last phase is where a group
// waits for them); [pp_piece_lo(ph), pp_piece_lo(ph + 1)) are the pieces of phase ph
constexpr int PP_TRACE_N = 1024;
#ifdef CREID_ABL_BUILD
#define PP_STAMP()                                                                                            \
  do {                                                                                                        \
    if (tr_on && tr_i < PP_TRACE_N) {                                                                         \
      const unsigned long long c_ = __builtin_readcyclecounter();                                             \
      if (lane == 0) pa.trace[(wave >> 2) * PP_TRACE_N + tr_i] = c_;                                          \
      ++tr_i;                                                                                                 \
    }                                                                                                         \
  } while (0)
#else
#define PP_STAMP() do { } while (0)
#endif

template <int NPH, int NP> __device__ constexpr int pp_piece_lo(int ph) {
  return ph <= 0 ? 0 : (ph >= NPH - 1 ? NP : (ph * NP + (NPH - 2)) / (NPH - 1));
}

struct PipeArgs {
  unsigned long long* trace;       // ablation build only: cycle stamps of waves 0 and 4 of workgroup 0 ([2][PP_TRACE_N])
  const unsigned short* src;
  const unsigned short* wgt;
  unsigned short* out;
  const unsigned short* add_src;   // nullable: residual (eval) / gradient accumulation (dgrad), full resolution
  float* bn_part;                  // nullable: per-128-row (sum, sumsq) partials of the fp32 accumulators
  int tiles_m, tiles_n;
};

// BM x BN tile, WN wave columns (WM = 8 / WN wave rows), KPH 16-wide k-slices per phase, GLM = main-loop form: 0 ping-pong wave
// groups (two barriers per phase, DMA pieces beside the fragment reads), 2 free-running (one barrier per k-tile, fragments
// double-buffered in registers).  (Form 1 -- pieces at the head of the multiply half -- and form 3 -- pieces between the MFMAs
// -- were measured slower / no faster and are gone.)
// EPI: 0 plain store (+ add_src), 1 BatchNorm statistics of the fp32 accumulators, 2 folded affine (+ residual) (+ ReLU) --
// compile-time, because the per-element work of the copy-out is VALU-bound (a runtime choice made the compiler evaluate
// every variant and select: ~450 instructions per 32-row piece, 3100 cycles of the 3300 a whole k-tile takes).
// (128 x 128: 32 accumulator registers per wave and a 64 KB ring -- two workgroups per CU, i.e. four waves per SIMD and <= 128
// registers, for the grids of 256 .. 1024 tiles the M = 8192 layers of the training batch give)
template <int BM, int BN, int WN, int KPH, int GLM, int EPI, typename ET>
__global__ __launch_bounds__(512, (BM * BN <= 128 * 128) ? 4 : 2) void igemm_bf16_pp_kernel(IGemmGeom g, PipeArgs pa) {
  constexpr int NT = 512, BK = 64, WM = 8 / WN;
  constexpr int MI = BM / WM / 32, NJ = BN / WN / 32;            // 32 x 32 accumulator blocks per wave
  constexpr int NAI = BM / 64, NBI = BN / 64, NP = NAI + NBI;     // DMA pieces (8 rows x 128 B) per wave and k-tile
  constexpr int NPH = 4 / KPH;                                    // phases per k-tile
  constexpr int TILE_A = BM * BK, TILE_B = BN * BK, STAGE = TILE_A + TILE_B;
  constexpr int PR = WM * 32;                                     // rows of one staged piece (one 32-row block of every wave row)
  constexpr int CPT = PR + 4;                                     // staging pitch (elements): 2 banks between columns
  constexpr int STG = BN * CPT;                                   // one staging buffer
  // ring depth: three slots in the free-running form where 160 KB allow it (128 x 256, 256 x 128: 3 x 48 KB) -- inside a captured
  // forward the operands come from HBM / the Infinity Cache, not from an L2 the previous launch of the same layer warmed, and
  // with two slots the once-per-k-tile wait saw less than one k-tile of prefetch distance (profiles/r04_plan_validation.md:
  // the long-K plans that won alone lost in situ)
  constexpr int NS = (GLM == 2 && BM * BN > 128 * 128 && 3 * STAGE * 2 + 12 * 1024 <= 160 * 1024) ? 3 : 2;
  constexpr int RING = NS * STAGE;
  constexpr int RED = 2 * (BM / 64) * 2 * BN;                     // fp32 column-sum scratch in 2-byte units
  // the staging area starts at ring slot 1: while a tile is copied out only slot 0 is in use (the next tile's first k-tile);
  // the column-sum scratch has its own 8 KB behind everything (written piece by piece: no registers held across the copy-out)
  constexpr int LDS_MAIN = (STAGE + 2 * STG > RING) ? STAGE + 2 * STG : RING;   // (staging: slots 1 .. while a tile is copied out)
  constexpr int LDS_ELEMS = LDS_MAIN + (EPI == 1 ? RED : 0);
  static_assert(LDS_ELEMS * 2 <= 160 * 1024, "LDS budget");
  static_assert(MI >= 2 && MI % 2 == 0 && NJ >= 1 && (KPH == 1 || KPH == 2), "shape");
  constexpr int CPR = BN / 8;                                     // 16-byte column chunks per row
  constexpr int NIT = (PR * CPR) / NT;                            // copy-out iterations per piece
  static_assert((PR * CPR) % NT == 0, "copy-out map");
  __shared__ __attribute__((aligned(1024))) unsigned short smem[LDS_ELEMS];
  unsigned short* stg = smem + STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                      // waves w and w + 4 share a SIMD: the two ping-pong groups
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, kh = lane >> 5;
  const int lr8 = lane >> 3, lcp = lane & 7;
  const int ntiles = pa.tiles_m * pa.tiles_n;
  const int nk = g.K / BK;
  const unsigned short* zpage = reinterpret_cast<const unsigned short*>(g_zero_page_pp);
#ifdef CREID_ABL_BUILD
  const bool tr_on = pa.trace != nullptr && blockIdx.x == 0 && (wave & 3) == 0;
  int tr_i = 0;
#endif

  // ---- per-tile gather state of the DMA pieces this wave issues (kept small: the 256 x 256 tile leaves ~100 registers beside
  // its accumulators).  Piece i of wave w covers tile rows (i * 8 + w) * 8 .. + 7: the swizzled source chunk of a lane does not
  // depend on i; rows beyond M get an output pixel far outside the image, i.e. the zero page from the bounds check.
  const int gch = (lcp ^ (((wave & 1) * 4 + (lr8 >> 1)) & 7)) << 3;
  int oyx[NAI], bpix[NAI];                                       // oy << 16 | (ox & 0xffff)
  const unsigned short* aptr[NAI];
  unsigned aok = 0;                                              // bit i: piece i reads a real source row (advance by 64 channels)
  const unsigned short* wp0;                                     // B piece i reads wp0 + i * 64 * K
  int cur_tap, cur_r, cur_s;
  auto tile_setup = [&](int tile) {
    const int tile_m = tile / pa.tiles_n, tile_n = tile - tile_m * pa.tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int r = (i * 8 + wave) * 8 + lr8, m = row0 + r;
      int b, rem, y, x;
      fast_divmod(m < g.M ? m : row0, g.OH * g.OW, g.inv_ohow, b, rem);
      fast_divmod(rem, g.OW, g.inv_ow, y, x);
      if (m >= g.M) y = -16384;
      bpix[i] = b * g.SH * g.SW;
      oyx[i] = (y << 16) | (x & 0xffff);
    }
    wp0 = pa.wgt + (int64_t)(col0 + wave * 8 + lr8) * g.K + gch;
    cur_tap = -1; cur_r = 0; cur_s = -1;
  };
  // pieces [lo, hi) of k-tile t into ring slot `buf` (A pieces first); the tap's source rows are recomputed when piece 0 of a
  // k-tile enters a new tap, inside a tap the pointers just advance by 64 channels
  auto issue = [&](int t, int buf, int lo, int hi) {
    if (CREID_ABL_ON(g.abl, 2) && t > 0) return;
    if (lo == 0) {
      const int tap = (t * BK) >> g.log2span;
      if (tap != cur_tap) {
        cur_tap = tap;
        if (++cur_s == g.kw) { cur_s = 0; ++cur_r; }
        aok = 0;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          int iy, ix;
          const bool ok = igemm_src_pixel(g, oyx[i] >> 16, (int)(short)(oyx[i] & 0xffff), cur_r, cur_s, iy, ix);
          aptr[i] = ok ? pa.src + (int64_t)(bpix[i] + iy * g.SW + ix) * g.pitch + gch : zpage;
          aok |= ok ? (1u << i) : 0u;
        }
      }
    }
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(smem + buf * STAGE + wave * 512));   // LDS byte address
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p < lo || p >= hi) continue;
      if (p < NAI) {
        pp_glds16(aptr[p], la + p * 8192);
        aptr[p] += ((aok >> p) & 1u) ? BK : 0;
      } else {
        pp_glds16(wp0 + (int64_t)(p - NAI) * 64 * g.K, la + TILE_A * 2 + (p - NAI) * 8192);
      }
    }
    if (hi == NP && lo < hi) wp0 += BK;
  };

  f32x16 acc[MI][NJ];
  int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  // first tile of this workgroup: the whole first k-tile, exposed once per workgroup
  if (tile < ntiles) { tile_setup(tile); issue(0, 0, 0, NP); }

  for (; tile < ntiles; tile += (int)gridDim.x) {
    const int tile_m = tile / pa.tiles_n, tile_n = tile - tile_m * pa.tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    PP_STAMP();
    if constexpr (GLM >= 2) {
      // free-running modes: the whole of k-tile 1 goes out NOW (ring slot 1 doubled as the copy-out staging area until the
      // barrier that ended the previous tile).  vmcnt(NP) then means "k-tile 0 has landed" whatever the previous tile's stores
      // do: loads return in order among themselves, so as long as one k-tile-0 piece is pending so are the NP younger ones.
      if (NS == 3 && nk > 2) { issue(1, 1, 0, NP); issue(2, 2, 0, NP); pp_wait_vm<2 * NP>(); }
      else if (nk > 1) { issue(1, 1, 0, NP); pp_wait_vm<NP>(); }
      else pp_wait_vm<0>();
    } else {
      pp_wait_vm<0>();                                           // k-tile 0 of this tile (and the previous tile's stores)
    }
    pp_barrier();
    PP_STAMP();
    if constexpr (GLM >= 2) {
      // ---- free-running form: ONE barrier per k-tile, every wave software-pipelined on its own.  Fragments are double-buffered
      // in registers (F0 / F1 alternate per 16-wide slice); slice s multiplies while the reads of slice s + 1 are in flight;
      // the k-tile barrier sits between the multiplies of slices 2 and 3: by then every wave has issued (and waited for) its
      // slice-3 reads, so ring slot t & 1 is free for k-tile t + 2, and every wave's pieces of k-tile t + 1 have landed, so
      // slice 3 multiplies while slice 0 of k-tile t + 1 is read.  Pieces of k-tile u are issued in slice 3 of k-tile u - 2
      // and slices 0, 1 of k-tile u - 1.
      constexpr int P0 = (NP * 3 + 7) / 8, P1 = (NP * 6 + 7) / 8;
      s16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
      auto rd = [&](int t, int sl, s16x8 (&a)[MI], s16x8 (&b)[NJ]) {
        const unsigned short* As = smem + (t % NS) * STAGE;
        const unsigned short* Bs = As + TILE_A;
        const int ch = 2 * sl + kh;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int r = wm * (BM / WM) + i * 32 + l31;
          a[i] = *reinterpret_cast<const s16x8*>(&As[r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = wn * (BN / WN) + j * 32 + l31;
          b[j] = *reinterpret_cast<const s16x8*>(&Bs[c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
        }
      };
      // multiply one slice behind the DMA pieces [lo, hi) of k-tile tq (riding BETWEEN the MFMAs, one piece per two, measured
      // no faster: profiles/r04_pp_structure.md)
      auto mm = [&](s16x8 (&a)[MI], s16x8 (&b)[NJ], bool doit, int tq, int slot, int lo, int hi) {
        if (doit) { issue(tq, slot, lo, hi); __builtin_amdgcn_sched_barrier(0); }
        if (CREID_ABL_ON(g.abl, 1)) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = ET::mfma(a[i], b[j], acc[i][j]);
      };
      // (top of the tile: the whole of k-tile 1 was requested behind k-tile 0, see below)
      rd(0, 0, a0, b0);
      // NS slots: k-tile u lives in slot u % NS; its pieces [0, P0) go out in slice 3 of k-tile u - NS (right behind the barrier
      // that frees the slot), [P0, NP) in slices 0, 1 of k-tile u - NS + 1; k-tiles 1 .. NS - 1 went out whole at the top
      for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
        const int tq = t + NS - 1;
        const bool iss = t > 0 && tq < nk;
        rd(t, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0, iss, tq, tq % NS, P0, P1);
        __builtin_amdgcn_sched_barrier(0);
        rd(t, 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1, iss, tq, tq % NS, P1, NP);
        __builtin_amdgcn_sched_barrier(0);
        rd(t, 3, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0, false, 0, 0, 0, 0);
        if (more) {
          // k-tile t + 1 must have landed; with three slots the NP pieces of k-tile t + 2 may still be on their way (loads only
          // in this loop: they return in order)
          if (NS == 3 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          pp_barrier();
          PP_STAMP();
          rd(t + 1, 0, a0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1, more && t + NS < nk, t + NS, t % NS, 0, P0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();                                              // everyone is done with the ring
    } else {
    if (grp == 1) pp_barrier();                                  // group 1 runs half a phase behind group 0

    for (int t = 0; t < nk; ++t) {
      const unsigned short* As = smem + (t & 1) * STAGE;
      const unsigned short* Bs = As + TILE_A;
      const bool more = t + 1 < nk;
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        // ---- fragment half: this group's operands of the phase (the other group multiplies meanwhile)
        s16x8 a[KPH][MI], b[KPH][NJ];
        if (CREID_ABL_ON(g.abl, 4)) {
#pragma unroll
          for (int kq = 0; kq < KPH; ++kq) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[kq][i] = s16x8{1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[kq][j] = s16x8{1, 2, 3, 4, 5, 6, 7, 8};
          }
        } else
#pragma unroll
        for (int kq = 0; kq < KPH; ++kq) {
          const int ch = 2 * (ph * KPH + kq) + kh;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int r = wm * (BM / WM) + i * 32 + l31;
            a[kq][i] = *reinterpret_cast<const s16x8*>(&As[r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int c = wn * (BN / WN) + j * 32 + l31;
            b[kq][j] = *reinterpret_cast<const s16x8*>(&Bs[c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
          }
        }
        if (GLM == 0 && more) issue(t + 1, (t + 1) & 1, pp_piece_lo<NPH, NP>(ph), pp_piece_lo<NPH, NP>(ph + 1));
        if (ph == NPH - 1 && grp == 1) {
          // group 1 reaches the barrier that frees slot t & 1 for the DMA of k-tile t + 2 and opens k-tile t + 1 for group 0:
          // its reads of slot t & 1 must have returned and its pieces of k-tile t + 1 must have landed
          if (more) pp_wait_vm<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        pp_barrier();
        PP_STAMP();
        // ---- multiply half
        __builtin_amdgcn_s_setprio(1);
        if (!CREID_ABL_ON(g.abl, 1))
#pragma unroll
        for (int kq = 0; kq < KPH; ++kq)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = ET::mfma(a[kq][i], b[kq][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (ph == NPH - 1 && grp == 0 && more) pp_wait_vm<0>();   // group 0's pieces of k-tile t + 1
        pp_barrier();
        PP_STAMP();
      }
    }
    if (grp == 0) pp_barrier();                                  // re-join: both groups are done with the ring
    }

    // ---- next tile's first k-tile is requested NOW: it lands while this tile is copied out
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) { tile_setup(next); issue(0, 0, 0, NP); }

    // ---- epilogue: 32-row blocks of every wave row, staged column-major (packed bf16, one ds_write_b64 per 4 rows of a column),
    // read back through the transposing LDS read as 16-byte row chunks; two staging buffers alternate (one barrier per piece)
    const int t4 = lane & 3, q4 = (lane >> 2) & 3, g4 = lane >> 4;
    float s1r[NJ], s2r[NJ];                                      // running column sums of the current 64-row sub-tile
    constexpr int G64 = (BM / WM) / 64;                           // 64-row sub-tiles per wave
    static_assert(G64 >= 1 && G64 == MI / 2, "wave rows");
    float* red = reinterpret_cast<float*>(smem + LDS_MAIN);       // [WM * G64][2][BN]
    constexpr bool affine = EPI == 2;
    const bool relu_now = affine && g.epi_relu && !pa.add_src;
    float esc[NJ], esh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = col0 + wn * (BN / WN) + j * 32 + l31;
      esc[j] = affine ? g.epi_scale[c] : 1.f;
      esh[j] = affine ? g.epi_shift[c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      unsigned short* sb = stg + (i & 1) * STG;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cl = wn * (BN / WN) + j * 32 + l31;
        float s1 = (i & 1) ? s1r[j] : 0.f, s2 = (i & 1) ? s2r[j] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rl = wm * 32 + 8 * q + 4 * kh;
          float v0 = acc[i][j][4 * q], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
          if constexpr (affine) {
            v0 = fmaf(v0, esc[j], esh[j]); v1 = fmaf(v1, esc[j], esh[j]); v2 = fmaf(v2, esc[j], esh[j]); v3 = fmaf(v3, esc[j], esh[j]);
            if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          } else if constexpr (EPI == 1) {
            s1 += v0; s2 = fmaf(v0, v0, s2);
            s1 += v1; s2 = fmaf(v1, v1, s2);
            s1 += v2; s2 = fmaf(v2, v2, s2);
            s1 += v3; s2 = fmaf(v3, v3, s2);
          }
          *reinterpret_cast<uint2*>(&sb[cl * CPT + rl]) = make_uint2(ET::pack2(v0, v1), ET::pack2(v2, v3));
        }
        s1r[j] = s1; s2r[j] = s2;
        if constexpr (EPI == 1) {
          if (i & 1) {                                            // a 64-row sub-tile is complete: lane halves, then to the scratch
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (kh == 0) { red[((wm * G64 + (i >> 1)) * 2 + 0) * BN + cl] = s1; red[((wm * G64 + (i >> 1)) * 2 + 1) * BN + cl] = s2; }
          }
        }
      }
      pp_lds_barrier();                                          // piece i staged
      u32x2 trlo[NIT], trhi[NIT];
      {
        const int sq = lane & 3, sj = (lane >> 2) & 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int Qs = (wave + 8 * it) * 16 + g4 * 4 + sq;
          const unsigned addr = (unsigned)(uintptr_t)&sb[((Qs % CPR) * 8 + sj) * CPT + 4 * (Qs / CPR)];
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                       : "=&v"(trlo[it]), "=&v"(trhi[it]) : "v"(addr), "i"(4 * CPT * 2) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(trlo[it]), "+v"(trhi[it]));
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int Q = (wave + 8 * it) * 16 + g4 * 4 + q4;
        const int ch = Q % CPR, rl = 4 * (Q / CPR) + t4;         // piece-local row: wave row rl / 32, row rl % 32 of its block i
        const int rr = row0 + (rl >> 5) * (BM / WM) + i * 32 + (rl & 31);
        if (rr < g.M) {
          uint4 v = make_uint4(trlo[it].x, trlo[it].y, trhi[it].x, trhi[it].y);
          const int64_t off = (int64_t)rr * g.N + col0 + ch * 8;
          if (pa.add_src) {
            const uint4 av = *reinterpret_cast<const uint4*>(pa.add_src + off);
            unsigned* vw = &v.x; const unsigned* aw = &av.x;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float lo = ET::lo(vw[q]) + ET::lo(aw[q]);
              float hi = ET::hi(vw[q]) + ET::hi(aw[q]);
              if (affine && g.epi_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
              vw[q] = ET::pack2(lo, hi);
            }
          }
          if (!CREID_ABL_ON(g.abl, 8)) *reinterpret_cast<uint4*>(pa.out + off) = v;
        }
      }
      PP_STAMP();
    }
    if constexpr (EPI == 1) {
      // column sums per 128 rows, in the tile kernels' association: (rows of one 64-row sub-tile: blocks, quads, 4 rows) ->
      // lane halves -> the two 64-row sub-tiles of a 128-row group
      pp_lds_barrier();
      constexpr int HALVES = BM / 128;
      for (int idx = tid; idx < HALVES * 2 * BN; idx += NT) {
        const int hv = idx / (2 * BN), which = (idx / BN) & 1, cl = idx % BN;
        if (row0 + hv * 128 < g.M)
          pa.bn_part[((int64_t)(tile_m * HALVES + hv) * 2 + which) * g.N + col0 + cl] =
              red[((2 * hv) * 2 + which) * BN + cl] + red[((2 * hv + 1) * 2 + which) * BN + cl];
      }
    }
    pp_lds_barrier();                                            // staging area (ring slot 1) is free again
  }
}

}  // namespace creid_pp
using namespace creid_pp;

// Launch plan kind 5 (tune.hpp) / CREID_IGEMM_PP: returns CREID_E_SHAPE when the launch is not covered (the caller falls back to
// the tile kernels).  variant = BM / 128 | (BN / 128) << 2 | KPH << 4 | MODE << 8 (MODE 2: free-running form; 0: 256 x 256, 2 slices per phase, pieces beside
// the fragment reads).
static unsigned long long* g_pp_trace_buf = nullptr;
#ifdef CREID_ABL_BUILD
extern "C" int creid_dbg_pp_trace(void* buf) { g_pp_trace_buf = (unsigned long long*)buf; return PP_TRACE_N; }
#endif

int launch_igemm_pp(const IGemmGeom& g, const void* src, const void* wgt, void* out, const void* add_src, float* bn_part,
                    int variant, int dtype, hipStream_t s) {
  if (!creid_is16(dtype)) return CREID_E_SHAPE;
  if (g.log2span < 6 || g.K % 64 != 0 || g.parity || g.add_compact || g.add_mask) return CREID_E_SHAPE;
  if (bn_part && (g.epi_scale || add_src)) return CREID_E_SHAPE;
  int bm = (variant & 3) * 128, bn = ((variant >> 2) & 3) * 128, kph = (variant >> 4) & 7, glm = (variant >> 8) & 3;
  if (bm == 0) bm = 256;
  if (bn == 0) bn = 256;
  if (kph == 0) kph = 2;
  if (g.N % bn != 0) return CREID_E_SHAPE;
  PipeArgs pa{g_pp_trace_buf, (const unsigned short*)src, (const unsigned short*)wgt, (unsigned short*)out, (const unsigned short*)add_src, bn_part,
              (g.M + bm - 1) / bm, g.N / bn};
  const int ntiles = pa.tiles_m * pa.tiles_n;
  static const int ncu = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  int cap = (bm * bn <= 128 * 128) ? 2 * ncu : ncu;
  { const char* e = CREID_KNOB_ENV("CREID_PP_WGS"); const int v = e ? atoi(e) : 0; if (v > 0) cap = v; }   // tests: few workgroups walk many tiles
  const dim3 grid((unsigned)(ntiles < cap ? ntiles : cap)), block(512);
  const int epi = g.epi_scale ? 2 : (bn_part ? 1 : 0);
#define CREID_PP_LAUNCH_E(BM_, BN_, WN_, KPH_, GLM_, EPI_)                                                  \
  do {                                                                                                      \
    if (dtype == CREID_F16) hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, EPI_, F16T>), grid, block, 0, s, g, pa); \
    else hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, EPI_, Bf16T>), grid, block, 0, s, g, pa); \
  } while (0)
#define CREID_PP_LAUNCH(BM_, BN_, WN_, KPH_, GLM_)                                                          \
  do {                                                                                                      \
    if (dtype == CREID_F16) {                                                                               \
      /* f16: eval-mode forward (folded affine), plain forward and (round 5: f16 training) the forward with BatchNorm partials */ \
      if (epi == 2) hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 2, F16T>), grid, block, 0, s, g, pa); \
      else if (epi == 1) hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 1, F16T>), grid, block, 0, s, g, pa); \
      else hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 0, F16T>), grid, block, 0, s, g, pa);        \
    }                                                                                                       \
    else if (epi == 2) hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 2, Bf16T>), grid, block, 0, s, g, pa); \
    else if (epi == 1) hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 1, Bf16T>), grid, block, 0, s, g, pa); \
    else hipLaunchKernelGGL((igemm_bf16_pp_kernel<BM_, BN_, WN_, KPH_, GLM_, 0, Bf16T>), grid, block, 0, s, g, pa);        \
  } while (0)
#define CREID_PP_KPH(BM_, BN_, WN_)                                                                        \
  do {                                                                                                     \
    /* the free-running 256 x 256 form with the BatchNorm-partials epilogue needs 257 registers (one spill): that one     \
       combination runs the ping-pong form (bit-identical results, tests/test_conv_pipe_gpu.py) and is never instantiated */ \
    if (glm == 2 && BM_ == 256 && BN_ == 256 && epi == 1) { if constexpr (BM_ == 256 && BN_ == 256) CREID_PP_LAUNCH_E(BM_, BN_, WN_, 1, 0, 1); } \
    else if (glm == 2 && BM_ == 256 && BN_ == 256) { if constexpr (BM_ == 256 && BN_ == 256) { if (epi == 2) CREID_PP_LAUNCH_E(BM_, BN_, WN_, 1, 2, 2); else CREID_PP_LAUNCH_E(BM_, BN_, WN_, 1, 2, 0); } } \
    else if (glm == 2) { if constexpr (!(BM_ == 256 && BN_ == 256)) CREID_PP_LAUNCH(BM_, BN_, WN_, 1, 2); }   \
    else if (glm != 0) return CREID_E_SHAPE;                                                               \
    else if (kph == 1) CREID_PP_LAUNCH(BM_, BN_, WN_, 1, 0);                                               \
    else if (kph == 2) CREID_PP_LAUNCH(BM_, BN_, WN_, 2, 0);                                               \
    else return CREID_E_SHAPE;                                                                             \
  } while (0)
  if (bm == 256 && bn == 256) CREID_PP_KPH(256, 256, 4);
  else if (bm == 128 && bn == 256) CREID_PP_KPH(128, 256, 4);
  else if (bm == 256 && bn == 128) CREID_PP_KPH(256, 128, 2);
  else if (bm == 128 && bn == 128) CREID_PP_KPH(128, 128, 4);
  else return CREID_E_SHAPE;
#undef CREID_PP_KPH
#undef CREID_PP_LAUNCH
#undef CREID_PP_LAUNCH_E
  return (int)hipGetLastError();
}
