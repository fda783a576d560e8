// Stages D + E fused for metric-only evaluation: CMC / mAP WITHOUT ever writing the m x n distance matrix or
// the ranked index matrix (utils/reid_metric.py:93-151 + utils/eval_reid.py:25-92 of the reference, whose
// `_commpute_batches_double` path exists because 50k x 200k x 4 B does not fit anywhere).
//
// eval_func only needs, per query, the kept-rank of each POSITIVE gallery entry (same pid, different camera):
//     rank(p) = #{kept j : (d_j, j) < (d_p, p)},   kept = everything except (same pid AND same camera)
// and kept = negatives (other pid; never removed) + positives.  So:
//   1. stream_poslist_kernel   per query: distances to its few positives (gathered rows), sorted by (d, index);
//   2. sqdist_count_f32_kernel the full Q.G^T contraction on the f32 MFMA pipe, tile by tile; the epilogue turns
//                              every NEGATIVE's distance into "how many positives of this query rank before it"
//                              (binary search in the query's LDS-resident positive list) and bumps an LDS
//                              histogram -- the tile is consumed in registers and never stored;
//   3. stream_finalize_kernel  prefix sums of the histogram -> rank of every positive -> AP, first match.
// Bit-consistency with the materialised path (creid_sqdist_matrix + creid_rank_rows + creid_cmc_ap_ranked): a
// v_mfma_f32_32x32x2_f32 accumulation is the sequential fmaf chain over k, which kernel 1 reproduces with plain
// fmaf in the same order, and both use the same "(qq + gg) - 2 dot" epilogue, so positives and negatives compare
// on identical bits and ties resolve by gallery index exactly like the stable rank kernel.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {
constexpr int SQ_TM = 64, SQ_TN = 256, SQ_BK = 16;
// LDS operand image of one 16-deep k-tile: [kh = k & 1][quarter = k >> 2][row][e = (k >> 1) & 1] -- the per-lane MFMA operand is
// A[i = lane & 31][k = 2 step + (lane >> 5)], so the two values a lane feeds to the two steps of a quarter are one 8-byte unit:
// a quarter's fragments are 1 + 4 ds_read_b64 per lane (32 lanes x 8 B = one conflict-free 256-B row of banks), and the staging
// side writes a global float4 (k = 4 q .. 4 q + 3) as two ds_write_b64: (x, z) -> kh 0, (y, w) -> kh 1.  Plane pitch = 2 rows +
// 8 dwords: the four quarter planes a 16-lane store group touches (4 rows x 4 quarters) cover 32 distinct banks.
constexpr int SQ_PA = SQ_TM * 2 + 8, SQ_PB = SQ_TN * 2 + 8;
constexpr int PL_MAXC = 128;

// float -> unsigned with the same order (negatives included; squared distances may be slightly negative)
__device__ __forceinline__ unsigned mono_key(float d) {
  const unsigned u = __float_as_uint(d);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
}  // namespace

// ----------------------------------------------------------------------------------------
// 1. positives.  One 128-thread workgroup per query.  Candidates = the gallery rows of the query's pid (CSR slice
//    of g_order, ascending gallery index), kept when their camera differs from the query's.  Thread c runs the fmaf
//    chain of candidate c over its own gallery row (16-B loads, consecutive k: every 128-B line is fetched from L2
//    once and then served by the L1); the query row is wave-uniform, so its values arrive through the scalar cache.
//    A chain is 2048 dependent FMAs (~5 us); thousands of them run side by side.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(PL_MAXC) void stream_poslist_kernel(
    const float* __restrict__ q, const float* __restrict__ g, const float* __restrict__ qq, const float* __restrict__ gg,
    int D, const int32_t* __restrict__ q_slot, const int64_t* __restrict__ csr_off, const int32_t* __restrict__ g_order,
    const int64_t* __restrict__ q_cams, const int64_t* __restrict__ g_cams, int cap, unsigned* __restrict__ pos_key,
    int32_t* __restrict__ pos_idx, int32_t* __restrict__ npos) {
  __shared__ int cand[PL_MAXC];
  __shared__ unsigned skey[PL_MAXC];
  __shared__ int s_n, s_wcnt[2];
  const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned* okey = pos_key + (int64_t)qi * cap;
  int32_t* oidx = pos_idx + (int64_t)qi * cap;
  for (int i = tid; i < cap; i += PL_MAXC) { okey[i] = 0xffffffffu; oidx[i] = 0x7fffffff; }   // padding for the search
  const int slot = q_slot[qi];
  if (slot < 0) { if (tid == 0) npos[qi] = 0; return; }
  const int64_t c0 = csr_off[slot], c1 = csr_off[slot + 1];
  const int64_t qc = q_cams[qi];
  // compaction of the kept candidates in gallery-index order (every pass handles 128 CSR entries)
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int64_t b = c0; b < c1; b += PL_MAXC) {
    const int64_t e = b + tid;
    int gi = -1;
    if (e < c1) { gi = g_order[e]; if (g_cams[gi] == qc) gi = -1; }
    const unsigned long long bal = __ballot(gi >= 0);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    const int base = s_n + (wave ? s_wcnt[0] : 0);
    const int tot = s_wcnt[0] + s_wcnt[1];
    if (gi >= 0) {
      const int p = base + __popcll(bal & lanemask_lt());
      if (p < PL_MAXC) cand[p] = gi;
    }
    __syncthreads();
    if (tid == 0) s_n += tot;
    __syncthreads();
  }
  const int nc = s_n;
  if (nc > cap || nc > PL_MAXC) { if (tid == 0) npos[qi] = -1; return; }   // the caller sends such queries to the general path
  if (nc == 0) { if (tid == 0) npos[qi] = 0; return; }
  const float* __restrict__ qrow = q + (int64_t)qi * D;
  int gi = 0;
  if (tid < nc) {
    gi = cand[tid];
    const float* __restrict__ grow = g + (int64_t)gi * D;
    float acc = 0.f;
    int k = 0;
    const int kend = D & ~15;
    // two 64-byte blocks of the row are in flight while a third is chained (the loop is bound by the load round trip
    // to the Infinity Cache, not by the 16 dependent FMAs of a block)
    float4 n0, n1, n2, n3, m0, m1, m2, m3;
    if (kend > 0) {
      n0 = *reinterpret_cast<const float4*>(grow); n1 = *reinterpret_cast<const float4*>(grow + 4);
      n2 = *reinterpret_cast<const float4*>(grow + 8); n3 = *reinterpret_cast<const float4*>(grow + 12);
      const int k1 = min(16, kend - 16);
      m0 = *reinterpret_cast<const float4*>(grow + k1); m1 = *reinterpret_cast<const float4*>(grow + k1 + 4);
      m2 = *reinterpret_cast<const float4*>(grow + k1 + 8); m3 = *reinterpret_cast<const float4*>(grow + k1 + 12);
    }
    for (; k < kend; k += 16) {                            // the MFMA's own k order: one sequential fmaf chain
      const float4 v0 = n0, v1 = n1, v2 = n2, v3 = n3;
      n0 = m0; n1 = m1; n2 = m2; n3 = m3;
      const int kn = min(k + 32, kend - 16);
      m0 = *reinterpret_cast<const float4*>(grow + kn); m1 = *reinterpret_cast<const float4*>(grow + kn + 4);
      m2 = *reinterpret_cast<const float4*>(grow + kn + 8); m3 = *reinterpret_cast<const float4*>(grow + kn + 12);
      acc = fmaf(qrow[k + 0], v0.x, acc); acc = fmaf(qrow[k + 1], v0.y, acc); acc = fmaf(qrow[k + 2], v0.z, acc); acc = fmaf(qrow[k + 3], v0.w, acc);
      acc = fmaf(qrow[k + 4], v1.x, acc); acc = fmaf(qrow[k + 5], v1.y, acc); acc = fmaf(qrow[k + 6], v1.z, acc); acc = fmaf(qrow[k + 7], v1.w, acc);
      acc = fmaf(qrow[k + 8], v2.x, acc); acc = fmaf(qrow[k + 9], v2.y, acc); acc = fmaf(qrow[k + 10], v2.z, acc); acc = fmaf(qrow[k + 11], v2.w, acc);
      acc = fmaf(qrow[k + 12], v3.x, acc); acc = fmaf(qrow[k + 13], v3.y, acc); acc = fmaf(qrow[k + 14], v3.z, acc); acc = fmaf(qrow[k + 15], v3.w, acc);
    }
    for (; k < D; ++k) acc = fmaf(qrow[k], grow[k], acc);
    skey[tid] = mono_key(fmaf(-2.0f, acc, qq[qi] + gg[gi]));                          // sqdist epilogue, same bits
  }
  __syncthreads();
  if (tid < nc) {                                      // rank by counting over (key, gallery index)
    const unsigned k = skey[tid];
    int pos = 0;
    for (int c = 0; c < nc; ++c) {
      const unsigned kc = skey[c];
      pos += (kc < k || (kc == k && cand[c] < gi)) ? 1 : 0;
    }
    okey[pos] = k; oidx[pos] = gi;
  }
  if (tid == 0) npos[qi] = nc;
}

// ----------------------------------------------------------------------------------------
// 2. streamed contraction + count.  Workgroup = (query tile of 64 rows, slice of the gallery); 4 waves as 2 x 2,
//    each 32 rows x 128 columns (1 x 4 MFMA 32x32 blocks) of a 64 x 256 tile; LDS operand image: see SQ_PA above.
//    Dynamic LDS: this query tile's positive keys [64][cap] and the histogram [64][cap].
// ----------------------------------------------------------------------------------------
// ABL: timing ablations of the k-loop (ablation build only; results wrong): 2 no global loads, 4 no barrier, 8 no LDS writes,
// 16 no fragment reads, 32 no MFMAs inside the loop, 64 global loads always from k-tiles 0 / 1 (cache-hot).
//
// Work split.  A row of the problem is one 64-query tile against the whole gallery, measured in UNITS of 64 gallery columns
// (U per row).  A workgroup owns a contiguous run of units and walks it in tiles of four units (256 columns); the last tile of a
// run may be 1-3 units wide (NJ < 4: each wave then multiplies NJ of its four 32-column blocks -- the blocks are dealt to the
// two column waves alternately, so a narrow tile costs NJ / 4 of a full one).
//   mode 0 (galleries beyond the Infinity Cache): per row `nsplit` runs of `upw` units, ids split-major, so the workgroups that
//          share an L2 walk the SAME gallery tiles at the same time (6250 x 200 000: 0.80 of the MFMA peak against 0.68 with
//          every workgroup streaming its own gallery range);
//   mode 1 (the gallery fits the Infinity Cache -- DukeMTMC 145 MB, 3000 x 15000 123 MB): the rows' units laid end to end and cut
//          into gridDim.x EQUAL runs (a run may end one row and begin the next): every resident slot gets the same number of
//          units, where per-row splitting left 2228 x 17661 at 5 tile-times for 4.72 tiles of work per slot (and 3000 x 15000 at
//          6 for 5.42).  Measured without the gallery sharing of mode 0: no loss at these sizes (profiles/r06_eval_kloop.md).
// (!FULLK -- a feature width that is not a multiple of 16 -- carries the zero-fill masks on top and runs one workgroup per CU.)
template <int ABL, bool FULLK>
__global__ __launch_bounds__(256, FULLK ? 2 : 1) void sqdist_count_f32_kernel(
    const float* __restrict__ q, const float* __restrict__ g, const float* __restrict__ qq, const float* __restrict__ gg,
    int m, int n, int D, const int64_t* __restrict__ q_pids, const int64_t* __restrict__ g_pids, int cap, int log2cap,
    const unsigned* __restrict__ pos_key, const int32_t* __restrict__ pos_idx, const int32_t* __restrict__ npos,
    unsigned* __restrict__ hist_out, int tiles_m, int U, int upw, int mode, int skip_count) {
  __shared__ __attribute__((aligned(16))) float As[2][2][4][SQ_PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][2][4][SQ_PB];
  __shared__ float s_qq[SQ_TM];
  __shared__ long long s_qpid[SQ_TM];
  __shared__ int s_np[SQ_TM];
  __shared__ unsigned s_kmax[SQ_TM];
  extern __shared__ __attribute__((aligned(16))) unsigned dyn[];
  unsigned* s_keys = dyn;                        // [64][cap]
  unsigned* s_hist = dyn + SQ_TM * cap;          // [64][cap]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  // XCD-aware order: consecutive ids land on different XCDs; every XCD gets a contiguous run of ids
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    bid = base + (bid >> 3);
  }
  long long g0, g1;                              // this workgroup's run, in units of the rows laid end to end
  if (mode == 0) {
    const int split = bid / tiles_m, tile_m = bid - split * tiles_m;
    const int u0 = split * upw;
    g0 = (long long)tile_m * U + u0;
    g1 = (long long)tile_m * U + min(U, u0 + upw);
  } else {
    const long long T = (long long)tiles_m * U;
    g0 = bid * T / gridDim.x;
    g1 = (bid + 1) * T / gridDim.x;
  }

  const int lrow = tid >> 2, lkc = tid & 3;
  // Staging addresses: one wave-uniform base per operand (SGPRs; the k advance is scalar arithmetic) + a 32-bit byte offset per
  // lane and row (clamped rows: at most 256 rows x D floats from the base), so a k-tile's fetch costs no vector ALU work.
  const char* abase = nullptr;
  unsigned aoff = 0u;
  const char* bbase = nullptr;
  unsigned boff[4];
  // Two register sets of staged k-tiles: a k-tile is fetched ~1.5 k-tiles before it is written to LDS.
  float4 ra[2], rb[2][4];
  unsigned rmask[2] = {0u, 0u};                    // !FULLK: all ones while the staged k-tile lies inside D
  auto set_tile = [&](int col0) {
    const int c0 = min(col0, n - 1);
    bbase = reinterpret_cast<const char*>(g + (int64_t)c0 * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) boff[i] = (unsigned)(min(col0 + lrow + 64 * i, n - 1) - c0) * (unsigned)D * 4u + 16u * lkc;
  };
  // part 0: the query row and gallery row groups 0 / 1; part 1: groups 2 / 3; part 2: all five; groups >= nj are not fetched
  // (k0 is wave-uniform; branch-free)
  auto gload = [&](auto S_, int k0, int part, int nj) {
    constexpr int S = decltype(S_)::value;
    const int ko = k0 < D ? k0 : 0;                // a k-tile beyond D (the pipeline fetches up to three ahead) reads k-tile 0
    if constexpr (FULLK) {
      const char* ab = abase + (int64_t)ko * 4;
      const char* bb = bbase + (int64_t)ko * 4;
      if (part != 1) ra[S] = *reinterpret_cast<const float4*>(ab + aoff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nj && (part == 2 || (i >> 1) == part)) rb[S][i] = *reinterpret_cast<const float4*>(bb + boff[i]);
    } else {                                       // ... and a lane whose 16 bytes lie beyond D reads its row's first 16 bytes;
      const bool in = k0 + 4 * lkc < D;            // lstore writes zeros for both
      rmask[S] = in ? 0xffffffffu : 0u;
      const unsigned lo = in ? (unsigned)ko * 4u : 0u - 16u * lkc;           // added to aoff / boff (which hold + 16 lkc): no wrap below 0
      if (part != 1) ra[S] = *reinterpret_cast<const float4*>(abase + (aoff + lo));
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nj && (part == 2 || (i >> 1) == part)) rb[S][i] = *reinterpret_cast<const float4*>(bbase + (boff[i] + lo));
    }
  };
  const int so = lrow * 2;                         // global k = 4 lkc + {0..3}: quarter lkc, steps 2 lkc and 2 lkc + 1
  auto st2 = [&](float* p, float v0, float v1, unsigned mask) {    // (x, z) -> kh 0, (y, w) -> kh 1
    if constexpr (FULLK) { p[0] = v0; p[1] = v1; }
    else { p[0] = __uint_as_float(__float_as_uint(v0) & mask); p[1] = __uint_as_float(__float_as_uint(v1) & mask); }
  };
  auto lstore_a = [&](auto S_, int buf) {
    constexpr int S = decltype(S_)::value;
    st2(&As[buf][0][lkc][so], ra[S].x, ra[S].z, rmask[S]);
    st2(&As[buf][1][lkc][so], ra[S].y, ra[S].w, rmask[S]);
  };
  auto lstore_b = [&](auto S_, int buf, int i0, int i1) {
    constexpr int S = decltype(S_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= i0 && i < i1) {
        st2(&Bs[buf][0][lkc][so + 128 * i], rb[S][i].x, rb[S][i].z, rmask[S]);
        st2(&Bs[buf][1][lkc][so + 128 * i], rb[S][i].y, rb[S][i].w, rmask[S]);
      }
  };
  // wave wn multiplies the 32-column blocks wn, wn + 2, wn + 4, wn + 6 of the tile (block b = gallery rows 32 b .. 32 b + 31)
  const int fa = (wm * 32 + l31) * 2, fb = (wn * 32 + l31) * 2;
  const int nk = (D + SQ_BK - 1) / SQ_BK;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int row0 = 0;

  // ---- one tile of NJ units at column col0; `next` >= 0: the following tile's column (its k-tile 0 is fetched under the epilogue)
  auto tile = [&](auto NJ_, int col0, int next) {
    constexpr int NJ = decltype(NJ_)::value;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // The k-loop keeps the matrix pipe fed from ONE wave: a k-tile is four phases of 2 NJ MFMAs (64 cycles each), and everything
    // else is issued in their shadow.  Phase q multiplies quarter q while quarter q + 1's fragments are read (phase 3: the
    // next k-tile's quarter 0 from the other buffer); phases 0 / 1 write the NEXT k-tile (staged in registers since the
    // iteration before last) to the other buffer, phases 2 / 3 refill those registers with the k-tile three ahead -- one global
    // load per MFMA gap: a VMEM issue holds the SIMD's issue port for ~40-60 cycles, hidden only under the 64 cycles of the
    // wave's own MFMA just before it; the one barrier per k-tile sits between phases 2 and 3, where phase 3's operands are
    // already on their way.  (Before round 6: reads, 32 MFMAs, writes, barrier in sequence -- the two waves of a SIMD fell into
    // step and idled together: 0.667 of the MFMA peak on 2228 x 17661, 0.708 on 6250 x 200 000.)
    float2 fa_[2], fb_[2][NJ];                     // fragment double buffer: [phase parity]
    auto frag = [&](auto F_, int buf, int qd) {
      constexpr int F = decltype(F_)::value;
      fa_[F] = *reinterpret_cast<const float2*>(&As[buf][kh][qd][fa]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb_[F][j] = *reinterpret_cast<const float2*>(&Bs[buf][kh][qd][fb + 128 * j]);
    };
    auto mma = [&](auto F_) {                      // steps in k order: step 2 q + e multiplies k = 2 step + kh
      constexpr int F = decltype(F_)::value;
      if constexpr (ABL & 32) return;
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[F].x, fb_[F][j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[F].y, fb_[F][j].y, acc[j], 0, 0, 0);
    };
    // Issue order inside a phase: nA MFMA gaps with cA instructions of class mA each, nB gaps with cB of class mB, the remaining
    // MFMAs bare (0x100 DS read, 0x200 DS write, 0x020 VMEM read; a phase has 2 NJ MFMAs, 1 + NJ fragment reads).
#define CREID_SQ_PHASE(sync, nA, mA, cA, nB, mB, cB)                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < nA; ++i_) {                                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, sync); __builtin_amdgcn_sched_group_barrier(mA, cA, sync); }     \
    _Pragma("unroll") for (int i_ = 0; i_ < nB; ++i_) {                                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, sync); __builtin_amdgcn_sched_group_barrier(mB, cB, sync); }     \
    if constexpr (2 * NJ - (nA) - (nB) > 0) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - (nA) - (nB), sync);   \
    __builtin_amdgcn_sched_barrier(0)
    auto body = [&](auto P_, int t) {              // k-tile t in buffer P = t & 1; stages k-tile t + 1, fetches k-tile t + 3
      constexpr int P = decltype(P_)::value;
      using SS = std::integral_constant<int, P ^ 1>;                // the register set that holds k-tile t + 1
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) frag(I1{}, P, 1);
      if constexpr (!(ABL & 8)) { lstore_a(SS{}, P ^ 1); lstore_b(SS{}, P ^ 1, 0, NJ < 2 ? NJ : 2); }
      mma(I0{});
      if constexpr (NJ == 4) { CREID_SQ_PHASE(0, 5, 0x100, 1, 3, 0x200, 2); }
      else if constexpr (NJ == 3) { CREID_SQ_PHASE(0, 4, 0x100, 1, 2, 0x200, 3); }
      else if constexpr (NJ == 2) { CREID_SQ_PHASE(0, 3, 0x100, 1, 1, 0x200, 6); }
      else { CREID_SQ_PHASE(0, 1, 0x100, 2, 1, 0x200, 4); }
      if constexpr (!(ABL & 16)) frag(I0{}, P, 2);
      if constexpr (!(ABL & 8)) lstore_b(SS{}, P ^ 1, 2, NJ);
      mma(I1{});
      if constexpr (NJ == 4) { CREID_SQ_PHASE(1, 5, 0x100, 1, 2, 0x200, 2); }
      else if constexpr (NJ == 3) { CREID_SQ_PHASE(1, 4, 0x100, 1, 1, 0x200, 2); }
      else if constexpr (NJ == 2) { CREID_SQ_PHASE(1, 3, 0x100, 1, 0, 0x200, 1); }
      else { CREID_SQ_PHASE(1, 2, 0x100, 1, 0, 0x200, 1); }
      if constexpr (!(ABL & 16)) frag(I1{}, P, 3);
      if constexpr (!(ABL & 2)) gload(SS{}, (ABL & 64) ? (t & 1) * SQ_BK : (t + 3) * SQ_BK, 0, NJ);
      mma(I0{});
      if constexpr (NJ == 4) { CREID_SQ_PHASE(2, 5, 0x100, 1, 3, 0x020, 1); }
      else if constexpr (NJ == 3) { CREID_SQ_PHASE(2, 2, 0x100, 2, 3, 0x020, 1); }
      else if constexpr (NJ == 2) { CREID_SQ_PHASE(2, 1, 0x100, 3, 3, 0x020, 1); }
      else { CREID_SQ_PHASE(2, 1, 0x100, 2, 1, 0x020, 2); }
      if constexpr (!(ABL & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) frag(I0{}, P ^ 1, 0);
      if constexpr (!(ABL & 2)) gload(SS{}, (ABL & 64) ? (t & 1) * SQ_BK : (t + 3) * SQ_BK, 1, NJ);
      mma(I1{});
      if constexpr (NJ == 4) { CREID_SQ_PHASE(3, 5, 0x100, 1, 2, 0x020, 1); }
      else if constexpr (NJ == 3) { CREID_SQ_PHASE(3, 4, 0x100, 1, 1, 0x020, 1); }
      else if constexpr (NJ == 2) { CREID_SQ_PHASE(3, 3, 0x100, 1, 0, 0x020, 1); }
      else { CREID_SQ_PHASE(3, 2, 0x100, 1, 0, 0x020, 1); }
    };
#undef CREID_SQ_PHASE
    auto last = [&](int buf) {                     // the tile's last k-tile: nothing left to stage
      frag(I1{}, buf, 1); mma(I0{});
      frag(I0{}, buf, 2); mma(I1{});
      frag(I1{}, buf, 3); mma(I0{});
      mma(I1{});
    };
    __syncthreads();                               // previous tile's readers are done with both LDS buffers
    lstore_a(I0{}, 0); lstore_b(I0{}, 0, 0, NJ);   // k-tile 0 was fetched during the previous tile's epilogue
    __syncthreads();
    gload(I1{}, SQ_BK, 2, NJ);                     // k-tiles 1 and 2 (zeros when there are none)
    gload(I0{}, 2 * SQ_BK, 2, NJ);
    frag(I0{}, 0, 0);
    int t = 0;
    for (; t + 2 < nk; t += 2) { body(I0{}, t); body(I1{}, t + 1); }
    if (t + 2 == nk) { body(I0{}, t); last(1); } else last(0);
    if (next >= 0) { set_tile(next); gload(I0{}, 0, 2, 4); }        // flies while the epilogue runs
    // ---- epilogue: the tile is consumed here (row-major walk: the row's metadata is read once per NJ columns)
    int rbase = wm * 32 + 4 * kh;                  // opaque per tile: the 16 rows' LDS addresses derived from it are recomputed here
    asm volatile("" : "+v"(rbase));                // instead of living in registers (or scratch) across the k-loop
    float gv[NJ];
    long long gp[NJ];
    bool okc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = col0 + (wn + 2 * j) * 32 + l31;
      okc[j] = c < n;
      gv[j] = okc[j] ? gg[c] : 0.f;
      gp[j] = okc[j] ? (long long)g_pids[c] : 0;
    }
    // Two accumulator rows x NJ column blocks = 2 NJ binary searches in flight per lane: the search is a chain of
    // dependent LDS reads (~100 cycles each), so it is the number of INDEPENDENT chains that sets the epilogue time.
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      int rl[2], np[2], lo[2][NJ];
      unsigned key[2][NJ];
      bool live[2][NJ];
      const unsigned* K[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        rl[h] = rbase + ((r + h) & 3) + 8 * ((r + h) >> 2);
        np[h] = (skip_count & 1) ? 0 : s_np[rl[h]];
        const long long qp = s_qpid[rl[h]];
        const float qv = s_qq[rl[h]];
        const unsigned kmax = s_kmax[rl[h]];
        K[h] = s_keys + (rl[h] << log2cap);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          key[h][j] = mono_key(fmaf(-2.0f, acc[j][r + h], qv + gv[j]));
          // positives / removed entries (same pid) are not counted; behind every positive: affects no rank
          live[h][j] = np[h] > 0 && okc[j] && gp[j] != qp && key[h][j] <= kmax;
          lo[h][j] = 0;
        }
      }
      if (np[0] == 0 && np[1] == 0) continue;                      // uniform per wave half
      for (int step = cap >> 1; step > 0; step >>= 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < NJ; ++j) lo[h][j] += (K[h][lo[h][j] + step - 1] < key[h][j]) ? step : 0;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          int l = lo[h][j];
          l += (K[h][l] < key[h][j]) ? 1 : 0;                      // l = #positives with key strictly below
          if (live[h][j]) {
            if (l < np[h] && K[h][l] == key[h][j]) {               // ties: by gallery index (rare)
              const int c = col0 + (wn + 2 * j) * 32 + l31;
              int rr_ = row0 + rl[h];
              asm volatile("" : "+v"(rr_));                        // keeps 16 rows' index pointers out of the k-loop's registers
              while (l < np[h] && K[h][l] == key[h][j] && pos_idx[(int64_t)rr_ * cap + l] < c) ++l;
            }
            if (l < np[h]) atomicAdd(&s_hist[(rl[h] << log2cap) + l], 1u);
          }
        }
    }
  };

  while (g0 < g1) {                               // the run's segments: one per row it touches
    const int row = (int)(g0 / U), u0 = (int)(g0 - (long long)row * U);
    const int u1 = (int)min((long long)U, u0 + (g1 - g0));
    g0 += u1 - u0;
    row0 = row * SQ_TM;
    __syncthreads();                               // the previous segment's histogram has been flushed
    int ts = tid;                                  // opaque: the set-up's LDS addresses are recomputed per segment instead of
    asm volatile("" : "+v"(ts));                   // occupying registers (or scratch) across the k-loops
    for (int i = ts; i < SQ_TM * cap; i += 256) {
      const int r = i >> log2cap, rr = row0 + r;
      s_keys[i] = rr < m ? pos_key[(int64_t)rr * cap + (i & (cap - 1))] : 0xffffffffu;
      s_hist[i] = 0u;
    }
    if (ts < SQ_TM) {
      const int rr = row0 + ts;
      const int np = rr < m ? npos[rr] : 0;
      s_np[ts] = np > 0 ? np : 0;
      s_qq[ts] = rr < m ? qq[rr] : 0.f;
      s_qpid[ts] = rr < m ? (long long)q_pids[rr] : 0;
      s_kmax[ts] = np > 0 ? pos_key[(int64_t)rr * cap + np - 1] : 0u;
    }
    abase = reinterpret_cast<const char*>(q + (int64_t)min(row0, m - 1) * D);
    aoff = (unsigned)(min(row0 + lrow, m - 1) - min(row0, m - 1)) * (unsigned)D * 4u + 16u * lkc;
    const int cend = u1 * 64;
    int col = u0 * 64;
    set_tile(col);
    gload(I0{}, 0, 2, 4);
    while (col < cend) {
      const int nj = min(4, (cend - col) >> 6), next = col + 256 < cend ? col + 256 : -1;
      if (nj == 4) tile(std::integral_constant<int, 4>{}, col, next);
      else if (nj == 3) tile(std::integral_constant<int, 3>{}, col, next);
      else if (nj == 2) tile(std::integral_constant<int, 2>{}, col, next);
      else tile(std::integral_constant<int, 1>{}, col, next);
      col += 256;
    }
    __syncthreads();
    for (int i = ts; i < SQ_TM * cap; i += 256) {
      const unsigned v = s_hist[i];
      const int rr = row0 + (i >> log2cap);
      if (v && rr < m) atomicAdd(&hist_out[(int64_t)rr * cap + (i & (cap - 1))], v);      // integer: order-independent
    }
  }
}

// ----------------------------------------------------------------------------------------
// 3. per-query AP / first match from the histogram (one thread per query; float64 like utils/eval_reid.py:75-79)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_finalize_kernel(const int32_t* __restrict__ npos,
                                                              const unsigned* __restrict__ hist, int m, int cap,
                                                              uint8_t* __restrict__ out_valid, double* __restrict__ out_ap,
                                                              int32_t* __restrict__ out_first) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= m) return;
  const int np = npos[qi];
  if (np <= 0) { out_valid[qi] = np < 0 ? 2 : 0; out_ap[qi] = 0.0; out_first[qi] = -1; return; }
  const unsigned* h = hist + (int64_t)qi * cap;
  long long before = 0;
  double ap = 0.0;
  for (int s = 0; s < np; ++s) {
    before += h[s];                                    // negatives ahead of positive s
    ap += (double)(s + 1) / (double)(before + s + 1);  // matches so far / 1-based kept position
  }
  out_valid[qi] = 1;
  out_ap[qi] = ap / (double)np;
  out_first[qi] = (int32_t)h[0];
}

// ----------------------------------------------------------------------------------------
// 0. the index the three kernels above walk, built ON THE DEVICE (utils/eval_reid.py:36-65 does the equivalent per query
//    on the host: `matches`, `remove`, `keep`): gallery grouped by pid as a CSR over the DENSE pid range [pmin, pmin + R),
//    slot = pid - pmin.  Counting sort: histogram -> exclusive scan -> scatter.  The order inside a pid group is whatever the
//    atomics give -- stream_poslist_kernel re-ranks its candidates by (distance, gallery index), so nothing downstream
//    depends on it.  Then per query: #positives = group size - #(same pid AND same camera), their maximum (-> LDS list
//    capacity) and the number of queries beyond PL_MAXC (-> general path).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_count_kernel(const int64_t* __restrict__ g_pids, int n, int64_t pmin, int R,
                                                         int32_t* __restrict__ count) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t s = g_pids[i] - pmin;
    if (s >= 0 && s < R) atomicAdd(&count[s], 1);
  }
}

// one workgroup: csr[0..R] = exclusive scan of count (int64, the layout stream_poslist_kernel reads); cursor[s] = 0
__global__ __launch_bounds__(1024) void plan_scan_kernel(const int32_t* __restrict__ count, int R, int64_t* __restrict__ csr,
                                                         int32_t* __restrict__ cursor, int32_t* __restrict__ stats) {
  __shared__ long long wsum[16];
  __shared__ long long carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_s = 0; stats[0] = 0; stats[1] = 0; }
  __syncthreads();
  for (int base = 0; base < R; base += 1024) {
    const int i = base + tid;
    const long long v = i < R ? (long long)count[i] : 0;
    long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    long long off = carry_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (i < R) { csr[i] = off + x - v; cursor[i] = 0; }
    __syncthreads();
    if (tid == 1023) carry_s = off + x;
    __syncthreads();
  }
  if (tid == 0) csr[R] = carry_s;
}

__global__ __launch_bounds__(256) void plan_scatter_kernel(const int64_t* __restrict__ g_pids, int n, int64_t pmin, int R,
                                                           const int64_t* __restrict__ csr, int32_t* __restrict__ cursor,
                                                           int32_t* __restrict__ g_order) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t s = g_pids[i] - pmin;
    if (s >= 0 && s < R) g_order[csr[s] + atomicAdd(&cursor[s], 1)] = i;
  }
}

// one wave per query
__global__ __launch_bounds__(256) void plan_query_kernel(const int64_t* __restrict__ q_pids, const int64_t* __restrict__ q_cams,
                                                         const int64_t* __restrict__ g_cams, const int64_t* __restrict__ csr,
                                                         const int32_t* __restrict__ g_order, int64_t pmin, int R, int m,
                                                         int32_t* __restrict__ q_slot, int32_t* __restrict__ n_pos,
                                                         int32_t* __restrict__ stats) {
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (qi >= m) return;
  const int64_t s = q_pids[qi] - pmin;
  int slot = -1, np = 0;
  if (s >= 0 && s < R) {
    const int64_t c0 = csr[s], c1 = csr[s + 1];
    if (c1 > c0) {
      slot = (int)s;
      const int64_t qc = q_cams[qi];
      int same = 0;
      for (int64_t e = c0 + lane; e < c1; e += 64) same += g_cams[g_order[e]] == qc ? 1 : 0;
      same = wave_sum_i(same);
      np = (int)(c1 - c0) - same;
    }
  }
  if (lane == 0) {
    q_slot[qi] = slot; n_pos[qi] = np;
    if (np > PL_MAXC) atomicAdd(&stats[1], 1);
    else atomicMax(&stats[0], np);
  }
}

extern "C" {

/* Device-side index for the streamed evaluation.  g_pids / g_cams int64[n], q_pids / q_cams int64[m] on the device; the pid
 * range [pmin, pmin + R) must cover every gallery pid (the caller knows min / max from the host arrays it uploaded).
 * Outputs: csr_off int64[R + 1], g_order int32[n], q_slot int32[m] (pid - pmin, or -1 when the pid has no gallery entry),
 * n_pos int32[m] (same pid, other camera), stats int32[2] = {max n_pos among queries with n_pos <= 128, #queries beyond}.
 * scratch: int32[2 * R].  Five launches, no host synchronisation. */
int creid_stream_plan(const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_cams, const int64_t* g_cams, int64_t m,
                      int64_t n, int64_t pmin, int64_t R, int64_t* csr_off, int32_t* g_order, int32_t* q_slot, int32_t* n_pos,
                      int32_t* stats, int32_t* scratch, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n > 0 && R > 0);
  CREID_CHECK_ARG(q_pids && g_pids && q_cams && g_cams && csr_off && g_order && q_slot && n_pos && stats && scratch);
  if (n > 0x7ffffff0LL || m > 0x7ffffff0LL || R > (1LL << 26)) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  int32_t* count = scratch;
  int32_t* cursor = scratch + R;
  hipError_t e = hipMemsetAsync(count, 0, (size_t)R * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  const unsigned gb = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(plan_count_kernel, dim3(gb), dim3(256), 0, s, g_pids, (int)n, pmin, (int)R, count);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, s, count, (int)R, csr_off, cursor, stats);
  hipLaunchKernelGGL(plan_scatter_kernel, dim3(gb), dim3(256), 0, s, g_pids, (int)n, pmin, (int)R, csr_off, cursor, g_order);
  if (m > 0)
    hipLaunchKernelGGL(plan_query_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, q_pids, q_cams, g_cams, csr_off, g_order,
                       pmin, (int)R, (int)m, q_slot, n_pos, stats);
  CREID_LAUNCH_RET();
}

int creid_stream_poslist(const float* q, const float* g, const float* qq, const float* gg, int64_t m, int64_t n,
                         int64_t D, const int32_t* q_slot, const int64_t* csr_off, const int32_t* g_order,
                         const int64_t* q_cams, const int64_t* g_cams, int32_t cap, uint32_t* pos_key, int32_t* pos_idx,
                         int32_t* npos, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n > 0 && D > 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(q && g && qq && gg && q_slot && csr_off && g_order && q_cams && g_cams && pos_key && pos_idx && npos);
  if (cap < 2 || cap > PL_MAXC || (cap & (cap - 1)) != 0) return CREID_E_SHAPE;
  if (n > 0x7ffffff0LL || m > 0x7ffffff0LL) return CREID_E_SHAPE;
  hipLaunchKernelGGL(stream_poslist_kernel, dim3((unsigned)m), dim3(PL_MAXC), 0, as_stream(stream), q, g, qq, gg, (int)D, q_slot,
                     csr_off, g_order, q_cams, g_cams, (int)cap, pos_key, pos_idx, npos);
  CREID_LAUNCH_RET();
}

int creid_stream_count(const float* q, const float* g, const float* qq, const float* gg, int64_t m, int64_t n, int64_t D,
                       const int64_t* q_pids, const int64_t* g_pids, int32_t cap, const uint32_t* pos_key,
                       const int32_t* pos_idx, const int32_t* npos, uint32_t* hist, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n > 0 && D > 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(q && g && qq && gg && q_pids && g_pids && pos_key && pos_idx && npos && hist);
  if (cap < 2 || cap > PL_MAXC || (cap & (cap - 1)) != 0 || D % 4 != 0) return CREID_E_SHAPE;
  if (n > 0x7ffffff0LL || m > 0x7ffffff0LL) return CREID_E_SHAPE;
  int log2cap = 0;
  while ((1 << log2cap) < cap) ++log2cap;
  const int tiles_m = (int)((m + SQ_TM - 1) / SQ_TM), tiles_n = (int)((n + SQ_TN - 1) / SQ_TN);
  const int U = (int)((n + 63) / 64);                           // units of 64 gallery columns per row of query tiles
  // enough workgroups for two per CU, but never fewer than ~4 gallery tiles per workgroup (per-tile restart cost)
  static const int target = [] { const char* e = getenv("CREID_STREAM_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  // timing ablation only (CREID_STREAM_NOEPI=1: contraction without the count epilogue -- results are then wrong)
  static const int skip_count = creid_ablation_env("CREID_STREAM_NOEPI");
  // mode 0: never MORE than `tper_max` gallery tiles per workgroup: the grid overshoots the 512 slots by up to tiles_m - 1
  // workgroups, which start when the first ones finish -- harmless when a workgroup is 5 tiles long, a whole second round on an
  // idle chip when it is 131 (6250 x 200 000: 588 workgroups, 66.0 ms; with <= 8 tiles per workgroup 46.1 ms; HBM-side traffic by
  // the counters the same under both rules -- profiles/r05_stream_grid.md)
  static const int tper_max = [] { const char* e = getenv("CREID_STREAM_TPER"); int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  int nsplit = (target + tiles_m - 1) / tiles_m;
  if (nsplit < (tiles_n + tper_max - 1) / tper_max) nsplit = (tiles_n + tper_max - 1) / tper_max;
  if (nsplit > tiles_n) nsplit = tiles_n;
  if (nsplit < 1) nsplit = 1;
  const int t_per = (tiles_n + nsplit - 1) / nsplit;
  nsplit = (tiles_n + t_per - 1) / t_per;                       // drop empty slices
  // mode 1 (equal runs of units over the resident slots): only while the gallery fits the Infinity Cache beside the queries, the
  // grid of mode 0 is a single round, and the equal run is shorter than mode 0's longest workgroup by more than the narrow tile
  // and the second segment cost (~a quarter tile: 2228 x 17661 -- 4.75 tiles against 5 -- measured EQUAL in both modes,
  // 3000 x 15000 -- 5.5 against 6 -- 5.6 % faster in mode 1; profiles/r06_eval_kloop.md).  CREID_STREAM_BALANCE=0 / 1 forces a
  // mode (the tests run both).
  const long long T = (long long)tiles_m * U;
  const long long wg1 = T / 4 < target ? (T / 4 > 0 ? T / 4 : 1) : target;
  const char* bal_e = CREID_KNOB_ENV("CREID_STREAM_BALANCE");
  const int bal = (bal_e && *bal_e) ? atoi(bal_e) : -1;
  const double run1 = (double)((T + wg1 - 1) / wg1) / 4.0 + 0.3;
  const int mode = bal >= 0 ? (bal != 0)
                            : ((double)n * (double)D * 4.0 <= 192e6 && (long long)tiles_m * nsplit <= target && run1 < (double)t_per);
  const int upw = 4 * t_per;
  const unsigned grid = mode == 0 ? (unsigned)(tiles_m * nsplit) : (unsigned)wg1;
  const size_t dyn = (size_t)2 * SQ_TM * cap * sizeof(unsigned);
#define CREID_COUNT_LAUNCH_(A, F)                                                                                     \
  do {                                                                                                                 \
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(sqdist_count_f32_kernel<A, F>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize,                  \
                                                          2 * SQ_TM * PL_MAXC * (int)sizeof(unsigned));                \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                    \
    hipLaunchKernelGGL((sqdist_count_f32_kernel<A, F>), dim3(grid), dim3(256), dyn, as_stream(stream),                 \
                       q, g, qq, gg, (int)m, (int)n, (int)D, q_pids, g_pids, (int)cap, log2cap, pos_key, pos_idx, npos, hist,  \
                       tiles_m, U, upw, mode, skip_count & 1);                                                         \
  } while (0)
#define CREID_COUNT_LAUNCH(A)                                                                                          \
  do { if (D % SQ_BK == 0) CREID_COUNT_LAUNCH_(A, true); else CREID_COUNT_LAUNCH_(0, false); } while (0)
#ifdef CREID_ABL_BUILD
  switch (skip_count >> 1) {
    case 1: CREID_COUNT_LAUNCH(2); break;
    case 2: CREID_COUNT_LAUNCH(4); break;
    case 3: CREID_COUNT_LAUNCH(6); break;
    case 4: CREID_COUNT_LAUNCH(8); break;
    case 7: CREID_COUNT_LAUNCH(14); break;
    case 8: CREID_COUNT_LAUNCH(16); break;
    case 15: CREID_COUNT_LAUNCH(30); break;
    case 16: CREID_COUNT_LAUNCH(32); break;
    case 32: CREID_COUNT_LAUNCH(64); break;
    default: CREID_COUNT_LAUNCH(0); break;
  }
#else
  CREID_COUNT_LAUNCH(0);
#endif
#undef CREID_COUNT_LAUNCH
#undef CREID_COUNT_LAUNCH_
  CREID_LAUNCH_RET();
}

int creid_stream_finalize(const int32_t* npos, const uint32_t* hist, int64_t m, int32_t cap, uint8_t* valid, double* ap,
                          int32_t* first, void* stream) {
  CREID_CHECK_ARG(m >= 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(npos && hist && valid && ap && first && cap >= 2);
  hipLaunchKernelGGL(stream_finalize_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, as_stream(stream), npos, hist,
                     (int)m, (int)cap, valid, ap, first);
  CREID_LAUNCH_RET();
}

}  // extern "C"
