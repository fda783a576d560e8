// Stage D/E: per-row argsort of the distance matrix (utils/reid_metric.py:129,132) and the
// CMC / AP scan over the ranked row (utils/eval_reid.py:44-90).
//
// rank_rows: one persistent 1024-thread workgroup per row slot; a stable LSD radix sort
// (4 passes x 8 bits) of (orderable(dist), index) pairs.  Each of the 16 waves owns a
// contiguous segment of the row, keeps a private digit histogram in LDS and scatters its
// segment in order with wave-level match (8 ballots) -- no barrier inside the scatter loop.
// Stability makes ties resolve by gallery index.  The row (n x 16 B of scratch) stays
// L2-resident; HBM traffic is the 4 B/pair read of the matrix and the 8 B/pair index write.
#include "common.hpp"
#include <stdlib.h>

namespace {
constexpr int RT = 1024;          // threads per workgroup
constexpr int RW = RT / 64;       // waves
constexpr int MAX_SLOTS = 512;    // persistent workgroups (row slots of scratch)

__device__ __forceinline__ unsigned orderable(float d) {
  d = d + 0.0f;  // -0 -> +0
  unsigned u = __float_as_uint(d);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
}  // namespace

__global__ __launch_bounds__(RT) void rank_rows_kernel(const float* __restrict__ dist, int64_t m, int64_t n,
                                                       int64_t ld, int64_t* __restrict__ out_idx,
                                                       unsigned* __restrict__ ws,
                                                       const uint8_t* __restrict__ only_flagged) {
  __shared__ unsigned hist[RW][256];   // per-wave digit counts, then per-wave scatter offsets
  __shared__ unsigned total[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned* keyA = ws + (size_t)blockIdx.x * 4 * n;
  unsigned* keyB = keyA + n;
  unsigned* idxA = keyB + n;
  unsigned* idxB = idxA + n;
  const int64_t seg = (n + RW - 1) / RW;
  const int64_t s0 = min((int64_t)wave * seg, n), s1 = min(s0 + seg, n);
  const unsigned long long lt = lanemask_lt();

  for (int64_t row = blockIdx.x; row < m; row += gridDim.x) {
    if (only_flagged && !only_flagged[row]) continue;     // (uniform) rows the LDS bucket kernel already ranked
    const float* drow = dist + row * ld;
    int64_t* orow = out_idx + row * n;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = pass * 8;
      const unsigned* kin = (pass & 1) ? keyA : keyB;   // pass1 reads A, pass2 reads B, pass3 reads A
      const unsigned* iin = (pass & 1) ? idxA : idxB;
      unsigned* kout = (pass & 1) ? keyB : keyA;        // pass0 writes A, pass1 writes B, ...
      unsigned* iout = (pass & 1) ? idxB : idxA;
      // (a) zero histograms
      for (int i = tid; i < RW * 256; i += RT) (&hist[0][0])[i] = 0;
      __syncthreads();
      // (b) per-wave histogram of its own segment
      for (int64_t i = s0 + lane; i < s1; i += 64) {
        const unsigned k = (pass == 0) ? orderable(drow[i]) : kin[i];
        atomicAdd(&hist[wave][(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      // (c) offsets: digit-major, wave-minor exclusive scan
      if (tid < 256) {
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < RW; ++w) { unsigned c = hist[w][tid]; hist[w][tid] = run; run += c; }
        total[tid] = run;
      }
      __syncthreads();
      if (wave == 0) {  // exclusive scan of 256 totals by one wave (4 per lane)
        unsigned t0 = total[4 * lane], t1 = total[4 * lane + 1], t2 = total[4 * lane + 2], t3 = total[4 * lane + 3];
        unsigned s = t0 + t1 + t2 + t3, incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        unsigned ex = incl - s;
        total[4 * lane] = ex; total[4 * lane + 1] = ex + t0; total[4 * lane + 2] = ex + t0 + t1;
        total[4 * lane + 3] = ex + t0 + t1 + t2;
      }
      __syncthreads();
      for (int i = tid; i < RW * 256; i += RT) (&hist[0][0])[i] += total[i & 255];
      __syncthreads();
      // (d) ordered scatter of this wave's segment, 64 elements per step
      volatile unsigned* off = hist[wave];
      for (int64_t base = s0; base < s1; base += 64) {
        const int64_t i = base + lane;
        const bool valid = i < s1;
        unsigned k = 0, v = 0;
        if (valid) {
          if (pass == 0) { k = orderable(drow[i]); v = (unsigned)i; }
          else { k = kin[i]; v = iin[i]; }
        }
        const unsigned dgt = (k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (dgt >> b) & 1u;
          const unsigned long long bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        const unsigned rank = __popcll(peers & lt);
        unsigned pos = 0;
        if (valid) pos = off[dgt] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) off[dgt] = pos + (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
          if (pass == 3) orow[pos] = (int64_t)v;
          else { kout[pos] = k; iout[pos] = v; }
        }
      }
      __syncthreads();  // all scatters visible (same CU, L1 write-through to L2 + block barrier)
      __threadfence_block();
    }
  }
}

// ----------------------------------------------------------------------------------------
// rank_rows for rows that fit in LDS (n <= ~22k: Market-1501 / DukeMTMC galleries): ONE pass over the row
// instead of four global radix passes.  The row's keys are mapped MONOTONICALLY onto 4096 buckets between the
// row's min and max key (bucket order == key order across buckets), counted, scattered into LDS grouped by
// bucket (order inside a bucket arbitrary), and every element then finds its rank inside its bucket by counting
// the members that precede it in (key, index) order -- a total order, so the result is exactly the stable
// argsort.  Buckets hold ~n/4096 elements (x5 at the mode of a bell-shaped row; members are compared four at a
// time with 16-byte LDS reads); a row whose largest bucket exceeds RL_MAX_BUCKET (pathological ties)
// is flagged and left to rank_rows_kernel.  HBM/L2 traffic: the 4 B/pair read (x3, L2-hot) and the 8 B/pair
// index write, nothing else.
// ----------------------------------------------------------------------------------------
constexpr int RL_NB = 4096, RL_MAX_BUCKET = 512, RL_KPT = 21;     // RL_KPT x 1024 threads >= n

__device__ __forceinline__ unsigned rl_bucket(unsigned k, unsigned kmin, float inv) {
  const unsigned b = (unsigned)((float)(k - kmin) * inv);       // float ops are monotone: bucket(k) non-decreasing in k
  return b < (unsigned)RL_NB ? b : (unsigned)(RL_NB - 1);
}

// Optional per-query evaluation (utils/eval_reid.py:36-90 for plain camera ids) while the ranked row is still in LDS: the
// separate CMC / AP pass re-read the int64 index matrix (8 B per pair, the largest stream of the evaluation) only to gather
// two labels per entry.  q_pids == nullptr: ranking only.
struct RankEval {
  const int64_t* q_pids; const int64_t* g_pids; const int64_t* q_cams; const int64_t* g_cams;
  uint8_t* valid; double* ap; int32_t* first;
};

__global__ __launch_bounds__(RT) void rank_rows_lds_kernel(const float* __restrict__ dist, int64_t m, int n, int64_t ld,
                                                           int64_t* __restrict__ out_idx,
                                                           uint8_t* __restrict__ fallback, RankEval ev) {
  extern __shared__ __attribute__((aligned(16))) unsigned rl_smem[];
  const int n4 = (n + 3) & ~3;
  unsigned* keys2 = rl_smem;                                    // [n4] keys grouped by bucket
  unsigned* off = keys2 + n4;                                   // [RL_NB + 1] bucket start offsets
  unsigned* cnt = off + RL_NB + 4;                              // [RL_NB] counters / cursors (16-B aligned)
  unsigned* red = cnt + RL_NB;                                  // [3 * RW] reduction scratch
  unsigned short* idx2 = reinterpret_cast<unsigned short*>(red + 3 * RW);   // [n] gallery index of each slot
  // fused evaluation: (keep, match) of every GALLERY entry for this query as two bitmaps, [ceil(n / 64)] 64-bit words each,
  // filled in phase A from coalesced label loads -- the evaluation then looks ranked entries up in LDS instead of gathering
  // two labels per entry from global memory (64 distinct cache lines per wave instruction: 18 us per row, measured)
  unsigned long long* bm_keep = reinterpret_cast<unsigned long long*>(idx2 + n4);
  unsigned long long* bm_match = bm_keep + ((n + 63) >> 6);
  for (int64_t row = blockIdx.x; row < m; row += gridDim.x) {
    // the thread index is re-derived behind an opaque copy in every row: nothing thread-dependent is hoisted out of the row
    // loop and kept live across its ~5000 instructions (at the 128-VGPR cap of a 1024-thread workgroup two such addresses
    // were spilled to scratch, next to 90 scalar registers parked in vector lanes)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const float* drow = dist + row * ld;
    if (ev.q_pids) {
      // (keeping all 21 label loads of a thread in flight next to the 21 distance loads costs 42 more live registers at the
      // 128-VGPR cap of a 1024-thread workgroup: 129 spills, the whole kernel 25 % slower -- measured; three per trip it is)
      const long long qp = ev.q_pids[row], qc = ev.q_cams[row];
#pragma unroll 3
      for (int j = 0; j < RL_KPT; ++j) {
        const int i = tid + j * RT;
        if (i - lane >= n) break;                                  // (wave-uniform)
        const bool in = i < n;
        const bool match = in && ev.g_pids[i] == qp;
        const bool keep = in && !(match && ev.g_cams[i] == qc);    // camera fetched for the rare matches only
        const unsigned long long km = __ballot(keep), mm = __ballot(match && keep);
        if (lane == 0) { bm_keep[i >> 6] = km; bm_match[i >> 6] = mm; }
      }
    }
    // ---- A: the row's keys go into registers once (RL_KPT independent loads in flight per thread; the three
    //         passes below would otherwise each pay ~n/1024 dependent L2 round trips), min / max key
    unsigned kreg[RL_KPT];
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j) {
      const int i = tid + j * RT;
      kreg[j] = i < n ? orderable(drow[i]) : 0u;
    }
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j)
      if (tid + j * RT < n) { kmin = min(kmin, kreg[j]); kmax = max(kmax, kreg[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
      kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
    }
    if (lane == 0) { red[wave] = kmin; red[RW + wave] = kmax; }
    for (int i = tid; i < RL_NB; i += RT) cnt[i] = 0;
    __syncthreads();
    kmin = red[0]; kmax = red[RW];
#pragma unroll
    for (int w = 1; w < RW; ++w) { kmin = min(kmin, red[w]); kmax = max(kmax, red[RW + w]); }
    const float inv = (float)RL_NB / ((float)(kmax - kmin) + 1.0f);
    // ---- B: bucket histogram
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j)
      if (tid + j * RT < n) atomicAdd(&cnt[rl_bucket(kreg[j], kmin, inv)], 1u);
    __syncthreads();
    // ---- exclusive scan of the 4096 counts (4 per thread), largest bucket
    const uint4 c4 = *reinterpret_cast<const uint4*>(&cnt[4 * tid]);
    const unsigned csum = (c4.x + c4.y) + (c4.z + c4.w);
    unsigned incl = csum, big = max(max(c4.x, c4.y), max(c4.z, c4.w));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, o, 64); if (lane >= o) incl += v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned)__shfl_xor((int)big, o, 64));
    __syncthreads();                                             // red[] reads above are done
    if (lane == 63) red[wave] = incl;
    if (lane == 0) red[2 * RW + wave] = big;
    __syncthreads();
    unsigned base = 0;
    big = 0;
#pragma unroll
    for (int w = 0; w < RW; ++w) { if (w < wave) base += red[w]; big = max(big, red[2 * RW + w]); }
    const unsigned ex = base + incl - csum;
    off[4 * tid] = ex; off[4 * tid + 1] = ex + c4.x; off[4 * tid + 2] = ex + c4.x + c4.y;
    off[4 * tid + 3] = ex + c4.x + c4.y + c4.z;
    if (tid == RT - 1) off[RL_NB] = ex + csum;
    *reinterpret_cast<uint4*>(&cnt[4 * tid]) = make_uint4(0u, 0u, 0u, 0u);     // becomes the scatter cursor
    __syncthreads();
    if (big > (unsigned)RL_MAX_BUCKET) {                         // uniform: leave the row to the radix kernel
      if (tid == 0) fallback[row] = 1;
      continue;
    }
    if (tid == 0) fallback[row] = 0;
    // ---- C: scatter (key, index) into bucket-grouped LDS
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j) {
      const int i = tid + j * RT;
      if (i < n) {
        const unsigned k = kreg[j];
        const unsigned b = rl_bucket(k, kmin, inv);
        const unsigned p = off[b] + atomicAdd(&cnt[b], 1u);
        keys2[p] = k; idx2[p] = (unsigned short)i;
      }
    }
    __syncthreads();
    // ---- D: rank inside the bucket by (key, index) -> final position (kept in registers) ...
    unsigned fpk[RL_KPT];                                        // (final position << 16) | gallery index: both < RL_KPT * RT <= 65536
    static_assert(RL_KPT * RT <= 65536, "packed (position, index) pairs");
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j) {
      const int p = tid + j * RT;
      fpk[j] = 0;
      if (p < n) {
        const unsigned k = keys2[p];
        const unsigned id = idx2[p];
        const unsigned b = rl_bucket(k, kmin, inv);
        const unsigned lo = off[b], hi = off[b + 1];
        unsigned r = 0;
        for (unsigned q = lo & ~3u; q < hi; q += 4) {           // four members per step (16-B + 8-B LDS reads)
          const uint4 kq = *reinterpret_cast<const uint4*>(&keys2[q]);
          const uint2 iq = *reinterpret_cast<const uint2*>(&idx2[q]);
          const unsigned i0 = iq.x & 0xffffu, i1 = iq.x >> 16, i2 = iq.y & 0xffffu, i3 = iq.y >> 16;
          r += (q + 0 >= lo && q + 0 < hi && (kq.x < k || (kq.x == k && i0 < id))) ? 1u : 0u;
          r += (q + 1 >= lo && q + 1 < hi && (kq.y < k || (kq.y == k && i1 < id))) ? 1u : 0u;
          r += (q + 2 >= lo && q + 2 < hi && (kq.z < k || (kq.z == k && i2 < id))) ? 1u : 0u;
          r += (q + 3 >= lo && q + 3 < hi && (kq.w < k || (kq.w == k && i3 < id))) ? 1u : 0u;
        }
        fpk[j] = ((lo + r) << 16) | id;
      }
    }
    __syncthreads();                                             // every read of keys2 / idx2 is done
    // ---- ... the ranked indices are assembled in LDS (over keys2) and leave as full 16-byte stores: the
    //      index matrix is 8 B per pair, the largest stream of the whole evaluation
#pragma unroll
    for (int j = 0; j < RL_KPT; ++j)
      if (tid + j * RT < n) keys2[fpk[j] >> 16] = fpk[j] & 0xffffu;
    __syncthreads();
    int64_t* orow = out_idx + row * (int64_t)n;
    if ((reinterpret_cast<uintptr_t>(orow) & 15) == 0) {
      const int pairs = n >> 1;
      for (int i = tid; i < pairs; i += RT) {
        const unsigned a = keys2[2 * i], b2 = keys2[2 * i + 1];
        longlong2 v; v.x = (long long)a; v.y = (long long)b2;
        *reinterpret_cast<longlong2*>(orow + 2 * i) = v;
      }
      if ((n & 1) && tid == 0) orow[n - 1] = (int64_t)keys2[n - 1];
    } else {
      for (int i = tid; i < n; i += RT) orow[i] = (int64_t)keys2[i];
    }
    if (ev.q_pids) {
      // cmc_ap_ranked_wide_kernel<false> on the LDS-resident ranked row (same wave segments, same ballots, same float64
      // summation order -> bit-identical per-query results); the bucket tables are dead by now and hold the scratch
      unsigned long long* masks = reinterpret_cast<unsigned long long*>(off);       // [groups][2] (16-B aligned)
      int* s_i = reinterpret_cast<int*>(cnt);                                       // [3][RW]
      double* s_d = reinterpret_cast<double*>(cnt + 64);                            // [RW]
      const int groups = (n + 63) >> 6, gseg = (groups + RW - 1) / RW;
      const int g0 = min(wave * gseg, groups), g1 = min(g0 + gseg, groups);
      const unsigned long long lt = lanemask_lt();
      int nkeep = 0, nmatch = 0;
      for (int gidx = g0; gidx < g1; ++gidx) {
        const int k = gidx * 64 + lane;
        bool keep = false, mk = false;
        if (k < n) {
          const unsigned gi = keys2[k];
          keep = (bm_keep[gi >> 6] >> (gi & 63u)) & 1ull;
          mk = (bm_match[gi >> 6] >> (gi & 63u)) & 1ull;
        }
        const unsigned long long km = __ballot(keep), mm = __ballot(mk);
        if (lane == 0) { masks[2 * gidx] = km; masks[2 * gidx + 1] = mm; }
        nkeep += __popcll(km); nmatch += __popcll(mm);
      }
      if (lane == 0) { s_i[wave] = nkeep; s_i[RW + wave] = nmatch; }
      __syncthreads();
      int bk = 0, bm = 0, tot_match = 0;
#pragma unroll
      for (int w = 0; w < RW; ++w) { if (w < wave) { bk += s_i[w]; bm += s_i[RW + w]; } tot_match += s_i[RW + w]; }
      double apv = 0.0;
      int firstv = 0x7fffffff;
      for (int gidx = g0; gidx < g1; ++gidx) {
        const unsigned long long km = masks[2 * gidx], mm = masks[2 * gidx + 1];
        if ((mm >> lane) & 1ull) {
          const int p = bk + __popcll(km & lt) + 1;        // 1-based kept position
          const int c = bm + __popcll(mm & lt) + 1;        // matches up to and including this one
          apv += (double)c / (double)p;
          firstv = min(firstv, p - 1);
        }
        bk += __popcll(km); bm += __popcll(mm);
      }
      apv = wave_sum_d(apv);
      firstv = wave_min_i(firstv);
      if (lane == 0) { s_d[wave] = apv; s_i[2 * RW + wave] = firstv; }
      __syncthreads();
      if (tid == 0) {
        double a = 0.0;
        int f = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < RW; ++w) { a += s_d[w]; f = min(f, s_i[2 * RW + w]); }
        const bool valid = tot_match > 0;
        ev.valid[row] = valid ? 1 : 0;
        ev.ap[row] = valid ? a / (double)tot_match : 0.0;
        ev.first[row] = valid ? f : -1;
      }
    }
    __syncthreads();                                             // LDS is reused by the next row
  }
}

// ----------------------------------------------------------------------------------------
// CMC / AP over a ranked row: one 256-thread workgroup per query, each wave owns a contiguous
// segment of rank positions; pass 1 counts kept / matched per segment, pass 2 re-walks with
// the exclusive bases and accumulates AP in float64.
// ----------------------------------------------------------------------------------------
template <bool CAMSETS>
__global__ __launch_bounds__(256) void cmc_ap_ranked_kernel(const int64_t* __restrict__ idx, int64_t m, int64_t n,
                                                            const int64_t* __restrict__ q_pids,
                                                            const int64_t* __restrict__ g_pids,
                                                            const int64_t* __restrict__ q_cams,
                                                            const int64_t* __restrict__ g_cams,
                                                            uint8_t* __restrict__ out_valid,
                                                            double* __restrict__ out_ap,
                                                            int32_t* __restrict__ out_first,
                                                            const uint8_t* __restrict__ only_rows) {
  __shared__ int s_keep[4], s_match[4], s_first[4];
  __shared__ double s_ap[4];
  const int64_t qi = blockIdx.x;
  if (only_rows && !only_rows[qi]) return;                 // rows the fused rank + evaluate kernel has already done
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t qp = q_pids[qi], qc = q_cams[qi];
  const int64_t* row = idx + qi * n;
  const int64_t seg = ((n + 3) / 4 + 63) / 64 * 64;
  const int64_t s0 = min((int64_t)wave * seg, n), s1 = min(s0 + seg, n);
  const unsigned long long lt = lanemask_lt();
  int nkeep = 0, nmatch = 0;
  for (int64_t b = s0; b < s1; b += 64) {
    const int64_t k = b + lane;
    bool keep = false, mk = false;
    if (k < s1) {
      const int64_t gi = row[k];
      const bool match = g_pids[gi] == qp;
      // CAMSETS: g_cams holds a bitmask of cameras per gallery entry; drop if the query's camera is in the set
      keep = !(match && (CAMSETS ? (((unsigned long long)g_cams[gi] >> qc) & 1ull) != 0 : g_cams[gi] == qc));
      mk = match && keep;
    }
    nkeep += __popcll(__ballot(keep));
    nmatch += __popcll(__ballot(mk));
  }
  if (lane == 0) { s_keep[wave] = nkeep; s_match[wave] = nmatch; }
  __syncthreads();
  int bk = 0, bm = 0, tot_match = 0;
  for (int w = 0; w < 4; ++w) { if (w < wave) { bk += s_keep[w]; bm += s_match[w]; } tot_match += s_match[w]; }
  double ap = 0.0;
  int first = 0x7fffffff;
  for (int64_t b = s0; b < s1; b += 64) {
    const int64_t k = b + lane;
    bool keep = false, mk = false;
    if (k < s1) {
      const int64_t gi = row[k];
      const bool match = g_pids[gi] == qp;
      keep = !(match && (CAMSETS ? (((unsigned long long)g_cams[gi] >> qc) & 1ull) != 0 : g_cams[gi] == qc));
      mk = match && keep;
    }
    const unsigned long long km = __ballot(keep), mm = __ballot(mk);
    if (mk) {
      const int p = bk + __popcll(km & lt) + 1;        // 1-based kept position
      const int c = bm + __popcll(mm & lt) + 1;        // matches up to and including this one
      ap += (double)c / (double)p;
      first = min(first, p - 1);
    }
    bk += __popcll(km); bm += __popcll(mm);
  }
  ap = wave_sum_d(ap);
  first = wave_min_i(first);
  if (lane == 0) { s_ap[wave] = ap; s_first[wave] = first; }
  __syncthreads();
  if (tid == 0) {
    const double a = ((s_ap[0] + s_ap[1]) + (s_ap[2] + s_ap[3]));
    const int f = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
    const bool valid = tot_match > 0;
    out_valid[qi] = valid ? 1 : 0;
    out_ap[qi] = valid ? a / (double)tot_match : 0.0;
    out_first[qi] = valid ? f : -1;
  }
}

// Same scan, 16 waves per query and ONE pass over the gallery gathers: the (keep, match) ballots of every
// 64-position group are parked in LDS (16 B per group), so the AP pass touches no global memory.  The kernel is
// pure dependent-load latency (ranked index -> pid / camera gathers): 4x the waves = 4x fewer serial steps.
template <bool CAMSETS>
__global__ __launch_bounds__(1024) void cmc_ap_ranked_wide_kernel(const int64_t* __restrict__ idx, int64_t m, int64_t n,
                                                                  const int64_t* __restrict__ q_pids,
                                                                  const int64_t* __restrict__ g_pids,
                                                                  const int64_t* __restrict__ q_cams,
                                                                  const int64_t* __restrict__ g_cams,
                                                                  uint8_t* __restrict__ out_valid,
                                                                  double* __restrict__ out_ap,
                                                                  int32_t* __restrict__ out_first,
                                                                  const uint8_t* __restrict__ only_rows) {
  constexpr int CW = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned long long cm_masks[];    // [groups][2]
  __shared__ int s_keep[CW], s_match[CW], s_first[CW];
  __shared__ double s_ap[CW];
  const int64_t qi = blockIdx.x;
  if (only_rows && !only_rows[qi]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t qp = q_pids[qi], qc = q_cams[qi];
  const int64_t* row = idx + qi * n;
  const int64_t groups = (n + 63) / 64;
  const int64_t gseg = (groups + CW - 1) / CW;
  const int64_t g0 = min((int64_t)wave * gseg, groups), g1 = min(g0 + gseg, groups);
  const unsigned long long lt = lanemask_lt();
  int nkeep = 0, nmatch = 0;
  for (int64_t gidx = g0; gidx < g1; ++gidx) {
    const int64_t k = gidx * 64 + lane;
    bool keep = false, mk = false;
    if (k < n) {
      const int64_t gi = row[k];
      const bool match = g_pids[gi] == qp;
      keep = !(match && (CAMSETS ? (((unsigned long long)g_cams[gi] >> qc) & 1ull) != 0 : g_cams[gi] == qc));
      mk = match && keep;
    }
    const unsigned long long km = __ballot(keep), mm = __ballot(mk);
    if (lane == 0) { cm_masks[2 * gidx] = km; cm_masks[2 * gidx + 1] = mm; }
    nkeep += __popcll(km); nmatch += __popcll(mm);
  }
  if (lane == 0) { s_keep[wave] = nkeep; s_match[wave] = nmatch; }
  __syncthreads();
  int bk = 0, bm = 0, tot_match = 0;
#pragma unroll
  for (int w = 0; w < CW; ++w) { if (w < wave) { bk += s_keep[w]; bm += s_match[w]; } tot_match += s_match[w]; }
  double ap = 0.0;
  int first = 0x7fffffff;
  for (int64_t gidx = g0; gidx < g1; ++gidx) {
    const unsigned long long km = cm_masks[2 * gidx], mm = cm_masks[2 * gidx + 1];
    if ((mm >> lane) & 1ull) {
      const int p = bk + __popcll(km & lt) + 1;        // 1-based kept position
      const int c = bm + __popcll(mm & lt) + 1;        // matches up to and including this one
      ap += (double)c / (double)p;
      first = min(first, p - 1);
    }
    bk += __popcll(km); bm += __popcll(mm);
  }
  ap = wave_sum_d(ap);
  first = wave_min_i(first);
  if (lane == 0) { s_ap[wave] = ap; s_first[wave] = first; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    int f = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < CW; ++w) { a += s_ap[w]; f = min(f, s_first[w]); }
    const bool valid = tot_match > 0;
    out_valid[qi] = valid ? 1 : 0;
    out_ap[qi] = valid ? a / (double)tot_match : 0.0;
    out_first[qi] = valid ? f : -1;
  }
}


// ----------------------------------------------------------------------------------------
// top-k per row (inference/get_similar.py:114-119 keeps `indices[:, :topk]` of a full argsort): the k smallest
// (distance, index) pairs of a row, in order, without ranking the other n - k entries.  One 1024-thread workgroup
// per row, three streaming reads of the row (L2-resident): key range -> 4096-bucket histogram over the range ->
// the bucket where the running count crosses k -> everything up to that bucket (k + a few entries) is collected
// into LDS as (key << 32 | index) and bitonic-sorted; 64-bit order == (distance, index) order, so ties resolve by
// gallery index like the stable rank kernel.  A row whose candidate set exceeds 4096 (massive ties) is flagged.
// ----------------------------------------------------------------------------------------
namespace {
constexpr int TK_T = 1024, TK_B = 4096, TK_CAP = 4096;
__device__ __forceinline__ unsigned tk_key(float d) {
  const unsigned u = __float_as_uint(d);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float tk_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
}  // namespace

__global__ __launch_bounds__(TK_T) void topk_rows_kernel(const float* __restrict__ dist, int n, int64_t ld, int k,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_dist,
                                                         uint8_t* __restrict__ flags) {
  __shared__ unsigned hist[TK_B];
  __shared__ unsigned long long cand[TK_CAP];
  __shared__ unsigned s_lo[16], s_hi[16], s_wsum[16];
  __shared__ unsigned s_cnt;
  __shared__ int s_bstar, s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = dist + (int64_t)blockIdx.x * ld;
  // A: key range
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int j = tid; j < n; j += TK_T) { const unsigned key = tk_key(row[j]); lo = min(lo, key); hi = max(hi, key); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64)); hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64)); }
  if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
  for (int i = tid; i < TK_B; i += TK_T) hist[i] = 0u;
  if (tid == 0) { s_cnt = 0u; s_bstar = TK_B - 1; s_total = n; }
  __syncthreads();
  lo = s_lo[0]; hi = s_hi[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
  const float scale = (float)TK_B / ((float)(hi - lo) + 1.0f);        // monotone map of [lo, hi] onto the buckets
  auto bucket = [&](unsigned key) { return min((int)((float)(key - lo) * scale), TK_B - 1); };
  // B: histogram, then the bucket where the running count reaches k
  for (int j = tid; j < n; j += TK_T) atomicAdd(&hist[bucket(tk_key(row[j]))], 1u);
  __syncthreads();
  unsigned h4[4], mine = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { h4[q] = hist[tid * 4 + q]; mine += h4[q]; }
  unsigned incl = mine;                                                // inclusive scan across the workgroup
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, o, 64); if (lane >= o) incl += v; }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wave; ++w) base += s_wsum[w];
  unsigned run = base + incl - mine;                                   // entries in the buckets before mine
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (run < (unsigned)k && run + h4[q] >= (unsigned)k) { s_bstar = tid * 4 + q; s_total = (int)(run + h4[q]); }
    run += h4[q];
  }
  __syncthreads();
  const int bstar = s_bstar, total = s_total;
  if (total > TK_CAP) { if (tid == 0) flags[blockIdx.x] = 1; return; }
  // C: collect and sort the candidates
  for (int j = tid; j < n; j += TK_T) {
    const unsigned key = tk_key(row[j]);
    if (bucket(key) <= bstar) cand[atomicAdd(&s_cnt, 1u)] = ((unsigned long long)key << 32) | (unsigned)j;
  }
  int S = 1;
  while (S < total) S <<= 1;
  __syncthreads();
  for (int i = total + tid; i < S; i += TK_T) cand[i] = ~0ull;
  __syncthreads();
  for (int sz = 2; sz <= S; sz <<= 1) {
    for (int st = sz >> 1; st > 0; st >>= 1) {
      for (int i = tid; i < (S >> 1); i += TK_T) {
        const int a = ((i / st) * st * 2) + (i % st), b = a + st;
        const bool up = ((a & sz) == 0);
        const unsigned long long x = cand[a], y = cand[b];
        if ((x > y) == up) { cand[a] = y; cand[b] = x; }
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < k; t += TK_T) {
    const unsigned long long c = cand[t];
    out_idx[(int64_t)blockIdx.x * k + t] = (int64_t)(c & 0xffffffffull);
    if (out_dist) out_dist[(int64_t)blockIdx.x * k + t] = tk_unkey((unsigned)(c >> 32));
  }
  if (tid == 0) flags[blockIdx.x] = 0;
}

// means over valid queries: single workgroup
__global__ __launch_bounds__(256) void eval_reduce_kernel(const uint8_t* __restrict__ valid,
                                                          const double* __restrict__ ap,
                                                          const int32_t* __restrict__ first, int64_t m, int max_rank,
                                                          float* __restrict__ out_cmc, double* __restrict__ out_map,
                                                          double* __restrict__ out_topk,
                                                          int64_t* __restrict__ out_nvalid) {
  __shared__ unsigned s_hist[64];   // first-match rank histogram, bins 0..63 (max_rank <= 64): the CMC curve is cut at
                                    // max_rank, the top-k hits (k up to 50) use the UNtruncated match row like
                                    // top_k_retrieval(orig_cmc) of utils/eval_reid.py:18-22,84
  __shared__ unsigned s_nvalid;
  __shared__ double s_sum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 64) s_hist[tid] = 0;
  if (tid == 0) s_nvalid = 0;
  __syncthreads();
  double sum = 0.0;
  unsigned nv = 0;
  for (int64_t i = tid; i < m; i += 256) {
    if (valid[i]) {
      ++nv;
      sum += ap[i];
      const int f = first[i];
      if (f < 64) atomicAdd(&s_hist[f], 1u);
    }
  }
  sum = wave_sum_d(sum);
  nv = (unsigned)wave_sum_i((int)nv);
  if (lane == 0) { s_sum[wave] = sum; atomicAdd(&s_nvalid, nv); }
  __syncthreads();
  if (tid == 0) {
    const unsigned nvalid = s_nvalid;
    *out_nvalid = (int64_t)nvalid;
    *out_map = ((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) / (double)nvalid;
    unsigned run = 0;
    const int ks[5] = {1, 5, 10, 20, 50};
    int kq = 0;
    for (int r = 0; r < 50 || r < max_rank; ++r) {
      run += s_hist[r];
      if (r < max_rank) out_cmc[r] = (float)run / (float)nvalid;   // float32 like the reference's all_cmc
      while (kq < 5 && ks[kq] == r + 1) { out_topk[kq] = (double)run / (double)nvalid; ++kq; }
    }
  }
}

extern "C" {

static size_t rank_radix_ws_bytes(int64_t m, int64_t n) {
  const int64_t slots = m < MAX_SLOTS ? m : MAX_SLOTS;
  return (size_t)slots * (size_t)n * 4 * sizeof(unsigned);
}

size_t creid_rank_rows_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  return rank_radix_ws_bytes(m, n) + (((size_t)m + 255) & ~(size_t)255);     // + per-row fallback flags
}

static int rank_rows_impl(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx, void* ws, size_t ws_bytes,
                          void* stream, RankEval ev) {
  CREID_CHECK_ARG(m >= 0 && n >= 0 && ld >= n);
  if (m == 0 || n == 0) return 0;
  CREID_CHECK_ARG(dist && out_idx && ws);
  if (n > 0xfffffff0LL) return CREID_E_SHAPE;
  if (ws_bytes < creid_rank_rows_workspace_bytes(m, n)) return CREID_E_WS;
  const int64_t slots = m < MAX_SLOTS ? m : MAX_SLOTS;
  hipStream_t s = as_stream(stream);
  uint8_t* flags = reinterpret_cast<uint8_t*>(ws) + rank_radix_ws_bytes(m, n);
  // rows that fit in LDS next to the bucket tables take the one-pass bucket kernel (CREID_RANK_LDS=0 disables)
  static const int use_lds = [] { const char* e = getenv("CREID_RANK_LDS"); return e ? atoi(e) : 1; }();
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  const size_t lds_bytes = n4 * 4 + (size_t)(2 * RL_NB + 4 + 3 * RW) * 4 + n4 * 2 +
                           (ev.q_pids ? (size_t)2 * ((n + 63) / 64) * 8 : 0);      // + the evaluation's two bitmaps
  const uint8_t* only_flagged = nullptr;
  if (use_lds && n >= 512 && n <= (int64_t)RL_KPT * RT && lds_bytes <= 156 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(rank_rows_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
        return (int)hipGetLastError();
      attr_set = true;
    }
    const int64_t wgs = m < 1024 ? m : 1024;
    hipLaunchKernelGGL(rank_rows_lds_kernel, dim3((unsigned)wgs), dim3(RT), lds_bytes, s, dist, m, (int)n, ld, out_idx,
                       flags, ev);
    only_flagged = flags;
  }
  hipLaunchKernelGGL(rank_rows_kernel, dim3((unsigned)slots), dim3(RT), 0, s, dist, m, n, ld, out_idx, (unsigned*)ws,
                     only_flagged);
  if (ev.q_pids) {
    // rows the one-pass kernel did not take (pathological ties, or a gallery that does not fit its LDS layout): the separate
    // scan over the index matrix, restricted to those rows
    if (m > 0x7fffffffLL || n > 0x7fffff00LL) return CREID_E_SHAPE;
    const size_t mask_bytes = (size_t)((n + 63) / 64) * 16;
    if (mask_bytes <= 48 * 1024)
      hipLaunchKernelGGL(cmc_ap_ranked_wide_kernel<false>, dim3((unsigned)m), dim3(1024), mask_bytes, s, out_idx, m, n, ev.q_pids,
                         ev.g_pids, ev.q_cams, ev.g_cams, ev.valid, ev.ap, ev.first, only_flagged);
    else
      hipLaunchKernelGGL(cmc_ap_ranked_kernel<false>, dim3((unsigned)m), dim3(256), 0, s, out_idx, m, n, ev.q_pids, ev.g_pids,
                         ev.q_cams, ev.g_cams, ev.valid, ev.ap, ev.first, only_flagged);
  }
  CREID_LAUNCH_RET();
}

int creid_rank_rows(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx, void* ws,
                    size_t ws_bytes, void* stream) {
  return rank_rows_impl(dist, m, n, ld, out_idx, ws, ws_bytes, stream, RankEval{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
}

/* creid_rank_rows + creid_cmc_ap_ranked in one pass where the row fits the one-pass rank kernel: the per-query (valid, AP,
 * first match) are computed while the ranked row is still in LDS; the int64 index matrix is written (the caller wants it) but
 * never read back.  Identical results to the two separate calls. */
int creid_rank_rows_eval(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx, void* ws, size_t ws_bytes,
                         const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_camids, const int64_t* g_camids,
                         uint8_t* out_valid, double* out_ap, int32_t* out_first, void* stream) {
  CREID_CHECK_ARG(q_pids && g_pids && q_camids && g_camids && out_valid && out_ap && out_first);
  return rank_rows_impl(dist, m, n, ld, out_idx, ws, ws_bytes, stream,
                        RankEval{q_pids, g_pids, q_camids, g_camids, out_valid, out_ap, out_first});
}

int creid_cmc_ap_ranked(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids, const int64_t* g_pids,
                        const int64_t* q_camids, const int64_t* g_camids, uint8_t* out_valid, double* out_ap,
                        int32_t* out_first, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n >= 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(idx && q_pids && g_pids && q_camids && g_camids && out_valid && out_ap && out_first);
  if (m > 0x7fffffffLL || n > 0x7fffff00LL) return CREID_E_SHAPE;
  const size_t mask_bytes = (size_t)((n + 63) / 64) * 16;
  if (mask_bytes <= 48 * 1024)
    hipLaunchKernelGGL(cmc_ap_ranked_wide_kernel<false>, dim3((unsigned)m), dim3(1024), mask_bytes, as_stream(stream), idx,
                       m, n, q_pids, g_pids, q_camids, g_camids, out_valid, out_ap, out_first, (const uint8_t*)nullptr);
  else
    hipLaunchKernelGGL(cmc_ap_ranked_kernel<false>, dim3((unsigned)m), dim3(256), 0, as_stream(stream), idx, m, n, q_pids,
                       g_pids, q_camids, g_camids, out_valid, out_ap, out_first, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_cmc_ap_ranked_camsets(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids, const int64_t* g_pids,
                                const int64_t* q_camids, const int64_t* g_cam_masks, uint8_t* out_valid, double* out_ap,
                                int32_t* out_first, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n >= 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(idx && q_pids && g_pids && q_camids && g_cam_masks && out_valid && out_ap && out_first);
  if (m > 0x7fffffffLL || n > 0x7fffff00LL) return CREID_E_SHAPE;
  const size_t mask_bytes = (size_t)((n + 63) / 64) * 16;
  if (mask_bytes <= 48 * 1024)
    hipLaunchKernelGGL(cmc_ap_ranked_wide_kernel<true>, dim3((unsigned)m), dim3(1024), mask_bytes, as_stream(stream), idx,
                       m, n, q_pids, g_pids, q_camids, g_cam_masks, out_valid, out_ap, out_first, (const uint8_t*)nullptr);
  else
    hipLaunchKernelGGL(cmc_ap_ranked_kernel<true>, dim3((unsigned)m), dim3(256), 0, as_stream(stream), idx, m, n, q_pids,
                       g_pids, q_camids, g_cam_masks, out_valid, out_ap, out_first, (const uint8_t*)nullptr);
  CREID_LAUNCH_RET();
}

int creid_eval_reduce(const uint8_t* valid, const double* ap, const int32_t* first, int64_t m, int32_t max_rank,
                      float* out_cmc, double* out_map, double* out_topk, int64_t* out_nvalid, void* stream) {
  CREID_CHECK_ARG(valid && ap && first && out_cmc && out_map && out_topk && out_nvalid && m >= 0);
  if (max_rank < 1 || max_rank > 64) return CREID_E_SHAPE;
  hipLaunchKernelGGL(eval_reduce_kernel, dim3(1), dim3(256), 0, as_stream(stream), valid, ap, first, m,
                     (int)max_rank, out_cmc, out_map, out_topk, out_nvalid);
  CREID_LAUNCH_RET();
}

int creid_topk_rows(const float* dist, int64_t m, int64_t n, int64_t ld, int32_t k, int64_t* out_idx, float* out_dist,
                    uint8_t* flags, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n > 0 && ld >= n && k >= 1);
  if (m == 0) return 0;
  CREID_CHECK_ARG(dist && out_idx && flags);
  if (k > n || k > 1024 || n > 0x7ffffff0LL || m > 0x7fffffffLL) return CREID_E_SHAPE;
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)m), dim3(TK_T), 0, as_stream(stream), dist, (int)n, ld, (int)k, out_idx,
                     out_dist, flags);
  CREID_LAUNCH_RET();
}

}  // extern "C"
