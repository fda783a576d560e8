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
                                                       unsigned* __restrict__ ws) {
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
                                                            int32_t* __restrict__ out_first) {
  __shared__ int s_keep[4], s_match[4], s_first[4];
  __shared__ double s_ap[4];
  const int64_t qi = blockIdx.x;
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

// means over valid queries: single workgroup
__global__ __launch_bounds__(256) void eval_reduce_kernel(const uint8_t* __restrict__ valid,
                                                          const double* __restrict__ ap,
                                                          const int32_t* __restrict__ first, int64_t m, int max_rank,
                                                          float* __restrict__ out_cmc, double* __restrict__ out_map,
                                                          double* __restrict__ out_topk,
                                                          int64_t* __restrict__ out_nvalid) {
  __shared__ unsigned s_hist[64];   // first-match rank histogram, bins 0..max_rank-1 (max_rank <= 64)
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
      if (f < max_rank) atomicAdd(&s_hist[f], 1u);
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
    for (int r = 0; r < max_rank; ++r) {
      run += s_hist[r];
      out_cmc[r] = (float)run / (float)nvalid;          // float32 like the reference's all_cmc
      while (kq < 5 && ks[kq] == r + 1) { out_topk[kq] = (double)run / (double)nvalid; ++kq; }
    }
    while (kq < 5) { out_topk[kq] = (double)run / (double)nvalid; ++kq; }  // k > max_rank: all kept
  }
}

extern "C" {

size_t creid_rank_rows_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  const int64_t slots = m < MAX_SLOTS ? m : MAX_SLOTS;
  return (size_t)slots * (size_t)n * 4 * sizeof(unsigned);
}

int creid_rank_rows(const float* dist, int64_t m, int64_t n, int64_t ld, int64_t* out_idx, void* ws,
                    size_t ws_bytes, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n >= 0 && ld >= n);
  if (m == 0 || n == 0) return 0;
  CREID_CHECK_ARG(dist && out_idx && ws);
  if (n > 0xfffffff0LL) return CREID_E_SHAPE;
  if (ws_bytes < creid_rank_rows_workspace_bytes(m, n)) return CREID_E_WS;
  const int64_t slots = m < MAX_SLOTS ? m : MAX_SLOTS;
  hipLaunchKernelGGL(rank_rows_kernel, dim3((unsigned)slots), dim3(RT), 0, as_stream(stream), dist, m, n, ld,
                     out_idx, (unsigned*)ws);
  CREID_LAUNCH_RET();
}

int creid_cmc_ap_ranked(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids, const int64_t* g_pids,
                        const int64_t* q_camids, const int64_t* g_camids, uint8_t* out_valid, double* out_ap,
                        int32_t* out_first, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n >= 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(idx && q_pids && g_pids && q_camids && g_camids && out_valid && out_ap && out_first);
  if (m > 0x7fffffffLL || n > 0x7fffff00LL) return CREID_E_SHAPE;
  hipLaunchKernelGGL(cmc_ap_ranked_kernel<false>, dim3((unsigned)m), dim3(256), 0, as_stream(stream), idx, m, n, q_pids,
                     g_pids, q_camids, g_camids, out_valid, out_ap, out_first);
  CREID_LAUNCH_RET();
}

int creid_cmc_ap_ranked_camsets(const int64_t* idx, int64_t m, int64_t n, const int64_t* q_pids, const int64_t* g_pids,
                                const int64_t* q_camids, const int64_t* g_cam_masks, uint8_t* out_valid, double* out_ap,
                                int32_t* out_first, void* stream) {
  CREID_CHECK_ARG(m >= 0 && n >= 0);
  if (m == 0) return 0;
  CREID_CHECK_ARG(idx && q_pids && g_pids && q_camids && g_cam_masks && out_valid && out_ap && out_first);
  if (m > 0x7fffffffLL || n > 0x7fffff00LL) return CREID_E_SHAPE;
  hipLaunchKernelGGL(cmc_ap_ranked_kernel<true>, dim3((unsigned)m), dim3(256), 0, as_stream(stream), idx, m, n, q_pids,
                     g_pids, q_camids, g_cam_masks, out_valid, out_ap, out_first);
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

}  // extern "C"
