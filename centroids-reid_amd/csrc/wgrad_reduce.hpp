// Split reduction of the weight-gradient partial tiles, as a DEVICE routine that any kernel can run in a few
// extra workgroups ("piggyback"): conv_wgrad.hip's split kernels leave `splits` fp32 partial copies
// ws[split][co][k] (k = (tap, c) tap-major); this sums them in a FIXED order (deterministic) and writes / adds
// the OIHW fp32 gradient.  Round 1 ran it as 53 stand-alone launches of 5-25 us (6.6 % of the training step,
// mostly launch latency and idle CUs); the data-gradient launch that follows every weight gradient now carries
// it in its first workgroups instead, where it overlaps with MFMA-bound tiles.
#pragma once
#include "common.hpp"

struct WRedJob {
  const float* ws;       // [splits][NCO][K] partials (null: no job)
  float* dw;             // OIHW fp32 gradient
  int splits, NCO, K;
  int cin, taps;         // K = taps * cin; taps = kh*kw (1 or 9); output index inside a channel: c * taps + tap
  int accumulate;
  int nblocks;           // workgroups that carry the job
  int rows_per_block;    // output channels per workgroup
  int split_lanes;       // 1x1 flat path: threads that share one output (power of two)
};

// One workgroup of NT threads reduces output channels [block*rows_per_block, ...).  Per channel: phase 1 sums the
// `splits` partial rows (K floats, contiguous) with (16-byte column groups) x (split lanes) threads and parks the
// row in LDS; phase 2 writes it out in OIHW order ([c][tap] instead of [tap][c] for 3x3) as one contiguous run.
// lds: >= NT*16 + K*4 bytes.
template <int NT>
__device__ __forceinline__ void wgrad_reduce_block(const WRedJob& j, int block, float* lds) {
  const int tid = threadIdx.x;
  const int nf4 = j.K >> 2;
  if (j.taps == 1) {
    // 1x1: the partial layout [co][c] IS the OIHW layout -- a flat stream of 16-byte outputs.  A workgroup covers
    // G = NT / SL outputs x SL split lanes (SL = j.split_lanes, chosen on the host so that a thread sums <= 8 splits):
    // every load of a pass is issued back to back (one memory round trip per workgroup, where the per-channel form
    // took one per channel), the lanes meet in LDS and lane 0 adds them in a fixed order.
    const int64_t total4 = (int64_t)j.NCO * nf4;
    const float4* wsp = reinterpret_cast<const float4*>(j.ws);
    float4* dw4 = reinterpret_cast<float4*>(j.dw);
    float4* red = reinterpret_cast<float4*>(lds);            // [SL][G]
    const int SL = j.split_lanes < NT ? j.split_lanes : NT, G = NT / SL;
    const int gi = tid % G, sl = tid / G;
    for (int64_t f0 = (int64_t)block * G; f0 < total4; f0 += (int64_t)j.nblocks * G) {   // uniform trip count
      const int64_t f = f0 + gi;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < total4) {
        int sp = sl;
        for (; sp + 7 * SL < j.splits; sp += 8 * SL) {
          float4 v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = wsp[(int64_t)(sp + q * SL) * total4 + f];
#pragma unroll
          for (int q = 0; q < 8; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
        }
        for (; sp < j.splits; sp += SL) {
          const float4 v = wsp[(int64_t)sp * total4 + f];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      if (SL > 1) {
        red[sl * G + gi] = acc;
        __syncthreads();
        if (sl == 0)
          for (int q = 1; q < SL; ++q) { const float4 v = red[q * G + gi]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
      if (sl == 0 && f < total4) {
        if (j.accumulate) { const float4 o = dw4[f]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        dw4[f] = acc;
      }
      if (SL > 1) __syncthreads();
    }
    return;
  }
  float4* red = reinterpret_cast<float4*>(lds);              // [SL][G]  (NT float4)
  float* row = lds + 4 * NT;                                 // [K]
  int SL = 1;                                                // split lanes: largest power of two <= splits that fits
  while (SL * 2 <= j.splits && SL * 2 * (nf4 < NT ? nf4 : NT) <= NT) SL *= 2;
  const int G = NT / SL;                                     // 16-byte column groups per pass
  const int gi = tid % G, sl = tid / G;
  const int64_t total4 = (int64_t)j.NCO * nf4;
  const float4* wsp = reinterpret_cast<const float4*>(j.ws);
  for (int r = 0; r < j.rows_per_block; ++r) {
    const int co = block * j.rows_per_block + r;
    if (co >= j.NCO) break;                                  // uniform
    for (int f0 = 0; f0 < nf4; f0 += G) {
      const int f = f0 + gi;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < nf4) {
        const float4* p = wsp + (int64_t)co * nf4 + f;
        int sp = sl;
        for (; sp + 7 * SL < j.splits; sp += 8 * SL) {       // 8 independent loads in flight
          const float4 a = p[(int64_t)sp * total4], b = p[(int64_t)(sp + SL) * total4];
          const float4 c = p[(int64_t)(sp + 2 * SL) * total4], d = p[(int64_t)(sp + 3 * SL) * total4];
          const float4 e = p[(int64_t)(sp + 4 * SL) * total4], f4 = p[(int64_t)(sp + 5 * SL) * total4];
          const float4 g = p[(int64_t)(sp + 6 * SL) * total4], h = p[(int64_t)(sp + 7 * SL) * total4];
          acc.x += ((a.x + b.x) + (c.x + d.x)) + ((e.x + f4.x) + (g.x + h.x));
          acc.y += ((a.y + b.y) + (c.y + d.y)) + ((e.y + f4.y) + (g.y + h.y));
          acc.z += ((a.z + b.z) + (c.z + d.z)) + ((e.z + f4.z) + (g.z + h.z));
          acc.w += ((a.w + b.w) + (c.w + d.w)) + ((e.w + f4.w) + (g.w + h.w));
        }
        for (; sp + 3 * SL < j.splits; sp += 4 * SL) {       // 4 independent loads in flight
          const float4 a = p[(int64_t)sp * total4], b = p[(int64_t)(sp + SL) * total4];
          const float4 c = p[(int64_t)(sp + 2 * SL) * total4], d = p[(int64_t)(sp + 3 * SL) * total4];
          acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
          acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; sp < j.splits; sp += SL) {
          const float4 a = p[(int64_t)sp * total4];
          acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
      }
      if (SL > 1) {
        red[sl * G + gi] = acc;
        __syncthreads();
        if (sl == 0 && f < nf4) {
          for (int q = 1; q < SL; ++q) { const float4 v = red[q * G + gi]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
          reinterpret_cast<float4*>(row)[f] = acc;
        }
        __syncthreads();
      } else if (f < nf4) {
        reinterpret_cast<float4*>(row)[f] = acc;
      }
    }
    __syncthreads();
    float* dst = j.dw + (int64_t)co * j.K;
    for (int o = tid; o < j.K; o += NT) {
      const int c = o / j.taps, tap = o - c * j.taps;        // OIHW inside the channel: [c][tap]
      const float v = row[tap * j.cin + c];
      dst[o] = j.accumulate ? dst[o] + v : v;
    }
    __syncthreads();
  }
}

// Host side: fill the job for a plain 1x1 / 3x3 convolution (kw_taps == kw, cpitch == cin); returns false when the
// layout is not the plain one (stem) and the stand-alone kernels must be used.
static inline bool wred_make_job(WRedJob& j, const float* ws, float* dw, int splits, int NCO, int K, int cin, int kh, int kw,
                                 int accumulate) {
  if (kh * kw * cin != K || (kh * kw != 1 && kh * kw != 9) || (K & 3) || K > 8192) return false;
  j.ws = ws; j.dw = dw; j.splits = splits; j.NCO = NCO; j.K = K; j.cin = cin; j.taps = kh * kw; j.accumulate = accumulate;
  // one output channel per workgroup: many short workgroups (one or two dependent load batches each, 2-4 us) that
  // run beside the carrying launch's tiles; several channels per workgroup made them the launch's long pole
  const int rpb = 1;
  j.rows_per_block = rpb;
  j.nblocks = (NCO + rpb - 1) / rpb;
  j.split_lanes = 1;
  if (j.taps == 1) {                                         // flat path: 512 / split_lanes outputs per workgroup
    while (j.split_lanes < 64 && splits > 8 * j.split_lanes) j.split_lanes *= 2;
    const int64_t total4 = (int64_t)NCO * (K >> 2);
    const int64_t per_wg = 512 / j.split_lanes;
    int64_t nb = (total4 + per_wg - 1) / per_wg;
    j.nblocks = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
  }
  return true;
}
