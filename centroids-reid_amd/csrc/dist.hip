// Stage D: row L2-normalise, row square-norms, query x gallery squared-L2 matrix.
//
// Replaces utils/reid_metric.py:25-33 (get_euclidean) and :113-115 (F.normalize) of the
// reference.  The contraction Q.G^T is MFMA-bound (AI ~ 560 FLOP/B at D=2048), so the
// fp32 kernel is a 128x128x16 LDS-tiled GEMM on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain,
// 157 TF peak) with the "+|q|^2 + |g|^2 - 2*" epilogue fused; the normalise / square-norm
// kernels are HBM-streaming (one wave per row, 16-B loads).
#include "common.hpp"

// ----------------------------------------------------------------------------------------
// l2norm: one wave per row.  y = x / max(sqrt(sum x^2), eps); optional sqnorm(y) output.
// ----------------------------------------------------------------------------------------
template <int OUT_DT>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, void* __restrict__ yv,
                                                          float* __restrict__ sqn, int64_t rows, int64_t D,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + row * D);
  const int nv = (int)(D >> 2);
  // rows up to 4096 elements stay in registers between the norm and the scaling (ONE read of the input instead of
  // two: the pass is HBM-bound); longer rows are streamed twice
  constexpr int RV = 16;
  const bool resident = nv <= 64 * RV;
  float4 keep[RV];
  float s = 0.f;
  if (resident) {
#pragma unroll
    for (int t = 0; t < RV; ++t) {
      const int i = lane + 64 * t;
      keep[t] = i < nv ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < RV; ++t) {          // same accumulation order as the streaming loop below
      const float4 v = keep[t];
      if (lane + 64 * t < nv) { s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s); }
    }
  } else {
    for (int i = lane; i < nv; i += 64) {
      float4 v = xr[i];
      s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
    }
  }
  s = wave_sum(s);
  const float denom = fmaxf(sqrtf(s), eps);
  float s2 = 0.f;
  auto emit = [&](int i, float4 v) {
    v.x /= denom; v.y /= denom; v.z /= denom; v.w /= denom;
    if constexpr (OUT_DT == CREID_F32) {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + row * D)[i] = v;
    } else if constexpr (OUT_DT == CREID_BF16) {
      ushort4 o;
      o.x = f32_to_bf16_bits(v.x); o.y = f32_to_bf16_bits(v.y);
      o.z = f32_to_bf16_bits(v.z); o.w = f32_to_bf16_bits(v.w);
      reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(yv) + row * D)[i] = o;
      v.x = bf16_bits_to_f32(o.x); v.y = bf16_bits_to_f32(o.y);
      v.z = bf16_bits_to_f32(o.z); v.w = bf16_bits_to_f32(o.w);
    } else {
      __half h0 = __float2half(v.x), h1 = __float2half(v.y), h2 = __float2half(v.z), h3 = __float2half(v.w);
      __half* yo = reinterpret_cast<__half*>(yv) + row * D + 4 * (int64_t)i;
      yo[0] = h0; yo[1] = h1; yo[2] = h2; yo[3] = h3;
      v.x = __half2float(h0); v.y = __half2float(h1); v.z = __half2float(h2); v.w = __half2float(h3);
    }
    s2 = fmaf(v.x, v.x, s2); s2 = fmaf(v.y, v.y, s2); s2 = fmaf(v.z, v.z, s2); s2 = fmaf(v.w, v.w, s2);
  };
  if (resident) {
#pragma unroll
    for (int t = 0; t < RV; ++t)
      if (lane + 64 * t < nv) emit(lane + 64 * t, keep[t]);
  } else {
    for (int i = lane; i < nv; i += 64) emit(i, xr[i]);
  }
  if (sqn) {
    s2 = wave_sum(s2);
    if (lane == 0) sqn[row] = s2;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const void* __restrict__ xv, float* __restrict__ out,
                                                         int64_t rows, int64_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  if constexpr (DT == CREID_F32) {
    const float* xr = reinterpret_cast<const float*>(xv) + row * D;
    for (int64_t i = lane; i < D; i += 64) { float v = xr[i]; s = fmaf(v, v, s); }
  } else if constexpr (DT == CREID_BF16) {
    const unsigned short* xr = reinterpret_cast<const unsigned short*>(xv) + row * D;
    for (int64_t i = lane; i < D; i += 64) { float v = bf16_bits_to_f32(xr[i]); s = fmaf(v, v, s); }
  } else {
    const __half* xr = reinterpret_cast<const __half*>(xv) + row * D;
    for (int64_t i = lane; i < D; i += 64) { float v = __half2float(xr[i]); s = fmaf(v, v, s); }
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// ----------------------------------------------------------------------------------------
// fp32 distance GEMM.  128x128 tile / 256 threads (4 waves as 2x2, 64x64 per wave = 2x2
// MFMA 32x32 tiles), BK = 16, LDS double-buffered.  Operand image of a k-tile (the same as stream_eval.hip's):
// [kh = k & 1][half = k >> 3][row][s4 = (k >> 1) & 3] -- the per-lane MFMA operand is A[i = lane&31][k = 2 step + (lane>>5)], so
// the values of four consecutive steps are one 16-byte unit: 8 ds_read_b128 per k-tile and lane instead of 32 ds_read_b32, and
// a staged global float4 is two ds_write_b64.  The k order of the accumulation is unchanged (bit-identical results).
// ----------------------------------------------------------------------------------------
namespace {
// plane pitch = 4 * rows + 16 dwords: the two halves a 16-lane ds_write_b64 group touches fall on disjoint banks
constexpr int DBM = 128, DBN = 128, DBK = 16, DPL = DBM * 4 + 16;
}

__global__ __launch_bounds__(256) void sqdist_f32_kernel(const float* __restrict__ q, const float* __restrict__ g,
                                                         const float* __restrict__ qq, const float* __restrict__ gg,
                                                         int m, int n, int D, float* __restrict__ out,
                                                         int64_t ldo, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[2][2][2][DPL];
  __shared__ __attribute__((aligned(16))) float Bs[2][2][2][DPL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give
  // each XCD a contiguous run of tiles that share the same query panel (L2 reuse of Q rows).
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    bid = base + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * DBM, col0 = tile_n * DBN;

  // staging map: each thread moves 2 float4 of A and 2 of B per k-tile
  const int lrow = tid >> 2, lkc = tid & 3;
  const float* ap[2];
  const float* bp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = min(row0 + lrow + 64 * i, m - 1);
    int rb = min(col0 + lrow + 64 * i, n - 1);
    ap[i] = q + (int64_t)ra * D + 4 * lkc;
    bp[i] = g + (int64_t)rb * D + 4 * lkc;
  }
  float4 ra[2], rb[2];
  unsigned rmask = 0u;                             // all ones while the staged k-tile lies inside D
  auto gload = [&](int k0) {                       // branch-free (a k-tile beyond D reads k-tile 0; lstore turns it into zeros)
    const bool in = k0 + 4 * lkc < D;
    const int ko = in ? k0 : 0;
    rmask = in ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(ap[i] + ko);
      rb[i] = *reinterpret_cast<const float4*>(bp[i] + ko);
    }
  };
  const int sh = lkc >> 1, so = lrow * 4 + 2 * (lkc & 1);          // global k = 4 lkc + {0..3}: steps 2 lkc, 2 lkc + 1
  auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & rmask); };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float2*>(&As[buf][0][sh][so + 256 * i]) = make_float2(mk(ra[i].x), mk(ra[i].z));
      *reinterpret_cast<float2*>(&As[buf][1][sh][so + 256 * i]) = make_float2(mk(ra[i].y), mk(ra[i].w));
      *reinterpret_cast<float2*>(&Bs[buf][0][sh][so + 256 * i]) = make_float2(mk(rb[i].x), mk(rb[i].z));
      *reinterpret_cast<float2*>(&Bs[buf][1][sh][so + 256 * i]) = make_float2(mk(rb[i].y), mk(rb[i].w));
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, kh = lane >> 5;
  const int fa = (wm * 64 + l31) * 4, fb = (wn * 64 + l31) * 4;
  // The k-loop is the two-phase pipeline of stream_eval.hip's sqdist_count_f32_kernel: 16 MFMAs per phase (64 cycles each) with
  // everything else issued in their shadow -- phase 1: second-half fragment reads + LDS writes of the next k-tile, barrier,
  // phase 2: global loads of the k-tile after that + first-half fragment reads of the next one.
  float4 a0[2], b0[2], a1[2], b1[2];
  auto frag = [&](int buf, int half, float4 (&a)[2], float4 (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const float4*>(&As[buf][kh][half][fa + 128 * i]);
      b[i] = *reinterpret_cast<const float4*>(&Bs[buf][kh][half][fb + 128 * i]);
    }
  };
  auto mma16 = [&](const float4 (&a)[2], const float4 (&b)[2]) {   // steps in k order: 4 half + s4 multiplies k = 2 step + kh
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float ac[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
        const float bc[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
        av[i] = ac[s4]; bv[i] = bc[s4];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
    }
  };
  const int nk = (D + DBK - 1) / DBK;
  gload(0);
  lstore(0);
  __syncthreads();
  gload(DBK);
  frag(0, 0, a0, b0);
  for (int t = 0; t + 1 < nk; ++t) {
    const int buf = t & 1;
    __builtin_amdgcn_sched_barrier(0);
    frag(buf, 1, a1, b1);
    lstore(buf ^ 1);
    mma16(a0, b0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    gload((t + 2) * DBK);
    frag(buf ^ 1, 0, a0, b0);
    mma16(a1, b1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x020, 1, 1); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  frag((nk - 1) & 1, 1, a1, b1);                   // last k-tile: nothing left to stage
  mma16(a0, b0);
  mma16(a1, b1);

  // epilogue: d = (qq + gg) - 2*dot  (-2*dot is exact, so fma == mul+add of the reference)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = col0 + wn * 64 + j * 32 + l31;
    const float gv = (c < n) ? gg[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (rr < m && c < n) out[(int64_t)rr * ldo + c] = fmaf(-2.0f, acc[i][j][r], qq[rr] + gv);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// 16-bit (bf16 / f16) distance GEMM: 128x128x64 tile on v_mfma_f32_32x32x16_{bf16,f16}.
// LDS is row-major [row][64] (128-B rows), 16-B chunks XOR-swizzled with (row>>1)&7 so that
// every ds_read_b128 lane group touches 16 distinct 16-B slots.
// ----------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void sqdist_h16_kernel(const unsigned short* __restrict__ q,
                                                         const unsigned short* __restrict__ g,
                                                         const float* __restrict__ qq, const float* __restrict__ gg,
                                                         int m, int n, int D, float* __restrict__ out,
                                                         int64_t ldo, int tiles_n) {
  constexpr int BK = 64;
  __shared__ __attribute__((aligned(16))) unsigned short As[2][128 * BK];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][128 * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    bid = base + (bid >> 3);
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int row0 = tile_m * 128, col0 = tile_n * 128;

  // staging: tile = 128 rows x 8 chunks(16 B) = 1024 chunks -> 4 per thread per operand
  const int lrow = tid >> 3, lch = tid & 7;
  const unsigned short* ap[4];
  const unsigned short* bp[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = lrow + 32 * i;
    ap[i] = q + (int64_t)min(row0 + r, m - 1) * D + 8 * lch;
    bp[i] = g + (int64_t)min(col0 + r, n - 1) * D + 8 * lch;
    soff[i] = r * BK + ((lch ^ ((r >> 1) & 7)) << 3);
  }
  uint4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (k0 + 8 * lch < D) {
        ra[i] = *reinterpret_cast<const uint4*>(ap[i] + k0);
        rb[i] = *reinterpret_cast<const uint4*>(bp[i] + k0);
      } else {
        ra[i] = make_uint4(0, 0, 0, 0);
        rb[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(&As[buf][soff[i]]) = ra[i];
      *reinterpret_cast<uint4*>(&Bs[buf][soff[i]]) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (D + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int l31 = lane & 31, kh = lane >> 5;
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) gload((t + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {  // 4 x K=16
      const int ch = 2 * kk + kh;
      s16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        a[i] = *reinterpret_cast<const s16x8*>(&As[buf][r * BK + ((ch ^ ((r >> 1) & 7)) << 3)]);
        const int c = wn * 64 + i * 32 + l31;
        b[i] = *reinterpret_cast<const s16x8*>(&Bs[buf][c * BK + ((ch ^ ((c >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (DT == CREID_BF16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a[i]),
                                                               __builtin_bit_cast(h16x8, b[j]), acc[i][j], 0, 0, 0);
        }
    }
    if (t + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = col0 + wn * 64 + j * 32 + l31;
    const float gv = (c < n) ? gg[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (rr < m && c < n) out[(int64_t)rr * ldo + c] = fmaf(-2.0f, acc[i][j][r], qq[rr] + gv);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
extern "C" {

int creid_abi_version(void) { return CREID_ABI_VERSION; }

int creid_l2norm_rows(const float* x, void* y, float* sqnorm, int64_t rows, int64_t D, int out_dtype, float eps,
                      void* stream) {
  CREID_CHECK_ARG(x && y && rows >= 0 && D > 0);
  if (D % 4 != 0) return CREID_E_SHAPE;
  if (rows == 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (out_dtype) {
    case CREID_F32: hipLaunchKernelGGL(l2norm_rows_kernel<CREID_F32>, grid, block, 0, s, x, y, sqnorm, rows, D, eps); break;
    case CREID_BF16: hipLaunchKernelGGL(l2norm_rows_kernel<CREID_BF16>, grid, block, 0, s, x, y, sqnorm, rows, D, eps); break;
    case CREID_F16: hipLaunchKernelGGL(l2norm_rows_kernel<CREID_F16>, grid, block, 0, s, x, y, sqnorm, rows, D, eps); break;
    default: return CREID_E_DTYPE;
  }
  CREID_LAUNCH_RET();
}

int creid_row_sqnorm(const void* x, float* out, int64_t rows, int64_t D, int dtype, void* stream) {
  CREID_CHECK_ARG(x && out && rows >= 0 && D > 0);
  if (rows == 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case CREID_F32: hipLaunchKernelGGL(row_sqnorm_kernel<CREID_F32>, grid, block, 0, s, x, out, rows, D); break;
    case CREID_BF16: hipLaunchKernelGGL(row_sqnorm_kernel<CREID_BF16>, grid, block, 0, s, x, out, rows, D); break;
    case CREID_F16: hipLaunchKernelGGL(row_sqnorm_kernel<CREID_F16>, grid, block, 0, s, x, out, rows, D); break;
    default: return CREID_E_DTYPE;
  }
  CREID_LAUNCH_RET();
}

int creid_sqdist_matrix(const void* q, const void* g, const float* qq, const float* gg, int64_t m, int64_t n,
                        int64_t D, int dtype, float* out, int64_t ldo, void* stream) {
  CREID_CHECK_ARG(q && g && qq && gg && out && m >= 0 && n >= 0 && D > 0 && ldo >= n);
  if (m == 0 || n == 0) return 0;
  if (m > 0x7fffff00LL || n > 0x7fffff00LL || D > 0x7fffff00LL) return CREID_E_SHAPE;
  const int tiles_m = (int)((m + 127) / 128), tiles_n = (int)((n + 127) / 128);
  const int64_t nwg = (int64_t)tiles_m * tiles_n;
  if (nwg > 0x7fffffffLL) return CREID_E_SHAPE;
  hipStream_t s = as_stream(stream);
  dim3 grid((unsigned)nwg), block(256);
  if (dtype == CREID_F32) {
    if (D % 4 != 0) return CREID_E_SHAPE;
    hipLaunchKernelGGL(sqdist_f32_kernel, grid, block, 0, s, (const float*)q, (const float*)g, qq, gg, (int)m,
                       (int)n, (int)D, out, ldo, tiles_n);
  } else if (dtype == CREID_BF16) {
    if (D % 8 != 0) return CREID_E_SHAPE;
    hipLaunchKernelGGL(sqdist_h16_kernel<CREID_BF16>, grid, block, 0, s, (const unsigned short*)q,
                       (const unsigned short*)g, qq, gg, (int)m, (int)n, (int)D, out, ldo, tiles_n);
  } else if (dtype == CREID_F16) {
    if (D % 8 != 0) return CREID_E_SHAPE;
    hipLaunchKernelGGL(sqdist_h16_kernel<CREID_F16>, grid, block, 0, s, (const unsigned short*)q,
                       (const unsigned short*)g, qq, gg, (int)m, (int)n, (int)D, out, ldo, tiles_n);
  } else {
    return CREID_E_DTYPE;
  }
  CREID_LAUNCH_RET();
}

}  // extern "C"
