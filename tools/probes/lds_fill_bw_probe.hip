// How fast can one CU fill LDS from L2-resident global memory on gfx950?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 16 B/lane)      mode 1: global_load_lds_dword (4 B/lane)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128      mode 3: global_load_dwordx4 only (no LDS write)
// Each workgroup (256 threads) streams `iters` x 32 KB out of a small (L2-hot) buffer; prints bytes/clk/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void fill(const unsigned* __restrict__ src, unsigned* __restrict__ out, int iters,
                                            int src_words) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[2][8192];   // 2 x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned acc = 0;
  const int wg_off = (blockIdx.x * 7919) % (src_words / 8192) * 8192;
  for (int it = 0; it < iters; ++it) {
    const unsigned* s = src + ((wg_off + it * 8192) % src_words);
    unsigned* l = lds[it & 1];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)   // 8 instr/wave x 4 waves x 1 KB = 32 KB
        __builtin_amdgcn_global_load_lds((gptr_t)(s + ((i * 4 + wave) * 64 + lane) * 4), (lptr_t)(l + (i * 4 + wave) * 256), 16, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i)  // 32 instr/wave x 4 waves x 256 B = 32 KB
        __builtin_amdgcn_global_load_lds((gptr_t)(s + (i * 4 + wave) * 64 + lane), (lptr_t)(l + (i * 4 + wave) * 64), 4, 0, 0);
    } else {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(s + ((i * 4 + wave) * 64 + lane) * 4);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(l + ((i * 4 + wave) * 64 + lane) * 4) = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i].x ^ v[i].w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (MODE != 3) acc += l[(tid * 33 + it) & 8191];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// LDS-DMA with an NS-deep ring: tile t+NS-1 is issued while tile t is consumed; wait = vmcnt((NS-2)*8)
template <int NS>
__global__ __launch_bounds__(256) void fill_ring(const unsigned* __restrict__ src, unsigned* __restrict__ out, int iters,
                                                 int src_words) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[NS][8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned acc = 0;
  const int wg_off = (blockIdx.x * 7919) % (src_words / 8192) * 8192;
  auto issue = [&](int it, int buf) {
    const unsigned* s = src + ((wg_off + it * 8192) % src_words);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(s + ((i * 4 + wave) * 64 + lane) * 4), (lptr_t)(lds[buf] + (i * 4 + wave) * 256), 16, 0, 0);
  };
  for (int p = 0; p < NS - 1; ++p) issue(p, p);
  int buf = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 8) : "memory");
    asm volatile("s_barrier" ::: "memory");
    issue(it + NS - 1, buf == 0 ? NS - 1 : buf - 1);
    acc += lds[buf][(tid * 33 + it) & 8191];
    buf = (buf + 1 == NS) ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = acc;
}

template <int NS>
static void run_ring(const unsigned* d, unsigned* o, int iters, int src_words) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(fill_ring<NS>, dim3(256), dim3(256), 0, 0, d, o, iters, src_words);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(fill_ring<NS>, dim3(256), dim3(256), 0, 0, d, o, iters, src_words);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * iters * 32768.0;
  printf("LDS-DMA ring NS=%d, 1 WG/CU: %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU  (%.0f ns per 32 KB tile per WG)\n", NS, ms,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e6 / iters);
}

// The implicit-GEMM access pattern: a DMA piece = 8 rows x 128 B, rows `row_bytes` apart (= channels x 2 B of
// an NHWC pixel); a k-step moves 128 B along the row.  All workgroups walk k in lockstep (rotate = 0) or start
// at a per-workgroup offset (rotate = 1).  16 KB (128 rows) per k-step per workgroup, ring of 2.
__global__ __launch_bounds__(256) void fill_rows(const unsigned char* __restrict__ src, unsigned* __restrict__ out,
                                                 int nrows, int row_bytes, int iters, int rotate) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[2][4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = row_bytes / 128;
  const int row0 = (blockIdx.x * 128) % nrows;
  const int kstart = rotate ? (blockIdx.x * 5) % nk : 0;
  unsigned acc = 0;
  auto issue = [&](int it, int buf) {
    const int k = (kstart + it) % nk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + (i * 4 + wave) * 8 + (lane >> 3);
      const unsigned char* p = src + (size_t)r * row_bytes + k * 128 + (lane & 7) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds[buf] + (i * 4 + wave) * 256), 16, 0, 0);
    }
  };
  issue(0, 0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    issue(it + 1, (it + 1) & 1);
    acc += lds[it & 1][(tid * 33 + it) & 4095];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = acc;
}

static void run_rows(const unsigned char* d, unsigned* o, int nrows, int row_bytes, int wgs_per_cu, int rotate) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1000, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(fill_rows, dim3(grid), dim3(256), 0, 0, d, o, nrows, row_bytes, iters, rotate);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(fill_rows, dim3(grid), dim3(256), 0, 0, d, o, nrows, row_bytes, iters, rotate);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * 16384.0;
  printf("rows x 128 B pieces, row pitch %5d B, %d WG/CU, rotate %d: %7.2f TB/s  %5.1f B/clk/CU  (%.0f ns per 16 KB k-step)\n",
         row_bytes, wgs_per_cu, rotate, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e6 / iters);
}

template <int MODE>
static void run(const char* name, const unsigned* d, unsigned* o, int wgs_per_cu, int iters, int src_words) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int ncu = 256, grid = ncu * wgs_per_cu;
  hipLaunchKernelGGL(fill<MODE>, dim3(grid), dim3(256), 0, 0, d, o, iters, src_words);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fill<MODE>, dim3(grid), dim3(256), 0, 0, d, o, iters, src_words);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * 32768.0;
  printf("%-34s wgs/cu %d: %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU @2.4GHz  (%.0f ns per 32 KB tile per WG)\n", name,
         wgs_per_cu, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e6 / iters);
}

int main() {
  const int src_words = 1 << 20;   // 4 MB: L2/MALL hot
  unsigned *d, *o;
  hipMalloc(&d, src_words * 4 + 65536 * 4); hipMalloc(&o, 64);
  hipMemset(d, 1, src_words * 4 + 65536 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0>("global_load_lds_dwordx4", d, o, w, 2000, src_words);
    run<1>("global_load_lds_dword", d, o, w, 500, src_words);
    run<2>("global_load_dwordx4 + ds_write_b128", d, o, w, 2000, src_words);
    run<3>("global_load_dwordx4 only", d, o, w, 2000, src_words);
  }
  {
    unsigned char* big;
    const int nrows = 8192;
    (void)hipMalloc(&big, (size_t)nrows * 4096 + 65536);
    (void)hipMemset(big, 1, (size_t)nrows * 4096 + 65536);
    for (int rb : {128, 256, 512, 1024, 2048, 4096})
      for (int rot = 0; rot < 2; ++rot) run_rows(big, o, nrows, rb, 2, rot);
  }
  run_ring<2>(d, o, 2000, src_words);
  run_ring<3>(d, o, 2000, src_words);
  run_ring<4>(d, o, 2000, src_words);
  return 0;
}
