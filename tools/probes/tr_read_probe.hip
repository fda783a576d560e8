// Probe the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out, int pitch_elems) {
  __shared__ __attribute__((aligned(16))) volatile unsigned short lds[64 * 256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 256; i += 64) lds[i] = (unsigned short)i;   // value = element index
  __syncthreads();
  // each 16-lane group g reads a 4(row) x 16(col) block: lane i -> row (i>>2), cols 4*(i&3)..+3
  const int g = lane >> 4, i = lane & 15;
  const int row = 4 * g + (i >> 2), col = 4 * (i & 3);
  const unsigned addr = (unsigned)(uintptr_t)(&lds[row * pitch_elems + col]);
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}

int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  for (int pitch : {16, 160}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
      printf("\n");
    }
  }
  return 0;
}
