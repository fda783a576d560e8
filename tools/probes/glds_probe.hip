// Probe __builtin_amdgcn_global_load_lds (16-byte) on gfx950: LDS destination = wave-uniform base + lane*16?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const unsigned* src, unsigned* out, int perm) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 64 * 4; i += 256) lds[i] = 0xdeadbeef;
  __syncthreads();
  // each lane picks its own global source (permuted), LDS base is per wave
  const int srcl = perm ? (lane ^ 5) : lane;
  const unsigned* g = src + (wave * 64 + srcl) * 4;
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                   (void __attribute__((address_space(3)))*)(lds + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 64 * 4; i += 256) out[i] = lds[i];
}

int main() {
  unsigned h[1024], *d, *o;
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int perm = 0; perm < 2; ++perm) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, o, perm);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int w = 0; w < 4 && ok; ++w)
      for (int l = 0; l < 64 && ok; ++l)
        for (int k = 0; k < 4; ++k) {
          const unsigned expect = (w * 64 + (perm ? (l ^ 5) : l)) * 4 + k;
          if (h[w * 256 + l * 4 + k] != expect) { printf("perm %d mismatch w%d l%d k%d got %u expect %u\n", perm, w, l, k, h[w * 256 + l * 4 + k], expect); ok = 0; break; }
        }
    printf("perm %d: %s\n", perm, ok ? "LDS[base + lane*16] <- each lane's own global 16 B : OK" : "FAIL");
  }
  return 0;
}
