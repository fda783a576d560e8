// Probe for VERDICT r05 item 1(b): can two INDEPENDENT kernels launched back to back on ONE stream overlap when the second is
// launched with hipExtAnyOrderLaunch (no AQL barrier bit)?  hip_ext.h says the flag is "not supported on AMD GFX9xx boards";
// this measures it on gfx950, eager and through a captured hipGraph, next to the two-stream (fork / join) form.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe && tools/probes/anyorder_probe
//
// Each kernel is 128 single-wave workgroups spinning ~T us (half the CUs): two of them fit side by side, so perfect overlap
// is 1.0 x T per pair and serial execution 2.0 x T.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin_kernel(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && cycles < 0) *sink = 1;
}

static float time_pairs(int mode, int pairs, long long cyc, hipStream_t s0, hipStream_t s1, bool graph) {
  // mode 0: plain launches, one stream; 1: second of each pair with hipExtAnyOrderLaunch; 2: second on another stream (fork / join)
  // 3 / 4: see below
  hipEvent_t e0, e1, fork, join;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
  int* sink = nullptr;
  auto body = [&]() {
    for (int p = 0; p < pairs; ++p) {
      if (mode == 2) { hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0); }   // fork BEFORE the first kernel: the two are independent
      hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(64), 0, s0, cyc, sink);
      if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(64), 0, s0, cyc, sink);
      else if (mode == 1) hipExtLaunchKernelGGL(spin_kernel, dim3(128), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, sink);
      else if (mode == 2) {
        hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(64), 0, s1, cyc, sink);
        hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0);
      } else {
        // mode 3: DEPENDENT side work without joins: B_p on the side stream waits for A_p (an edge out of the main chain), the main
        // chain never waits until the end -- the shape of "weight gradients on a side stream"; mode 4: one such edge every 4th pair
        if (mode == 3 || p % 4 == 3) { hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0); }
        hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(64), 0, s1, cyc, sink);
      }
    }
    if (mode >= 3) { hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0); }
  };
  float ms = 0.f;
  if (!graph) {
    body(); hipStreamSynchronize(s0);
    hipEventRecord(e0, s0); body(); hipEventRecord(e1, s0); hipStreamSynchronize(s0);
  } else {
    hipGraph_t g; hipGraphExec_t ge;
    if (hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1.f;
    body();
    if (hipStreamEndCapture(s0, &g) != hipSuccess) return -1.f;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1.f;
    hipGraphLaunch(ge, s0); hipStreamSynchronize(s0);
    hipEventRecord(e0, s0); hipGraphLaunch(ge, s0); hipEventRecord(e1, s0); hipStreamSynchronize(s0);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / pairs;
}

int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const char* names[5] = {"plain, one stream", "second kernel hipExtAnyOrderLaunch, one stream", "second kernel on a second stream (fork / join)",
                          "B_p on a side stream after A_p, one join at the end", "same, side stream forked every 4th pair only"};
  for (int us : {5, 20, 50}) {
    const long long cyc = (long long)us * rate_khz / 1000;
    printf("kernel ~%d us (128 one-wave workgroups; wall clock %d kHz): us per PAIR\n", us, rate_khz);
    for (int graph = 0; graph < 2; ++graph)
      for (int mode = 0; mode < 5; ++mode)
        printf("  %-6s %-50s %8.1f\n", graph ? "graph" : "eager", names[mode], time_pairs(mode, 20, cyc, s0, s1, graph));
  }
  return 0;
}
