// Does a long straight-line loop body run slower when the four waves of a CU belong to four one-wave workgroups than when they are
// the four waves of ONE workgroup?  (The record-reading backward does: 0.54 vs 0.44 ms at 4096 rollouts, with the waves spread
// evenly over the SIMDs either way -- wave_placement.hip.)  The loop here is BODY independent-chain FMAs (8 bytes each), no memory.
//   hipcc --offload-arch=gfx950 -O3 ifetch_lockstep.hip -o ifetch_lockstep && ./ifetch_lockstep
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BODY>
__global__ void __launch_bounds__(256) k_body(float* sink, int iters) {
  float a = threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  asm volatile("v_mov_b32 v207, 0" ::: "v207");      // 208 registers, as the backward
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < BODY / 4; ++j) {
      a = __builtin_fmaf(a, 0.999f, 0.001f); b = __builtin_fmaf(b, 0.998f, 0.002f);
      c = __builtin_fmaf(c, 0.997f, 0.003f); d = __builtin_fmaf(d, 0.996f, 0.004f);
    }
  }
  if (a + b + c + d == 12345.f) sink[0] = a;
}
template <int BODY>
static void run(int waves, int block) {
  float* sink; hipMalloc(&sink, 4);
  const int iters = 400000 / BODY * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_body<BODY>, dim3(waves * 64 / block), dim3(block), 0, 0, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
  }
  printf("  body %5d instructions (%5d bytes)  %5d waves  workgroups of %3d threads  %8.3f ms\n", BODY, BODY * 8, waves, block, best);
  hipFree(sink);
}
int main() {
  for (int waves : {256, 512, 768, 1024, 2048})
    for (int block : {64, 256}) { run<64>(waves, block); run<1024>(waves, block); run<4096>(waves, block); }
  return 0;
}
