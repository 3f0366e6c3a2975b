// Issue cost of packed FP32 (v_pk_fma_f32) vs scalar v_fma_f32 for ONE wave: dependent and independent chains.
//   hipcc --offload-arch=gfx950 -O3 pk_latency.hip -o pk_latency && ./pk_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 4096
__global__ void k_scalar_dep(float* out, long long* cyc, float a, float b) {
  float x = threadIdx.x;
  long long t0 = clock64();
#pragma unroll 64
  for (int i = 0; i < REP; ++i) x = __builtin_fmaf(x, a, b);
  long long t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x % 64 == 0) { cyc[1 + 2 * (threadIdx.x / 64)] = t0; cyc[2 + 2 * (threadIdx.x / 64)] = t1; }
}
__global__ void k_pk_dep(float* out, long long* cyc, float a, float b) {
  f2 x = {(float)threadIdx.x, 1.0f}; f2 av = {a, a}, bv = {b, b};
  long long t0 = clock64();
#pragma unroll 64
  for (int i = 0; i < REP; ++i) x = __builtin_elementwise_fma(x, av, bv);
  long long t1 = clock64();
  out[threadIdx.x] = x.x + x.y; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_scalar_ind(float* out, long long* cyc, float a, float b) {
  float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP / 4; ++i) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
  long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3; if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x % 64 == 0) { cyc[1 + 2 * (threadIdx.x / 64)] = t0; cyc[2 + 2 * (threadIdx.x / 64)] = t1; }
}
__global__ void k_pk_ind(float* out, long long* cyc, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.0f}, x1 = {2, 3}, x2 = {4, 5}, x3 = {6, 7}; f2 av = {a, a}, bv = {b, b};
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP / 4; ++i) { x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv); x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv); }
  long long t1 = clock64();
  out[threadIdx.x] = x0.x + x1.y + x2.x + x3.y; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_scalar_ind_masked(float* out, long long* cyc, float a, float b, int lanes) {
  // same chain with only `lanes` of the 64 lanes active: does the SIMD skip 16-lane passes whose EXEC bits are all zero?
  if ((int)threadIdx.x >= lanes) return;
  float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP / 4; ++i) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
  long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 4 * 2048); hipMalloc(&cyc, 8 * 128);
  struct { const char* n; void (*k)(float*, long long*, float, float); } ks[] = {{"v_fma_f32 dependent", k_scalar_dep}, {"v_pk_fma_f32 dependent", k_pk_dep},
      {"v_fma_f32 4 independent", k_scalar_ind}, {"v_pk_fma_f32 4 independent", k_pk_ind}};
  for (auto& k : ks) {
    long long c = 0;
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k.k, dim3(1), dim3(64), 0, 0, out, cyc, 0.999f, 0.001f); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-28s %lld clock64 ticks for %d instructions = %.2f ticks each\n", k.n, c, REP, (double)c / REP);
  }
  for (int lanes : {64, 32, 16, 1}) {
    long long c = 0;
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_scalar_ind_masked, dim3(1), dim3(64), 0, 0, out, cyc, 0.999f, 0.001f, lanes); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    printf("v_fma_f32 4 independent, %2d active lanes: %.2f ticks each\n", lanes, (double)c / REP);
  }
  // several waves per SIMD: does the SIMD keep issuing one VALU instruction per 4 cycles when it has 2 / 4 / 8 waves to choose from?
  // (span = last wave's end - first wave's start: the arbiter favours the oldest wave, so wave 0 alone says nothing)
  for (int waves : {4, 8, 16}) {        // one workgroup on one CU: waves / 4 per SIMD
    long long st[1 + 2 * 32];
    auto span = [&]() { long long lo = st[1], hi = st[2]; for (int w = 0; w < waves; ++w) { lo = st[1 + 2 * w] < lo ? st[1 + 2 * w] : lo; hi = st[2 + 2 * w] > hi ? st[2 + 2 * w] : hi; } return hi - lo; };
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_scalar_ind, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.999f, 0.001f); hipMemcpy(st, cyc, sizeof(st), hipMemcpyDeviceToHost); }
    printf("v_fma_f32 4 independent, %2d waves on one CU (%d per SIMD): %.2f ticks per instruction and SIMD issue slot\n", waves, waves / 4, (double)span() / REP / (waves / 4));
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_scalar_dep, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.999f, 0.001f); hipMemcpy(st, cyc, sizeof(st), hipMemcpyDeviceToHost); }
    printf("v_fma_f32 dependent,     %2d waves on one CU (%d per SIMD): %.2f ticks per instruction and SIMD issue slot\n", waves, waves / 4, (double)span() / REP / (waves / 4));
  }
  return 0;
}
