// Issue cost of DPP instructions for ONE wave (the adjoint chain of the component-parallel backward is ~25 % DPP):
// v_mov_b32_dpp quad_perm, v_add_f32_dpp (DPP folded into the add), the dot3 pattern (mul + two DPP adds), dependent and independent.
//   hipcc --offload-arch=gfx950 -O3 dpp_latency.hip -o dpp_latency && ./dpp_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kRot1 = 0x09 | (0x3 << 6);      // quad_perm [1,2,0,3]
constexpr int kRot2 = 0x12 | (0x3 << 6);      // quad_perm [2,0,1,3]
#define TIMED(body) long long t0 = clock64(); body; long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
__global__ void k_mov_dep(float* out, long long* cyc, float a) {
  float x = threadIdx.x;
  TIMED(_Pragma("unroll 64") for (int i = 0; i < REP; ++i) { x = dpp<kRot1>(x); asm volatile("" : "+v"(x)); })
  out[threadIdx.x] = x;
}
__global__ void k_mov_ind(float* out, long long* cyc, float a) {
  float x0 = threadIdx.x, x1 = 1 + x0, x2 = 2 + x0, x3 = 3 + x0;
  TIMED(_Pragma("unroll 16") for (int i = 0; i < REP / 4; ++i) { x0 = dpp<kRot1>(x0); x1 = dpp<kRot1>(x1); x2 = dpp<kRot1>(x2); x3 = dpp<kRot1>(x3);
                                                                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); })
  out[threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void k_add_dep(float* out, long long* cyc, float a) {      // x = x + dpp(x): v_add_f32_dpp
  float x = threadIdx.x;
  TIMED(_Pragma("unroll 64") for (int i = 0; i < REP; ++i) { x = x * a + dpp<kRot1>(x); })
  out[threadIdx.x] = x;
}
__global__ void k_add_ind(float* out, long long* cyc, float a) {
  float x0 = threadIdx.x, x1 = 1 + x0, x2 = 2 + x0, x3 = 3 + x0;
  TIMED(_Pragma("unroll 16") for (int i = 0; i < REP / 4; ++i) { x0 = a + dpp<kRot1>(x0); x1 = a + dpp<kRot1>(x1); x2 = a + dpp<kRot1>(x2); x3 = a + dpp<kRot1>(x3); })
  out[threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void k_dot3_dep(float* out, long long* cyc, float a) {     // the lane-sum of a 3-vector: mul, add_dpp, add_dpp (REP / 4 dots = 3 * REP / 4 instructions)
  float x = threadIdx.x;
  TIMED(_Pragma("unroll 16") for (int i = 0; i < REP / 4; ++i) { float t = x * a; t = t + dpp<kRot1>(t); x = t + dpp<kRot2>(x * a); })
  out[threadIdx.x] = x;
}
__global__ void k_fma_dep(float* out, long long* cyc, float a) {
  float x = threadIdx.x;
  TIMED(_Pragma("unroll 64") for (int i = 0; i < REP; ++i) x = __builtin_fmaf(x, a, 0.001f);)
  out[threadIdx.x] = x;
}
__global__ void k_lds(float* out, long long* cyc, float a) {          // ten 16-byte LDS reads per lane back to back, then one use of each
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 buf[10 * 64];
  for (int i = threadIdx.x; i < 640; i += 64) buf[i] = f4{a, a, a, a};
  __syncthreads();
  float x = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < REP / 16; ++i) {
    f4 s = buf[threadIdx.x];
#pragma unroll
    for (int p = 1; p < 10; ++p) s += buf[p * 64 + threadIdx.x];
    x += s.x + s.y + s.z + s.w;
    asm volatile("" ::: "memory");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = x;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 4 * 2048); hipMalloc(&cyc, 8 * 128);
  struct { const char* n; void (*k)(float*, long long*, float); double per; } ks[] = {
      {"v_fma_f32 dependent", k_fma_dep, REP}, {"v_mov_b32_dpp dependent", k_mov_dep, REP}, {"v_mov_b32_dpp 4 independent", k_mov_ind, REP},
      {"fma-like + dpp operand, dependent (mul + add_dpp per trip)", k_add_dep, REP}, {"v_add_f32_dpp 4 independent", k_add_ind, REP},
      {"dot3 pattern dependent (per dot3)", k_dot3_dep, REP / 4}, {"10 x ds_read_b128 + 14 adds (per trip)", k_lds, REP / 16}};
  for (auto& k : ks) {
    long long c = 0;
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k.k, dim3(1), dim3(64), 0, 0, out, cyc, 0.999f); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-60s %.2f ticks each\n", k.n, (double)c / k.per);
  }
  return 0;
}
