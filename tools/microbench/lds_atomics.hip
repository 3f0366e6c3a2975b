// What does an LDS float atomic cost (gfx950)?  One workgroup per CU, W waves, every lane issues REP x 8 LDS operations on a 128 KB
// table with different lane -> address patterns; prints cycles per wave instruction and CU (the CU's LDS pipe is shared by its waves).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 512
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) unsigned lds_u;
// OP 0: ds_write_b32   1: ds_add_u32   2: ds_add_f32   3: ds_add_rtn_f32 (result used)   4: ds_pk_add_f16?? (skipped)
// PAT 0: consecutive (conflict-free)   1: pseudo-random cells of a 128 x 128 window   2: 16 distinct random addresses x 4 lanes
//     3: all lanes one address          4: random, row stride 129 (padded)
template <int OP, int PAT, int ACTIVE = 64>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc) {
  __shared__ unsigned long long tab64[16384];
  float* tab = reinterpret_cast<float*>(tab64);
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
  if (PAT == 2) s = (blockIdx.x * 1024 + (threadIdx.x >> 2)) * 2654435761u + 12345u;
  float acc = 0.f;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned idx;
      s = s * 1664525u + 1013904223u;
      if (PAT == 0) idx = (wave * 64 + lane + q * 1024 + i * 64) & 32767u;
      else if (PAT == 3) idx = (wave * 97 + q) & 32767u;
      else if (PAT == 4) idx = (((s >> 9) & 127u) * 129u + ((s >> 20) & 127u)) & 32767u;
      else idx = (s >> 10) & 16383u;
      const float v = 1.0f + (float)q;
      if (ACTIVE < 64 && (lane % (64 / ACTIVE)) != 0) continue;      // ACTIVE lanes of the wave issue the operation
      if (OP == 5) { __hip_atomic_fetch_add(&tab64[idx >> 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); continue; }
      if (OP == 0) tab[idx] = v;
      else if (OP == 1) __hip_atomic_fetch_add((unsigned*)&tab[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (OP == 2) __builtin_amdgcn_ds_faddf((lds_f*)&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
      else acc += __builtin_amdgcn_ds_faddf((lds_f*)&tab[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = clock64();
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + tab[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int PAT, int ACTIVE = 64>
static void run(const char* name, int waves, float* out, long long* cyc) {
  const int blocks = 256;
  std::vector<long long> h(blocks);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<OP, PAT, ACTIVE>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc); hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost); }
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-64s %2d waves per CU: %7.1f ticks per wave instruction and CU\n", name, waves, s / blocks / REP / 8 / waves);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8 * 256);
  for (int w : {4, 8}) {
    run<0, 0>("ds_write_b32, consecutive", w, out, cyc);
    run<0, 1>("ds_write_b32, random cells", w, out, cyc);
    run<1, 0>("ds_add_u32, consecutive", w, out, cyc);
    run<1, 1>("ds_add_u32, random cells", w, out, cyc);
    run<2, 0>("ds_add_f32, consecutive", w, out, cyc);
    run<2, 1>("ds_add_f32, random cells of a 128 x 128 window", w, out, cyc);
    run<2, 4>("ds_add_f32, random cells, row stride 129", w, out, cyc);
    run<2, 2>("ds_add_f32, 16 random addresses x 4 lanes", w, out, cyc);
    run<2, 3>("ds_add_f32, one address", w, out, cyc);
    run<3, 1>("ds_add_rtn_f32, random cells", w, out, cyc);
    run<2, 1, 32>("ds_add_f32, random cells, 32 active lanes", w, out, cyc);
    run<2, 1, 16>("ds_add_f32, random cells, 16 active lanes", w, out, cyc);
    run<2, 1, 4>("ds_add_f32, random cells, 4 active lanes", w, out, cyc);
    run<2, 1, 1>("ds_add_f32, random cells, 1 active lane", w, out, cyc);
    run<5, 1>("ds_add_u64, random cells", w, out, cyc);
  }
  return 0;
}
