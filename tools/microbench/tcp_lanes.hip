// What does the vector memory path (TA / TCP) charge a load instruction for: active lanes, distinct addresses or distinct
// lines?  16 waves per CU on every CU issue independent loads from a 256 KB table (L2-resident, like the rollout's map) with
// different lane -> address patterns; prints cycles per load instruction and CU.
//   hipcc --offload-arch=gfx950 -O3 tcp_lanes.hip -o tcp_lanes && ./tcp_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 2048
template <int MODE, int WIDTH>
__global__ void __launch_bounds__(1024) k(const float* __restrict__ tab, float* out, long long* cyc, unsigned mask) {
  const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned idx;
  if (MODE == 0) idx = wave * 97;                                   // all 64 lanes the same address
  else if (MODE == 1) idx = wave * 97 + (lane >> 2) * 1031;         // 16 distinct lines, each address 4 times (a rollout's lanes)
  else if (MODE == 2) idx = wave * 97 + lane * 1031;                // 64 distinct lines
  else if (MODE == 3) idx = wave * 64 + lane * WIDTH;               // consecutive elements (fully coalesced)
  else idx = wave * 97 + (lane >> 2) * 1031;                        // MODE 4: as 1, but only one lane in four active
  idx &= mask;
  float acc = 0;
  const bool on = MODE != 4 || (lane & 3) == 0;
  long long t0 = clock64();
  if (on) {
#pragma unroll 8
    for (int i = 0; i < REP; ++i) {
      if (WIDTH == 1) acc += tab[idx];
      else if (WIDTH == 2) { float2 v = *(const float2*)(tab + (idx & ~1u)); acc += v.x + v.y; }
      else { float4 v = *(const float4*)(tab + (idx & ~3u)); acc += v.x + v.y + v.z + v.w; }
      idx = (idx * 5 + 1 + (WIDTH > 1 ? WIDTH * 4 : 0)) & mask;     // next pseudo-random address: independent of the loaded data
      if (MODE == 3) idx = (idx + 64 * WIDTH) & mask;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int WIDTH>
static void run(const char* name, const float* tab, float* out, long long* cyc) {
  const int blocks = 256;   // one 16-wave workgroup per CU
  std::vector<long long> h(blocks);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE, WIDTH>), dim3(blocks), dim3(1024), 0, 0, tab, out, cyc, 65535u); hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost); }
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-58s %6.1f ticks per load instruction and CU\n", name, s / blocks / REP / 16);
}
int main() {
  float *tab, *out; long long* cyc;
  hipMalloc(&tab, 4 * 65536 + 64); hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8 * 256);
  hipMemset(tab, 0, 4 * 65536 + 64);
  run<0, 1>("dword,   64 lanes, one address", tab, out, cyc);
  run<1, 1>("dword,   64 lanes, 16 addresses x 4 lanes", tab, out, cyc);
  run<4, 1>("dword,   16 lanes active (1 in 4), 16 addresses", tab, out, cyc);
  run<2, 1>("dword,   64 lanes, 64 lines", tab, out, cyc);
  run<3, 1>("dword,   64 lanes, consecutive", tab, out, cyc);
  run<2, 2>("dwordx2, 64 lanes, 64 lines", tab, out, cyc);
  run<2, 4>("dwordx4, 64 lanes, 64 lines", tab, out, cyc);
  run<1, 2>("dwordx2, 64 lanes, 16 addresses x 4 lanes", tab, out, cyc);
  run<4, 2>("dwordx2, 16 lanes active, 16 addresses", tab, out, cyc);
  run<3, 4>("dwordx4, 64 lanes, consecutive", tab, out, cyc);
  return 0;
}
