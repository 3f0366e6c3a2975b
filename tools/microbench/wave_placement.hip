// Where does the dispatcher put the waves of a launch of one-wave workgroups?  Every wave records its (XCC, SE, CU, SIMD) from
// the hardware-id registers while all waves of the launch are resident; the host prints the histogram of waves per SIMD / CU.
//   hipcc --offload-arch=gfx950 -O3 wave_placement.hip -o wave_placement && ./wave_placement [block] [big: 1 = the kernel takes 208 VGPRs] [LDS bytes per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#define K_PLACE_BODY \
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     /* HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] */ \
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   /* XCC_ID */ \
  const unsigned long long t_start = wall_clock64();                 /* 100 MHz constant clock: when did this wave start? */ \
  float x = threadIdx.x; \
  for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);   /* stay resident until the whole grid is placed */ \
  if (threadIdx.x % 64 == 0) { const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) / 64; ids[2 * w] = hw; ids[2 * w + 1] = xcc; \
    if (starts) { starts[2 * w] = t_start; starts[2 * w + 1] = wall_clock64(); } } \
  if (x == 12345.f) sink[0] = x;
template <int BIG> __global__ void k_place(unsigned* ids, float* sink, int spin, unsigned long long* starts);
template <> __global__ void k_place<1>(unsigned* ids, float* sink, int spin, unsigned long long* starts) { asm volatile("v_mov_b32 v207, 0" ::: "v207"); K_PLACE_BODY }      // 208 registers: a SIMD holds two such waves
template <> __global__ void k_place<0>(unsigned* ids, float* sink, int spin, unsigned long long* starts) {
  K_PLACE_BODY
}
int main(int argc, char** argv) {
  const int block = argc > 1 ? atoi(argv[1]) : 64;
  const int big = argc > 2 ? atoi(argv[2]) : 0;
  const int lds = argc > 3 ? atoi(argv[3]) : 0;      // dynamic LDS bytes per workgroup (nobody uses them)
  for (int waves0_ : {256, 512, 768, 1024, 1536, 2048, 4096}) {
    int waves0 = waves0_;
    if (block == 192 && waves0 == 2048) waves0 = 1536;      // 512 workgroups of three waves
    const int waves = waves0 / (block / 64) * (block / 64);
    unsigned* ids; float* sink; unsigned long long* starts; hipMalloc(&ids, 8 * waves); hipMalloc(&sink, 4); hipMalloc(&starts, 16 * waves);
    if (big) hipLaunchKernelGGL(k_place<1>, dim3(waves * 64 / block), dim3(block), lds, 0, ids, sink, 200000, starts);
    else hipLaunchKernelGGL(k_place<0>, dim3(waves * 64 / block), dim3(block), lds, 0, ids, sink, 200000, starts);
    std::vector<unsigned> h(2 * waves); hipMemcpy(h.data(), ids, 8 * waves, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> ht(2 * waves); hipMemcpy(ht.data(), starts, 16 * waves, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0, e0 = ~0ull, e1 = 0;
    for (int w = 0; w < waves; ++w) { t0 = std::min(t0, ht[2 * w]); t1 = std::max(t1, ht[2 * w]); e0 = std::min(e0, ht[2 * w + 1]); e1 = std::max(e1, ht[2 * w + 1]); }
    printf("      first to last wave START %.2f us, first to last wave END %.2f us, first start to last end %.2f us (100 MHz clock)\n", (t1 - t0) * 0.01, (e1 - e0) * 0.01, (e1 - t0) * 0.01);
    std::map<unsigned, int> per_simd, per_cu, per_xcc;
    for (int w = 0; w < waves; ++w) {
      const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      per_cu[cu_key]++; per_simd[(cu_key << 2) | simd]++; per_xcc[xcc]++;
    }
    // workgroups of several waves: do the waves of ONE workgroup sit on different SIMDs?
    const int wpb = block / 64;
    int shared_simd = 0, split_cu = 0;
    for (int g = 0; wpb > 1 && g + wpb <= waves; g += wpb) {
      bool same = false, other_cu = false;
      for (int i = 0; i < wpb; ++i)
        for (int j = i + 1; j < wpb; ++j) {
          same |= ((h[2 * (g + i)] >> 4) & 3) == ((h[2 * (g + j)] >> 4) & 3);
          other_cu |= ((h[2 * (g + i)] >> 8) & 0xff) != ((h[2 * (g + j)] >> 8) & 0xff);
        }
      shared_simd += same; split_cu += other_cu;
    }
    if (wpb > 1) printf("      workgroups of %d waves: %d of %d have two waves on one SIMD (%d span CUs)\n", wpb, shared_simd, waves / wpb, split_cu);
    if (wpb == 3 && waves / wpb <= 512) {      // two 3-wave workgroups per CU (the streaming backward at 2048 rollouts): which SIMDs, in wave order?
      std::map<unsigned, std::vector<int>> by_cu;      // CU -> workgroup ids
      for (int g = 0; g + wpb <= waves; g += wpb) {
        const unsigned hw = h[2 * g], xcc = h[2 * g + 1] & 0xf;
        by_cu[(xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)].push_back(g / wpb);
      }
      std::map<std::string, int> pat;
      for (auto& kv : by_cu) {
        std::string k;
        for (int wg : kv.second) { k += "("; for (int i = 0; i < wpb; ++i) k += char('0' + ((h[2 * (wg * wpb + i)] >> 4) & 3)); k += ")"; }
        pat[k]++;
      }
      printf("      SIMDs of the waves, per CU, workgroups in id order:");
      for (auto& kv : pat) printf("  %s x%d", kv.first.c_str(), kv.second);
      printf("\n");
    }
    std::map<int, int> hs, hc;
    for (auto& kv : per_simd) hs[kv.second]++;
    for (auto& kv : per_cu) hc[kv.second]++;
    printf("%5d waves (block %d%s): %zu XCCs, %zu CUs, %zu SIMDs used;  SIMDs by wave count:", waves, block, big ? ", 208 VGPRs" : "", per_xcc.size(), per_cu.size(), per_simd.size());
    for (auto& kv : hs) printf(" %dx%d", kv.second, kv.first);
    printf(";  CUs by wave count:");
    for (auto& kv : hc) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
    hipFree(ids); hipFree(sink); hipFree(starts);
  }
  return 0;
}
