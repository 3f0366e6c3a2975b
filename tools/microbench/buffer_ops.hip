// Device check of the raw-buffer accessors of the component-parallel kernels (monoforce_amd/csrc/rollout_cp_common.h): rows
// read / written through a descriptor (per-lane voffset + scalar soffset) must equal plain global accesses, element by element.
// (Found with it: __builtin_bit_cast on an ext_vector ELEMENT expression reads element 0 -- the accessors cast whole vectors.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/buffer_ops tools/microbench/buffer_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../monoforce_amd/csrc/rollout_cp_common.h"
using namespace mf::cp;
namespace mf { void set_error(const std::string&) {} }
__global__ void k(const float* a, float* o, int n, int stride_bytes) {
  const Rsrc ra = make_rsrc(a), ro = make_rsrc(o);
  const unsigned voff = threadIdx.x * 24;       // 6 floats per lane and step: one float3 row + one float2 row + one scalar
  for (int i = 0; i < n; ++i) {
    const unsigned soff = (unsigned)i * (unsigned)stride_bytes;
    float x, y, z, c0, c1;
    bload3(ra, voff, soff, &x, &y, &z);
    bload2(ra, voff + 12, soff, &c0, &c1);
    const float s = bload1(ra, voff + 20, soff);
    bstore3(ro, voff, soff, x + 1.0f, y + 2.0f, z + 3.0f);
    bstore2(ro, voff + 12, soff, c0 + 4.0f, c1 + 5.0f);
    bstore1(ro, voff + 20, soff, s + 6.0f);
  }
}
int main() {
  const int n = 5, lanes = 64, stride = lanes * 24;
  std::vector<float> h(n * lanes * 6), out(h.size(), -1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i * 0.5f;
  float *da, *dout;
  if (hipMalloc(&da, h.size() * 4) != hipSuccess || hipMalloc(&dout, h.size() * 4) != hipSuccess) return 2;
  (void)hipMemcpy(da, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0xff, h.size() * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(lanes), 0, 0, da, dout, n, stride);
  (void)hipMemcpy(out.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const float e = h[i] + (float)(i % 6 + 1);
    if (out[i] != e) { if (bad < 5) printf("i=%zu got %g want %g\n", i, out[i], e); ++bad; }
  }
  printf("buffer accessors: %d mismatches of %zu\n", bad, h.size());
  return bad != 0;
}
