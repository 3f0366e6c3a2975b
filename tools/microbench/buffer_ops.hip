// Device check of raw-buffer row accessors (descriptor + per-lane voffset + scalar soffset) against plain global accesses --
// the variant of the component-parallel kernels' row addressing that was measured and not kept (rollout_cp_common.h).
// Found with it: __builtin_bit_cast applied to an ext_vector ELEMENT expression (`bit_cast<float>(v.y)`) reads element 0 with
// this compiler; cast the whole vector, then take elements.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/buffer_ops tools/microbench/buffer_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
using Rsrc = __amdgpu_buffer_rsrc_t;
__device__ Rsrc make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000); }
__global__ void k(const float* a, float* o, int n, int stride_bytes) {
  const Rsrc ra = make_rsrc(a), ro = make_rsrc(o);
  const unsigned voff = threadIdx.x * 24;       // 6 floats per lane and step: a float3 row, a float2 row, a scalar
  for (int i = 0; i < n; ++i) {
    const unsigned soff = (unsigned)i * (unsigned)stride_bytes;
    const f32x3 r = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(ra, voff, soff, 2));
    const f32x2 c = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, voff + 12, soff, 2));
    const float s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voff + 20, soff, 2));
    const f32x3 r2 = {r.x + 1.0f, r.y + 2.0f, r.z + 3.0f};
    const f32x2 c2 = {c.x + 4.0f, c.y + 5.0f};
    __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, r2), ro, voff, soff, 2);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c2), ro, voff + 12, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s + 6.0f), ro, voff + 20, soff, 2);
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
