// interpolate_grid, float32 fast math (hardware reciprocal square root, FMA contraction) -- the arithmetic of the *_fast rollout kernels.
#include "interp_grid_kernel.h"

namespace mf {
void launch_interp_fast_f32(const InterpArgs<float>& a, hipStream_t st) { launch_interp<float, true>(a, st); }
}  // namespace mf
