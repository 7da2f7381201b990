/*
 * oracle/ref_shim/ref_host.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Host driver of the reference's own kernels (see cuda_runtime.h in this directory).  The two .inc files are
 * GENERATED into oracle/_ref/ by the Makefile from /root/reference (never committed):
 *   gen_sg_kernels.inc       sg.cu from `using namespace math;` up to the first torch entry point
 *                            (= sg.cu:16-175: constants, square(), evaluate_gaussian_{fwd,bwd}_kernel)
 *   gen_raydirs_kernel.inc   utils_kernel.cu from `using namespace math;` up to the backward kernel
 *                            (= utils_kernel.cu:9-51: compute_raydirs_forward_kernel)
 * The launch geometry below repeats the reference's launchers: sg.cu:206-207,260-261 (grid {N, ceil(D/128)},
 * block 128) and utils_kernel.cu:107-113 (16x16 blocks over W x N*H).
 */
#include <cuda_runtime.h>

thread_local dim3 blockIdx, blockDim, threadIdx, gridDim;

#include "helper_math.h"

namespace ref_sg {
using namespace math;
#include "gen_sg_kernels.inc"
}
namespace ref_utils {
using namespace math;
#include "gen_raydirs_kernel.inc"
}

template <typename F>
static void launch(dim3 grid, dim3 block, F&& body) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int by = 0; by < (int)grid.y; ++by)
    for (int bx = 0; bx < (int)grid.x; ++bx) {
      gridDim = grid; blockDim = block; blockIdx = dim3(bx, by, 0);
      for (unsigned ty = 0; ty < block.y; ++ty)
        for (unsigned tx = 0; tx < block.x; ++tx) {
          threadIdx = dim3(tx, ty, 0);
          body();
        }
    }
}

extern "C" {

/* sgutilslib.evaluate_gaussian_fwd (sg.cu:177-226) */
void ref_sg_fwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas, const float* light_values,
                const float* light_pts, const float* prim_pts, const int* n_lights, int w_type, float* integral) {
  const unsigned bs = 128;
  launch(dim3(N, (D + bs - 1) / bs, 1), dim3(bs), [&] {
    ref_sg::evaluate_gaussian_fwd_kernel((const float3*)lobe_dirs, lobe_sigmas, (const float3*)light_values,
                                         (const float3*)light_pts, (const float3*)prim_pts, n_lights, (float3*)integral,
                                         N, D, L, w_type);
  });
}

/* sgutilslib.evaluate_gaussian_bwd (sg.cu:228-278); grad_light_values may be NULL, else pre-zeroed [N,L,3] */
void ref_sg_bwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas, const float* light_values,
                const float* light_pts, const float* prim_pts, const int* n_lights, const float* grad_integral,
                int w_type, float* grad_dirs, float* grad_sigmas, float* grad_light_values) {
  const unsigned bs = 128;
  launch(dim3(N, (D + bs - 1) / bs, 1), dim3(bs), [&] {
    ref_sg::evaluate_gaussian_bwd_kernel((const float3*)lobe_dirs, lobe_sigmas, (const float3*)light_values,
                                         (const float3*)light_pts, (const float3*)prim_pts, n_lights,
                                         (const float3*)grad_integral, (float3*)grad_dirs, grad_sigmas,
                                         (float3*)grad_light_values, N, D, L, w_type);
  });
}

/* utilslib.compute_raydirs_forward (utils.cpp:46-82, utils_kernel.cu:96-128); pixelcoords may be NULL */
void ref_raydirs(int N, int H, int W, float* viewpos, float* viewrot, float* focal, float* princpt, float* pixelcoords,
                 float volradius, float* raypos, float* raydir, float* tminmax) {
  dim3 block(16, 16);
  launch(dim3((W + 15) / 16, (N * H + 15) / 16), block, [&] {
    ref_utils::compute_raydirs_forward_kernel(N, H, W, (float3*)viewpos, (float3*)viewrot, (float2*)focal,
                                              (float2*)princpt, (float2*)pixelcoords, volradius, (float3*)raypos,
                                              (float3*)raydir, (float2*)tminmax);
  });
}

}  // extern "C"
