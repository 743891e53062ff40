// Input side of the render path: observer rays generated on the device.
//
// Restates get_rays, sunerf/data/ray_sampling.py:7-36: helioprojective pixel angles (Tx, Ty) [rad] ->
//   direction = (sin Tx, -sin Ty cos Tx, -cos Tx cos Ty)      evaluated in fp64, rounded to fp32 (np.stack(..., dtype=float32))
//   rays_d[r] = sum_c direction[c] * c2w[r][c]               fp32 products, summed left to right (np.sum over 3 elements)
//   rays_o    = c2w[:3, 3] tiled
// and the per-ray time column that evaluation/loader.py:92,214 builds with ones_like(...) * time.
// One thread per pixel; 48 bytes written per ray, nothing read but the two angles.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

struct RayArgs {
  const double* tx; const double* ty;
  int per_pixel, width;
  int64_t pix_begin, n_pix;
  float c2w[12];
  float* rays_o; float* rays_d;
  float* times; float time_value;
};

__global__ __launch_bounds__(256) void observer_rays_kernel(RayArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pix) return;
  const int64_t p = a.pix_begin + i;
  const double Tx = a.per_pixel ? a.tx[p] : a.tx[p % a.width];
  const double Ty = a.per_pixel ? a.ty[p] : a.ty[p / a.width];
  const double sx = sin(Tx), cx = cos(Tx), sy = sin(Ty), cy = cos(Ty);
  const float d0 = (float)sx, d1 = (float)(-sy * cx), d2 = (float)(-cx * cy);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.rays_d[i * 3 + r] = (d0 * a.c2w[4 * r + 0] + d1 * a.c2w[4 * r + 1]) + d2 * a.c2w[4 * r + 2];
    a.rays_o[i * 3 + r] = a.c2w[4 * r + 3];
  }
  if (a.times) a.times[i] = a.time_value;
}

}  // namespace

extern "C" int sunerf_observer_rays(const double* tx, const double* ty, int per_pixel, int width, int64_t pix_begin,
                                    int64_t n_pix, const float* c2w_host, float time_value, float* rays_o, float* rays_d,
                                    float* times, void* stream) {
  if (n_pix < 0 || pix_begin < 0 || width < 1 || !c2w_host) return SUNERF_E_BADARG;
  if (n_pix == 0) return 0;
  if (!tx || !ty || !rays_o || !rays_d) return SUNERF_E_BADARG;
  RayArgs a;
  a.tx = tx; a.ty = ty; a.per_pixel = per_pixel != 0; a.width = width; a.pix_begin = pix_begin; a.n_pix = n_pix;
  for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host[i];
  a.rays_o = rays_o; a.rays_d = rays_d; a.times = times; a.time_value = time_value;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(observer_rays_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
