// Sample placement along rays and hierarchical resampling (sunerf/train/sampling.py).
// HBM-bound elementwise / per-ray kernels; compiled with -ffp-contract=off so that every product and sum is
// rounded separately, exactly like the reference's chain of aten ops.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

// ---- StratifiedSampler.forward sampling.py:68-98 / SphericalSampler.forward sampling.py:16-49 ----------------
// one thread per (ray, sample); the per-ray quadratic is recomputed per thread (a dozen flops) so that the
// z_vals store is fully coalesced.
__global__ void sample_z_kernel(int kind, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                const float* __restrict__ t_vals, const float* __restrict__ t_rand, int64_t n_rays,
                                int S, float distance, float solar_R, float* __restrict__ z_vals) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * S) return;
  const int64_t ray = idx / S;
  const int i = (int)(idx - ray * S);
  const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
  const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
  const float oo = (ox * ox + oy * oy) + oz * oz;              // rays_o.pow(2).sum(-1)
  const float a = (dx * dx + dy * dy) + dz * dz;               // rays_d.pow(2).sum(-1)
  const float b = ((2.f * ox) * dx + (2.f * oy) * dy) + (2.f * oz) * dz;   // (2*o*d).sum(-1)
  const float c = oo - solar_R * solar_R;
  const float dist_inner = (-b - sqrtf(b * b - (4.f * a) * c)) / (2.f * a);
  float dist_near, dist_far;
  if (kind == SUNERF_SAMPLER_STRATIFIED) {
    const float dist_o = sqrtf(oo);
    dist_near = dist_o - distance;
    dist_far = dist_o + distance;
  } else {
    const float c2 = oo - distance * distance;
    const float disc = sqrtf(b * b - (4.f * a) * c2);
    dist_near = (-b - disc) / (2.f * a);
    dist_far = (-b + disc) / (2.f * a);
  }
  if (!(dist_inner != dist_inner)) dist_far = dist_inner;      // ~isnan(dist_inner): stop at the solar surface
  auto zval = [&](int k) {
    const float t = t_vals[k];
    return dist_near * (1.f - t) + dist_far * t;
  };
  float z = zval(i);
  if (t_rand) {  // sampling.py:93-98 in-bin jitter
    const float upper = (i + 1 < S) ? .5f * (zval(i + 1) + z) : z;
    const float lower = (i > 0) ? .5f * (z + zval(i - 1)) : z;
    z = lower + (upper - lower) * t_rand[idx];
  }
  z_vals[idx] = z;
}

// ---- HierarchicalSampler.forward / sample_pdf sampling.py:111-169 -------------------------------------------
// One thread per ray: the CDF is a sequential running sum (torch.cumsum semantics) and the inverse-CDF
// lookup is discontinuous in the CDF values, so it is kept sequential rather than re-associated by a parallel
// scan.  Per-thread scratch (cdf, bins) lives in LDS, laid out [element][thread] (conflict-free).
// Work per ray is O(S_c + S_f); the kernel moves (2*S_c + 2*S_f + S_c) * 4 bytes per ray.
constexpr int RS_THREADS = 64;

__global__ __launch_bounds__(RS_THREADS) void hier_resample_kernel(
    const float* __restrict__ z_vals, const float* __restrict__ weights, const float* __restrict__ u_in,
    int u_per_ray, int64_t n_rays, int Sc, int Sf, float* __restrict__ new_z_out, float* __restrict__ z_comb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int64_t ray = (int64_t)blockIdx.x * RS_THREADS + tid;
  if (ray >= n_rays) return;
  const int nb = Sc - 1;                 // number of bins (mid points) == cdf entries
  float* cdf = lds;                      // [nb][RS_THREADS]
  float* bins = lds + (size_t)nb * RS_THREADS;
  float* nz = bins + (size_t)nb * RS_THREADS;   // [Sf][RS_THREADS] new samples (for the merge)
  const float* z = z_vals + ray * Sc;
  const float* w = weights + ray * Sc;
  // pdf = (w[1:-1] + 1e-5) / sum(w[1:-1] + 1e-5)      (Sc-2 entries)
  // torch's CPU cumsum accumulates fp32 inputs in fp64 and rounds every output to fp32; its sum is a cascaded
  // (more-accurate-than-sequential) fp32 sum.  fp64 accumulators reproduce the former exactly and the latter to
  // the last bit in almost all cases (62 fp64 adds per ray: free on this HBM-bound kernel).
  double wsum_d = 0.0;
  for (int i = 1; i < Sc - 1; ++i) wsum_d += (double)(w[i] + 1e-5f);
  const float wsum = (float)wsum_d;
  double run = 0.0;
  cdf[0 * RS_THREADS + tid] = 0.f;
  for (int i = 1; i < Sc - 1; ++i) {
    run += (double)((w[i] + 1e-5f) / wsum);
    cdf[i * RS_THREADS + tid] = (float)run;
  }
  float zprev = z[0];
  bool z_sorted = true;
  for (int i = 0; i < nb; ++i) {
    const float zn = z[i + 1];
    bins[i * RS_THREADS + tid] = .5f * (zn + zprev);
    z_sorted = z_sorted && (zn >= zprev);
    zprev = zn;
  }
  // inverse CDF
  bool sorted = true;
  float last = -INFINITY;
  for (int j = 0; j < Sf; ++j) {
    const float u = u_per_ray ? u_in[ray * Sf + j] : u_in[j];
    // searchsorted(cdf, u, right=True): first index with cdf[idx] > u   (binary search, nb entries)
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid * RS_THREADS + tid] > u) hi = mid; else lo = mid + 1;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    const float c0 = cdf[below * RS_THREADS + tid], c1 = cdf[above * RS_THREADS + tid];
    const float b0 = bins[below * RS_THREADS + tid], b1 = bins[above * RS_THREADS + tid];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    const float s = b0 + t * (b1 - b0);
    new_z_out[ray * Sf + j] = s;
    nz[j * RS_THREADS + tid] = s;
    sorted = sorted && (s >= last);
    last = s;
  }
  // z_vals_combined = sort(cat[z_vals, new_z])  (sampling.py:122)
  float* out = z_comb + ray * (Sc + Sf);
  if (sorted && z_sorted) {  // the normal case: merge of two ascending runs
    int i = 0, j = 0;
    for (int k = 0; k < Sc + Sf; ++k) {
      const float a = (i < Sc) ? z[i] : INFINITY;
      const float b = (j < Sf) ? nz[j * RS_THREADS + tid] : INFINITY;
      if (j >= Sf || (i < Sc && a <= b)) { out[k] = a; ++i; } else { out[k] = b; ++j; }
    }
  } else {  // perturb=True (random u) or non-ascending coarse z (|d| far from 1): general insertion sort, rare path
    for (int k = 0; k < Sc + Sf; ++k) {
      const float v = (k < Sc) ? z[k] : nz[(k - Sc) * RS_THREADS + tid];
      int m = k - 1;
      while (m >= 0 && out[m] > v) { out[m + 1] = out[m]; --m; }
      out[m + 1] = v;
    }
  }
}

}  // namespace

extern "C" int sunerf_sample_z(int sampler_kind, const float* rays_o, const float* rays_d, const float* t_vals,
                               const float* t_rand, int64_t n_rays, int n_samples, float distance, float solar_R,
                               float* z_vals, void* stream) {
  if (n_rays < 0 || n_samples < 1) return SUNERF_E_BADARG;
  if (sampler_kind != SUNERF_SAMPLER_STRATIFIED && sampler_kind != SUNERF_SAMPLER_SPHERICAL) return SUNERF_E_UNSUPPORTED;
  if (n_rays == 0) return 0;      // an empty batch is valid (its tensors have null data pointers)
  if (!rays_o || !rays_d || !t_vals || !z_vals) return SUNERF_E_BADARG;
  const int64_t total = n_rays * n_samples;
  const int threads = 256;
  const int64_t blocks = (total + threads - 1) / threads;
  if (blocks > 0x7fffffffLL) return SUNERF_E_UNSUPPORTED;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(sample_z_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, sampler_kind,
                     rays_o, rays_d, t_vals, t_rand, n_rays, n_samples, distance, solar_R, z_vals);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_hier_resample(const float* z_vals, const float* weights, const float* u, int u_per_ray,
                                    int64_t n_rays, int n_coarse, int n_fine, float* new_z, float* z_comb,
                                    void* stream) {
  if (n_rays < 0 || n_coarse < 3 || n_fine < 1) return SUNERF_E_BADARG;
  if (n_rays == 0) return 0;
  if (!z_vals || !weights || !u || !new_z || !z_comb) return SUNERF_E_BADARG;
  const size_t lds = ((size_t)2 * (n_coarse - 1) + n_fine) * RS_THREADS * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  if (n_rays == 0) return 0;
  const int64_t blocks = (n_rays + RS_THREADS - 1) / RS_THREADS;
  if (blocks > 0x7fffffffLL) return SUNERF_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)hier_resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(hier_resample_kernel, dim3((unsigned)blocks), dim3(RS_THREADS), lds, (hipStream_t)stream, z_vals,
                     weights, u, u_per_ray, n_rays, n_coarse, n_fine, new_z, z_comb);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
