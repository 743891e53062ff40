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
constexpr int RS_THREADS = 256;
constexpr int RS_G = 8;                         // lanes per ray
constexpr int RS_RAYS = RS_THREADS / RS_G;      // rays per workgroup

// 8 lanes per ray.  The CDF (a 62-step fp64 running sum whose rounding decides which side of the reference's thresholds a
// sample falls on) is built by the group's first lane exactly as a single thread would; the S_f inverse-CDF searches and
// the merge -- as rank computations: position of z_i = i + #{new < z_i}, position of new_j = j + #{z <= new_j}, the order a
// stable two-run merge produces -- are dealt out over the 8 lanes.  Lanes of a group sit in one wave, LDS operations of a
// wave complete in order, so no barrier separates the phases.
// BINS_GIVEN (HierarchicalSampler.sample_pdf called by itself, sampling.py:128-169): `z_vals` holds the nb = Sc - 1 bin
// positions of every ray and `weights` their nb - 1 weights as they are (no mid points, no [1:-1] slice, no merge).
template <bool BINS_GIVEN>
__global__ __launch_bounds__(RS_THREADS) void hier_resample_kernel(
    const float* __restrict__ z_vals, const float* __restrict__ weights, const float* __restrict__ u_in,
    int u_per_ray, int64_t n_rays, int Sc, int Sf, float* __restrict__ new_z_out, float* __restrict__ z_comb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, g = tid & (RS_G - 1), sub = tid / RS_G;
  const int64_t ray_raw = (int64_t)blockIdx.x * RS_RAYS + sub;
  const bool ray_ok = ray_raw < n_rays;
  const int64_t ray = ray_ok ? ray_raw : n_rays - 1;      // surplus groups shadow the last ray (no stores)
  const int nb = Sc - 1;                 // number of bins (mid points) == cdf entries
  float* cdf = lds + (size_t)sub * (2 * nb + Sf + Sc);    // [nb]
  float* bins = cdf + nb;                // [nb]
  float* nz = bins + nb;                 // [Sf] new samples
  float* zc = nz + Sf;                   // [Sc] coarse samples
  const float* z = z_vals + ray * (BINS_GIVEN ? nb : Sc);
  const float* w = BINS_GIVEN ? weights + ray * (nb - 1) - 1 : weights + ray * Sc;   // w[1 .. Sc-2] are the pdf's entries
  if (!BINS_GIVEN)
    for (int i = g; i < Sc; i += RS_G) zc[i] = z[i];
  bool z_sorted = true;
  if (g == 0) {
    // pdf = (w[1:-1] + 1e-5) / sum(w[1:-1] + 1e-5)      (Sc-2 entries)
    // torch's CPU cumsum accumulates fp32 inputs in fp64 and rounds every output to fp32; its sum is a cascaded
    // (more-accurate-than-sequential) fp32 sum.  fp64 accumulators reproduce the former exactly and the latter to
    // the last bit in almost all cases.
    double wsum_d = 0.0;
    for (int i = 1; i < Sc - 1; ++i) wsum_d += (double)(w[i] + 1e-5f);
    const float wsum = (float)wsum_d;
    double run = 0.0;
    cdf[0] = 0.f;
    for (int i = 1; i < Sc - 1; ++i) {
      run += (double)((w[i] + 1e-5f) / wsum);
      cdf[i] = (float)run;
    }
  }
  for (int i = g; i < nb; i += RS_G) {
    if (BINS_GIVEN) {
      bins[i] = z[i];
    } else {
      const float z0 = z[i], z1 = z[i + 1];
      bins[i] = .5f * (z1 + z0);
      z_sorted = z_sorted && (z1 >= z0);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // inverse CDF: sample j of this lane
  for (int j = g; j < Sf; j += RS_G) {
    const float u = u_per_ray ? u_in[ray * Sf + j] : u_in[j];
    // searchsorted(cdf, u, right=True): first index with cdf[idx] > u   (binary search, nb entries)
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bins[below], b1 = bins[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    const float sv = b0 + t * (b1 - b0);
    if (ray_ok) new_z_out[ray * Sf + j] = sv;
    nz[j] = sv;
  }
  if (BINS_GIVEN) return;
  __builtin_amdgcn_wave_barrier();
  bool sorted = z_sorted;
  for (int j = g; j < Sf; j += RS_G)
    if (j > 0) sorted = sorted && (nz[j] >= nz[j - 1]);
  // both runs ascending?  (and-reduce over the group's 8 lanes)
  int ok = sorted ? 1 : 0;
  ok &= __shfl_xor(ok, 1);
  ok &= __shfl_xor(ok, 2);
  ok &= __shfl_xor(ok, 4);
  if (!ray_ok) return;
  // z_vals_combined = sort(cat[z_vals, new_z])  (sampling.py:122)
  float* out = z_comb + ray * (Sc + Sf);
  if (ok) {  // the normal case: merge of two ascending runs, by rank
    for (int i = g; i < Sc; i += RS_G) {
      const float v = zc[i];
      int lo = 0, hi = Sf;                       // #{new < v}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nz[mid] < v) lo = mid + 1; else hi = mid;
      }
      out[i + lo] = v;
    }
    for (int j = g; j < Sf; j += RS_G) {
      const float v = nz[j];
      int lo = 0, hi = Sc;                       // #{z <= v}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (zc[mid] <= v) lo = mid + 1; else hi = mid;
      }
      out[j + lo] = v;
    }
  } else if (g == 0) {  // perturb=True (random u) or non-ascending coarse z (|d| far from 1): insertion sort, rare path
    for (int k = 0; k < Sc + Sf; ++k) {
      const float v = (k < Sc) ? zc[k] : nz[k - Sc];
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
  const size_t lds = ((size_t)2 * (n_coarse - 1) + n_fine + n_coarse) * RS_RAYS * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  if (n_rays == 0) return 0;
  const int64_t blocks = (n_rays + RS_RAYS - 1) / RS_RAYS;
  if (blocks > 0x7fffffffLL) return SUNERF_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)hier_resample_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(hier_resample_kernel<false>, dim3((unsigned)blocks), dim3(RS_THREADS), lds, (hipStream_t)stream, z_vals,
                     weights, u, u_per_ray, n_rays, n_coarse, n_fine, new_z, z_comb);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_sample_pdf(const float* bins, const float* weights, const float* u, int u_per_ray, int64_t n_rays,
                                 int n_bins, int n_fine, float* samples, void* stream) {
  if (n_rays < 0 || n_bins < 2 || n_fine < 1) return SUNERF_E_BADARG;
  if (n_rays == 0) return 0;
  if (!bins || !weights || !u || !samples) return SUNERF_E_BADARG;
  const int sc = n_bins + 1;             // the kernel's LDS layout is that of n_bins + 1 coarse samples
  const size_t lds = ((size_t)2 * n_bins + n_fine + sc) * RS_RAYS * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  const int64_t blocks = (n_rays + RS_RAYS - 1) / RS_RAYS;
  if (blocks > 0x7fffffffLL) return SUNERF_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)hier_resample_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(hier_resample_kernel<true>, dim3((unsigned)blocks), dim3(RS_THREADS), lds, (hipStream_t)stream, bins,
                     weights, u, u_per_ray, n_rays, sc, n_fine, samples, (float*)nullptr);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
