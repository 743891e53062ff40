// Density / temperature radiative-transfer integral (run_density_temperature.py path), forward and backward.
//
// Replaces DensityTemperatureRadiativeTransfer.raw2outputs / regularization, sunerf/rendering/density_temperature.py:192-274,
// the "+ base" of NeRF_DT.forward (sunerf/model/model.py:181-185) and the epilogues of base_tracing.py:99-110 for the DT
// subclass.  The MLP itself runs in the fused render kernel (render_fwd.hip) whose raw (N,S,2) output is the input here.
//
//   rho = exp(relu(raw0 + base_rho));  logT = relu(raw1 + base_T);  kappa_w = relu(log_abs[w])
//   A_j   = cumulative_trapezoid(rho kappa_w, z)_j                         j = 0..S-2   (density_temperature.py:261)
//   I_w   = trapezoid(exp(-A_j) rho_j^2 R_w(logT_j), z_j; j = 0..S-2) * vol_c * pixel_intensity_factor   (:263-265)
//   weights = relu(inf0) / (sum + 1e-10);  regularizing quantity = relu(inf0)
// R_w = linear interpolation of the AIA temperature response (x exposure time), 0 outside the table (Interp1D(...,
// extrap=0), restated from its documented semantics: parity unpinned for that sub-step) and 0 for an absent channel
// (wavelength entry <= 0).  The reference's per-wavelength Python loop with two host syncs per channel
// (density_temperature.py:245-256) becomes one launch.
//
// Layout: 8 lanes per ray (lane w < 7: channel w of the row; lane 7: weights / maps), one thread walks the S samples.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int DT_THREADS = 64;      // 8 rays per block
constexpr int NCH = 7;

struct DtArgs {
  const float* raw;          // (N,S,2) MLP output
  const float* z_vals;       // (N,S)
  const float* rays_o;       // (N,3)
  const float* rays_d;
  const float* wavelengths;  // (N,W) channel wavelength in Angstrom, <= 0: absent
  const float* table_logt;   // (7,101) fp32
  const float* table_resp;   // (7,101) fp32, response x exposure time
  const float* log_abs;      // (7,) in the order 94,131,171,193,211,304,335
  const float* vol_c;        // (1,)
  float base_rho, base_t, pixel_factor, reg_radius;
  int64_t n_rays;
  int S, W;
  // forward outputs
  float* image;              // (N,W)
  float* weights;            // (N,S)
  float* reg_q;              // (N,S)  relu(inf0)
  float* height_map;         // (N,) or null
  float* absorption_map;     // (N,) or null
  float* regularization;     // (N,S) or null
  // backward
  const float* g_image;      // (N,W)
  const float* g_reg;        // (N,S) or null
  float* g_raw;              // (N,S,2)
  float* g_log_abs;          // (7,) accumulated
  float* g_vol_c;            // (1,) accumulated
  unsigned* g_absmax_bits;
};

__device__ __forceinline__ int channel_of(float wl) {
  const float w[NCH] = {94.f, 131.f, 171.f, 193.f, 211.f, 304.f, 335.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) if (wl == w[c]) return c;
  return -1;
}

// linear interpolation on the 101-point grid; returns value and slope (both 0 outside [x0, x100])
__device__ __forceinline__ void response(const float* lt, const float* rs, float x, float& val, float& slope) {
  val = 0.f; slope = 0.f;
  if (!(x >= lt[0] && x <= lt[100])) return;
  int i = (int)((x - lt[0]) * 20.f);              // grid step 0.05
  i = max(0, min(99, i));
  while (i < 99 && lt[i + 1] <= x) ++i;            // searchsorted(right=True) - 1, clamped to the last interval
  while (i > 0 && lt[i] > x) --i;
  const float x0 = lt[i], x1 = lt[i + 1], y0 = rs[i], y1 = rs[i + 1];
  slope = (y1 - y0) / (x1 - x0);
  val = y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

__global__ __launch_bounds__(DT_THREADS) void dt_integral_fwd_kernel(DtArgs a) {
  const int tid = threadIdx.x, sub = tid & 7;
  const int64_t ray = (int64_t)blockIdx.x * 8 + (tid >> 3);
  if (ray >= a.n_rays) return;
  const int S = a.S;
  const float* z = a.z_vals + ray * S;
  const float* r = a.raw + ray * S * 2;
  if (sub < NCH) {
    if (sub >= a.W) return;
    const int ch = channel_of(a.wavelengths[ray * a.W + sub]);
    const float kappa = ch >= 0 ? fmaxf(a.log_abs[ch], 0.f) : 0.f;
    const float* lt = a.table_logt + (ch >= 0 ? ch : 0) * 101;
    const float* rs = a.table_resp + (ch >= 0 ? ch : 0) * 101;
    float A = 0.f, ab_prev = 0.f, e_prev = 0.f, term_prev = 0.f, z_prev = 0.f, z_prev2 = 0.f, trap = 0.f;
    for (int i = 0; i < S; ++i) {
      const float rho = expf(fmaxf(r[2 * i] + a.base_rho, 0.f));
      const float logt = fmaxf(r[2 * i + 1] + a.base_t, 0.f);
      float R = 0.f, dR;
      if (ch >= 0) response(lt, rs, logt, R, dR);
      const float ab = rho * kappa, e = rho * rho * R, zi = z[i];
      if (i >= 1) {
        A += (ab + ab_prev) * (zi - z_prev) / 2.f;          // cumulative_trapezoid: ((y1 + y0) * dx) / 2, running sum
        const float term = expf(-A) * e_prev;               // term_{i-1} = exp(-A_{i-1}) * emission_{i-1}
        if (i >= 2) trap += (term + term_prev) * (z_prev - z_prev2);
        term_prev = term;
      }
      ab_prev = ab; e_prev = e; z_prev2 = z_prev; z_prev = zi;
    }
    a.image[ray * a.W + sub] = trap / 2.f * a.vol_c[0] * a.pixel_factor;   // trapezoid: sum((y1 + y0) * dx) / 2
  } else {
    // lane 7: weights = relu(inf0) / (sum + 1e-10), maps, regularization (density_temperature.py:268-274)
    const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    float sum = 0.f;
    for (int i = 0; i < S; ++i) sum += fmaxf(r[2 * i] + a.base_rho, 0.f);
    const float denom = sum + 1e-10f;
    float hm = 0.f, am = 0.f;
    for (int i = 0; i < S; ++i) {
      const float q = fmaxf(r[2 * i] + a.base_rho, 0.f);
      const float w = q / denom;
      a.weights[ray * S + i] = w;
      a.reg_q[ray * S + i] = q;
      if (a.regularization || a.height_map) {
        const float zi = z[i];
        const float px = ox + dx * zi, py = oy + dy * zi, pz = oz + dz * zi;
        const float pd = sqrtf((px * px + py * py) + pz * pz);
        hm += w * pd;
        if (a.regularization) a.regularization[ray * S + i] = fmaxf(pd - a.reg_radius, 0.f) * fmaxf(q, 0.f);
      }
      am += 1.f - q;
    }
    if (a.height_map) a.height_map[ray] = hm;
    if (a.absorption_map) a.absorption_map[ray] = am;
  }
}

// backward: same 8-lanes-per-ray layout; lane w keeps exp(-A_j) of its channel in LDS ([S][64]), walks the samples in
// reverse accumulating the suffix sums of d loss / d A_j; the per-sample raw gradients of the 8 lanes are combined with
// shuffles (no atomics); log_abs / vol_c gradients are block-reduced and added atomically (7 + 1 values per block).
__global__ __launch_bounds__(DT_THREADS) void dt_integral_bwd_kernel(DtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];    // [S][DT_THREADS] exp(-A_j)
  const int tid = threadIdx.x, sub = tid & 7;
  const int64_t ray_raw = (int64_t)blockIdx.x * 8 + (tid >> 3);
  const bool ray_ok = ray_raw < a.n_rays;
  const int64_t ray = ray_ok ? ray_raw : a.n_rays - 1;
  const int S = a.S;
  const float* z = a.z_vals + ray * S;
  const float* r = a.raw + ray * S * 2;
  const bool chan_lane = sub < NCH && sub < a.W && ray_ok;
  int ch = -1;
  float kappa = 0.f, g_I = 0.f;
  if (chan_lane) {
    ch = channel_of(a.wavelengths[ray * a.W + sub]);
    kappa = ch >= 0 ? fmaxf(a.log_abs[ch], 0.f) : 0.f;
    g_I = a.g_image[ray * a.W + sub];
  }
  const float* lt = a.table_logt + (ch >= 0 ? ch : 0) * 101;
  const float* rs = a.table_resp + (ch >= 0 ? ch : 0) * 101;
  const float C = a.vol_c[0] * a.pixel_factor;
  // ---- forward sweep: exp(-A_j) for j = 0..S-2, and the trapezoid value (for d/d vol_c) ----
  float trap = 0.f;
  if (chan_lane) {
    float A = 0.f, ab_prev = 0.f, e_prev = 0.f, term_prev = 0.f, z_prev = 0.f, z_prev2 = 0.f;
    for (int i = 0; i < S; ++i) {
      const float rho = expf(fmaxf(r[2 * i] + a.base_rho, 0.f));
      const float logt = fmaxf(r[2 * i + 1] + a.base_t, 0.f);
      float R = 0.f, dR;
      if (ch >= 0) response(lt, rs, logt, R, dR);
      const float ab = rho * kappa, e = rho * rho * R, zi = z[i];
      if (i >= 1) {
        A += (ab + ab_prev) * (zi - z_prev) / 2.f;
        const float ea = expf(-A);
        lds[(i - 1) * DT_THREADS + tid] = ea;
        const float term = ea * e_prev;
        if (i >= 2) trap += (term + term_prev) * (z_prev - z_prev2);
        term_prev = term;
      }
      ab_prev = ab; e_prev = e; z_prev2 = z_prev; z_prev = zi;
    }
    trap *= 0.5f;
  }
  const float g_trap = g_I * C;
  float g_kappa = 0.f;
  // geometry for the regularization gradient (lane 7)
  const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  // ---- reverse sweep over k = S-1 .. 0 ----
  float G = 0.f;            // G_k = sum_{j >= k} g_A_j  (G_{S-1} = 0)
  float local_max = 0.f;
  for (int k = S - 1; k >= 0; --k) {
    const float raw0 = r[2 * k], raw1 = r[2 * k + 1];
    const float inf0 = raw0 + a.base_rho, inf1 = raw1 + a.base_t;
    float g0 = 0.f, g1 = 0.f;
    if (chan_lane) {
      const float zk = z[k];
      // trapezoid weight of term_j on the grid z[0..S-2]:  wt_j = (dz_{j-1} + dz_j) / 2 with missing neighbours dropped
      auto wt = [&](int j) {
        float w = 0.f;
        if (j >= 1) w += z[j] - z[j - 1];
        if (j <= S - 3) w += z[j + 1] - z[j];
        return 0.5f * w;
      };
      auto point = [&](int j, float& rho, float& R, float& dR) {
        rho = expf(fmaxf(r[2 * j] + a.base_rho, 0.f));
        R = 0.f; dR = 0.f;
        if (ch >= 0) response(lt, rs, fmaxf(r[2 * j + 1] + a.base_t, 0.f), R, dR);
      };
      float rho, R, dR;
      point(k, rho, R, dR);
      // G_{k-1} = G_k + g_A_{k-1},  g_A_j = -g_term_j * term_j
      float G_km1 = G;
      if (k >= 1) {
        float rho1, R1, dR1;
        point(k - 1, rho1, R1, dR1);
        const float term = lds[(k - 1) * DT_THREADS + tid] * rho1 * rho1 * R1;
        G_km1 = G - g_trap * wt(k - 1) * term;
      }
      float g_ab = 0.f;
      if (k <= S - 2) g_ab += 0.5f * (z[k + 1] - zk) * G;
      if (k >= 1) g_ab += 0.5f * (zk - z[k - 1]) * G_km1;
      float g_e = 0.f;
      if (k <= S - 2) g_e = g_trap * wt(k) * lds[k * DT_THREADS + tid];
      const float g_rho = g_ab * kappa + g_e * 2.f * rho * R;
      g_kappa += g_ab * rho;
      if (inf0 > 0.f) g0 = g_rho * rho;
      if (inf1 > 0.f) g1 = g_e * rho * rho * dR;
      G = G_km1;
    } else if (sub == 7 && ray_ok) {
      // regularization_k = relu(|p_k| - R) * relu(relu(inf0))
      const float gr = a.g_reg ? a.g_reg[ray * S + k] : 0.f;
      if (gr != 0.f && inf0 > 0.f) {
        const float zk = z[k];
        const float px = ox + dx * zk, py = oy + dy * zk, pz = oz + dz * zk;
        g0 = gr * fmaxf(sqrtf((px * px + py * py) + pz * pz) - a.reg_radius, 0.f);
      }
    }
    // combine the 8 lanes of the ray
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) { g0 += __shfl_xor(g0, d, 8); g1 += __shfl_xor(g1, d, 8); }
    if (sub == 0 && ray_ok) {
      a.g_raw[(ray * S + k) * 2 + 0] = g0;
      a.g_raw[(ray * S + k) * 2 + 1] = g1;
      local_max = fmaxf(local_max, fmaxf(fabsf(g0), fabsf(g1)));
    }
  }
  // ---- parameter gradients: log_abs (through kappa = relu(log_abs)) and vol_c ----
  if (chan_lane && ch >= 0) {
    if (a.log_abs[ch] > 0.f && g_kappa != 0.f) atomicAdd(a.g_log_abs + ch, g_kappa);
    const float gv = g_I * trap * a.pixel_factor;
    if (gv != 0.f) atomicAdd(a.g_vol_c, gv);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, d));
  if (tid == 0 && local_max > 0.f && local_max < INFINITY) atomicMax(a.g_absmax_bits, __float_as_uint(local_max));
}

int check_common(const DtArgs& a) {
  if (a.n_rays == 0 && a.S >= 3 && a.W >= 1 && a.W <= NCH) return 0;
  if (!a.raw || !a.z_vals || !a.rays_o || !a.rays_d || !a.wavelengths || !a.table_logt || !a.table_resp || !a.log_abs || !a.vol_c)
    return SUNERF_E_BADARG;
  if (a.n_rays < 0 || a.S < 3) return SUNERF_E_BADARG;
  if (a.W < 1 || a.W > NCH) return SUNERF_E_UNSUPPORTED;
  return 0;
}

}  // namespace

namespace {

// SimpleStar.forward, sunerf/model/stellar_model.py:53-102 (Pascoe et al. 2019 eq. 4 and 6), at the sample points
// o + d z of sampling.py:100: the analytic density / temperature field that stands in for a trained NeRF_DT when synthetic
// observations are rendered (evaluation/image_render.py:236-268).  raw = (ln rho, log10 T).
struct StarArgs {
  const float* rays_o; const float* rays_d; const float* z_vals;
  int64_t n_rays; int S;
  float rho_0, h0, T0, Rs, t_photosphere;
  float* raw;
};

__global__ __launch_bounds__(256) void simple_star_kernel(StarArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.n_rays * a.S) return;
  const int64_t ray = idx / a.S;
  const float z = a.z_vals[idx];
  const float x = a.rays_o[ray * 3 + 0] + a.rays_d[ray * 3 + 0] * z;
  const float y = a.rays_o[ray * 3 + 1] + a.rays_d[ray * 3 + 1] * z;
  const float w = a.rays_o[ray * 3 + 2] + a.rays_d[ray * 3 + 2] * z;
  const float radius = sqrtf((x * x + y * y) + w * w);
  float rho = a.rho_0, temp = a.t_photosphere;
  if (radius > 1.f) {
    rho = a.rho_0 * expf((1.f / a.h0) * (1.f / radius - 1.f));
    temp = radius <= a.Rs ? (radius - 1.f) * ((a.T0 - a.t_photosphere) / (a.Rs - 1.f)) + a.t_photosphere : a.T0;
  }
  // a NaN radius (missed-sphere rays of SphericalSampler) fails every comparison of the reference's masks and leaves the
  // zero-initialised rho / temp: log(0) = -inf
  if (!(radius == radius)) { rho = 0.f; temp = 0.f; }
  a.raw[idx * 2 + 0] = logf(rho);
  a.raw[idx * 2 + 1] = log10f(temp);
}

}  // namespace

extern "C" int sunerf_simple_star_field(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays,
                                        int n_samples, float rho_0, float h0, float T0, float Rs, float t_photosphere,
                                        float* raw, void* stream) {
  if (n_rays < 0 || n_samples < 1) return SUNERF_E_BADARG;
  if (n_rays == 0) return 0;
  if (!rays_o || !rays_d || !z_vals || !raw) return SUNERF_E_BADARG;
  StarArgs a;
  a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.n_rays = n_rays; a.S = n_samples;
  a.rho_0 = rho_0; a.h0 = h0; a.T0 = T0; a.Rs = Rs; a.t_photosphere = t_photosphere; a.raw = raw;
  const int64_t total = n_rays * n_samples;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(simple_star_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_dt_integral_fwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                                      const float* wavelengths, int n_wavelengths, const float* table_logt,
                                      const float* table_resp, const float* log_abs, const float* vol_c, float base_log_density,
                                      float base_log_temperature, float pixel_intensity_factor, float reg_radius,
                                      int64_t n_rays, int n_samples, float* image, float* weights, float* reg_q,
                                      float* height_map, float* absorption_map, float* regularization, void* stream) {
  DtArgs a = {};
  a.raw = raw; a.z_vals = z_vals; a.rays_o = rays_o; a.rays_d = rays_d; a.wavelengths = wavelengths; a.W = n_wavelengths;
  a.table_logt = table_logt; a.table_resp = table_resp; a.log_abs = log_abs; a.vol_c = vol_c; a.base_rho = base_log_density;
  a.base_t = base_log_temperature; a.pixel_factor = pixel_intensity_factor; a.reg_radius = reg_radius; a.n_rays = n_rays;
  a.S = n_samples; a.image = image; a.weights = weights; a.reg_q = reg_q; a.height_map = height_map;
  a.absorption_map = absorption_map; a.regularization = regularization;
  if (int rc = check_common(a)) return rc;
  if (n_rays == 0) return 0;
  if (!image || !weights || !reg_q) return SUNERF_E_BADARG;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(dt_integral_fwd_kernel, dim3((unsigned)((n_rays + 7) / 8)), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_dt_integral_bwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                                      const float* wavelengths, int n_wavelengths, const float* table_logt,
                                      const float* table_resp, const float* log_abs, const float* vol_c, float base_log_density,
                                      float base_log_temperature, float pixel_intensity_factor, float reg_radius,
                                      int64_t n_rays, int n_samples, const float* g_image, const float* g_reg, float* g_raw,
                                      float* g_log_abs, float* g_vol_c, void* g_absmax, void* stream) {
  DtArgs a = {};
  a.raw = raw; a.z_vals = z_vals; a.rays_o = rays_o; a.rays_d = rays_d; a.wavelengths = wavelengths; a.W = n_wavelengths;
  a.table_logt = table_logt; a.table_resp = table_resp; a.log_abs = log_abs; a.vol_c = vol_c; a.base_rho = base_log_density;
  a.base_t = base_log_temperature; a.pixel_factor = pixel_intensity_factor; a.reg_radius = reg_radius; a.n_rays = n_rays;
  a.S = n_samples; a.g_image = g_image; a.g_reg = g_reg; a.g_raw = g_raw; a.g_log_abs = g_log_abs; a.g_vol_c = g_vol_c;
  a.g_absmax_bits = (unsigned*)g_absmax;
  if (int rc = check_common(a)) return rc;
  if (!g_image || !g_raw || !g_log_abs || !g_vol_c || !g_absmax) return SUNERF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if ((e = hipMemsetAsync(g_absmax, 0, 4, st)) != hipSuccess) return (int)e;
  if ((e = hipMemsetAsync(g_log_abs, 0, NCH * sizeof(float), st)) != hipSuccess) return (int)e;
  if ((e = hipMemsetAsync(g_vol_c, 0, sizeof(float), st)) != hipSuccess) return (int)e;
  if (n_rays == 0) return 0;
  const size_t lds = (size_t)n_samples * DT_THREADS * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute((const void*)dt_integral_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(dt_integral_bwd_kernel, dim3((unsigned)((n_rays + 7) / 8)), dim3(DT_THREADS), lds, st, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
