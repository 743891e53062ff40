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
// Layout: 32 lanes per ray, one sample per lane and 32-sample chunk (coalesced reads of raw / z, coalesced writes), the
// channels looped inside.  Everything sequential along the ray in the reference -- the running optical depth
// (cumulative_trapezoid), the trapezoid sum of the attenuated emission and, in the backward pass, the suffix sums of the
// optical-depth gradients -- is a chunk-wise scan over the 32 lanes with a scalar carry from chunk to chunk, like the
// emission integral (render_fwd.hip, render_bwd.hip).  (The first version walked the samples with ONE thread per ray and
// channel: 1.2 ms for the backward of 8192 rays x 256 samples x 7 channels, 10 % of a config-5 training step.)
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int DT_THREADS = 256;
constexpr int DT_RAYS = DT_THREADS / 32;      // rays per workgroup (and per step of its walk over the batch)
constexpr int DT_MAX_GRID = 1024;             // backward: workgroups of the grid
constexpr int NCH = 7;
constexpr int NTAB = NCH * 101;

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

__device__ __forceinline__ float scan_up32(float v, int n) {      // inclusive prefix sum over the 32 lanes of a ray
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float o = __shfl_up(v, d, 32);
    if (n >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float scan_down32(float v, int n) {    // inclusive suffix sum
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float o = __shfl_down(v, d, 32);
    if (n + d < 32) v += o;
  }
  return v;
}
__device__ __forceinline__ float sum32(float v) {
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 32);
  return v;
}

// channel set-up of one ray (lane-uniform): table rows, absorption coefficient
struct Channels {
  int ch[NCH];
  float kappa[NCH];
  __device__ __forceinline__ void init(const DtArgs& a, int64_t ray) {
#pragma unroll
    for (int w = 0; w < NCH; ++w) {
      ch[w] = w < a.W ? channel_of(a.wavelengths[ray * a.W + w]) : -1;
      kappa[w] = ch[w] >= 0 ? fmaxf(a.log_abs[ch[w]], 0.f) : 0.f;
    }
  }
};

// One forward sweep along the ray: per channel the optical depth A_i = cumulative_trapezoid(rho kappa, z) and the trapezoid
// sum of term_j = exp(-A_{j+1}) rho_j^2 R(logT_j), j = 0..S-2 (density_temperature.py:261-265).  `ea` (backward only):
// receives exp(-A_{j+1}) at [j * NCH + w].  Returns the trapezoid sums (x 2, as the reference's running sum) in trap[].
template <bool KEEP>
__device__ __forceinline__ void forward_sweep(const DtArgs& a, const float* tab, const Channels& C, const float* r, const float* z,
                                              int n, float* ea, float trap[NCH]) {
  const int S = a.S, n_chunks = (S + 31) >> 5;
  float A_c[NCH], ab_c[NCH], e_c[NCH], T_c[NCH];
#pragma unroll
  for (int w = 0; w < NCH; ++w) { A_c[w] = ab_c[w] = e_c[w] = T_c[w] = 0.f; trap[w] = 0.f; }
  float z_c1 = 0.f, z_c2 = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    const bool valid = i < S;
    const int ii = valid ? i : S - 1;
    const float zi = z[ii];
    const f32x2 rr = *(const f32x2*)(r + 2 * ii);
    float zp = __shfl_up(zi, 1, 32);
    if (n == 0) zp = z_c1;
    float zpp = __shfl_up(zp, 1, 32);
    if (n == 0) zpp = z_c2;
    const float rho = expf(fmaxf(rr[0] + a.base_rho, 0.f));
    const float logt = fmaxf(rr[1] + a.base_t, 0.f);
#pragma unroll
    for (int w = 0; w < NCH; ++w) {
      if (w >= a.W) break;
      float R = 0.f, dR;
      if (C.ch[w] >= 0) response(tab + C.ch[w] * 101, tab + NTAB + C.ch[w] * 101, logt, R, dR);
      const float ab = rho * C.kappa[w], e = rho * rho * R;
      float abp = __shfl_up(ab, 1, 32), ep = __shfl_up(e, 1, 32);
      if (n == 0) { abp = ab_c[w]; ep = e_c[w]; }
      const float inc = (valid && i >= 1) ? (ab + abp) * (zi - zp) / 2.f : 0.f;     // ((y1 + y0) * dx) / 2
      const float A = A_c[w] + scan_up32(inc, n);
      const float ex = expf(-A);
      if (KEEP && valid && i >= 1) ea[(size_t)(i - 1) * NCH + w] = ex;
      const float T = (valid && i >= 1) ? ex * ep : 0.f;                            // term_{i-1}
      float Tp = __shfl_up(T, 1, 32);
      if (n == 0) Tp = T_c[w];
      if (valid && i >= 2) trap[w] += (T + Tp) * (zp - zpp);
      A_c[w] = __shfl(A, 31, 32); ab_c[w] = __shfl(ab, 31, 32); e_c[w] = __shfl(e, 31, 32); T_c[w] = __shfl(T, 31, 32);
    }
    z_c2 = __shfl(zp, 31, 32);
    z_c1 = __shfl(zi, 31, 32);
  }
#pragma unroll
  for (int w = 0; w < NCH; ++w) trap[w] = sum32(trap[w]);
}

__global__ __launch_bounds__(DT_THREADS) void dt_integral_fwd_kernel(DtArgs a) {
  __shared__ float tab[2 * NTAB];
  const int tid = threadIdx.x, n = tid & 31, sub = tid >> 5;
  for (int i = tid; i < NTAB; i += DT_THREADS) { tab[i] = a.table_logt[i]; tab[NTAB + i] = a.table_resp[i]; }
  __syncthreads();
  const int64_t ray = (int64_t)blockIdx.x * DT_RAYS + sub;
  if (ray >= a.n_rays) return;                     // (a whole 32-lane group leaves: the shuffles are 32 wide)
  const int S = a.S, n_chunks = (S + 31) >> 5;
  const float* z = a.z_vals + ray * S;
  const float* r = a.raw + ray * S * 2;
  Channels C;
  C.init(a, ray);
  float trap[NCH];
  forward_sweep<false>(a, tab, C, r, z, n, nullptr, trap);
  if (n < a.W) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NCH; ++w) if (w == n) t = trap[w];
    a.image[ray * a.W + n] = t / 2.f * a.vol_c[0] * a.pixel_factor;              // trapezoid: sum((y1 + y0) * dx) / 2
  }
  // weights = relu(inf0) / (sum + 1e-10), maps, regularization (density_temperature.py:268-274)
  const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  float sum = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    if (i < S) sum += fmaxf(r[2 * i] + a.base_rho, 0.f);
  }
  const float denom = sum32(sum) + 1e-10f;
  float hm = 0.f, am = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    if (i >= S) continue;
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
  hm = sum32(hm); am = sum32(am);
  if (n == 0) {
    if (a.height_map) a.height_map[ray] = hm;
    if (a.absorption_map) a.absorption_map[ray] = am;
  }
}

// backward: a forward sweep keeps exp(-A_{j+1}) of every channel in LDS ([ray][j][channel]); the reverse sweep forms the
// suffix sums G_k = sum_{j >= k} dL/dA_j chunk by chunk (descending) and from them the gradients of the two raw outputs of
// every sample, summed over the channels in the lane (no atomics).  log_abs / vol_c gradients: per-ray lane sums ->
// workgroup sums in LDS -> 8 atomic adds per workgroup.
__global__ __launch_bounds__(DT_THREADS) void dt_integral_bwd_kernel(DtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];    // tables | g_kappa[7] g_vol | [DT_RAYS][S][NCH] exp(-A)
  float* tab = lds;
  float* acc8 = lds + 2 * NTAB;
  const int tid = threadIdx.x, n = tid & 31, sub = tid >> 5;
  for (int i = tid; i < NTAB; i += DT_THREADS) { tab[i] = a.table_logt[i]; tab[NTAB + i] = a.table_resp[i]; }
  if (tid < 8) acc8[tid] = 0.f;
  __syncthreads();
  float local_max = 0.f;
  // a workgroup walks over groups of DT_RAYS rays (grid <= DT_MAX_GRID) and sends its sums ONCE: 8 atomic adds and one atomic
  // max per workgroup instead of per ray group / per wave
  const int64_t n_groups = (a.n_rays + DT_RAYS - 1) / DT_RAYS;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
  const int64_t ray_raw = grp * DT_RAYS + sub;
  const bool ray_ok = ray_raw < a.n_rays;
  const int64_t ray = ray_ok ? ray_raw : a.n_rays - 1;
  const int S = a.S, n_chunks = (S + 31) >> 5;
  float* ea = lds + 2 * NTAB + 8 + (size_t)sub * S * NCH;
  const float* z = a.z_vals + ray * S;
  const float* r = a.raw + ray * S * 2;
  Channels C;
  C.init(a, ray);
  float trap[NCH];
  forward_sweep<true>(a, tab, C, r, z, n, ea, trap);
  float g_trap[NCH], g_kappa[NCH], G_c[NCH];
  const float Cf = a.vol_c[0] * a.pixel_factor;
#pragma unroll
  for (int w = 0; w < NCH; ++w) {
    g_trap[w] = (ray_ok && w < a.W) ? a.g_image[ray * a.W + w] * Cf : 0.f;
    g_kappa[w] = 0.f;
    G_c[w] = 0.f;
  }
  const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  // trapezoid weight of term_j on the grid z[0..S-2]:  wt_j = (dz_{j-1} + dz_j) / 2 with missing neighbours dropped
  auto wt = [&](int j, float zm, float z0, float zp1) {
    float w = 0.f;
    if (j >= 1) w += z0 - zm;
    if (j <= S - 3) w += zp1 - z0;
    return 0.5f * w;
  };
  for (int c = n_chunks - 1; c >= 0; --c) {
    const int k = 32 * c + n;
    const bool valid = k < S;
    const int kk = valid ? k : S - 1;
    const float zk = z[kk], zm = z[kk >= 1 ? kk - 1 : 0], zp1 = z[kk + 1 < S ? kk + 1 : S - 1];
    const f32x2 rr = *(const f32x2*)(r + 2 * kk);
    const float inf0 = rr[0] + a.base_rho, inf1 = rr[1] + a.base_t;
    const float rho = expf(fmaxf(inf0, 0.f)), logt = fmaxf(inf1, 0.f);
    const float wk = wt(kk, zm, zk, zp1);
    // lane 0 also needs dL/dA of the sample below its chunk (k - 1): its term is recomputed here
    const bool below = n == 0 && k >= 1;
    float rho_b = 0.f, logt_b = 0.f, w_b = 0.f;
    if (below) {
      const f32x2 rb = *(const f32x2*)(r + 2 * (k - 1));
      rho_b = expf(fmaxf(rb[0] + a.base_rho, 0.f));
      logt_b = fmaxf(rb[1] + a.base_t, 0.f);
      w_b = wt(k - 1, z[k >= 2 ? k - 2 : 0], zm, zk);
    }
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int w = 0; w < NCH; ++w) {
      if (w >= a.W) break;
      float R = 0.f, dR = 0.f;
      if (C.ch[w] >= 0) response(tab + C.ch[w] * 101, tab + NTAB + C.ch[w] * 101, logt, R, dR);
      const bool has_term = valid && k <= S - 2;
      const float eak = has_term ? ea[(size_t)k * NCH + w] : 0.f;
      const float gA = has_term ? -g_trap[w] * wk * (eak * rho * rho * R) : 0.f;       // dL/dA_k = -g_term_k * term_k
      const float G = G_c[w] + scan_down32(gA, n);                                      // G_k = sum_{j >= k} dL/dA_j
      float gA_b = __shfl_up(gA, 1, 32);                                                 // dL/dA_{k-1}
      if (n == 0) {
        gA_b = 0.f;
        if (below) {
          float Rb = 0.f, dRb;
          if (C.ch[w] >= 0) response(tab + C.ch[w] * 101, tab + NTAB + C.ch[w] * 101, logt_b, Rb, dRb);
          gA_b = -g_trap[w] * w_b * (ea[(size_t)(k - 1) * NCH + w] * rho_b * rho_b * Rb);
        }
      }
      if (valid) {
        float g_ab = 0.f;
        if (k <= S - 2) g_ab += 0.5f * (zp1 - zk) * G;
        if (k >= 1) g_ab += 0.5f * (zk - zm) * (G + gA_b);                               // G_{k-1}
        const float g_e = has_term ? g_trap[w] * wk * eak : 0.f;
        const float g_rho = g_ab * C.kappa[w] + g_e * 2.f * rho * R;
        g_kappa[w] += g_ab * rho;
        if (inf0 > 0.f) g0 += g_rho * rho;
        if (inf1 > 0.f) g1 += g_e * rho * rho * dR;
      }
      G_c[w] = __shfl(G, 0, 32);
    }
    if (valid && ray_ok) {
      // regularization_k = relu(|p_k| - R) * relu(relu(inf0))
      const float gr = a.g_reg ? a.g_reg[ray * S + k] : 0.f;
      if (gr != 0.f && inf0 > 0.f) {
        const float px = ox + dx * zk, py = oy + dy * zk, pz = oz + dz * zk;
        g0 += gr * fmaxf(sqrtf((px * px + py * py) + pz * pz) - a.reg_radius, 0.f);
      }
      const f32x2 gg = {g0, g1};
      *(f32x2*)(a.g_raw + ((size_t)ray * S + k) * 2) = gg;
      local_max = fmaxf(local_max, fmaxf(fabsf(g0), fabsf(g1)));
    }
  }
  // ---- parameter gradients: log_abs (through kappa = relu(log_abs)) and vol_c ----
  float g_vol = 0.f;
#pragma unroll
  for (int w = 0; w < NCH; ++w) {
    const float gk = sum32(g_kappa[w]);
    if (n == 0 && ray_ok && w < a.W && C.ch[w] >= 0) {
      if (a.log_abs[C.ch[w]] > 0.f && gk != 0.f) atomicAdd(acc8 + C.ch[w], gk);
      g_vol += a.g_image[ray * a.W + w] * (0.5f * trap[w]) * a.pixel_factor;
    }
  }
  if (n == 0 && g_vol != 0.f) atomicAdd(acc8 + 7, g_vol);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, d));
  __shared__ float wave_max[DT_THREADS / 64];
  if ((tid & 63) == 0) wave_max[tid >> 6] = local_max;
  __syncthreads();
  if (tid < NCH && acc8[tid] != 0.f) atomicAdd(a.g_log_abs + tid, acc8[tid]);
  if (tid == 7 && acc8[7] != 0.f) atomicAdd(a.g_vol_c, acc8[7]);
  if (tid == 0) {
    float m = wave_max[0];
#pragma unroll
    for (int w = 1; w < DT_THREADS / 64; ++w) m = fmaxf(m, wave_max[w]);
    if (m > 0.f && m < INFINITY) atomicMax(a.g_absmax_bits, __float_as_uint(m));
  }
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
  hipLaunchKernelGGL(dt_integral_fwd_kernel, dim3((unsigned)((n_rays + DT_RAYS - 1) / DT_RAYS)), dim3(DT_THREADS), 0,
                     (hipStream_t)stream, a);
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
  if ((char*)g_vol_c == (char*)g_log_abs + NCH * sizeof(float) && (char*)g_absmax == (char*)g_vol_c + sizeof(float)) {
    // the three small outputs in one buffer (what the Python wrapper passes): one clear instead of three
    if ((e = hipMemsetAsync(g_log_abs, 0, (NCH + 2) * sizeof(float), st)) != hipSuccess) return (int)e;
  } else {
    if ((e = hipMemsetAsync(g_absmax, 0, 4, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(g_log_abs, 0, NCH * sizeof(float), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(g_vol_c, 0, sizeof(float), st)) != hipSuccess) return (int)e;
  }
  if (n_rays == 0) return 0;
  const size_t lds = ((size_t)2 * NTAB + 8 + (size_t)DT_RAYS * n_samples * NCH) * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute((const void*)dt_integral_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  SUNERF_CLEAR_ERROR();
  int64_t groups = (n_rays + DT_RAYS - 1) / DT_RAYS;
  if (groups > DT_MAX_GRID) groups = DT_MAX_GRID;
  hipLaunchKernelGGL(dt_integral_bwd_kernel, dim3((unsigned)groups), dim3(DT_THREADS), lds, st, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
