// Output side of the render path: the training loss of the reference's Lightning modules and
// the clip + Adam update, as three small HBM-bound kernels that never synchronise with the host.
//
//   loss_kernel      sunerf/model/sunerf.py:105-125 (+ sunerf/train/scaling.py:17-28): finite check of the outputs,
//                    asinh image scaling, MSE(coarse) + MSE(fine), regularization.mean(), PSNR, and d loss / d image
//   grad_norm_kernel torch.nn.utils.clip_grad_norm_ (what Lightning's gradient_clip_val = 0.5 calls, run_emission.py:72)
//   adam_kernel      torch.optim.Adam single-tensor update (sunerf.py:31), operation order of torch/optim/adam.py
//
// All reductions are two-stage and ordered (block partials in fp64 -> the last block sums them in index order), so the
// results do not depend on scheduling.  ~4 bytes of HBM traffic per element and pass; the kernels are launched with a
// fixed grid of at most TS_BLOCKS workgroups and stride over their input.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int TS_THREADS = 256;
constexpr int TS_BLOCKS = 128;
constexpr int TS_MAX_EXTRA = 8;

// workspace: [0] ticket (unsigned), then from byte 64: TS_BLOCKS x 4 doubles of block partials
struct Workspace {
  unsigned* ticket;
  double* partial;
  __host__ __device__ explicit Workspace(void* p) : ticket((unsigned*)p), partial((double*)((char*)p + 64)) {}
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();   // sh may still be read by a previous call
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < TS_THREADS / 64; ++w) t += sh[w];
  return t;
}

// true for exactly one workgroup of the grid: the one that finishes last (all partials are then visible to it)
__device__ __forceinline__ bool last_block(unsigned* ticket) {
  __shared__ unsigned last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (last) __threadfence();
  return last != 0;
}

// Strided sweep of a float array with two 16-byte loads in flight per thread (a scalar grid-stride loop leaves every
// thread with one 4-byte load per memory round trip: ~1 us each, 20 us per launch at training batch sizes).
// `f` sees every element exactly once; the visiting order is fixed by (n, grid), so sums stay reproducible.
typedef float ts_f4 __attribute__((ext_vector_type(4)));
template <class F>
__device__ __forceinline__ void sweep(const float* p, int64_t n, int64_t tid, int64_t stride, F f) {
  if ((((uintptr_t)p) & 15) != 0) {
    for (int64_t i = tid; i < n; i += stride) f(p[i]);
    return;
  }
  const ts_f4* p4 = (const ts_f4*)p;
  const int64_t n4 = n >> 2;
  int64_t i = tid;
  for (; i + stride < n4; i += 2 * stride) {
    const ts_f4 u = p4[i], v = p4[i + stride];
    f(u[0]); f(u[1]); f(u[2]); f(u[3]); f(v[0]); f(v[1]); f(v[2]); f(v[3]);
  }
  for (; i < n4; i += stride) {
    const ts_f4 u = p4[i];
    f(u[0]); f(u[1]); f(u[2]); f(u[3]);
  }
  for (int64_t j = (n4 << 2) + tid; j < n; j += stride) f(p[j]);
}

struct LossArgs {
  const float* coarse; const float* fine; const float* target;
  int64_t n;                   // elements of each image (N rays x W channels)
  const float* reg; int64_t n_reg;
  const float* extra[TS_MAX_EXTRA]; int64_t extra_n[TS_MAX_EXTRA]; int n_extra;
  int scaling;                 // 0: plain MSE (density-temperature module), 1: ImageAsinhScaling on both sides
  float vmax, a, normalization;
  float lambda_image, lambda_reg;
  float* g_coarse; float* g_fine;
  float* stats;                // [loss, coarse mse, fine mse, regularization mean, psnr, non-finite count, 0, 0]
  void* workspace;
};

__device__ __forceinline__ float scale_image(const LossArgs& a, float x) {
  // scaling.py:26-28: image / vmax, then asinh(image / a) / normalization (three roundings, like the reference)
  if (!a.scaling) return x;
  const float u = (x / a.vmax) / a.a;
  return asinhf(u) / a.normalization;
}
__device__ __forceinline__ float scale_image_grad(const LossArgs& a, float x) {
  if (!a.scaling) return 1.f;
  const float u = (x / a.vmax) / a.a;
  return 1.f / (sqrtf(u * u + 1.f) * a.normalization * a.a * a.vmax);
}

__global__ __launch_bounds__(TS_THREADS) void loss_kernel(LossArgs a) {
  __shared__ double sh[TS_THREADS / 64];
  const int64_t tid = (int64_t)blockIdx.x * TS_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * TS_THREADS;
  float sq_c = 0.f, sq_f = 0.f, sum_r = 0.f;
  unsigned bad = 0;
  const float gscale = a.lambda_image * 2.f / (float)a.n;     // d (lambda * mean(diff^2)) / d diff = lambda * 2 diff / n
  for (int64_t i = tid; i < a.n; i += stride) {
    const float c = a.coarse[i], f = a.fine[i], t = a.target[i];
    bad += !isfinite(c) + !isfinite(f);
    const float ts = scale_image(a, t);
    const float dc = scale_image(a, c) - ts, df = scale_image(a, f) - ts;
    sq_c += dc * dc;
    sq_f += df * df;
    a.g_coarse[i] = gscale * dc * scale_image_grad(a, c);
    a.g_fine[i] = gscale * df * scale_image_grad(a, f);
  }
  sweep(a.reg, a.n_reg, tid, stride, [&](float r) {
    bad += !isfinite(r);
    sum_r += r;
  });
  for (int k = 0; k < a.n_extra; ++k)
    sweep(a.extra[k], a.extra_n[k], tid, stride, [&](float x) { bad += !isfinite(x); });

  Workspace ws(a.workspace);
  const double b0 = block_sum((double)sq_c, sh), b1 = block_sum((double)sq_f, sh), b2 = block_sum((double)sum_r, sh),
               b3 = block_sum((double)bad, sh);
  if (threadIdx.x == 0) {
    double* p = ws.partial + (size_t)blockIdx.x * 4;
    p[0] = b0; p[1] = b1; p[2] = b2; p[3] = b3;
  }
  if (!last_block(ws.ticket)) return;
  // final sums by the whole last workgroup in a fixed order (thread t takes blocks t, t + 256, ...; then the block tree):
  // a single thread walking 512 x 4 uncached partials costs ~0.3 ms
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += TS_THREADS) {
    const volatile double* p = ws.partial + (size_t)b * 4;
    s0 += p[0]; s1 += p[1]; s2 += p[2]; s3 += p[3];
  }
  s0 = block_sum(s0, sh); s1 = block_sum(s1, sh); s2 = block_sum(s2, sh); s3 = block_sum(s3, sh);
  if (threadIdx.x == 0) {
    const float mse_c = (float)(s0 / (double)a.n), mse_f = (float)(s1 / (double)a.n);
    const float reg = a.n_reg > 0 ? (float)(s2 / (double)a.n_reg) : 0.f;
    a.stats[0] = a.lambda_image * (mse_c + mse_f) + a.lambda_reg * reg;    // sunerf.py:118-119
    a.stats[1] = mse_c;
    a.stats[2] = mse_f;
    a.stats[3] = reg;
    a.stats[4] = -10.f * log10f(mse_f);                                   // sunerf.py:122
    a.stats[5] = (float)s3;
    a.stats[6] = 0.f; a.stats[7] = 0.f;
    *ws.ticket = 0;       // ready for the next launch on this workspace
  }
}

struct NormArgs {
  const float* grads; int64_t n;
  float grad_scale, max_norm;
  const float* skip_if;     // optional device scalar: a value > 0 (non-finite outputs seen on ANY rank) skips the update
  long long* step_dev;      // optional device counter of APPLIED updates (a skipped step does not advance the bias correction)
  long long step_host;      // step number to use when there is no device counter
  float* norm_out;          // [total norm (after grad_scale), clip coefficient, skipped (0 / 1), 0]
  void* workspace;          // bytes 8..15: step number of this update (read by adam_kernel)
};

__global__ __launch_bounds__(TS_THREADS) void grad_norm_kernel(NormArgs a) {
  __shared__ double sh[TS_THREADS / 64];
  const int64_t tid = (int64_t)blockIdx.x * TS_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * TS_THREADS;
  float sq = 0.f;
  sweep(a.grads, a.n, tid, stride, [&](float x) {
    const float g = x * a.grad_scale;
    sq += g * g;
  });
  Workspace ws(a.workspace);
  const double b = block_sum((double)sq, sh);
  if (threadIdx.x == 0) ws.partial[(size_t)blockIdx.x * 4] = b;
  if (!last_block(ws.ticket)) return;
  double s = 0;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += TS_THREADS) s += ((const volatile double*)ws.partial)[(size_t)k * 4];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(s);
    // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1; max_norm <= 0 disables clipping
    float coef = 1.f;
    if (a.max_norm > 0.f) coef = fminf(a.max_norm / (total + 1e-6f), 1.f);
    // The update is skipped -- on every rank alike, because both inputs are results of the all-reduce -- when a non-finite
    // output was counted anywhere (sunerf.py:105-107 would have asserted) or when the gradient itself is not finite (a NaN
    // norm would otherwise become a NaN clip coefficient and reach every parameter).
    const bool skip = (a.skip_if && !(*a.skip_if <= 0.f)) || !isfinite(total);
    long long step = a.step_host;
    if (a.step_dev) {
      step = *a.step_dev + 1;
      if (!skip) *a.step_dev = step;
    }
    a.norm_out[0] = total;
    a.norm_out[1] = coef;
    a.norm_out[2] = skip ? 1.f : 0.f;
    a.norm_out[3] = 0.f;
    *(long long*)((char*)a.workspace + 8) = step;
    *ws.ticket = 0;
  }
}

struct AdamArgs {
  float* params; float* grads; float* exp_avg; float* exp_avg_sq; int64_t n;
  float grad_scale;
  const float* norm;          // [total, clip coefficient, skipped] from grad_norm_kernel, or null (legacy: no norm pass)
  const float* skip_if;       // legacy path only (norm == null): a value > 0 leaves everything untouched
  const void* workspace;      // bytes 8..15: step number (norm != null)
  long long step_host;
  double lr, beta1, beta2;
  float one_minus_beta1, beta2f, one_minus_beta2, eps;
};

__global__ __launch_bounds__(TS_THREADS) void adam_kernel(AdamArgs a) {
  long long step = a.step_host;
  float coef = 1.f;
  if (a.norm) {
    if (a.norm[2] > 0.f) return;
    coef = a.norm[1];
    step = *(const long long*)((const char*)a.workspace + 8);
  } else if (a.skip_if && !(*a.skip_if <= 0.f)) {
    return;
  }
  // scalars exactly as torch/optim/adam.py forms them (python floats = doubles, rounded to fp32 when they meet a tensor)
  const double bc1 = 1.0 - pow(a.beta1, (double)step), bc2 = 1.0 - pow(a.beta2, (double)step);
  const float step_size = (float)(a.lr / bc1);
  const float bias_correction2_sqrt = (float)sqrt(bc2);
  const int64_t tid = (int64_t)blockIdx.x * TS_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * TS_THREADS;
  for (int64_t i = tid; i < a.n; i += stride) {
    const float g = (a.grads[i] * a.grad_scale) * coef;
    float m = a.exp_avg[i], v = a.exp_avg_sq[i];
    m = m + a.one_minus_beta1 * (g - m);                      // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2f + (a.one_minus_beta2 * g) * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / bias_correction2_sqrt + a.eps;
    a.params[i] = a.params[i] + (-step_size * m) / denom;     // param.addcdiv_(exp_avg, denom, value=-step_size)
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
    a.grads[i] = g;                                           // the clipped, averaged gradient (what the reference leaves in .grad)
  }
}

int blocks_for(int64_t n) {
  const int64_t b = (n + TS_THREADS - 1) / TS_THREADS;
  return (int)(b < 1 ? 1 : (b > TS_BLOCKS ? TS_BLOCKS : b));
}

}  // namespace

extern "C" size_t sunerf_train_workspace_bytes(void) { return 64 + (size_t)TS_BLOCKS * 4 * sizeof(double); }

extern "C" int sunerf_training_loss(const float* coarse_image, const float* fine_image, const float* target_image, int64_t n,
                                    const float* regularization, int64_t n_reg, const float* const* finite_check_host,
                                    const int64_t* finite_check_sizes_host, int n_finite_check, int scaling, float vmax,
                                    float a, float lambda_image, float lambda_regularization, float* g_coarse,
                                    float* g_fine, float* stats, void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 1 || n_reg < 0 || n_finite_check < 0 || n_finite_check > TS_MAX_EXTRA) return SUNERF_E_BADARG;
  if (!coarse_image || !fine_image || !target_image || !g_coarse || !g_fine || !stats || !workspace) return SUNERF_E_BADARG;
  if (n_reg > 0 && !regularization) return SUNERF_E_BADARG;
  if (scaling != 0 && scaling != 1) return SUNERF_E_UNSUPPORTED;
  if (scaling == 1 && !(vmax > 0.f && a > 0.f)) return SUNERF_E_BADARG;
  if (workspace_bytes < sunerf_train_workspace_bytes()) return SUNERF_E_WORKSPACE;
  LossArgs k;
  k.coarse = coarse_image; k.fine = fine_image; k.target = target_image; k.n = n; k.reg = regularization; k.n_reg = n_reg;
  k.n_extra = n_finite_check;
  int64_t most = n > n_reg ? n : n_reg;
  for (int i = 0; i < n_finite_check; ++i) {
    if (!finite_check_host || !finite_check_sizes_host || finite_check_sizes_host[i] < 0) return SUNERF_E_BADARG;
    if (finite_check_sizes_host[i] > 0 && !finite_check_host[i]) return SUNERF_E_BADARG;
    k.extra[i] = finite_check_host[i];
    k.extra_n[i] = finite_check_sizes_host[i];
    if (k.extra_n[i] > most) most = k.extra_n[i];
  }
  k.scaling = scaling; k.vmax = vmax; k.a = a;
  k.normalization = scaling ? (float)asinh(1.0 / (double)a) : 1.f;   // scaling.py:21: np.arcsinh(1 / a) stored as fp32
  k.lambda_image = lambda_image; k.lambda_reg = lambda_regularization;
  k.g_coarse = g_coarse; k.g_fine = g_fine; k.stats = stats; k.workspace = workspace;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(loss_kernel, dim3(blocks_for(most)), dim3(TS_THREADS), 0, (hipStream_t)stream, k);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                                     double beta1, double beta2, double eps, float max_norm, float grad_scale, int64_t step,
                                     const float* skip_if_positive, float* norm_out, void* workspace,
                                     size_t workspace_bytes, void* step_counter, void* stream) {
  if (n < 0 || (step < 1 && !step_counter)) return SUNERF_E_BADARG;
  if (n == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq) return SUNERF_E_BADARG;
  const bool norm_pass = norm_out && workspace;
  if ((max_norm > 0.f || step_counter) && !norm_pass) return SUNERF_E_BADARG;
  if (norm_pass && workspace_bytes < sunerf_train_workspace_bytes()) return SUNERF_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  SUNERF_CLEAR_ERROR();
  if (norm_pass) {
    NormArgs na;
    na.grads = grads; na.n = n; na.grad_scale = grad_scale; na.max_norm = max_norm; na.norm_out = norm_out;
    na.skip_if = skip_if_positive; na.step_dev = (long long*)step_counter; na.step_host = step;
    na.workspace = workspace;
    hipLaunchKernelGGL(grad_norm_kernel, dim3(blocks_for(n)), dim3(TS_THREADS), 0, st, na);
    SUNERF_CHECK_LAUNCH();
  }
  AdamArgs a;
  a.params = params; a.grads = grads; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.n = n; a.grad_scale = grad_scale;
  a.norm = norm_pass ? norm_out : nullptr;
  a.skip_if = skip_if_positive;
  a.workspace = workspace;
  a.step_host = step;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
  a.one_minus_beta1 = (float)(1.0 - beta1);
  a.beta2f = (float)beta2;
  a.one_minus_beta2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  // no reduction here: as many workgroups as the elements need (the reductions above are capped at TS_BLOCKS partials)
  const int64_t adam_blocks = (n + TS_THREADS - 1) / TS_THREADS;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(adam_blocks > 4096 ? 4096 : adam_blocks)), dim3(TS_THREADS), 0, st, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
