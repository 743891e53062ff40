// fp32 backward of the sine MLP for SMALL batches (gfx950): parameter gradients from the gradient w.r.t. the raw output,
// with every product and sum in fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered fp32 accumulation) and the
// forward activations RECOMPUTED in fp32 from the query points -- nothing is read from the fp16 activation stash.
//
// Replaces what torch.autograd derives from sunerf/model/model.py:44-57 + 123-132 (reference root), like sunerf_mlp_dgrad +
// sunerf_mlp_wgrad / sunerf_mlp_backward_pipe, for the case those kernels are not built for.  They run the chain
// dZ_{l-1} = (W_l^T dZ_l) cos(Z_{l-1}) on single fp16 operands (dZ, cos, H rounded to 11 bits): every term of a gradient sum
// carries ~2^-12 of relative error.  For a training batch (>= 1e5 samples) that averages out -- every tensor within 1e-3 of the
// fp32 reference, tests/test_gpu_backward.py, test_gpu_e2e.py -- but a sum over a few hundred samples that cancels to a few
// per cent of its terms (bias gradients of tiny batches: tests/tools/fuzz_parity.py cases 23 / 37 / 57, 34 ... 1100 samples) keeps
// 2^-12 x its condition number, 2e-3 ... 3e-2.  tests/tools/bias_conditioning.py reproduces those numbers on the CPU and shows
// that fp32 summation of the bias terms alone changes little (the error sits in the chain's operands, not in the last sum).
// So the small batches (<= 4096 samples by default: sunerf_hip/ops.py:EXACT_BACKWARD_SAMPLES) get the arithmetic of the reference
// instead: 4e-7 ... 5e-5 on every tensor.  Chosen by sample count in sunerf_hip/ops.py:mlp_backward.
//
// Layer-major, plain global-memory GEMMs (one 32 x 32 output tile per wave, operands straight from L2, requested a group of
// products ahead): at these sizes the working set (18 x N x 256 floats) is 75 MB at most; 25 GEMM launches (+ 10 small ones) per backward.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int EPI_SINCOS = 0;   // out0 = sin(acc + bias), out1 = cos(acc + bias)           (forward layer)
constexpr int EPI_MULC = 1;     // out0 = acc * mul                                           (data gradient: dZ = dH * cos)
constexpr int EPI_PART = 2;     // out0[slice][m][n] = acc over this block's slice of K       (weight gradient partial sums)

struct GemmArgs {
  const float* A; long a_sm, a_sk;     // A(m, k) = A[m * a_sm + k * a_sk]
  const float* B; long b_sk, b_sn;     // B(k, n) = B[k * b_sk + n * b_sn]
  int M, N, K;
  const float* bias;                   // EPI_SINCOS: [N]
  float* out0; float* out1; long ldo;  // row-major [M][ldo] (EPI_PART: [slice][M][ldo])
  const float* mul; long ldm;          // EPI_MULC: [M][ldm]
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = (a.N + 31) / 32, tiles_m = (a.M + 31) / 32;
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile >= (long)tiles_m * tiles_n) return;
  const int tm = (int)(tile / tiles_n), tn = (int)(tile % tiles_n);
  const int m = tm * 32 + r, n = tn * 32 + r;
  const bool mok = m < a.M, nok = n < a.N;
  const float* pa = a.A + (long)(mok ? m : 0) * a.a_sm;
  const float* pb = a.B + (long)(nok ? n : 0) * a.b_sn;
  int k0 = 0, k1 = a.K;
  if (EPI == EPI_PART) {
    const int per = ((a.K + (int)gridDim.y - 1) / (int)gridDim.y + 1) & ~1;     // even: a k pair never straddles two slices
    k0 = (int)blockIdx.y * per;
    k1 = k0 + per < a.K ? k0 + per : a.K;
  }
  // lane half h supplies k + h of a 2-deep product (A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]).  The
  // operands of GROUP products are requested together, one group ahead of the matrix instructions that consume them: with one
  // load pair and one instruction per trip the loop ran at the latency of an L2 read per 64 matrix cycles (90 us per GEMM at
  // 4096 samples, 25 GEMM launches (+ 10 small ones) per backward).
  constexpr int GROUP = 8;
  f32x16 acc = {0};
  float av[GROUP], bv[GROUP];
  auto fetch = [&](int k, float* fa, float* fb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int kk = k + 2 * u + h;
      const bool kok = kk < k1;
      const int kc = kok ? kk : k0;
      const float x = pa[(long)kc * a.a_sk], y = pb[(long)kc * a.b_sk];
      fa[u] = (mok && kok) ? x : 0.f;
      fb[u] = (nok && kok) ? y : 0.f;
    }
  };
  if (k0 < k1) fetch(k0, av, bv);
  for (int k = k0; k < k1; k += 2 * GROUP) {
    float an[GROUP], bn[GROUP];
    const bool more = k + 2 * GROUP < k1;
    if (more) fetch(k + 2 * GROUP, an, bn);
#pragma unroll
    for (int u = 0; u < GROUP; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    if (more) {
#pragma unroll
      for (int u = 0; u < GROUP; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
    }
  }
  if (!nok) return;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = tm * 32 + acc_row(g, h);
    if (row >= a.M) continue;
    const float v = acc[g];
    if (EPI == EPI_SINCOS) {
      const float z = v + a.bias[n];
      a.out0[(long)row * a.ldo + n] = sinf(z);
      a.out1[(long)row * a.ldo + n] = cosf(z);
    } else if (EPI == EPI_MULC) {
      a.out0[(long)row * a.ldo + n] = v * a.mul[(long)row * a.ldm + n];
    } else {
      a.out0[((long)blockIdx.y * a.M + row) * a.ldo + n] = v;
    }
  }
}

// query points -> the 84 encoder features of PositionalEncoding.forward (model.py:123-132): [x, sin(x 2^k / 2) k-major, cos(...)]
__global__ void encode_kernel(const float* rays_o, const float* rays_d, const float* times, const float* z_vals,
                              const float* points, long n_rays, int S, float* enc /* [n_rays * S][84] */) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * S) return;
  float v[4];
  if (points) {
    for (int c = 0; c < 4; ++c) v[c] = points[i * 4 + c];
  } else {
    const long ray = i / S;
    const float z = z_vals[i];
    // sampling.py:100: product and sum rounded separately (-ffp-contract=off for the whole library)
    for (int c = 0; c < 3; ++c) v[c] = rays_o[ray * 3 + c] + rays_d[ray * 3 + c] * z;
    v[3] = times[ray];
  }
  float* e = enc + i * SUNERF_ENC_DIM;
  for (int c = 0; c < 4; ++c) e[c] = v[c];
  for (int k = 0; k < 10; ++k) {
    const float f = k == 0 ? 0.5f : (float)(1 << (k - 1));      // 2^k / scale_factor, exact
    for (int c = 0; c < 4; ++c) {
      const float arg = v[c] * f;
      e[4 + 4 * k + c] = sinf(arg);
      e[44 + 4 * k + c] = cosf(arg);
    }
  }
}

// One launch per layer behind the weight-gradient GEMM (they were two: 44 -> 35 launches per backward):
//   blocks [0, ceil(cols / 64)):  db[o] (+)= sum over samples of dZ[s][o], accumulated in fp64 (the sum may cancel to a small fraction
//                                 of its terms); 64 columns per block, 16 row groups, four independent loads in flight per thread
//   the other blocks:             dW[o][j] (+)= sum over slices of partial[slice][o][j]
constexpr int COLSUM_GROUPS = 16;
constexpr int FINISH_THREADS = 64 * COLSUM_GROUPS;
__global__ __launch_bounds__(FINISH_THREADS) void finish_layer_kernel(const float* partial, int slices, long count, float* dw, const float* dz,
                                                                      long n, int ld, int cols, float* db, int accumulate) {
  const int col_blocks = (cols + 63) / 64;
  if ((int)blockIdx.x >= col_blocks) {
    const long i = (long)(blockIdx.x - col_blocks) * FINISH_THREADS + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += partial[(long)k * count + i];
    dw[i] = accumulate ? dw[i] + s : s;
    return;
  }
  __shared__ double part[COLSUM_GROUPS][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < cols) {
    long i = q;
    for (; i + 3 * COLSUM_GROUPS < n; i += 4 * COLSUM_GROUPS) {
      const float v0 = dz[i * ld + c], v1 = dz[(i + COLSUM_GROUPS) * ld + c], v2 = dz[(i + 2 * COLSUM_GROUPS) * ld + c],
                  v3 = dz[(i + 3 * COLSUM_GROUPS) * ld + c];
      s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; i < n; i += COLSUM_GROUPS) s0 += (double)dz[i * ld + c];
  }
  part[q][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && c < cols) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < COLSUM_GROUPS; ++k) t += part[k][threadIdx.x];
    const float v = (float)t;
    db[c] = accumulate ? db[c] + v : v;
  }
}

constexpr int K_SLICES = 16;

struct ExactLayout {
  size_t enc, act, dz, partial, total;      // act: [layer][H | cos][N][D]; dz: [2][N][D]
  size_t per;                               // floats per [N][D] tensor
  ExactLayout(int64_t n, int D, int n_linear) {
    auto up = [](size_t v) { return (v + 63) / 64 * 64; };
    per = up((size_t)n * D);
    size_t off = 0;
    enc = off; off += up((size_t)n * SUNERF_ENC_DIM);
    act = off; off += (size_t)(n_linear - 1) * 2 * per;
    dz = off; off += 2 * per;
    partial = off; off += (size_t)K_SLICES * D * (D > SUNERF_ENC_DIM ? D : SUNERF_ENC_DIM);
    total = off * sizeof(float);
  }
};

template <int EPI>
int launch_gemm(const GemmArgs& a, int slices, hipStream_t st) {
  const long tiles = (long)((a.M + 31) / 32) * ((a.N + 31) / 32);
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3((unsigned)((tiles + 3) / 4), (unsigned)slices), dim3(256), 0, st, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t sunerf_mlp_backward_exact_workspace_bytes(int64_t n_points, int d_filter, int n_linear) {
  if (n_points < 1 || d_filter < 1 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return 0;
  return ExactLayout(n_points, d_filter, n_linear).total;
}

extern "C" int sunerf_mlp_backward_exact(const float* const* weights_host, const float* const* biases_host, int n_linear,
                                         int d_filter, int d_out, const float* rays_o, const float* rays_d, const float* times,
                                         const float* z_vals, const float* points, int64_t n_rays, int n_samples,
                                         const float* g_raw, void* workspace, size_t workspace_bytes,
                                         float* const* grad_weights_host, float* const* grad_biases_host, int accumulate,
                                         void* stream) {
  if (!weights_host || !biases_host || !grad_weights_host || !grad_biases_host || !g_raw || !workspace) return SUNERF_E_BADARG;
  if (n_rays <= 0 || n_samples < 1 || d_filter < 1 || d_out < 1) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return SUNERF_E_UNSUPPORTED;
  if (!points && (!rays_o || !rays_d || !times || !z_vals)) return SUNERF_E_BADARG;
  for (int i = 0; i < n_linear; ++i)
    if (!weights_host[i] || !biases_host[i] || !grad_weights_host[i] || !grad_biases_host[i]) return SUNERF_E_BADARG;
  const int64_t N = n_rays * n_samples;
  if (N > (int64_t)1 << 24) return SUNERF_E_UNSUPPORTED;
  const int D = d_filter, n_act = n_linear - 1;
  const ExactLayout L(N, D, n_linear);
  if (workspace_bytes < L.total) return SUNERF_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* enc = ws + L.enc;
  auto H = [&](int l) { return ws + L.act + (size_t)(2 * l) * L.per; };
  auto C = [&](int l) { return ws + L.act + (size_t)(2 * l + 1) * L.per; };
  float* dzb[2] = {ws + L.dz, ws + L.dz + L.per};
  int rc;

  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rays_o, rays_d, times, z_vals, points,
                     (long)n_rays, n_samples, enc);
  SUNERF_CHECK_LAUNCH();
  // forward, fp32: H_l = sin(X W_l^T + b_l), cos kept for the backward
  for (int l = 0; l < n_act; ++l) {
    const int K = l == 0 ? SUNERF_ENC_DIM : D;
    GemmArgs a = {};
    a.A = l == 0 ? enc : H(l - 1); a.a_sm = K; a.a_sk = 1;
    a.B = weights_host[l]; a.b_sk = 1; a.b_sn = K;
    a.M = (int)N; a.N = D; a.K = K; a.bias = biases_host[l];
    a.out0 = H(l); a.out1 = C(l); a.ldo = D;
    if ((rc = launch_gemm<EPI_SINCOS>(a, 1, st))) return rc;
  }
  // backward: dZ of the out layer is g_raw itself
  const float* dz = g_raw;
  int dz_cols = d_out, flip = 0;
  for (int i = n_linear - 1; i >= 0; --i) {
    const int cols = i == 0 ? SUNERF_ENC_DIM : D;           // fan-in of layer i
    const float* X = i == 0 ? enc : H(i - 1);
    // dW_i[o][j] = sum_s dZ_i[s][o] X[s][j]
    GemmArgs w = {};
    w.A = dz; w.a_sm = 1; w.a_sk = dz_cols;
    w.B = X; w.b_sk = cols; w.b_sn = 1;
    w.M = dz_cols; w.N = cols; w.K = (int)N;
    w.out0 = ws + L.partial; w.ldo = cols;
    if ((rc = launch_gemm<EPI_PART>(w, K_SLICES, st))) return rc;
    const long count = (long)dz_cols * cols;
    hipLaunchKernelGGL(finish_layer_kernel, dim3((unsigned)((dz_cols + 63) / 64 + (count + FINISH_THREADS - 1) / FINISH_THREADS)),
                       dim3(FINISH_THREADS), 0, st, ws + L.partial, K_SLICES, count, grad_weights_host[i], dz, (long)N, dz_cols, dz_cols,
                       grad_biases_host[i], accumulate);
    SUNERF_CHECK_LAUNCH();
    if (i == 0) break;
    // dZ_{i-1}[s][j] = (sum_o dZ_i[s][o] W_i[o][j]) cos(Z_{i-1})[s][j]
    GemmArgs d = {};
    d.A = dz; d.a_sm = dz_cols; d.a_sk = 1;
    d.B = weights_host[i]; d.b_sk = D; d.b_sn = 1;
    d.M = (int)N; d.N = D; d.K = dz_cols;
    d.out0 = dzb[flip]; d.ldo = D; d.mul = C(i - 1); d.ldm = D;
    if ((rc = launch_gemm<EPI_MULC>(d, 1, st))) return rc;
    dz = dzb[flip]; dz_cols = D; flip ^= 1;
  }
  return 0;
}
