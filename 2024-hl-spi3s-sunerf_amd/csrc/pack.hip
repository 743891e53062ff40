// Weight packing: nn.Linear fp32 parameters -> fp16 hi/lo MFMA A-fragment image (layout: sunerf_common.h).
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

struct PackArgs {
  const float* W[SUNERF_MAX_LAYERS];
  const float* b[SUNERF_MAX_LAYERS];
  int n_linear, D, d_out;
  int fp8c;          // hidden / out layers in the fp8c stream format (SUNERF_PRECISION_FAST)
  char* packed;
};

// max |w| (after the 1/(2 pi) scaling of the non-output layers) of every hidden / out layer -> absmax[l] (float bits)
__global__ void pack_absmax_kernel(PackArgs a) {
  const PackedLayout L(a.D, a.n_linear);
  unsigned* absmax = (unsigned*)(a.packed + L.absmax_off());
  const int l = 1 + blockIdx.y;
  const int rows = (l == a.n_linear - 1) ? a.d_out : a.D;
  float m = 0.f;
  // rows * D is a multiple of 4 (D % 32 == 0) and nn.Linear weights are 16-byte aligned: one 16-byte load per iteration
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t n = (size_t)rows * a.D;
  if ((((uintptr_t)a.W[l]) & 15) == 0) {
    const f4* w4 = (const f4*)a.W[l];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) {
      const f4 w = w4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      m = fmaxf(m, fabsf(a.W[l][i]));
  }
  if (l < a.n_linear - 1) m *= 0.15915494309189535f;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  // one atomic per workgroup: atomics on one address serialise in L2 (~10 ns each; 2048 of them were the kernel's 20 us)
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f && m < INFINITY) atomicMax(absmax + l, __float_as_uint(m));
  }
}

__device__ __forceinline__ int scale_exponent(unsigned absmax_bits) {
  const float m = __uint_as_float(absmax_bits);
  if (!(m > 0.f)) return 0;
  int e;
  frexpf(m, &e);            // m = f 2^e, f in [0.5, 1)  =>  |w| < 2^e
  // at most 21: a weight whose fp16 head is SUBNORMAL leaves a remainder of up to 2^-25 whatever the layer's scale, and
  // 2^-25 2^(sh+11) must stay below the e4m3 maximum (a layer of max |w| < 2e-4 used to overflow to NaN here)
  const int sh = 8 - e;
  return sh > 21 ? 21 : sh;
}
__device__ __forceinline__ unsigned char to_fp8(float x) {   // OCP e4m3, round to nearest even, saturating
  x = fminf(fmaxf(x, -448.f), 448.f);   // (the conversion itself turns an overflow into NaN)
  return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xff);
}

// one thread per (hi, lo) pair of the image
__global__ void pack_mlp_kernel(PackArgs a) {
  const PackedLayout L(a.D, a.n_linear);
  const size_t n_pairs0 = (size_t)L.NT * SUNERF_KS0 * 512;                       // in layer
  const size_t n_pairs_h = (size_t)(a.n_linear - 2) * L.NT * L.KS * 512;         // hidden->hidden
  const size_t n_pairs_o = (size_t)L.KS * 512;                                   // out layer
  const size_t n_bias = L.n_bias();
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = n_pairs0 + n_pairs_h + n_pairs_o + n_bias;
  if (idx >= total) return;

  if (idx >= n_pairs0 + n_pairs_h + n_pairs_o) {  // biases (fp32)
    size_t i = idx - (n_pairs0 + n_pairs_h + n_pairs_o);
    float* dst = (float*)(a.packed + L.bias_off());
    int l = (int)(i / a.D), f = (int)(i % a.D);
    float v;
    if (l < a.n_linear - 1) v = a.b[l][f] * 0.15915494309189535f;   // revolutions (see sunerf_common.h)
    else { int r = (int)(i - (size_t)(a.n_linear - 1) * a.D); v = r < a.d_out ? a.b[a.n_linear - 1][r] : 0.f; }
    dst[i] = v;
    return;
  }

  int l, U, s, lane, e, ks, in_dim;
  size_t r = idx;
  if (r < n_pairs0) {
    l = 0; ks = SUNERF_KS0; in_dim = SUNERF_ENC_DIM;
  } else if (r < n_pairs0 + n_pairs_h) {
    r -= n_pairs0; ks = L.KS; in_dim = a.D;
    l = 1 + (int)(r / ((size_t)L.NT * ks * 512)); r %= (size_t)L.NT * ks * 512;
  } else {
    r -= n_pairs0 + n_pairs_h; l = a.n_linear - 1; ks = L.KS; in_dim = a.D;
  }
  U = (int)(r / ((size_t)ks * 512)); r %= (size_t)ks * 512;
  s = (int)(r / 512); r %= 512;
  lane = (int)(r / 8); e = (int)(r % 8);
  const int m = lane & 31, h = lane >> 5;
  const int col = (l == 0) ? kmap_encoding(s, h, e) : kmap_hidden(s, h, e);
  const int row = 32 * U + m;
  const int n_rows = (l == a.n_linear - 1) ? a.d_out : a.D;
  float w = 0.f;
  if (col >= 0 && row < n_rows) w = a.W[l][(size_t)row * in_dim + col];
  if (l < a.n_linear - 1) w *= 0.15915494309189535f;   // pre-activation in revolutions for v_sin_f32
  const _Float16 hi = (_Float16)w;                  // round to nearest
  if (l >= 1 && a.fp8c) {                           // fp8c format (sunerf_common.h)
    const int sh = scale_exponent(((const unsigned*)(a.packed + L.absmax_off()))[l]);
    if (U == 0 && s == 0 && lane == 0 && e == 0) ((int*)(a.packed + L.scale_off()))[l] = sh;
    char* grp = a.packed + L.block_off(l, U) + (size_t)(s >> 2) * SUNERF_GROUP_BYTES;
    const int sl = s & 3;
    *(_Float16*)(grp + sl * 1024 + lane * 16 + e * 2) = hi;
    const size_t byte = (size_t)(sl >> 1) * 1024 + lane * 16 + (sl & 1) * 8 + e;
    grp[4096 + byte] = (char)to_fp8(ldexpf(w - (float)hi, sh + 11));
    grp[6144 + byte] = (char)to_fp8(ldexpf((float)hi, sh));
    return;
  }
  const _Float16 lo = (_Float16)(w - (float)hi);    // exact remainder, rounded to nearest (may be subnormal)
  _Float16* blk = (_Float16*)(a.packed + L.block_off(l, U));
  blk[((size_t)(s * 2 + 0) * 64 + lane) * 8 + e] = hi;
  blk[((size_t)(s * 2 + 1) * 64 + lane) * 8 + e] = lo;
}

}  // namespace

extern "C" size_t sunerf_packed_mlp_bytes(int d_filter, int n_linear) {
  if (d_filter <= 0 || d_filter % 32 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return 0;
  return PackedLayout(d_filter, n_linear).total_bytes();
}

extern "C" int sunerf_pack_mlp(const float* const* weights_host, const float* const* biases_host, int n_linear,
                               int d_filter, int d_out, int precision, void* packed, void* stream) {
  if (!weights_host || !biases_host || !packed) return SUNERF_E_BADARG;
  if (d_filter <= 0 || d_filter % 32 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS || d_out < 1 || d_out > 32)
    return SUNERF_E_UNSUPPORTED;
  PackArgs a;
  for (int i = 0; i < n_linear; ++i) {
    if (!weights_host[i] || !biases_host[i]) return SUNERF_E_BADARG;
    a.W[i] = weights_host[i];
    a.b[i] = biases_host[i];
  }
  if (precision != SUNERF_PRECISION_FAST && precision != SUNERF_PRECISION_EXACT && precision != SUNERF_PRECISION_HALF)
    return SUNERF_E_BADARG;
  a.n_linear = n_linear; a.D = d_filter; a.d_out = d_out; a.packed = (char*)packed;
  a.fp8c = precision == SUNERF_PRECISION_FAST;
  const PackedLayout L(d_filter, n_linear);
  const size_t total = (size_t)L.NT * SUNERF_KS0 * 512 + (size_t)(n_linear - 2) * L.NT * L.KS * 512 +
                       (size_t)L.KS * 512 + L.n_bias();
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  SUNERF_CLEAR_ERROR();
  if (a.fp8c) {
    hipError_t e = hipMemsetAsync(a.packed + L.scale_off(), 0, 128, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pack_absmax_kernel, dim3(64, n_linear - 1), dim3(256), 0, (hipStream_t)stream, a);
    SUNERF_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(pack_mlp_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_abi_version(void) { return SUNERF_ABI_VERSION; }
