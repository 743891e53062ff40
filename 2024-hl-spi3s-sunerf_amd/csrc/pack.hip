// Weight packing: nn.Linear fp32 parameters -> fp16 hi/lo MFMA A-fragment image (layout: sunerf_common.h).
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

struct PackArgs {
  const float* W[SUNERF_MAX_LAYERS];
  const float* b[SUNERF_MAX_LAYERS];
  int n_linear, D, d_out;
  char* packed;
};

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
                               int d_filter, int d_out, void* packed, void* stream) {
  if (!weights_host || !biases_host || !packed) return SUNERF_E_BADARG;
  if (d_filter <= 0 || d_filter % 32 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS || d_out < 1 || d_out > 32)
    return SUNERF_E_UNSUPPORTED;
  PackArgs a;
  for (int i = 0; i < n_linear; ++i) {
    if (!weights_host[i] || !biases_host[i]) return SUNERF_E_BADARG;
    a.W[i] = weights_host[i];
    a.b[i] = biases_host[i];
  }
  a.n_linear = n_linear; a.D = d_filter; a.d_out = d_out; a.packed = (char*)packed;
  const PackedLayout L(d_filter, n_linear);
  const size_t total = (size_t)L.NT * SUNERF_KS0 * 512 + (size_t)(n_linear - 2) * L.NT * L.KS * 512 +
                       (size_t)L.KS * 512 + L.n_bias();
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(pack_mlp_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_abi_version(void) { return SUNERF_ABI_VERSION; }
