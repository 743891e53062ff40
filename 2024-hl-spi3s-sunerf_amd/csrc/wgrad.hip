// Weight / bias gradients of the sine MLP from the activation and dZ stashes (gfx950).
//
//   dW_l[j][k] = sum_samples dZ_l[j][n] * X_l[k][n],   X_0 = encoding, X_l = H_{l-1};   db_l[j] = sum_samples dZ_l[j][n]
//
// Both operands sit in HBM in MFMA B-fragment order (sunerf_common.h: lane = sample, 8 features per lane), while the
// contraction here runs over SAMPLES.  No explicit transposition is needed: a fragment is copied to LDS as it is and
// read back with ds_read_b64_tr_b16, which hands every lane 4 consecutive samples of ONE feature; two such reads form
// the 8-deep k slice of a 32x32x16 MFMA operand.  Feature order inside a tile is the fragment order
// (index 16 s + 8 h + e); the final reduce kernel maps it back to nn.Linear rows / columns.
//
// Decomposition: one workgroup (4 waves) accumulates the full D x D (or D x 96, 32 x D) gradient of one layer over a
// contiguous slice of chunks; wave w owns the 4 x 4 tile quadrant (w >> 1, w & 1) = 256 accumulator registers.  The
// kernel is HBM-bound (1 KiB of fragments per sample and layer against 64 MFMA-cycles), so the staging is a plain
// register-prefetched double buffer.  Partials go to a workspace and are summed by reduce_grads_kernel.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int WG_THREADS = 256;

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const char* act_stash;
  const char* dz_stash;
  const float* g_raw;
  const unsigned* g_absmax_bits;
  float* partial;        // [n_linear][split][tile_r 8][tile_c 8][reg 16][lane 64]
  int64_t n_chunks_total;
  int64_t n_rays;
  int S, n_chunks;       // samples per ray, chunks per ray
  int n_linear, split;
};

__device__ __forceinline__ float gscale_from_bits(unsigned bits) {
  const float m = __uint_as_float(bits);
  if (!(m > 0.f)) return 1.f;
  int e;
  frexpf(m, &e);
  return ldexpf(1.f, 10 - e);
}

// transposed read of one MFMA operand (32 features x 16 samples) from a fragment pair staged in LDS.
//   frags: LDS address of fragment 2T (1 KiB each, fragment 2T+1 follows); ks = k-step (samples 16 ks .. 16 ks + 15)
__device__ __forceinline__ half8 tr_operand(const char* frags, int lane, int ks) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const char* f = frags + (g & 1) * 1024;                  // lanes 0-15 / 32-47: features 0-15, else 16-31 of the tile
  const int n0 = 16 * ks + 8 * (g >> 1);                   // lanes >= 32 hold k elements 8..15
  const char* addr = f + ((p >> 1) * 32 + n0 + q) * 16 + (p & 1) * 8;
  half4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(uintptr_t)addr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:64" : "=v"(hi) : "v"((unsigned)(uintptr_t)addr) : "memory");   // samples n0+4..n0+7
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) :: "memory");
  half8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <int D>
__global__ __launch_bounds__(WG_THREADS, 1) void wgrad_kernel(WgradArgs a) {
  constexpr int NT = D / 32, KS = D / 16;
  constexpr int QR = NT >= 4 ? 4 : NT;     // tiles per quadrant side (D = 64: 2 x 2 tiles, single quadrant per side...)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A fragments (KS KiB) | B fragments (max(KS, 6) KiB)
  const StashLayout SL(D, a.n_linear);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int layer = blockIdx.x / a.split, split = blockIdx.x % a.split;
  const int n_act = a.n_linear - 1;
  const bool is_out = layer == a.n_linear - 1;
  const bool is_in = layer == 0;
  const int a_frags = is_out ? 2 : KS;                 // out layer: one 32-row tile built from g_raw
  const int b_frags = is_in ? SUNERF_KS0 : KS;
  const int row_tiles = is_out ? 1 : NT;
  const int col_tiles = is_in ? SUNERF_KS0 / 2 : NT;
  char* ldsA = smem;
  char* ldsB = smem + KS * 1024;
  const size_t dz_chunk_bytes = (size_t)n_act * KS * 1024;
  const float gscale = gscale_from_bits(*a.g_absmax_bits);

  // tiles of this wave: rows [r0, r0 + nr), cols [c0, c0 + nc)
  const int rq = wave >> 1, cq = wave & 1;
  const int r0 = rq * QR, c0 = cq * QR;
  const int nr = max(0, min(QR, row_tiles - r0)), nc = max(0, min(QR, col_tiles - c0));

  f32x16 acc[QR][QR];
#pragma unroll
  for (int i = 0; i < QR; ++i)
#pragma unroll
    for (int j = 0; j < QR; ++j) acc[i][j] = (f32x16){0};

  const int64_t per = (a.n_chunks_total + a.split - 1) / a.split;
  const int64_t cbeg = (int64_t)split * per, cend = min(a.n_chunks_total, cbeg + per);

  for (int64_t chunk = cbeg; chunk < cend; ++chunk) {
    // ---- stage the chunk's fragments: A = dZ_layer (or g_raw), B = X_layer ----
    __syncthreads();   // previous chunk's readers are done
    const char* srcB = a.act_stash + chunk * SL.chunk_bytes() + (is_in ? 0 : SL.h_off(layer - 1));
    for (int off = tid * 16; off < b_frags * 1024; off += WG_THREADS * 16) *(f32x4*)(ldsB + off) = *(const f32x4*)(srcB + off);
    if (!is_out) {
      const char* srcA = a.dz_stash + chunk * dz_chunk_bytes + (size_t)layer * KS * 1024;
      for (int off = tid * 16; off < a_frags * 1024; off += WG_THREADS * 16) *(f32x4*)(ldsA + off) = *(const f32x4*)(srcA + off);
    } else {
      // fragment pair of the 32-"feature" tile whose features 0 / 1 are d loss / d raw[..., 0 / 1] (fragment order:
      // feature index 16 s + 8 h + e -> s = 0, h = 0, e = 0 / 1), 32 samples
      const int64_t ray = chunk / a.n_chunks;
      const int c = (int)(chunk % a.n_chunks);
      for (int idx = tid; idx < 128; idx += WG_THREADS) {    // 2 fragments x 64 lanes
        const int s = idx >> 6, l = idx & 63, n = l & 31, h = l >> 5;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int i = 32 * c + n;
        if (s == 0 && h == 0 && i < a.S) {
          const f32x2 g = *(const f32x2*)(a.g_raw + ((size_t)ray * a.S + i) * 2);
          v[0] = (_Float16)(g[0] * gscale);
          v[1] = (_Float16)(g[1] * gscale);
        }
        *(half8*)(ldsA + idx * 16) = v;
      }
    }
    __syncthreads();
    // ---- 2 k-steps of 16 samples ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 af[QR], bf[QR];
#pragma unroll
      for (int i = 0; i < QR; ++i) af[i] = (i < nr) ? tr_operand(ldsA + (size_t)(2 * (r0 + i)) * 1024, lane, ks) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < QR; ++j) bf[j] = (j < nc) ? tr_operand(ldsB + (size_t)(2 * (c0 + j)) * 1024, lane, ks) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QR; ++j)
          if (i < nr && j < nc) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- partial sums -> workspace [layer][split][tr 8][tc 8][reg][lane] (unused tiles are never read) ----
  float* out = a.partial + ((size_t)layer * a.split + split) * 64 * 1024;
#pragma unroll
  for (int i = 0; i < QR; ++i)
#pragma unroll
    for (int j = 0; j < QR; ++j)
      if (i < nr && j < nc) {
        float* t = out + ((size_t)(r0 + i) * 8 + (c0 + j)) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r * 64 + lane] = acc[i][j][r];
      }
}

// feature of fragment-order index f (= 16 s + 8 h + e) on the activation side / the encoding side
__device__ __forceinline__ int frag_feature_hidden(int f) { return kmap_hidden(f >> 4, (f >> 3) & 1, f & 7); }
__device__ __forceinline__ int frag_feature_enc(int f) { return kmap_encoding(f >> 4, (f >> 3) & 1, f & 7); }

struct ReduceArgs {
  const float* partial;
  const unsigned* g_absmax_bits;
  float* gW[SUNERF_MAX_LAYERS];
  int n_linear, D, d_out, split;
  int accumulate;     // 0: overwrite grads, 1: add to them
};

// one thread per element of every dW: sums the split partials, unscales, writes nn.Linear layout [out][in]
__global__ void reduce_grads_kernel(ReduceArgs a) {
  const int layer = blockIdx.y;
  const int D = a.D;
  const int rows = (layer == a.n_linear - 1) ? a.d_out : D;
  const int cols = (layer == 0) ? SUNERF_ENC_DIM : D;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // enumerate in tile order so that reads of the partials are coalesced: idx = ((tr*8 + tc)*16 + reg)*64 + lane
  const int row_tiles = (layer == a.n_linear - 1) ? 1 : D / 32;
  const int col_tiles = (layer == 0) ? SUNERF_KS0 / 2 : D / 32;
  if (idx >= row_tiles * col_tiles * 1024) return;
  const int lane = idx & 63, reg = (idx >> 6) & 15, t = idx >> 10;
  const int tr = t / col_tiles, tc = t % col_tiles;
  const int fa = 32 * tr + acc_row(reg, lane >> 5);     // fragment-order index on the dZ side
  const int fb = 32 * tc + (lane & 31);                 // ... on the X side
  const int j = (layer == a.n_linear - 1) ? fa : frag_feature_hidden(fa);   // out layer: feature index = output index
  const int k = (layer == 0) ? frag_feature_enc(fb) : frag_feature_hidden(fb);
  if (j >= rows || k < 0 || k >= cols) return;
  const float* p = a.partial + (size_t)layer * a.split * 64 * 1024 + ((size_t)tr * 8 + tc) * 1024 + reg * 64 + lane;
  float sum = 0.f;
  for (int s = 0; s < a.split; ++s) sum += p[(size_t)s * 64 * 1024];
  const float m = __uint_as_float(*a.g_absmax_bits);
  float inv = 1.f;
  if (m > 0.f) { int e; frexpf(m, &e); inv = ldexpf(1.f, e - 10); }
  float* dst = a.gW[layer] + (size_t)j * cols + k;
  *dst = a.accumulate ? *dst + sum * inv : sum * inv;
}

// ---- bias gradients: db_l[j] = sum over samples of dZ_l[j][n] -----------------------------------------------------
// grid (n_act, BG_SPLIT): every workgroup sums a slice of chunks for one layer; lane = (sample, half) as stored, so the
// per-lane partial sums over chunks are reduced over the 32 samples at the end and added atomically (few values).
constexpr int BG_SPLIT = 64;
struct BiasArgs {
  const char* dz_stash;
  const float* g_raw;
  const unsigned* g_absmax_bits;
  float* gb[SUNERF_MAX_LAYERS];
  int64_t n_chunks_total, n_samples_total;
  int n_linear, D, d_out;
};

__global__ __launch_bounds__(256) void bias_grad_kernel(BiasArgs a) {
  const int layer = blockIdx.x;            // activation layers 0..n_act-1; layer n_act = out layer (from g_raw)
  const int n_act = a.n_linear - 1;
  const int KS = a.D / 16;
  const int tid = threadIdx.x;
  const float m = __uint_as_float(*a.g_absmax_bits);
  float inv = 1.f;
  if (m > 0.f) { int e; frexpf(m, &e); inv = ldexpf(1.f, e - 10); }
  if (layer == n_act) {   // out layer: plain column sums of g_raw (unscaled fp32)
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.y * 256 + tid; i < a.n_samples_total; i += (int64_t)gridDim.y * 256) {
      const f32x2 g = *(const f32x2*)(a.g_raw + i * 2);
      s0 += g[0]; s1 += g[1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); }
    if ((tid & 63) == 0) { atomicAdd(a.gb[layer] + 0, s0); if (a.d_out > 1) atomicAdd(a.gb[layer] + 1, s1); }
    return;
  }
  // threads = (fragment s, lane): D/16 fragments x 64 lanes; loop when that exceeds the block
  const size_t dz_chunk_bytes = (size_t)n_act * KS * 1024;
  for (int fl = tid; fl < KS * 64; fl += 256) {
    const int s = fl >> 6, lane = fl & 63;
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t chunk = blockIdx.y; chunk < a.n_chunks_total; chunk += gridDim.y) {
      const half8 v = *(const half8*)(a.dz_stash + chunk * dz_chunk_bytes + ((size_t)layer * KS + s) * 1024 + lane * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum[e] += (float)v[e];
    }
    // reduce over the 32 samples (lanes of one half)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) sum[e] += __shfl_xor(sum[e], d, 32);
    }
    if ((lane & 31) == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(a.gb[layer] + kmap_hidden(s, lane >> 5, e), sum[e] * inv);
    }
  }
}

}  // namespace

extern "C" size_t sunerf_wgrad_workspace_bytes(int n_linear, int split) {
  if (n_linear < 2 || split < 1) return 0;
  return (size_t)n_linear * split * 64 * 1024 * sizeof(float);
}

extern "C" int sunerf_mlp_wgrad(int d_filter, int n_linear, int d_out, const void* act_stash, const void* dz_stash,
                                const float* g_raw, const void* g_absmax, int64_t n_rays, int n_samples,
                                void* workspace, int split, float* const* grad_weights_host,
                                float* const* grad_biases_host, int accumulate, void* stream) {
  if (!act_stash || !dz_stash || !g_raw || !g_absmax || !workspace || !grad_weights_host || !grad_biases_host) return SUNERF_E_BADARG;
  if (n_rays < 0 || n_samples < 2 || split < 1) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS || d_out < 1 || d_out > 2) return SUNERF_E_UNSUPPORTED;
  if (d_filter != 64 && d_filter != 128 && d_filter != 256) return SUNERF_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (n_samples + 31) / 32;
  WgradArgs a;
  a.act_stash = (const char*)act_stash; a.dz_stash = (const char*)dz_stash; a.g_raw = g_raw;
  a.g_absmax_bits = (const unsigned*)g_absmax; a.partial = (float*)workspace;
  a.n_chunks_total = n_rays * n_chunks; a.n_rays = n_rays; a.S = n_samples; a.n_chunks = n_chunks;
  a.n_linear = n_linear; a.split = split;
  const int ks = d_filter / 16;
  const size_t lds = ((size_t)ks + (ks > SUNERF_KS0 ? ks : SUNERF_KS0)) * 1024;
  hipError_t e;
  SUNERF_CLEAR_ERROR();
  if (n_rays > 0) {
    const unsigned grid = (unsigned)(n_linear * split);
    switch (d_filter) {
      case 64: hipLaunchKernelGGL(wgrad_kernel<64>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
      case 128: hipLaunchKernelGGL(wgrad_kernel<128>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
      default: hipLaunchKernelGGL(wgrad_kernel<256>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
    }
    SUNERF_CHECK_LAUNCH();
  }
  ReduceArgs r;
  BiasArgs b;
  for (int i = 0; i < n_linear; ++i) {
    if (!grad_weights_host[i] || !grad_biases_host[i]) return SUNERF_E_BADARG;
    r.gW[i] = grad_weights_host[i];
    b.gb[i] = grad_biases_host[i];
    if (!accumulate) {
      const size_t nb = (size_t)((i == n_linear - 1) ? d_out : d_filter) * sizeof(float);
      e = hipMemsetAsync(grad_biases_host[i], 0, nb, st);
      if (e != hipSuccess) return (int)e;
      if (n_rays == 0) {
        const size_t nw = nb * ((i == 0) ? SUNERF_ENC_DIM : d_filter);
        e = hipMemsetAsync(grad_weights_host[i], 0, nw, st);
        if (e != hipSuccess) return (int)e;
      }
    }
  }
  if (n_rays == 0) return 0;
  r.partial = (const float*)workspace; r.g_absmax_bits = (const unsigned*)g_absmax; r.n_linear = n_linear; r.D = d_filter;
  r.d_out = d_out; r.split = split; r.accumulate = accumulate;
  hipLaunchKernelGGL(reduce_grads_kernel, dim3(64 * 1024 / 256, n_linear), dim3(256), 0, st, r);
  SUNERF_CHECK_LAUNCH();
  b.dz_stash = (const char*)dz_stash; b.g_raw = g_raw; b.g_absmax_bits = (const unsigned*)g_absmax;
  b.n_chunks_total = n_rays * n_chunks; b.n_samples_total = n_rays * n_samples; b.n_linear = n_linear; b.D = d_filter;
  b.d_out = d_out;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(n_linear, BG_SPLIT), dim3(256), 0, st, b);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
