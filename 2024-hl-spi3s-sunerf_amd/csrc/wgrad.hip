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
#include "grad_common.h"

namespace {

constexpr int WG_THREADS = 256;

struct WgradArgs {
  const char* act_stash;
  const char* dz_stash;
  const float* g_raw;
  const unsigned* g_absmax_bits;
  float* partial;        // [n_linear][split][tile_r T][tile_c T + 1 (T = bias column)][reg 16][lane 64], T = max(8, D/32)
  int64_t n_chunks_total;
  int64_t n_rays;
  int S, n_chunks;       // samples per ray, chunks per ray
  int n_linear, split;
  int lds_bytes;
};

__device__ __forceinline__ float gscale_from_bits(unsigned bits) { return sunerf_gscale(bits); }

constexpr int NBUF = 4;   // LDS ring of chunk buffers (3 chunks of HBM latency cover)

// KIND: 0 = in layer (X = encoding, 6 fragments), 1 = hidden layer, 2 = out layer (dZ built from g_raw)
// a workgroup covers at most 8 x 8 tiles (4 waves x 4 x 4 tiles = all 256 AGPRs): D = 512 is split over 2 x 2 workgroups
__host__ __device__ constexpr int wg_quads(int D) { return D > 256 ? 2 : 1; }

template <int D, int KIND>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, char* smem, int layer, int split, int qa, int qb) {
  constexpr int NQ = wg_quads(D), T = wg_tiles(D);
  constexpr int NT = D / 32 / NQ, KS = D / 16 / NQ;         // tiles / fragments per side of THIS workgroup's block
  constexpr int KSF = D / 16;                               // fragments per layer in the stashes
  constexpr int QR = NT >= 4 ? 4 : NT;                      // tiles per quadrant side
  constexpr int BFR = KS > SUNERF_KS0 ? KS : SUNERF_KS0;    // B-side fragments per buffer
  constexpr int BUF = (KS + BFR) * 1024;                    // one chunk buffer: A fragments | B fragments
  constexpr int A_FRAGS = KIND == 2 ? 0 : KS;               // out layer: the A tile is built from g_raw, not DMA'd
  constexpr int B_FRAGS = KIND == 0 ? SUNERF_KS0 : KS;
  constexpr int ROW_TILES = KIND == 2 ? 1 : NT;
  constexpr int COL_TILES = KIND == 0 ? SUNERF_KS0 / 2 : NT;
  // the in layer has 3 column tiles and the out layer 1 row tile: their other workgroup blocks are empty
  if ((KIND == 0 && qb > 0) || (KIND == 2 && qa > 0)) return;
  constexpr int PIECES = A_FRAGS + B_FRAGS;
  constexpr int PW = (PIECES + 3) / 4;                      // DMA instructions per wave and chunk (uniform)
  const StashLayout SL(D, a.n_linear);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_act = a.n_linear - 1;
  const size_t dz_chunk_bytes = (size_t)n_act * KSF * 1024;
  const float gscale = gscale_from_bits(*a.g_absmax_bits);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned dummy = lds0 + NBUF * BUF;

  // tiles of this wave: rows [r0, r0 + nr), cols [c0, c0 + nc); waves of column quadrant 0 also own the bias column
  const int rq = wave >> 1, cq = wave & 1;
  const int r0 = rq * QR, c0 = cq * QR;
  const int nr = max(0, min(QR, ROW_TILES - r0)), nc = max(0, min(QR, COL_TILES - c0));
  const bool do_bias = cq == 0 && qb == 0 && nr > 0;

  f32x16 acc[QR][QR];     // the 16 tiles fill the 256 AGPRs exactly
  float bsum[QR];         // db: per-lane partial sums of the A operand (row = lane & 31) over its 8-sample k slices
#pragma unroll
  for (int i = 0; i < QR; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < QR; ++j) acc[i][j] = (f32x16){0};
  }

  const int64_t per = (a.n_chunks_total + a.split - 1) / a.split;
  const int64_t cbeg = (int64_t)split * per, cend = min(a.n_chunks_total, cbeg + per);
  const int64_t n_my = cend > cbeg ? cend - cbeg : 0;

  // out-of-shape tiles read LDS that no DMA fills: define it (a NaN there would poison the shared bias tile via NaN * 0)
  for (int off = tid * 16; off < a.lds_bytes; off += WG_THREADS * 16) *(f32x4*)(smem + off) = (f32x4){0, 0, 0, 0};
  __syncthreads();

  // DMA of one chunk: PIECES pieces of 1 KiB dealt round-robin to the 4 waves; every wave issues the same number PW of
  // instructions (surplus ones re-read piece 0 into a dummy slot) so that vmcnt accounting is uniform
  const char* srcA0 = a.dz_stash + ((size_t)layer * KSF + (size_t)qa * KS) * 1024 + lane * 16;
  const char* srcB0 = a.act_stash + (KIND == 0 ? 0 : SL.h_off(layer - 1) + (size_t)qb * KS * 1024) + lane * 16;
  const size_t act_chunk_bytes = SL.chunk_bytes();
  auto issue_chunk = [&](int64_t chunk, int buf) {
    const char* srcA = srcA0 + chunk * dz_chunk_bytes;
    const char* srcB = srcB0 + chunk * act_chunk_bytes;
    const unsigned dst0 = lds0 + buf * BUF;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int p = k * 4 + wave;
      const bool real = p < PIECES;
      const bool isA = p < A_FRAGS;
      const char* src = !real ? srcB : (isA ? srcA + (size_t)p * 1024 : srcB + (size_t)(p - A_FRAGS) * 1024);
      const unsigned dst = !real ? dummy : (isA ? dst0 + p * 1024 : dst0 + KS * 1024 + (p - A_FRAGS) * 1024);
      const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
      // the stashes are read exactly once by this kernel (d <= 256): non-temporal, so the stream does not displace the
      // partial sums / weights other kernels left in L2 and lands sooner; at d = 512 two workgroups of an XCD read every
      // byte and the second one should find it in L2
      if (NQ == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(dst_u) : "memory");
      else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst_u) : "memory");
    }
  };
  // prologue: NBUF-1 chunks in flight (surplus issues re-read the first chunk: keeps the op count uniform)
  // (a workgroup whose slice is empty -- fewer chunks than workgroups -- must still read inside the stash)
  const int64_t safe = min(cbeg, a.n_chunks_total - 1);
  for (int k = 0; k < NBUF - 1; ++k) issue_chunk(k < n_my ? cbeg + k : safe, k);
  const unsigned laneoff = tr_lane_offset(lane);

  for (int64_t it = 0; it < n_my; ++it) {
    const int buf = (int)(it & (NBUF - 1));
    const int64_t chunk = cbeg + it;
    // chunk `it` has landed when at most the (NBUF-2) younger chunks are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PW * (NBUF - 2)) : "memory");
    if (KIND == 2) {
      // A tile of the out layer: "features" 0 / 1 = d loss / d raw[..., 0 / 1] (fragment order index 16 s + 8 h + e ->
      // s = 0, h = 0, e = 0 / 1), 32 samples; written by the first 128 threads (2 fragments x 64 lanes)
      if (tid < 128) {
        const int64_t ray = chunk / a.n_chunks;
        const int c = (int)(chunk % a.n_chunks);
        const int s = tid >> 6, n = lane & 31, h = lane >> 5;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int i = 32 * c + n;
        if (s == 0 && h == 0 && i < a.S) {
          const f32x2 g = *(const f32x2*)(a.g_raw + ((size_t)ray * a.S + i) * 2);
          v[0] = (_Float16)(g[0] * gscale);
          v[1] = (_Float16)(g[1] * gscale);
        }
        *(half8*)(smem + (size_t)buf * BUF + tid * 16) = v;
      }
    }
    __syncthreads();   // everyone's pieces of chunk `it` landed; everyone finished chunk it-1 (its buffer is refilled next)
    {
      const int64_t nxt = it + NBUF - 1;
      issue_chunk(nxt < n_my ? cbeg + nxt : safe, (int)(nxt & (NBUF - 1)));
    }
    const unsigned bufA = lds0 + buf * BUF + laneoff, bufB = bufA + KS * 1024;
    // ---- per k-step (16 samples): batch of transposed operand reads, one wait, 16 (+4 bias) MFMAs ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // all QR x QR tiles are computed unconditionally (tiles outside the layer's shape read staged-but-unused LDS and
      // are never stored): runtime guards around the MFMAs cost more registers than the few wasted MFMAs of the in / out
      // layers cost time in this HBM-bound kernel
      half4 alo[QR], ahi[QR], blo[QR], bhi[QR];
#pragma unroll
      for (int i = 0; i < QR; ++i) tr_issue(bufA + (2 * (r0 + i)) * 1024 + ks * 256, alo[i], ahi[i]);
#pragma unroll
      for (int j = 0; j < QR; ++j) tr_issue(bufB + (2 * (c0 + j)) * 1024 + ks * 256, blo[j], bhi[j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < QR; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
#pragma unroll
      for (int j = 0; j < QR; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
      half8 bf[QR];
#pragma unroll
      for (int j = 0; j < QR; ++j) bf[j] = join(blo[j], bhi[j]);
#pragma unroll
      for (int i = 0; i < QR; ++i) {
        const half8 af = join(alo[i], ahi[i]);
#ifndef SUNERF_DBG_WGRAD_NOMFMA
#define SUNERF_DBG_WGRAD_NOMFMA 0     // timing experiments only: stream the stashes, multiply (almost) nothing
#endif
        if (!SUNERF_DBG_WGRAD_NOMFMA || (i == 0 && ks == 0))
#pragma unroll
        for (int j = 0; j < QR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[j], acc[i][j], 0, 0, 0);
        // db = sum over samples of dZ: four v_dot2_f32_f16 against (1, 1) per operand (fp32 accumulate)
        const half2v one2 = {(_Float16)1, (_Float16)1};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const half2v pr = {af[e], af[e + 1]};
          bsum[i] = __builtin_amdgcn_fdot2(pr, one2, bsum[i], false);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus prefetches target our LDS: drain before exit
  // ---- partial sums -> workspace [layer][split][tr T][tc T + 1][reg][lane]; tc = T is the bias column ----
  float* out = a.partial + ((size_t)layer * a.split + split) * (T * (T + 1)) * 1024;
  const int gr0 = qa * NT + r0, gc0 = qb * NT + c0;      // global tile coordinates of this wave's block
#pragma unroll
  for (int i = 0; i < QR; ++i) {
    if (i < nr) {
#pragma unroll
      for (int j = 0; j < QR; ++j)
        if (j < nc) {
          float* t = out + ((size_t)(gr0 + i) * (T + 1) + (gc0 + j)) * 1024;
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r * 64 + lane] = acc[i][j][r];
        }
    }
  }
  if (do_bias) {   // lanes l and l + 32 hold the two k halves of row l: combine, store 32 sums per row tile
#pragma unroll
    for (int i = 0; i < QR; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32);
      if (i < nr && lane < 32) out[((size_t)(gr0 + i) * (T + 1) + T) * 1024 + lane] = v;
    }
  }
}

template <int D>
__global__ __launch_bounds__(WG_THREADS, 1) void wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF chunk buffers + 1 KiB dummy target + slack
  constexpr int NQ = wg_quads(D);
  int q, ls;
  if (NQ == 1) {
    q = 0; ls = blockIdx.x;
  } else {
    // the 2 x 2 workgroups of one (layer, slice) read the same dZ / H halves pairwise: put them on the SAME XCD (workgroup
    // i runs on XCD i % 8) so that the second reader of a chunk hits that XCD's L2 instead of going to HBM again
    const int xcd = blockIdx.x & 7, g = blockIdx.x >> 5;
    q = (blockIdx.x >> 3) & 3;
    ls = g * 8 + xcd;
    if (ls >= a.n_linear * a.split) return;
  }
  const int layer = ls / a.split, split = ls % a.split;
  const int qa = q / NQ, qb = q % NQ;
  if (layer == 0) wgrad_body<D, 0>(a, smem, layer, split, qa, qb);
  else if (layer == a.n_linear - 1) wgrad_body<D, 2>(a, smem, layer, split, qa, qb);
  else wgrad_body<D, 1>(a, smem, layer, split, qa, qb);
}

}  // namespace

extern "C" size_t sunerf_wgrad_workspace_bytes(int d_filter, int n_linear, int split) {
  if (n_linear < 2 || split < 1 || d_filter < 32 || d_filter % 32) return 0;
  const size_t T = wg_tiles(d_filter);
  return (size_t)n_linear * split * T * (T + 1) * 1024 * sizeof(float);
}

extern "C" int sunerf_mlp_wgrad(int d_filter, int n_linear, int d_out, const void* packedT, const void* act_stash,
                                const void* dz_stash, const float* g_raw, const void* g_absmax, int64_t n_rays, int n_samples,
                                void* workspace, int split, float* const* grad_weights_host,
                                float* const* grad_biases_host, int accumulate, void* stream) {
  if (!grad_weights_host || !grad_biases_host) return SUNERF_E_BADARG;
  if (n_rays < 0 || n_samples < 2 || split < 1) return SUNERF_E_BADARG;
  if (n_rays > 0 && (!packedT || !act_stash || !dz_stash || !g_raw || !g_absmax || !workspace)) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS || d_out < 1 || d_out > 2) return SUNERF_E_UNSUPPORTED;
  if (d_filter != 64 && d_filter != 128 && d_filter != 256 && d_filter != 512) return SUNERF_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (n_samples + 31) / 32;
  ReduceArgs r;
  for (int i = 0; i < n_linear; ++i) {
    if (!grad_weights_host[i] || !grad_biases_host[i]) return SUNERF_E_BADARG;
    r.gW[i] = grad_weights_host[i];
    r.gb[i] = grad_biases_host[i];
  }
  hipError_t e;
  if (n_rays == 0) {
    if (!accumulate)
      for (int i = 0; i < n_linear; ++i) {
        const size_t nb = (size_t)((i == n_linear - 1) ? d_out : d_filter) * sizeof(float);
        if ((e = hipMemsetAsync(grad_biases_host[i], 0, nb, st)) != hipSuccess) return (int)e;
        if ((e = hipMemsetAsync(grad_weights_host[i], 0, nb * ((i == 0) ? SUNERF_ENC_DIM : d_filter), st)) != hipSuccess) return (int)e;
      }
    return 0;
  }
  WgradArgs a;
  a.act_stash = (const char*)act_stash; a.dz_stash = (const char*)dz_stash; a.g_raw = g_raw;
  a.g_absmax_bits = (const unsigned*)g_absmax; a.partial = (float*)workspace;
  a.n_chunks_total = n_rays * n_chunks; a.n_rays = n_rays; a.S = n_samples; a.n_chunks = n_chunks;
  a.n_linear = n_linear; a.split = split;
  const int nq = wg_quads(d_filter);
  const int ks = d_filter / 16 / nq;       // fragments per side of one workgroup's block
  // ring + 1 KiB dummy DMA target + slack for the unguarded operand reads of out-of-shape tiles
  const size_t lds = (size_t)NBUF * ((size_t)ks + (ks > SUNERF_KS0 ? ks : SUNERF_KS0)) * 1024 + 1024 + 16 * 1024;
  a.lds_bytes = (int)lds;
  const void* fn = d_filter == 64 ? (const void*)wgrad_kernel<64> : d_filter == 128 ? (const void*)wgrad_kernel<128>
                 : d_filter == 256 ? (const void*)wgrad_kernel<256> : (const void*)wgrad_kernel<512>;
  if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
  // every (layer, split) slot of the workspace that the reduce kernel reads is written by exactly one workgroup; slots of
  // workgroups without chunks hold zeros from their zero-initialised accumulators
  SUNERF_CLEAR_ERROR();
  // d = 512: groups of 8 (layer, slice) pairs x 4 blocks, see wgrad_kernel
  const unsigned grid = nq == 1 ? (unsigned)(n_linear * split) : (unsigned)((n_linear * split + 7) / 8 * 32);
  switch (d_filter) {
    case 64: hipLaunchKernelGGL(wgrad_kernel<64>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
    case 128: hipLaunchKernelGGL(wgrad_kernel<128>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
    case 256: hipLaunchKernelGGL(wgrad_kernel<256>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
    default: hipLaunchKernelGGL(wgrad_kernel<512>, dim3(grid), dim3(WG_THREADS), lds, st, a); break;
  }
  SUNERF_CHECK_LAUNCH();
  const size_t slot = (size_t)wg_tiles(d_filter) * (wg_tiles(d_filter) + 1) * 1024;
  for (int i = 0; i < n_linear; ++i) {
    r.partial[i] = (const float*)workspace + (size_t)i * split * slot;
    r.split[i] = split;
    r.slot[i] = slot;
    r.bias[i] = r.partial[i] + (size_t)wg_tiles(d_filter) * 1024;          // the bias column of the same slots
    r.bias_split[i] = split;
    r.bias_slot[i] = slot;
    r.bias_tr[i] = (size_t)(wg_tiles(d_filter) + 1) * 1024;
  }
  r.status = nullptr;
  r.sticky = nullptr;
  r.g_absmax_bits = (const unsigned*)g_absmax; r.n_linear = n_linear; r.D = d_filter;
  // the boosts sunerf_pack_mlp_t folded into the transposed image the data gradient went through (its tail holds their source)
  r.sumsq = (const float*)((const char*)packedT + sunerf_packed_mlp_t_bytes(d_filter, n_linear) - SUNERF_MAX_LAYERS * sizeof(float));
  r.d_out = d_out; r.accumulate = accumulate;
  const unsigned tt = (unsigned)wg_tiles(d_filter);
  hipLaunchKernelGGL(reduce_grads_kernel, dim3(tt * (tt + 1) * 1024 / 256, n_linear), dim3(256), 0, st, r);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
