// Shared pieces of the weight-gradient kernels (wgrad.hip: one kernel per step over the dZ stash; bwd_pipe.hip: the
// layer-pipelined backward that never writes dZ to HBM): transposed operand reads from fragments staged in LDS, the
// partial-sum workspace layout, and the kernel that sums the partials into nn.Linear layouts.
#pragma once
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// transposed read of one MFMA operand (32 features x 16 samples) from a fragment pair staged in LDS: two
// ds_read_b64_tr_b16 (4 samples each); issue-only -- the caller waits once for a whole batch of operands.
//   addr: LDS byte address of fragment 2T (1 KiB each, fragment 2T+1 follows) + tr_lane_offset(lane) + 256 * k-step
__device__ __forceinline__ unsigned tr_lane_offset(int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // lanes 0-15 / 32-47: features 0-15 of the tile (fragment 2T), else 16-31 (fragment 2T+1); lanes >= 32: k 8..15
  return (g & 1) * 1024 + ((p >> 1) * 32 + 8 * (g >> 1) + q) * 16 + (p & 1) * 8;
}
__device__ __forceinline__ void tr_issue(unsigned addr, half4& lo, half4& hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:64"
               : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
}
// the same with the fragment-pair / k-step part of the address as an immediate (one address VGPR for a whole batch of operands)
template <int OFF>
__device__ __forceinline__ void tr_issue_imm(unsigned base, half4& lo, half4& hi) {
  static_assert(OFF >= 0 && OFF + 64 < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi) : "v"(base), "i"(OFF), "i"(OFF + 64) : "memory");
}
__device__ __forceinline__ half8 join(half4 lo, half4 hi) {
  half8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// tiles per side of the partial-sum workspace (bias column = index wg_tiles)
__host__ __device__ constexpr int wg_tiles(int D) { return D / 32 > 8 ? D / 32 : 8; }

// feature of fragment-order index f (= 16 s + 8 h + e) on the activation side / the encoding side
__device__ __forceinline__ int frag_feature_hidden(int f) { return kmap_hidden(f >> 4, (f >> 3) & 1, f & 7); }
__device__ __forceinline__ int frag_feature_enc(int f) { return kmap_encoding(f >> 4, (f >> 3) & 1, f & 7); }

// Partial sums of layer l: `split[l]` slots of `slot[l]` floats each at `partial[l]`; inside a slot
// [tile_r][tile_c T + 1 (T = bias column)][reg 16][lane 64], T = max(8, D/32), in MFMA accumulator order.
struct ReduceArgs {
  const float* partial[SUNERF_MAX_LAYERS];
  int split[SUNERF_MAX_LAYERS];
  size_t slot[SUNERF_MAX_LAYERS];
  // bias sums of layer l: 32 floats per row tile at bias[l] + tr * bias_tr[l], in `bias_split[l]` slots `bias_slot[l]` floats apart.
  // (The two-kernel backward keeps them in the bias column of the partial-sum slots: bias = partial + T * 1024, bias_tr =
  // (T + 1) * 1024; the pipelined backward sums db_l in fp32 where dZ_l is FORMED -- the stage above, or the prologue -- so some
  // layers' sums live in another workgroup's slots.)
  const float* bias[SUNERF_MAX_LAYERS];
  int bias_split[SUNERF_MAX_LAYERS];
  size_t bias_slot[SUNERF_MAX_LAYERS];
  size_t bias_tr[SUNERF_MAX_LAYERS];
  const unsigned* g_absmax_bits;
  float* gW[SUNERF_MAX_LAYERS];
  float* gb[SUNERF_MAX_LAYERS];
  int n_linear, D, d_out;
  int accumulate;          // 0: overwrite grads, 1: add to them
  const float* sumsq;      // per-layer sums of squares at the tail of the transposed image (the boosts folded into W^T), or null
  const unsigned* status;  // optional: a non-zero word (the pipelined backward gave up) turns every gradient into NaN, so that
                           // the optimiser's non-finite guard skips the step instead of applying garbage
  unsigned* sticky;        // optional: receives max(sticky, status).  The status word is cleared in front of every launch and one
                           // workspace serves all launches of a stream (fine model, then coarse: a give-up of the first would be
                           // erased by the second before the host looks); this word is only ever raised, the host reads and clears it
};

// one thread per element of every dW / db: sums the split partials, unscales, writes nn.Linear layouts
__global__ void reduce_grads_kernel(ReduceArgs a) {
  const int layer = blockIdx.y;
  const int D = a.D;
  if (a.status && a.sticky && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned st = *a.status;
    if (st) atomicMax(a.sticky, st);
  }
  const int rows = (layer == a.n_linear - 1) ? a.d_out : D;
  const int cols = (layer == 0) ? SUNERF_ENC_DIM : D;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // enumerate in tile order so that reads of the partials are coalesced: idx = ((tr*(T+1) + tc)*16 + reg)*64 + lane
  const int T = wg_tiles(D);
  const int row_tiles = (layer == a.n_linear - 1) ? 1 : D / 32;
  const int col_tiles = (layer == 0) ? SUNERF_KS0 / 2 : D / 32;
  if (idx >= row_tiles * (T + 1) * 1024) return;
  const int lane = idx & 63, reg = (idx >> 6) & 15, t = idx >> 10;
  const int tr = t / (T + 1), tc = t % (T + 1);
  if (tc != T && tc >= col_tiles) return;
  int k = 0;
  int fa_row = acc_row(reg, lane >> 5);
  if (tc == T) {                                        // bias slot: 32 plain sums per row tile (reg 0, lanes 0..31)
    if (reg != 0 || lane >= 32) return;
    fa_row = lane;
  } else {
    const int fb = 32 * tc + (lane & 31);               // fragment-order index on the X side
    k = (layer == 0) ? frag_feature_enc(fb) : frag_feature_hidden(fb);
    if (k < 0 || k >= cols) return;
  }
  const int fa = 32 * tr + fa_row;                      // fragment-order index on the dZ side
  const int j = (layer == a.n_linear - 1) ? fa : frag_feature_hidden(fa);   // out layer: feature index = output index
  if (j >= rows) return;
  const bool is_bias = tc == T;
  const size_t slot = is_bias ? a.bias_slot[layer] : a.slot[layer];
  const float* p = is_bias ? a.bias[layer] + (size_t)tr * a.bias_tr[layer] + lane
                           : a.partial[layer] + ((size_t)tr * (T + 1) + tc) * 1024 + reg * 64 + lane;
  // (four independent loads in flight: one dependent load per partial made this kernel latency-bound at 80 us)
  const int ns = is_bias ? a.bias_split[layer] : a.split[layer];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 4 <= ns; s += 4) {
    const float v0 = p[(size_t)s * slot], v1 = p[(size_t)(s + 1) * slot], v2 = p[(size_t)(s + 2) * slot], v3 = p[(size_t)(s + 3) * slot];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; s < ns; ++s) s0 += p[(size_t)s * slot];
  const float sum = (s0 + s1) + (s2 + s3);
  // dZ of this layer carries the boosts of every layer above it (sunerf_common.h: sunerf_bwd_boost)
  int boost = 0;
  if (a.sumsq)
    for (int l = layer + 1; l < a.n_linear; ++l) boost += sunerf_bwd_boost(a.sumsq[l], D);
  const float inv = ldexpf(sunerf_gscale_inv(*a.g_absmax_bits), -boost);
  float v = sum * inv;
  if (a.status && *a.status != 0u) v = __uint_as_float(0x7fc00000u);
  float* dst = (tc == T) ? a.gb[layer] + j : a.gW[layer] + (size_t)j * cols + k;
  *dst = a.accumulate ? *dst + v : v;
}

}  // namespace
