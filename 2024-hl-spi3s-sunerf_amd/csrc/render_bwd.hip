// Backward pass of the fused emission renderer (gfx950).
//
//   loss gradients w.r.t. image (N) and regularization (N,S)
//     -> integral_bwd_kernel : d/d raw of emission.py:14-54 + base_tracing.py:43-44,108        -> g_raw (N,S,2)
//     -> dgrad_pair_kernel   : back-propagation through the sine MLP (model.py:44-57), two 32-sample chunks per wave:
//                              dZ_l = (W_{l+1}^T dZ_{l+1}) * cos(Z_l), written to the dZ stash (fp16 fragments)
//     -> wgrad (wgrad.hip)   : dW_l = sum_samples dZ_l H_{l-1}^T,  db_l = sum_samples dZ_l
// No gradient flows to the ray geometry / sample positions (sampling.py:120 detaches the resampled z and the
// stratified z has no parameters), so layer 0 needs no data gradient.
//
// Numerics: dZ and the activations enter the matrix products as single fp16 operands, W^T as hi + lo (fp32 accumulate): the reference tolerance for gradients
// is 1e-3 relative per tensor and the rounding errors (2^-12 relative, unbiased) average out over the samples.  fp16 has
// a narrow exponent range, so g_raw is multiplied by a power of two `gscale` chosen from max|g_raw| of the batch
// (computed on the device, no host round trip) and dW / db are multiplied by 1/gscale at the end (sunerf_common.h:
// SUNERF_GSCALE_LOG2); the hidden data gradients saturate at +-65504 instead of overflowing.
#include "sunerf_common.h"
#include "weight_ring.h"
#include "../../include/sunerf_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// integral backward: 32 lanes per ray, one sample per lane and 32-sample chunk (coalesced reads of raw / z and writes of
// g_raw; the two sweeps along the ray are chunk-wise scans with a scalar carry, like the forward kernel's integral)
// ---------------------------------------------------------------------------------------------------------------
constexpr int IB_THREADS = 256;                // 8 rays per workgroup
constexpr int IB_RAYS = IB_THREADS / 32;
constexpr int IB_MAX_GRID = 2048;              // 8 workgroups per CU

__global__ __launch_bounds__(IB_THREADS) void integral_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ g_image, const float* __restrict__ g_reg,
    const float* __restrict__ g_weights, const float* __restrict__ g_absorption,
    float g_reg_const, float reg_radius, int64_t n_rays, int S, float* __restrict__ g_raw,
    unsigned* __restrict__ g_absmax_bits) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [IB_RAYS][3][Sp]: emerging intensity, a, dist
  const int tid = threadIdx.x, n = tid & 31, sub = tid >> 5;
  float local_max = 0.f;
  // a workgroup walks over groups of IB_RAYS rays (grid <= IB_MAX_GRID): ONE atomic per workgroup on the batch maximum -- with
  // one per wave and 4096 workgroups the 16384 atomics on that single word took most of the kernel's 200 us
  const int64_t n_groups = (n_rays + IB_RAYS - 1) / IB_RAYS;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
  const int64_t ray_raw = grp * IB_RAYS + sub;
  const bool ray_ok = ray_raw < n_rays;
  const int64_t ray = ray_ok ? ray_raw : n_rays - 1;
  const int n_chunks = (S + 31) >> 5, Sp = n_chunks * 32;
  float* em_s = lds + (size_t)sub * 3 * Sp;
  float* a_s = em_s + Sp;
  float* dist_s = a_s + Sp;
  const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
  const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
  const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
  const float* z = z_vals + ray * S;
  const float* r = raw + ray * S * 2;
  const float gi = g_image[ray];
  // ---- sweep 1 (along the ray): emerging_i = I_i * T_i, T the exclusive product of (a + 1e-10) ----
  float carry_T = 1.f, carry_z = 0.f;
  float sum_em = 0.f, dot_gw = 0.f;      // per-lane partial sums (only used with g_weights)
  const float z0 = z[0], z1 = z[1];
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    const bool valid = i < S;
    const float zi = z[valid ? i : S - 1];
    float zprev = __shfl_up(zi, 1, 32);
    if (n == 0) zprev = carry_z;
    const float dzv = (i == 0) ? (z1 - z0) : (zi - zprev);
    const float dist = dzv * dnorm;
    const f32x2 rr = *(const f32x2*)(r + 2 * (valid ? i : S - 1));
    const float a = expf(-fmaxf(rr[1], 0.f) * dist);
    float pr = valid ? (a + 1e-10f) : 1.f;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const float o = __shfl_up(pr, d, 32);
      if (n >= d) pr *= o;
    }
    float excl = __shfl_up(pr, 1, 32);
    if (n == 0) excl = 1.f;
    const float em_i = valid ? expf(rr[0]) * dist * (carry_T * excl) : 0.f;
    em_s[i] = em_i;
    a_s[i] = a;
    dist_s[i] = dist;
    if (g_weights) {
      sum_em += em_i;
      dot_gw += valid ? g_weights[ray * S + i] * em_i : 0.f;
    }
    carry_T *= __shfl(pr, 31, 32);
    carry_z = __shfl(zi, 31, 32);
  }
  // weights = em / (sum em + 1e-10) (emission.py:49-50):  sum_k gw_k d weights_k / d em_i = gw_i / D - dot / D^2
  float inv_den = 0.f, dot_over_den2 = 0.f;
  if (g_weights) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { sum_em += __shfl_xor(sum_em, d, 32); dot_gw += __shfl_xor(dot_gw, d, 32); }
    inv_den = 1.f / (sum_em + 1e-10f);
    dot_over_den2 = dot_gw * inv_den * inv_den;
  }
  // ---- sweep 2 (against the ray): suffix_i = sum_{k > i} emerging_k ----
  float carry_suffix = 0.f;
  for (int c = n_chunks - 1; c >= 0; --c) {
    const int i = 32 * c + n;
    const bool valid = i < S;
    const float em = em_s[i];
    // gradient arriving at em_i: the image's (gi) and, when the caller differentiates the weights output too, theirs
    const float ge = g_weights ? gi + (valid ? g_weights[ray * S + i] : 0.f) * inv_den - dot_over_den2 : gi;
    const float wv = g_weights ? ge * em : em;         // (without g_weights: the sums of em itself, scaled by gi afterwards)
    float incl = wv;                                   // inclusive suffix sum within the chunk
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const float o = __shfl_down(incl, d, 32);
      if (n + d < 32) incl += o;
    }
    const float suffix = carry_suffix + (incl - wv);
    carry_suffix += __shfl(incl, 0, 32);
    if (valid && ray_ok) {
      const float a = a_s[i], dist = dist_s[i];
      const float r1 = r[2 * i + 1];
      // image = sum_i em_i :  d/d r0_i = em_i ;  d/d a_i = suffix_i / (a_i + 1e-10)
      const float g0 = g_weights ? wv : gi * em;
      float ga = (g_weights ? suffix : gi * suffix) / (a + 1e-10f);
      if (g_absorption) ga += g_absorption[ray * S + i];        // the 'regularizing_quantity' output is a itself
      // regularization_i = relu(|p_i| - R) (1 - a_i)   (base_tracing.py:43-44, D2 resolved)
      const float gr = g_reg ? g_reg[ray * S + i] : g_reg_const;
      if (gr != 0.f) {
        const float zi = z[i];
        const float px = ox + dx * zi, py = oy + dy * zi, pz = oz + dz * zi;
        const float pd = sqrtf((px * px + py * py) + pz * pz);
        ga -= gr * fmaxf(pd - reg_radius, 0.f);
      }
      // a = exp(-relu(r1) dist)
      const float g1 = (r1 > 0.f) ? ga * (-dist * a) : 0.f;
      const f32x2 gg = {g0, g1};
      *(f32x2*)(g_raw + ((size_t)ray * S + i) * 2) = gg;
      local_max = fmaxf(local_max, fmaxf(fabsf(g0), fabsf(g1)));
    }
  }
  }
  // max |g_raw| of the batch (bit pattern of a non-negative float orders like an unsigned integer)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, d));
  __shared__ float wave_max[IB_THREADS / 64];
  if ((tid & 63) == 0) wave_max[tid >> 6] = local_max;
  __syncthreads();
  if (tid == 0) {
    float m = wave_max[0];
#pragma unroll
    for (int w = 1; w < IB_THREADS / 64; ++w) m = fmaxf(m, wave_max[w]);
    if (m > 0.f && m < INFINITY) atomicMax(g_absmax_bits, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// integral forward on a GIVEN raw tensor: EmissionRadiativeTransfer.raw2outputs (emission.py:14-54) as a stand-alone entry
// point (the render kernel has the same arithmetic fused behind its MLP).  32 lanes per ray, chunk-wise product scan.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(IB_THREADS) void integral_fwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_d, int64_t n_rays, int S,
    float* __restrict__ image, float* __restrict__ weights, float* __restrict__ absorption) {
  const int tid = threadIdx.x, n = tid & 31, sub = tid >> 5;
  const int64_t ray = (int64_t)blockIdx.x * IB_RAYS + sub;
  if (ray >= n_rays) return;                       // (a whole 32-lane group leaves: the shuffles below are 32 wide)
  const int n_chunks = (S + 31) >> 5;
  const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
  const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
  const float* z = z_vals + ray * S;
  const float* r = raw + ray * S * 2;
  float carry_T = 1.f, carry_z = 0.f, sum_em = 0.f;
  const float z0 = z[0], z1 = z[1];
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    const bool valid = i < S;
    const float zi = z[valid ? i : S - 1];
    float zprev = __shfl_up(zi, 1, 32);
    if (n == 0) zprev = carry_z;
    const float dist = ((i == 0) ? (z1 - z0) : (zi - zprev)) * dnorm;      // emission.py:19-24: first distance duplicated
    const f32x2 rr = *(const f32x2*)(r + 2 * (valid ? i : S - 1));
    const float a = expf(-fmaxf(rr[1], 0.f) * dist);
    float pr = valid ? (a + 1e-10f) : 1.f;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const float o = __shfl_up(pr, d, 32);
      if (n >= d) pr *= o;
    }
    float excl = __shfl_up(pr, 1, 32);
    if (n == 0) excl = 1.f;
    const float em = valid ? expf(rr[0]) * dist * (carry_T * excl) : 0.f;
    float s = em;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
    sum_em += s;
    if (valid) {
      weights[ray * S + i] = em;                   // un-normalised; finalised below by the same lane
      absorption[ray * S + i] = a;
    }
    carry_T *= __shfl(pr, 31, 32);
    carry_z = __shfl(zi, 31, 32);
  }
  const float denom = sum_em + 1e-10f;
  for (int c = 0; c < n_chunks; ++c) {
    const int i = 32 * c + n;
    if (i < S) weights[ray * S + i] = weights[ray * S + i] / denom;
  }
  if (n == 0) image[ray] = sum_em;
}

// ---------------------------------------------------------------------------------------------------------------
// dgrad
// ---------------------------------------------------------------------------------------------------------------
// Transposed weight image "packedT" (pack_mlp_t_kernel), fp16, consumption order (A fragment: lane (m, h) element e =
// W_l[out feature kmapT][in feature 32U + m]):
//   [out_layer^T : D/32 tiles x 1 k-step x 1 KiB : K slot (h=0,e) = output e of the network (e < d_out), rest zero]
//   for l = n_linear-2 down to 1: [W_l^T : D/32 tiles x D/16 k-steps x (hi 1 KiB | lo 1 KiB), K order = kmap_hidden]
// The hidden weights are split hi + lo (two MFMAs per k-step): a single fp16 weight (2^-12 relative) is a SYSTEMATIC
// error that every sample shares and that accumulates over the layers (measured 6e-4 on dW_0 of an 8-layer net);
// the fp16 rounding of dZ itself is per-sample noise that averages out in the weight gradients.
constexpr int DG_WAVES = 4;
constexpr int DG_THREADS = DG_WAVES * 64;

struct DgradArgs {
  const char* packedT;
  const float* g_raw;          // (N, S, 2)
  const unsigned* g_absmax_bits;
  const char* stash;           // activation stash of the forward pass (cos fragments are read)
  char* dz_stash;              // out: [chunk][n_act layers][KS fragments] fp16 dZ (scaled by gscale)
  int64_t n_rays;
  int S;
  int n_linear;
};

__device__ __forceinline__ float gscale_from_bits(unsigned bits) { return sunerf_gscale(bits); }

__device__ __forceinline__ void pin_agpr(half8& f) { asm volatile("" : "+a"(f)); }

// dZ = dH * cos for one accumulator tile -> two fp16 fragments (the next layer's B operand and the dZ stash)
__device__ __forceinline__ void dz_tile(const f32x16& acc, const half8& c0, const half8& c1, half8& d0, half8& d1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    d0[j] = (_Float16)sunerf_sat16(acc[j] * (float)c0[j]);
    d1[j] = (_Float16)sunerf_sat16(acc[8 + j] * (float)c1[j]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dgrad: TWO 32-sample chunks per wave and weight tile
// ---------------------------------------------------------------------------------------------------------------
// With one chunk per wave every A fragment pair (hi, lo: 2 KiB of LDS reads) feeds two MFMAs, i.e. 128 KiB of LDS reads
// per workgroup and tile against 1024 MFMA cycles -- exactly the 128 B/clk the LDS delivers, so the matrix pipe can be at
// most ~50 % busy.  Here a wave back-propagates two consecutive chunks of its ray at once: each A fragment feeds four
// MFMAs and the two independent accumulators hide the MFMA dependency latency.  The W^T stream is DMA'd page-wise into the
// 4-page LDS ring the forward kernel uses (weight_ring.h); stash traffic goes through buffer instructions.
// Register budget, d <= 256: 4 sets of dZ fragments (2 chunks x (layer input, layer output)) = up to 256 AGPRs.
// d = 512 (SPILL): one set per chunk fills the 256 AGPRs.  The layer output only goes to the dZ stash (where it has to go
// anyway) and the LAST tile of a layer -- whose k-step s is the last reader of input fragment s -- pulls fragment s of
// that output back in as the next layer's input (each lane re-reads exactly the bytes it stored itself).
template <int D>
__global__ __launch_bounds__(DG_THREADS, 1) void dgrad_pair_kernel(DgradArgs a) {
  using sunerf_ring::Ring;
  constexpr bool SPILL = D > 256;
  constexpr int NT = D / 32, KS = D / 16;
  constexpr int PS = Ring<D>::PAGE_STEPS, NB = KS / PS;            // k-steps per page, pages per tile
  constexpr int PAGE = Ring<D>::PAGE, PIECES = Ring<D>::PIECES;
  constexpr int PF = 2;                             // A fragments requested PF k-steps ahead (across page boundaries)
  constexpr int ACQ = PS - PF;                      // k-step of a page at which the next page is acquired
  constexpr int CW = 2;                             // cos window in tiles (per chunk): 2 x 2 x 2 fragments = 32 registers (4: no faster)
  // the epilogue of a tile (dZ = dH * cos, fp16 pack, stash store) is dealt out over the first k-steps of the NEXT tile as
  // 16 pair micro-ops (2 chunks x 8 register pairs); the last tile of a layer produces the next layer's input fragments
  // KS-2 and KS-1, so everything must be done before k-step KS-2
  constexpr int EPI_PER = (16 + (KS - 2) - 1) / (KS - 2);
  static_assert(PIECES <= ACQ && NT % CW == 0 && KS >= 4 && KS % PS == 0, "pieces are issued before the acquire");
  static_assert(!SPILL || EPI_PER == 1, "reload order assumes one epilogue micro-op per k-step");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ring of 4 pages
  const StashLayout SL(D, a.n_linear);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int n_act = a.n_linear - 1;
  const int n_chunks = (a.S + 31) >> 5;
  const int n_pairs = (n_chunks + 1) >> 1;
  const int64_t n_groups = (a.n_rays + DG_WAVES - 1) / DG_WAVES;
  const float gscale = gscale_from_bits(*a.g_absmax_bits);
  const char* wT_out = a.packedT;
  const char* wT_hidden = a.packedT + (size_t)NT * 1024;
  const unsigned dz_chunk_bytes = (unsigned)n_act * KS * 1024;
  const int n_hidden = a.n_linear - 2;
  const int n_pages = n_hidden * NT * NB;            // pages of the W^T stream per chunk pair
  const size_t spare = (size_t)a.n_rays * n_chunks;  // chunk id of the stash slot nobody reads

  Ring<D> ring;
  const char* cur = smem + lane * 16;                // lane-adjusted base of the page being read
  int slot = 0;
  half8 fhi[PF], flo[PF];
  if (n_pages > 0) {
    ring.init(wT_hidden, (size_t)n_pages * PAGE, (unsigned)(uintptr_t)smem, wave, lane);
    ring.template issue_page<0>();
    ring.template issue_page<0>();
    ring.template issue_page<0>();
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < PF; ++s) { fhi[s] = *(const half8*)(cur + s * 2048); flo[s] = *(const half8*)(cur + s * 2048 + 1024); }
  }

  for (int64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
    const int64_t ray_raw = group * DG_WAVES + wave;
    const bool ray_ok = ray_raw < a.n_rays;
    const int64_t ray = ray_ok ? ray_raw : a.n_rays - 1;
    for (int pr = 0; pr < n_pairs; ++pr) {
      // chunk ids of the pair; a missing second chunk (odd chunk count) and waves without a ray work on the spare slot
      size_t cid[2];
      bool valid[2];
      int si[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = 2 * pr + q;
        const bool have = ray_ok && c < n_chunks;
        cid[q] = have ? (size_t)ray_raw * n_chunks + c : spare;
        si[q] = 32 * c + n;
        valid[q] = have && si[q] < a.S;
      }
      // wave-uniform resource descriptors: activation stash (cos is read) and dZ stash of the two chunks
      const Rsrc sb0 = make_rsrc(a.stash + cid[0] * SL.chunk_bytes(), (unsigned)SL.chunk_bytes());
      const Rsrc sb1 = make_rsrc(a.stash + cid[1] * SL.chunk_bytes(), (unsigned)SL.chunk_bytes());
      const Rsrc dz0 = make_rsrc(a.dz_stash + cid[0] * (size_t)dz_chunk_bytes, dz_chunk_bytes);
      const Rsrc dz1 = make_rsrc(a.dz_stash + cid[1] * (size_t)dz_chunk_bytes, dz_chunk_bytes);

      // cos window: slot U % CW of chunk q holds the cos fragments of tile U of the layer being consumed; after its use
      // it takes tile U + CW -- of the same layer while U + CW < NT, else tile U + CW - NT of the next layer down
      half8 cw0[2 * CW], cw1[2 * CW];
      {
        const int c0 = (int)SL.c_off(n_act - 1);
#pragma unroll
        for (int f = 0; f < 2 * CW; ++f) { cw0[f] = buf_load_nt(sb0, c0 + f * 1024); cw1[f] = buf_load_nt(sb1, c0 + f * 1024); }
      }
      // window slot `ws` (a literal) is free again after the epilogue of tile Up of the layer whose cos is lc: it takes tile
      // Up + CW -- of the same layer while Up + CW < NT, else tile Up + CW - NT of the next layer down (clamped to layer 0 at
      // the very end: never consumed).  Up may be a run-time value (d = 512): only offsets depend on it.
      auto refill = [&](int ws, int Up, int lc) __attribute__((always_inline)) {
        const bool same = Up + CW < NT;
        const int Un = same ? Up + CW : Up + CW - NT;
        const int ln = same ? lc : (lc - 1 >= 0 ? lc - 1 : 0);
        const int off = (int)SL.c_off(ln) + (2 * Un) * 1024;
        cw0[2 * ws] = buf_load_nt(sb0, off);
        cw0[2 * ws + 1] = buf_load_nt(sb0, off + 1024);
        cw1[2 * ws] = buf_load_nt(sb1, off);
        cw1[2 * ws + 1] = buf_load_nt(sb1, off + 1024);
      };

      // pending epilogue: accumulators of the tile just finished; t?? collect the packed dZ fragments
      f32x16 prev0, prev1;
      half8 t00, t01, t10, t11;
      // pair micro-op k (0..15, a literal) of the pending tile Up (cos window slot ws, cos layer lc, output block offset lo):
      // chunk k & 1, register pair k >> 1.  Finished fragments go to the stash and, when `y0` is given (then Up must be a
      // compile-time constant), into the AGPR-resident operand set.  The stores are NOT non-temporal at d = 512: the same
      // wave reads them back one layer later.
      auto epi_op = [&](int k, int ws, int Up, int lc, int lo, bool to_regs, half8* y0, half8* y1) __attribute__((always_inline)) {
        const int q = k & 1, pp = k >> 1;                     // constants after unrolling
        const f32x16& acc = q ? prev1 : prev0;
        const half8& c = (q ? cw1 : cw0)[2 * ws + (pp >> 2)];
        float v0 = sunerf_sat16(acc[2 * pp] * (float)c[(2 * pp) & 7]);
        float v1 = sunerf_sat16(acc[2 * pp + 1] * (float)c[(2 * pp + 1) & 7]);
        const f32x2 vv = {v0, v1};
        half2v pk = __builtin_convertvector(vv, half2v);
        asm volatile("" : "+v"(pk));                           // anchor: keep the micro-op in its k-step
        half8& t = q ? (pp < 4 ? t10 : t11) : (pp < 4 ? t00 : t01);
        t[(2 * pp) & 7] = pk[0];
        t[(2 * pp + 1) & 7] = pk[1];
        if ((pp & 3) == 3) {                                   // a fragment is complete
          const int f = 2 * Up + (pp >> 2);
          if (SPILL) buf_store(t, q ? dz1 : dz0, lo + f * 1024);
          else buf_store_nt(t, q ? dz1 : dz0, lo + f * 1024);
          if (to_regs) {   // (a flag, not `y0 != nullptr`: a null test of a private-array address keeps the array in memory)
            pin_agpr(t);
            (q ? y1 : y0)[f] = t;
          }
        }
        if (k == 15) refill(ws, Up, lc);                       // both chunks are done with this window slot
      };

      half8 dzo0 = {0, 0, 0, 0, 0, 0, 0, 0}, dzo1 = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid[0] && h == 0) {
        const f32x2 g = *(const f32x2*)(a.g_raw + ((size_t)ray * a.S + si[0]) * 2);
        dzo0[0] = (_Float16)(g[0] * gscale); dzo0[1] = (_Float16)(g[1] * gscale);
      }
      if (valid[1] && h == 0) {
        const f32x2 g = *(const f32x2*)(a.g_raw + ((size_t)ray * a.S + si[1]) * 2);
        dzo1[0] = (_Float16)(g[0] * gscale); dzo1[1] = (_Float16)(g[1] * gscale);
      }
      constexpr int YK = SPILL ? 1 : KS;                       // d = 512 has no second set
      half8 xa0[KS], xa1[KS], xb0[YK], xb1[YK];
      // ---- out layer: dH_{L-1} = W_out^T dZ_out, one k-step per tile, A fragments straight from L2; the epilogue of its
      // last tile is left pending for the first hidden tile ----
      {
        const int lo = (n_act - 1) * KS * 1024;
#pragma unroll
        for (int U = 0; U < NT; ++U) {
          const half8 aT = *(const half8*)(wT_out + U * 1024 + lane * 16);
          f32x16 acc0 = {0}, acc1 = {0};
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aT, dzo0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aT, dzo1, acc1, 0, 0, 0);
          prev0 = acc0; prev1 = acc1;
          if (U < NT - 1 || n_hidden == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) epi_op(k, U % CW, U, n_act - 1, lo, true, xa0, xa1);
          }
        }
      }
      // ---- hidden layers, l = n_linear-2 ... 1 : dZ_{l-1} = (W_l^T dZ_l) * cos(Z_{l-1}), both chunks per A fragment ----
      // One tile.  x: dZ_l (input; its last two fragments are still being produced by the pending epilogue during the
      // layer's first tile), y: dZ_{l-1} (output set in registers; nullptr at d = 512, where the output lives in the stash
      // and replaces x).  `first` / `last` / `ws` (window slot of the PENDING tile) are literals at every call site; U
      // itself may be a run-time value at d = 512 (only offsets depend on it there).
      auto tile = [&](int l, int U, bool first, bool last, int ws, half8* x0, half8* x1, half8* y0, half8* y1) __attribute__((always_inline)) {
        const int lo = (l - 1) * KS * 1024;     // this layer's output block in the dZ stash
        const int lo_in = l * KS * 1024;        // the previous layer's (our input's) block
        constexpr int RL = 3;
        half8 rl0[RL], rl1[RL];
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int hb = 0; hb < NB; ++hb) {
          const char* nxt = smem + ((slot + 1) & 3) * PAGE + lane * 16;
#pragma unroll
          for (int s = 0; s < PS; ++s) {
            const int ks = hb * PS + s;
            const int r = ks % PF;
#ifndef SUNERF_DBG_DGRAD_HI_ONLY
#define SUNERF_DBG_DGRAD_HI_ONLY 0     // experiment: single fp16 W^T (systematic 2^-12 weight error, see the header comment)
#endif
            if (!SUNERF_DBG_DGRAD_HI_ONLY) {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(flo[r], x0[ks], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(flo[r], x1[ks], acc1, 0, 0, 0);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fhi[r], x0[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fhi[r], x1[ks], acc1, 0, 0, 0);
            // pending epilogue: tile U-1 of this layer, or (first) the last tile of the layer above, whose output is our x
#pragma unroll
            for (int e = 0; e < EPI_PER; ++e) {
              const int k = ks * EPI_PER + e;
              if (k < 16) {
                if (first) epi_op(k, ws, NT - 1, l, lo_in, true, x0, x1);
                else epi_op(k, ws, U - 1, l - 1, lo, !SPILL, y0, y1);
              }
            }
            if (SPILL && last) {
              // last tile: k-step ks was the last reader of input fragment ks -> it becomes fragment ks of this layer's
              // output (stored by the epilogue of tile ks/2: at the latest by this tile's k-step 15, for ks = 28, 29).
              // Requested at k-step ks, committed to the operand registers RL k-steps later (an immediate commit would
              // park the wave for an L2 round trip in every k-step).
              if (ks >= RL && ks - RL < KS - 2) {
                x0[ks - RL] = rl0[(ks - RL) % RL];
                x1[ks - RL] = rl1[(ks - RL) % RL];
                pin_agpr(x0[ks - RL]); pin_agpr(x1[ks - RL]);
              }
              if (ks < KS - 2) {
                rl0[ks % RL] = buf_load(dz0, lo + ks * 1024);
                rl1[ks % RL] = buf_load(dz1, lo + ks * 1024);
              }
            }
            if (s < PIECES) {   // page + 3 of the stream -> the ring slot everyone left at the last acquire
              if (s == 0) ring.template issue_piece<0>();
              if (s == 1 && PIECES > 1) ring.template issue_piece<(PIECES > 1 ? 1 : 0)>();
              if (s == 2 && PIECES > 2) ring.template issue_piece<(PIECES > 2 ? 2 : 0)>();
              if (s == 3 && PIECES > 3) ring.template issue_piece<(PIECES > 3 ? 3 : 0)>();
              if (s == 4 && PIECES > 4) ring.template issue_piece<(PIECES > 4 ? 4 : 0)>();
              if (s == 5 && PIECES > 5) ring.template issue_piece<(PIECES > 5 ? 5 : 0)>();
              if (s == 6 && PIECES > 6) ring.template issue_piece<(PIECES > 6 ? 6 : 0)>();
              if (s == 7 && PIECES > 7) ring.template issue_piece<(PIECES > 7 ? 7 : 0)>();
            }
            if (s == ACQ) {
              // next page: this wave's pieces of it have landed once at most the 2 younger pages (and whatever was
              // issued after them) are outstanding; then everyone's.  Raw barrier: the cos / dZ traffic stays in flight.
              asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * PIECES) : "memory");
              __builtin_amdgcn_s_barrier();
            }
            {
              const int sn = s + PF;
              const char* q = sn < PS ? cur + sn * 2048 : nxt + (sn - PS) * 2048;
              fhi[r] = *(const half8*)q;
              flo[r] = *(const half8*)(q + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          slot = (slot + 1) & 3;
          cur = nxt;
        }
        if (SPILL && last) {   // the last RL requests
#pragma unroll
          for (int ks = KS; ks < KS + RL; ++ks)
            if (ks - RL >= 0 && ks - RL < KS - 2) {
              x0[ks - RL] = rl0[(ks - RL) % RL];
              x1[ks - RL] = rl1[(ks - RL) % RL];
              pin_agpr(x0[ks - RL]); pin_agpr(x1[ks - RL]);
            }
        }
        prev0 = acc0; prev1 = acc1;
      };
      auto hidden_layer = [&](int l, half8* x0, half8* x1, half8* y0, half8* y1) __attribute__((always_inline)) {
        if constexpr (SPILL) {
          // four tile bodies instead of sixteen: first, a run-time loop over pairs of middle tiles (window slots 0, 1), last
          static_assert(CW == 2 && NT % 2 == 0, "tile pairing follows the two-slot cos window");
          tile(l, 0, true, false, (NT - 1) % CW, x0, x1, x0, x1);
          for (int U = 1; U < NT - 1; U += 2) {
            tile(l, U, false, false, 0, x0, x1, x0, x1);       // pending: tile U - 1 (even)
            tile(l, U + 1, false, false, 1, x0, x1, x0, x1);   // pending: tile U (odd)
          }
          tile(l, NT - 1, false, true, 0, x0, x1, x0, x1);
        } else {
#pragma unroll
          for (int U = 0; U < NT; ++U) tile(l, U, U == 0, false, (U == 0 ? NT - 1 : U - 1) % CW, x0, x1, y0, y1);
        }
      };
      if constexpr (SPILL) {
        for (int l = a.n_linear - 2; l >= 1; --l) hidden_layer(l, xa0, xa1, xa0, xa1);
        if (n_hidden > 0) {   // flush: the last tile of layer 1 has no successor in this pair
#pragma unroll
          for (int k = 0; k < 16; ++k) epi_op(k, (NT - 1) % CW, NT - 1, 0, 0, false, xa0, xa1);
        }
      } else {
        int l = a.n_linear - 2;
        int last = 0;   // which set holds the output of the last hidden layer: 0 = xb, 1 = xa
        for (; l - 1 >= 1; l -= 2) {
          hidden_layer(l, xa0, xa1, xb0, xb1);
          hidden_layer(l - 1, xb0, xb1, xa0, xa1);
          last = 1;
        }
        if (l >= 1) { hidden_layer(l, xa0, xa1, xb0, xb1); last = 0; }
        if (n_hidden > 0) {   // flush: the last tile of layer 1 has no successor in this pair
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            if (last) epi_op(k, (NT - 1) % CW, NT - 1, 0, 0, true, xa0, xa1);
            else epi_op(k, (NT - 1) % CW, NT - 1, 0, 0, true, xb0, xb1);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prefetched pages target our LDS: drain before the workgroup ends
}

// ---- packing of the transposed image ------------------------------------------------------------------------
struct PackTArgs {
  const float* W[SUNERF_MAX_LAYERS];
  int n_linear, D, d_out;
  char* packedT;
  float* sumsq;      // [SUNERF_MAX_LAYERS] at the tail of packedT: sum of squares of every layer's weight (layers >= 1)
};

// sum of squares of the hidden / out layers' weights (the per-layer boosts of the backward chain: sunerf_common.h);
// one workgroup of 1024 threads per layer writes its layer's sum: no atomics, nothing to clear beforehand
__global__ __launch_bounds__(1024) void layer_sumsq_kernel(PackTArgs a) {
  const int l = 1 + blockIdx.x;
  const size_t n = (size_t)((l == a.n_linear - 1) ? a.d_out : a.D) * a.D;
  float acc = 0.f;
  typedef float f4 __attribute__((ext_vector_type(4)));
  if ((((uintptr_t)a.W[l]) & 15) == 0) {          // rows * D is a multiple of 4 (D % 32 == 0)
    const f4* w4 = (const f4*)a.W[l];
    const size_t n4 = n / 4;
    size_t i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {      // four independent 16-byte loads in flight per lane: one CU streams a layer
      const f4 w0 = w4[i], w1 = w4[i + 1024], w2 = w4[i + 2048], w3 = w4[i + 3072];
      acc += (w0[0] * w0[0] + w0[1] * w0[1] + w0[2] * w0[2] + w0[3] * w0[3]) + (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2] + w1[3] * w1[3])
           + (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2] + w2[3] * w2[3]) + (w3[0] * w3[0] + w3[1] * w3[1] + w3[2] * w3[2] + w3[3] * w3[3]);
    }
    for (; i < n4; i += 1024) {
      const f4 w = w4[i];
      acc += w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
    }
  } else {
    for (size_t i = threadIdx.x; i < n; i += 1024) acc += a.W[l][i] * a.W[l][i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += part[i];
    a.sumsq[l] = t;
  }
}

__global__ void pack_mlp_t_kernel(PackTArgs a) {
  const int NT = a.D / 32, KS = a.D / 16;
  const size_t n_out = (size_t)NT * 512;                               // halfs of the out^T part
  const size_t n_hid = (size_t)(a.n_linear - 2) * NT * KS * 512;     // (hi, lo) pairs
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out + n_hid) return;
  _Float16* dst = (_Float16*)a.packedT;
  float w = 0.f;
  if (idx < n_out) {
    const int U = (int)(idx / 512), lane = (int)(idx % 512) / 8, e = (int)(idx % 8);
    const int m = lane & 31, h = lane >> 5;
    if (h == 0 && e < a.d_out) w = a.W[a.n_linear - 1][(size_t)e * a.D + 32 * U + m];
    dst[idx] = (_Float16)ldexpf(w, sunerf_bwd_boost(a.sumsq[a.n_linear - 1], a.D));
    return;
  } else {
    size_t r = idx - n_out;
    const int li = (int)(r / ((size_t)NT * KS * 512));   // 0 -> layer n_linear-2, 1 -> n_linear-3, ...
    r %= (size_t)NT * KS * 512;
    const int l = a.n_linear - 2 - li;
    const int U = (int)(r / ((size_t)KS * 512)); r %= (size_t)KS * 512;
    const int s = (int)(r / 512); r %= 512;
    const int lane = (int)(r / 8), e = (int)(r % 8);
    const int m = lane & 31, h = lane >> 5;
    // A[m][k] = W_l^T[in = 32U+m][out = kmap_hidden(s,h,e)] = W_l[out][in]
    w = ldexpf(a.W[l][(size_t)kmap_hidden(s, h, e) * a.D + 32 * U + m], sunerf_bwd_boost(a.sumsq[l], a.D));
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    _Float16* blk = dst + n_out + ((size_t)li * NT + U) * KS * 1024;     // block of KS k-steps x 1024 halfs
    blk[((size_t)(s * 2 + 0) * 64 + lane) * 8 + e] = hi;
    blk[((size_t)(s * 2 + 1) * 64 + lane) * 8 + e] = lo;
  }
}

}  // namespace

extern "C" size_t sunerf_packed_mlp_t_bytes(int d_filter, int n_linear) {
  if (d_filter <= 0 || d_filter % 32 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return 0;
  const size_t NT = d_filter / 32, KS = d_filter / 16;
  return NT * 1024 + (size_t)(n_linear - 2) * NT * KS * 2048 + SUNERF_MAX_LAYERS * sizeof(float);   // + the layers' sums of squares
}

extern "C" int sunerf_pack_mlp_t(const float* const* weights_host, int n_linear, int d_filter, int d_out, void* packedT,
                                 void* stream) {
  if (!weights_host || !packedT) return SUNERF_E_BADARG;
  if (d_filter <= 0 || d_filter % 32 || n_linear < 2 || n_linear > SUNERF_MAX_LAYERS || d_out < 1 || d_out > 8)
    return SUNERF_E_UNSUPPORTED;
  PackTArgs a;
  for (int i = 0; i < n_linear; ++i) {
    if (!weights_host[i]) return SUNERF_E_BADARG;
    a.W[i] = weights_host[i];
  }
  a.n_linear = n_linear; a.D = d_filter; a.d_out = d_out; a.packedT = (char*)packedT;
  const size_t NTh = d_filter / 32, KSh = d_filter / 16;
  const size_t total = NTh * 512 + (size_t)(n_linear - 2) * NTh * KSh * 512;   // threads: out halfs + hidden (hi, lo) pairs
  a.sumsq = (float*)((char*)packedT + NTh * 1024 + (size_t)(n_linear - 2) * NTh * KSh * 2048);
  const int threads = 256;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(layer_sumsq_kernel, dim3(n_linear - 1), dim3(1024), 0, (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  hipLaunchKernelGGL(pack_mlp_t_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                     (hipStream_t)stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t sunerf_dz_stash_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear) {
  if (n_rays < 0 || n_samples < 1 || d_filter < 32 || d_filter % 32 || n_linear < 2) return 0;
  const int64_t chunks = n_rays * ((n_samples + 31) / 32) + 1;
  return (size_t)chunks * (size_t)(n_linear - 1) * (d_filter / 16) * 1024;
}

extern "C" int sunerf_emission_integral_fwd(const float* raw, const float* z_vals, const float* rays_d, int64_t n_rays,
                                            int n_samples, float* image, float* weights, float* absorption, void* stream) {
  if (n_rays < 0 || n_samples < 2) return SUNERF_E_BADARG;
  if (n_rays == 0) return 0;
  if (!raw || !z_vals || !rays_d || !image || !weights || !absorption) return SUNERF_E_BADARG;
  const int64_t blocks = (n_rays + IB_RAYS - 1) / IB_RAYS;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(integral_fwd_kernel, dim3((unsigned)blocks), dim3(IB_THREADS), 0, (hipStream_t)stream, raw, z_vals, rays_d,
                     n_rays, n_samples, image, weights, absorption);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_emission_integral_bwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                                            const float* g_image, const float* g_reg, const float* g_weights,
                                            const float* g_absorption, float g_reg_const, float reg_radius,
                                            int64_t n_rays, int n_samples, float* g_raw, void* g_absmax, void* stream) {
  if (n_rays < 0 || n_samples < 2 || !g_absmax) return SUNERF_E_BADARG;
  if (n_rays > 0 && (!raw || !z_vals || !rays_o || !rays_d || !g_image || !g_raw)) return SUNERF_E_BADARG;
  const size_t lds = (size_t)IB_RAYS * 3 * (size_t)((n_samples + 31) / 32 * 32) * sizeof(float);
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  hipError_t e = hipMemsetAsync(g_absmax, 0, 4, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (n_rays == 0) return 0;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute((const void*)integral_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  int64_t blocks = (n_rays + IB_RAYS - 1) / IB_RAYS;
  if (blocks > IB_MAX_GRID) blocks = IB_MAX_GRID;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(integral_bwd_kernel, dim3((unsigned)blocks), dim3(IB_THREADS), lds, (hipStream_t)stream, raw, z_vals,
                     rays_o, rays_d, g_image, g_reg, g_weights, g_absorption, g_reg_const, reg_radius, n_rays, n_samples, g_raw,
                     (unsigned*)g_absmax);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

template <int D>
static int launch_dgrad(const DgradArgs& a, hipStream_t stream) {
  const int64_t n_groups = (a.n_rays + DG_WAVES - 1) / DG_WAVES;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  cus = sunerf_grid_cap("SUNERF_GRID_CAP_DGRAD", cus);
  const unsigned grid = (unsigned)(n_groups < cus ? n_groups : cus);
  const size_t lds = (size_t)sunerf_ring::Ring<D>::RING;
  hipError_t e = hipFuncSetAttribute((const void*)dgrad_pair_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(dgrad_pair_kernel<D>, dim3(grid), dim3(DG_THREADS), lds, stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_mlp_dgrad(const void* packedT, int d_filter, int n_linear, const float* g_raw, const void* g_absmax,
                                const void* act_stash, void* dz_stash, int64_t n_rays, int n_samples, void* stream) {
  if (n_rays < 0 || n_samples < 2) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return SUNERF_E_UNSUPPORTED;
  if (n_rays == 0) return 0;
  if (!packedT || !g_raw || !g_absmax || !act_stash || !dz_stash) return SUNERF_E_BADARG;
  DgradArgs a;
  a.packedT = (const char*)packedT; a.g_raw = g_raw; a.g_absmax_bits = (const unsigned*)g_absmax;
  a.stash = (const char*)act_stash; a.dz_stash = (char*)dz_stash; a.n_rays = n_rays; a.S = n_samples; a.n_linear = n_linear;
  switch (d_filter) {
    case 64: return launch_dgrad<64>(a, (hipStream_t)stream);
    case 128: return launch_dgrad<128>(a, (hipStream_t)stream);
    case 256: return launch_dgrad<256>(a, (hipStream_t)stream);
    case 512: return launch_dgrad<512>(a, (hipStream_t)stream);
    default: return SUNERF_E_UNSUPPORTED;
  }
}
