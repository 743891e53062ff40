// Fused forward render pass for gfx950:  points -> (x,y,z,t) -> positional encoding -> sine MLP -> emission /
// absorption integral, one launch, nothing per-sample wider than 8 bytes ever written to HBM (inference).
//
// Replaces (per coarse or fine pass) sampling.py:100, base_tracing.py:64-65/83-84/118-129, model.py:123-132,
// model.py:44-57, emission.py:14-54, base_tracing.py:135-156 and the epilogues base_tracing.py:99-110.
//
// Work decomposition
//   workgroup = 4 waves (one per SIMD, ~1 wave/SIMD occupancy by register budget); each wave owns ONE ray at a
//   time and walks its samples in chunks of 32 (one MFMA column tile).  The 4 waves run the layer sequence in
//   lock-step because they share the weight stream: each (layer, 32-row tile) block of fp16 hi/lo A fragments
//   is staged global(L2) -> LDS once per workgroup and read by all 4 waves with one ds_read_b128 per MFMA operand.
//   Activations stay in registers for the whole MLP (sunerf_common.h explains the transposed formulation).
//   The per-ray integral is a per-wavefront exclusive scan (product) over each 32-sample chunk with a scalar
//   carry between chunks.
#include "sunerf_common.h"
#include "../../include/sunerf_hip.h"

namespace {

constexpr int WAVES = 4;
constexpr int THREADS = WAVES * 64;
constexpr int NSLOT = 3;  // LDS ring slots for weight blocks

struct RenderArgs {
  const char* packed;
  const float* rays_o;
  const float* rays_d;
  const float* times;
  const float* z_vals;
  int64_t n_rays;
  int S;
  int n_linear;
  float* image;
  float* weights;
  float* absorption;
  float* raw;
  float* height_map;
  float* absorption_map;
  float* regularization;
  float reg_radius;
  char* stash;
};

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// x = hi + lo with hi, lo fp16 (round-to-nearest both): |x - hi - lo| <= 2^-22 |x| (2^-25 absolute in the fp16
// subnormal range, which the gfx950 MFMA honours)
__device__ __forceinline__ void split2(float x0, float x1, half2v& hi, half2v& lo) {
  hi[0] = (_Float16)x0; hi[1] = (_Float16)x1;
  lo[0] = (_Float16)(x0 - (float)hi[0]); lo[1] = (_Float16)(x1 - (float)hi[1]);
}

// sin(x), x in radians, via the hardware v_sin_f32 (argument in revolutions, measured max abs error 1.3e-7 on
// [-40, 40] revolutions on gfx950; valid to +-256 revolutions, far beyond any hidden pre-activation)
__device__ __forceinline__ float sin_rad(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }

// 48 encoding slots of this lane half (kmap_encoding): 40 x sin or cos (2^(k-1) x_c), then raw x_c / zero pad.
// The argument is reduced exactly: x/(2 pi) is formed as an unevaluated sum p + e (|error| ~ 2^-48 |p|), the octave
// scaling by 2^(k-1) is exact, frac() is exact, so the fractional revolution handed to v_sin_f32 carries ~2^-25
// absolute error even at the top frequency (the reference's fp32 argument 2^8 x is itself exact).
template <typename F>
__device__ __forceinline__ void encode_point(const float v[4], int h, F&& emit /* (slot q, value) */) {
  constexpr float INV2PI_HI = 0.15915494309189535f;
  constexpr float INV2PI_LO = (float)(0.15915494309189533576888 - (double)INV2PI_HI);
  float p[4], e[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    p[c] = v[c] * INV2PI_HI;
    e[c] = __builtin_fmaf(v[c], INV2PI_HI, -p[c]);
    e[c] = __builtin_fmaf(v[c], INV2PI_LO, e[c]);
  }
  const float phase = h ? 0.25f : 0.f;  // cos(x) = sin(x + 1/4 revolution)
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float sc = (k == 0) ? 0.5f : (float)(1 << (k - 1));   // f_k / scale_factor = 2^k / 2
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float s = p[c] * sc;
      float f = s - __builtin_rintf(s);
      f = __builtin_fmaf(e[c], sc, f) + phase;
      emit(4 * k + c, __builtin_amdgcn_sinf(f));
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) emit(40 + c, h ? 0.f : v[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) emit(44 + c, 0.f);
}

__device__ __forceinline__ float shfl_up32(float v, int d, int n) {  // within the 32-lane half; n = lane & 31
  const float o = __shfl_up(v, d, 32);
  return o;
}

// ---- weight stream: global (L2-resident packed image) -> LDS ring, asynchronous LDS-DMA ------------------------
// The (layer, tile) blocks are consumed in a fixed cyclic order (all blocks of the MLP once per 32-sample chunk),
// so the stream is a ring of NSLOT slots filled two blocks ahead by `global_load_lds_dwordx4` (no VGPR staging):
// each of the 4 waves issues a quarter of a block's 1 KiB pieces.  Per block step:
//     s_waitcnt vmcnt(#pieces of the NEXT block)   -> this wave's pieces of the CURRENT block have landed
//     s_barrier                                    -> everyone's pieces landed; everyone is done reading the
//                                                     previous block, whose slot is the one refilled next
//     issue the DMA of block (current + 2)
//     compute on the current slot
// The DMA is inline asm on purpose: hipcc does not count it, so it neither drains it with vmcnt(0) at barriers nor
// in front of unrelated LDS reads; our counted waits stay valid when compiler-issued loads/stores interleave
// (extra younger operations only make `vmcnt(N)` stricter).
template <int D>
struct Ring {
  static constexpr int BLK0 = SUNERF_KS0 * 2048;
  static constexpr int BLK = (D / 16) * 2048;
  static constexpr int SLOT = BLK > BLK0 ? BLK : BLK0;   // ring slot size
  static constexpr int CNT0 = BLK0 / 1024 / WAVES;   // DMA pieces per wave, in-layer block
  static constexpr int CNT = BLK / 1024 / WAVES;     // hidden / out block
  static_assert(BLK0 % (1024 * WAVES) == 0 && BLK % (1024 * WAVES) == 0, "blocks must split evenly over the waves");
  const char* packed;
  unsigned lds_base;   // LDS byte address of slot 0
  int nb, nt;          // blocks per MLP pass, of which the first nt are in-layer blocks
  int cur, pf;         // next block to consume / to prefetch (0..nb-1)
  int cslot, pslot;    // their ring slots
  int wave, lane;

  __device__ __forceinline__ void issue() {
    const bool small = pf < nt;
    const size_t off = small ? (size_t)pf * BLK0 : (size_t)nt * BLK0 + (size_t)(pf - nt) * BLK;
    const int cnt = small ? CNT0 : CNT;
    const char* src = packed + off + (size_t)(wave * cnt) * 1024 + lane * 16;
    unsigned dst = lds_base + pslot * SLOT + wave * cnt * 1024;
    for (int i = 0; i < cnt; ++i) {
      unsigned keep;
      const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst_u) : "memory");
      src += 1024; dst += 1024;
    }
    pf = (pf + 1 == nb) ? 0 : pf + 1;
    pslot = (pslot + 1 == NSLOT) ? 0 : pslot + 1;
  }
  __device__ __forceinline__ void prologue() { issue(); issue(); }
  // returns the LDS byte offset (from slot 0) of the block to consume
  __device__ __forceinline__ int acquire() {
    const int nxt = (cur + 1 == nb) ? 0 : cur + 1;
    if (nxt < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CNT0) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CNT) : "memory");
    __builtin_amdgcn_s_barrier();
    issue();
    const int off = cslot * SLOT;
    cur = nxt;
    cslot = (cslot + 1 == NSLOT) ? 0 : cslot + 1;
    return off;
  }
};

// one 32-row output tile: acc = bias + W_tile * X   (three fp16 MFMAs per k-step)
template <int KSTEPS>
__device__ __forceinline__ f32x16 tile_mma(const char* slot, const float* bias_tile, int lane, int h,
                                           const half8* xhi, const half8* xlo) {
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *(const f32x4*)(bias_tile + 8 * j + 4 * h);
    acc[4 * j + 0] = b[0]; acc[4 * j + 1] = b[1]; acc[4 * j + 2] = b[2]; acc[4 * j + 3] = b[3];
  }
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const half8 ahi = *(const half8*)(slot + s * 2048 + lane * 16);
    const half8 alo = *(const half8*)(slot + s * 2048 + 1024 + lane * 16);
    acc = mfma16(alo, xhi[s], acc);
    acc = mfma16(ahi, xlo[s], acc);
    acc = mfma16(ahi, xhi[s], acc);
  }
  return acc;
}

// sin() + hi/lo split of one accumulator tile into the two k-step fragments it forms for the next layer
__device__ __forceinline__ void activate_tile(const f32x16& acc, half8& hi0, half8& lo0, half8& hi1, half8& lo1) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    half2v a, b;
    split2(sin_rad(acc[j]), sin_rad(acc[j + 1]), a, b);
    hi0[j] = a[0]; hi0[j + 1] = a[1]; lo0[j] = b[0]; lo0[j + 1] = b[1];
    split2(sin_rad(acc[8 + j]), sin_rad(acc[8 + j + 1]), a, b);
    hi1[j] = a[0]; hi1[j + 1] = a[1]; lo1[j] = b[0]; lo1[j + 1] = b[1];
  }
}

template <int D>
struct Mlp {
  static constexpr int NT = D / 32;
  static constexpr int KS = D / 16;
  static constexpr int XK = KS > SUNERF_KS0 ? KS : SUNERF_KS0;  // fragments per register set
  static constexpr int BLK0 = SUNERF_KS0 * 2048;
  static constexpr int BLK = KS * 2048;

  // one layer: X (KIN k-steps) -> Y (KS k-steps); weight blocks come through the LDS ring
  template <int KIN>
  static __device__ __forceinline__ void layer(Ring<D>& ring, const char* slots, const float* bias, int lane, int h,
                                               const half8* xhi, const half8* xlo, half8* yhi, half8* ylo) {
#pragma unroll
    for (int U = 0; U < NT; ++U) {
      const char* slot = slots + ring.acquire();
      const f32x16 acc = tile_mma<KIN>(slot, bias + 32 * U, lane, h, xhi, xlo);
      activate_tile(acc, yhi[2 * U], ylo[2 * U], yhi[2 * U + 1], ylo[2 * U + 1]);
    }
  }

  static __device__ __forceinline__ f32x16 out_layer(Ring<D>& ring, const char* slots, const float* bias, int lane,
                                                     int h, const half8* xhi, const half8* xlo) {
    const char* slot = slots + ring.acquire();
    return tile_mma<KS>(slot, bias, lane, h, xhi, xlo);
  }
};

template <int D>
__global__ __launch_bounds__(THREADS, 1) void render_fwd_kernel(RenderArgs a) {
  using M = Mlp<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PackedLayout L(D, a.n_linear);
  char* slot = smem;                                            // ring of NSLOT weight-block slots
  float* bias = (float*)(smem + (size_t)NSLOT * Ring<D>::SLOT);  // n_bias floats
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;

  {  // biases -> LDS once
    const float* gb = (const float*)(a.packed + L.bias_off());
    for (int i = tid; i < (int)L.n_bias(); i += THREADS) bias[i] = gb[i];
  }
  __syncthreads();
  Ring<D> ring;
  ring.packed = a.packed;
  ring.lds_base = (unsigned)(uintptr_t)slot;
  ring.nt = M::NT;
  ring.nb = M::NT * (a.n_linear - 1) + 1;
  ring.cur = 0; ring.pf = 0; ring.cslot = 0; ring.pslot = 0;
  ring.wave = __builtin_amdgcn_readfirstlane(wave); ring.lane = lane;
  ring.prologue();
  const int S = a.S;
  const int n_chunks = (S + 31) >> 5;
  const int64_t n_groups = (a.n_rays + WAVES - 1) / WAVES;

  for (int64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
    const int64_t ray_raw = group * WAVES + wave;
    const bool ray_ok = ray_raw < a.n_rays;
    const int64_t ray = ray_ok ? ray_raw : a.n_rays - 1;
    const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    const float tm = a.times[ray];
    const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);   // torch.norm(rays_d), emission.py:26
    const float* zrow = a.z_vals + ray * S;

    float carry_T = 1.f;      // product of (absorption + 1e-10) over all previous samples of the ray
    float carry_z = 0.f;      // z of the previous chunk's last sample
    float sum_em = 0.f, sum_abs = 0.f, sum_h = 0.f;

    for (int c = 0; c < n_chunks; ++c) {
      const int i = 32 * c + n;
      const bool valid = i < S;
      const float z = zrow[valid ? i : S - 1];
      // sampling.py:100 -- product and sum rounded separately
      const float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;
      const float v[4] = {px, py, pz, tm};

      half8 xa_hi[M::XK], xa_lo[M::XK], xb_hi[M::XK], xb_lo[M::XK];
      {  // positional encoding straight into the in-layer B fragments (k-steps 0..5 of xb)
        encode_point(v, h, [&](int q, float val) {
          const _Float16 hi = (_Float16)val;
          xb_hi[q >> 3][q & 7] = hi;
          xb_lo[q >> 3][q & 7] = (_Float16)(val - (float)hi);
        });
      }
      // in layer: 84(96) -> D
      M::template layer<SUNERF_KS0>(ring, slot, bias, lane, h, xb_hi, xb_lo, xa_hi, xa_lo);
      // hidden layers, ping-pong between the two register sets
      int l = 1;
      for (; l + 1 < a.n_linear - 1; l += 2) {
        M::template layer<M::KS>(ring, slot, bias + (size_t)l * D, lane, h, xa_hi, xa_lo, xb_hi, xb_lo);
        M::template layer<M::KS>(ring, slot, bias + (size_t)(l + 1) * D, lane, h, xb_hi, xb_lo, xa_hi, xa_lo);
      }
      f32x16 out;
      const float* obias = bias + (size_t)(a.n_linear - 1) * D;
      if (l < a.n_linear - 1) {
        M::template layer<M::KS>(ring, slot, bias + (size_t)l * D, lane, h, xa_hi, xa_lo, xb_hi, xb_lo);
        out = M::out_layer(ring, slot, obias, lane, h, xb_hi, xb_lo);
      } else {
        out = M::out_layer(ring, slot, obias, lane, h, xa_hi, xa_lo);
      }

      // ---- emission / absorption integral for this chunk (emission.py:14-54); lanes 0..31 hold rows 0,1 ----
      const float r0 = out[0], r1 = out[1];   // meaningful on h == 0 lanes
      // dists: z_i - z_{i-1}, first one duplicated (emission.py:21-22)
      float zprev = __shfl_up(z, 1, 32);
      const float znext = __shfl_down(z, 1, 32);
      if (n == 0) zprev = carry_z;
      float dzv = (i == 0) ? (znext - z) : (z - zprev);
      const float dist = dzv * dnorm;
      const float inten = expf(r0) * dist;
      const float absn = expf(-fmaxf(r1, 0.f) * dist);
      float pr = valid ? (absn + 1e-10f) : 1.f;
      // inclusive product scan over the 32 lanes of the half
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const float o = __shfl_up(pr, d, 32);
        if (n >= d) pr *= o;
      }
      float excl = __shfl_up(pr, 1, 32);
      if (n == 0) excl = 1.f;
      const float T = carry_T * excl;
      const float em = valid ? inten * T : 0.f;
      const float pdist = sqrtf((px * px + py * py) + pz * pz);   // base_tracing.py:102
      float s_em = em, s_abs = valid ? (1.f - absn) : 0.f, s_h = em * pdist;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) {
        s_em += __shfl_xor(s_em, d, 32);
        s_abs += __shfl_xor(s_abs, d, 32);
        s_h += __shfl_xor(s_h, d, 32);
      }
      sum_em += s_em; sum_abs += s_abs; sum_h += s_h;
      carry_T *= __shfl(pr, 31, 32);
      carry_z = __shfl(z, 31, 32);
      if (ray_ok && valid && h == 0) {
        const int64_t o = ray * S + i;
        a.weights[o] = em;  // un-normalised; finalised below by the same lane
        a.absorption[o] = absn;
        if (a.raw) { a.raw[o * 2 + 0] = r0; a.raw[o * 2 + 1] = r1; }
        if (a.regularization) a.regularization[o] = fmaxf(pdist - a.reg_radius, 0.f) * (1.f - absn);
      }
    }
    // ---- per-ray finalisation: normalise weights (emission.py:49-50), per-ray outputs ----
    if (ray_ok && h == 0) {
      const float denom = sum_em + 1e-10f;
      for (int c = 0; c < n_chunks; ++c) {
        const int i = 32 * c + n;
        if (i < S) { const int64_t o = ray * S + i; a.weights[o] = a.weights[o] / denom; }
      }
      if (n == 0) {
        a.image[ray] = sum_em;
        if (a.height_map) a.height_map[ray] = sum_h / denom;
        if (a.absorption_map) a.absorption_map[ray] = sum_abs;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two prefetches still in flight target our LDS: drain
}

template <int D>
int launch_render(const RenderArgs& a, hipStream_t stream) {
  const PackedLayout L(D, a.n_linear);
  const size_t lds = (size_t)NSLOT * Ring<D>::SLOT + L.n_bias() * 4;
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  hipError_t e = hipFuncSetAttribute((const void*)render_fwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t n_groups = (a.n_rays + WAVES - 1) / WAVES;
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const unsigned grid = (unsigned)(n_groups < cus ? n_groups : cus);
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(render_fwd_kernel<D>, dim3(grid), dim3(THREADS), lds, stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t sunerf_act_stash_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear) {
  if (n_rays < 0 || n_samples < 1 || d_filter < 32 || n_linear < 2) return 0;
  const int64_t chunks = n_rays * ((n_samples + 31) / 32);
  return (size_t)chunks * (size_t)(n_linear - 1) * d_filter * 32 * 4;
}

extern "C" int sunerf_emission_render_fwd(const void* packed, int d_filter, int n_linear, const float* rays_o,
                                          const float* rays_d, const float* times, const float* z_vals,
                                          int64_t n_rays, int n_samples, float* image, float* weights,
                                          float* absorption, float* raw, float* height_map, float* absorption_map,
                                          float* regularization, float reg_radius, void* act_stash, void* stream) {
  if (!packed || !rays_o || !rays_d || !times || !z_vals || !image || !weights || !absorption) return SUNERF_E_BADARG;
  if (n_rays < 0 || n_samples < 2) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return SUNERF_E_UNSUPPORTED;
  if (n_rays == 0) return 0;
  RenderArgs a;
  a.packed = (const char*)packed; a.rays_o = rays_o; a.rays_d = rays_d; a.times = times; a.z_vals = z_vals;
  a.n_rays = n_rays; a.S = n_samples; a.n_linear = n_linear; a.image = image; a.weights = weights;
  a.absorption = absorption; a.raw = raw; a.height_map = height_map; a.absorption_map = absorption_map;
  a.regularization = regularization; a.reg_radius = reg_radius; a.stash = (char*)act_stash;
  switch (d_filter) {
    case 64: return launch_render<64>(a, (hipStream_t)stream);
    case 128: return launch_render<128>(a, (hipStream_t)stream);
    case 256: return launch_render<256>(a, (hipStream_t)stream);
    default: return SUNERF_E_UNSUPPORTED;
  }
}
