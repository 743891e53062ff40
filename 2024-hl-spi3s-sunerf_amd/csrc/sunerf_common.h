// Shared definitions of the gfx950 SuNeRF renderer kernels (device + host).
//
// Packed MLP image (produced by pack.hip, consumed by render_fwd.hip / render_bwd.hip)
// ----------------------------------------------------------------------------------
// The field MLP of the reference (NeRF, sunerf/model/model.py:12-57) is evaluated TRANSPOSED on the
// matrix cores:  H_{l+1}^T = W_l * H_l^T  with  A = W_l (rows = output features),  B = H_l^T (columns =
// samples), v_mfma_f32_32x32x16_f16.  With that orientation the 32x32 fp32 accumulator tile of layer l
// (column = sample on the lane, rows = features in the 16 registers) IS, after sin() and conversion to
// fp16, the B operand of layer l+1 -- activations never leave the register file.
//
// fp32 accuracy on fp16 matrix cores: every operand x is split x = hi + lo (both fp16, |x-hi-lo| <= 2^-22 |x|,
// fp16 subnormals are honoured by the MFMA on gfx950) and a product is formed from three MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulate).  That is 3/16 of the cost of the exact-fp32 MFMA.
//
// A "block" is the A-operand data of one (layer, 32-row output tile): for each 16-deep k-step two 1 KiB
// fragments (hi, lo) in lane order (lane l = 16 bytes at offset 16*l), so that one ds_read_b128 per lane
// fetches an MFMA A operand and a block can be copied global -> LDS linearly.
//
//   k order inside a k-step (lane half h = lane>>5, element e = 0..7), hidden layers:
//       input feature = 32*(s>>1) + 16*(s&1) + 8*(e>>2) + 4*h + (e&3)
//     i.e. exactly the feature that accumulator register 8*(s&1)+e of tile s>>1 holds on lane half h.
//   in_layer (positional encoding, 84 inputs padded to 96 = 6 k-steps), slot q = 8*s + e:
//       q < 40      : lane half 0 -> sin feature (reference column 4 + q), half 1 -> cos (column 44 + q)
//       40 <= q < 44: lane half 0 -> raw coordinate q-40 (reference column q-40), half 1 -> zero
//       q >= 44     : zero
//
// Byte layout of the packed image for (D = d_filter, n_linear Linear layers):
//   weight STREAM, in consumption order, every k-step exactly 2048 B (hi fragment, lo fragment):
//     [in_layer : D/32 tiles x 6 k-steps][hidden l=1..n_linear-2 : D/32 tiles x D/16 k-steps each]
//     [out_layer: 1 tile x D/16 k-steps, rows >= d_out zero]
//   then [bias fp32: (n_linear-1)*D, then 32 (out layer)]
//   The render kernels treat the stream as a byte FIFO that is DMA'd page by page (page = (D/16)*2048 B) into an
//   LDS ring of 4 pages; the stream length is a multiple of the page size (and, for D >= 128, of the ring size, so
//   that every MLP pass starts at ring offset 0 and all LDS read addresses are compile-time constants).
//   All layers except out_layer are stored pre-multiplied by 1/(2 pi) (weights and biases): the accumulator then
//   holds the pre-activation in REVOLUTIONS, which is what the hardware v_sin_f32 takes -- one multiply per
//   activation saved.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define SUNERF_KS0 6  // k-steps of the in layer (96 = 84 padded)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PackedLayout {
  int D, n_linear, NT, KS;
  size_t blk0, blk;
  __host__ __device__ PackedLayout(int d, int nl) : D(d), n_linear(nl), NT(d / 32), KS(d / 16),
                                                     blk0((size_t)SUNERF_KS0 * 2048), blk((size_t)(d / 16) * 2048) {}
  __host__ __device__ size_t block_off(int l, int U) const {
    if (l == 0) return (size_t)U * blk0;
    size_t base = (size_t)NT * blk0;
    if (l < n_linear - 1) return base + ((size_t)(l - 1) * NT + U) * blk;
    return base + (size_t)(n_linear - 2) * NT * blk;
  }
  __host__ __device__ size_t pass_bytes() const { return (size_t)NT * blk0 + (size_t)(n_linear - 2) * NT * blk + blk; }
  __host__ __device__ size_t ring_bytes() const { return 4 * blk; }
  __host__ __device__ size_t stream_bytes() const { return pass_bytes(); }   // always a multiple of the page size
  __host__ __device__ size_t bias_off() const { return stream_bytes(); }
  __host__ __device__ size_t n_bias() const { return (size_t)(n_linear - 1) * D + 32; }
  // "fp8c" stream format (below; d <= 256, SUNERF_PRECISION_FAST): per-layer scale exponents sh[l] (int[16]) and the max |w| they derive from
  // (float bit patterns, uint[16]; scratch of the pack kernels)
  __host__ __device__ size_t scale_off() const { return bias_off() + n_bias() * 4; }
  __host__ __device__ size_t absmax_off() const { return scale_off() + 64; }
  __host__ __device__ size_t total_bytes() const { return absmax_off() + 64; }
};

// "fp8c" stream format of the hidden and out layers (d <= 256).  A product x*w with x = xh + xl, w = wh + wl (fp16 heads,
// exact fp32 remainders) is formed as  xh*wh  [v_mfma_f32_32x32x16_f16]  +  fp8(xh)*fp8(wl)  +  fp8(xl)*fp8(wh)
// [v_mfma_scale_f32_32x32x64_f8f6f4, e4m3, block scales]: the two correction terms are 2^-12 of the product, so their fp8
// rounding (2^-4) is a 2^-16 relative error (measured 1.3e-6 on the output of an 8 x 256 net) -- and one 64-deep fp8
// instruction replaces four 16-deep fp16 ones at twice the rate.  Per 64 input features ("group" = 4 k-steps) the stream
// holds 8 KiB:
//   [4 x 1 KiB : wh fragments of the 4 k-steps, as in the classic format]
//   [2 x 1 KiB : fp8(wl * 2^(sh+11)), lane l = 16 bytes per KiB; byte j = 8 sl + e of the lane's 32 <-> k-step 4g + sl,
//                element e (same feature order as the fp16 fragments); first KiB = bytes 0..15, second = bytes 16..31]
//   [2 x 1 KiB : fp8(wh * 2^sh), same arrangement]
// with sh = 8 - ceil(log2 max|w|) per layer (|wh 2^sh| < 256, |wl 2^(sh+11)| <= 128; e4m3 saturates at 448).  The in
// layer (96 = 6 k-steps, not a multiple of 4) keeps the classic hi | lo format.
constexpr int SUNERF_GROUP_BYTES = 8192;

// Activation stash written by the training forward pass and read by the backward kernels.  Everything is kept in
// MFMA B-fragment order (1 KiB per fragment: lane l = 16 bytes = 8 x 16 bit at offset 16 l; element e of lane half h of
// fragment s is feature kmap_hidden(s, h, e) of sample l & 31), i.e. exactly the registers the forward kernel holds,
// so a fragment is one fully coalesced 1 KiB store.  Per 32-sample chunk, two formats (include/sunerf_hip.h: SUNERF_STASH_*):
//   FP16  [enc : 6 fragments, fp16(hi) of the encoded input (kmap_encoding order)]
//         for every activation layer l = 0 .. n_linear-2:  [H_l : D/16 fragments, fp16(sin)] [C_l : D/16 fragments, fp16(cos)]
//         -- what the two-kernel backward (render_bwd.hip, wgrad.hip) reads: 8.2 KB per sample of an 8 x 256 network
//   PHASE [enc : 6 fragments] and for every activation layer [P_l : D/16 fragments, 16-bit PHASE of the pre-activation]  [r4]
//         -- what the layer-pipelined backward (bwd_pipe.hip) reads: 4.1 KB per sample.  The phase is frac(z / 2 pi) in [0, 1)
//         as an unsigned normalised 16-bit code (v_cvt_pknorm_u16_f32: round(65535 f)).  The backward needs sin(z) (weight
//         gradients) AND cos(z) (data gradients): one phase gives both to 2^-17 of a revolution = 4.8e-5 absolute (the fp16
//         sin / cos fragments carry 2^-12 relative each), in half the bytes.  Decoding is v_cvt_f32_u32 on a 16-bit half + v_mul +
//         v_sin / v_cos -- done by the data-gradient waves of bwd_pipe.hip in the time they used to wait at the barrier.
struct StashLayout {
  int KS, n_act;
  bool phase;
  __host__ __device__ StashLayout(int d, int n_linear, bool phase_format = false) : KS(d / 16), n_act(n_linear - 1), phase(phase_format) {}
  __host__ __device__ size_t layer_bytes() const { return (size_t)(phase ? 1 : 2) * KS * 1024; }
  __host__ __device__ size_t chunk_bytes() const { return (size_t)SUNERF_KS0 * 1024 + (size_t)n_act * layer_bytes(); }
  // first fragment of layer l's block: H_l (FP16 format) / P_l (PHASE format)
  __host__ __device__ size_t h_off(int l) const { return (size_t)SUNERF_KS0 * 1024 + (size_t)l * layer_bytes(); }
  __host__ __device__ size_t c_off(int l) const { return h_off(l) + (size_t)KS * 1024; }      // FP16 format only
};
constexpr float SUNERF_PHASE_SCALE = 1.f / 65535.f;      // phase code -> revolutions

// input feature (column of the nn.Linear weight) held by (k-step s, lane half h, element e); -1 = zero pad
__host__ __device__ inline int kmap_hidden(int s, int h, int e) {
  return 32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
}
__host__ __device__ inline int kmap_encoding(int s, int h, int e) {
  int q = 8 * s + e;
  if (q < 40) return 4 + 40 * h + q;
  if (q < 44 && h == 0) return q - 40;
  return -1;
}
// row of a 32x32 accumulator tile held by register g on lane half h
__host__ __device__ inline int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (e.g. a probe made by the
// caller's framework): clear it before a launch so that the check after the launch sees only our own.
// Experiment knob (tools/experiments/overlap_probe.py): SUNERF_GRID_CAP_FWD / _DGRAD limit a persistent kernel's grid
// below the CU count so that two kernels can run side by side on disjoint CUs.  Read at every launch; unset = all CUs.
static inline int sunerf_grid_cap(const char* name, int cus) {
  const char* s = getenv(name);
  const int v = s ? atoi(s) : 0;
  return (v > 0 && v < cus) ? v : cus;
}
#define SUNERF_CLEAR_ERROR() (void)hipGetLastError()
#define SUNERF_CHECK_LAUNCH()                         \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)

// Backward pass: the loss gradient w.r.t. the raw MLP output is multiplied by a power of two chosen from max|g_raw| of the
// batch (found on the device) before it enters the fp16 matrix products, and dW / db are multiplied by its inverse at the
// end.  max|g_raw| is placed at 2^SUNERF_GSCALE_LOG2 = 16: the data gradients of the hidden layers then have 2^12 of
// head-room before fp16 saturates at 65504 -- a network whose layers amplify gradients (trained / large weights: gain g
// per layer grows dZ by g^7 over an 8-layer net; all hidden weights x 4 is 2^8.4) stays finite, where the former
// placement at 2^10 left 2^6.  Downwards nothing is lost that matters: gfx950's MFMA honours fp16 subnormals, so a value
// 2^-18 of the maximum still carries an absolute error of only 2^-29 of the maximum.
#define SUNERF_GSCALE_LOG2 4
#if defined(__HIPCC__)
__device__ __forceinline__ int sunerf_gscale_exponent(unsigned absmax_bits) {   // e with max|g_raw| = f 2^e, f in [0.5, 1)
  const float m = __uint_as_float(absmax_bits);
  int e = SUNERF_GSCALE_LOG2;
  if (m > 0.f) frexpf(m, &e);
  return e;
}
__device__ __forceinline__ float sunerf_gscale(unsigned absmax_bits) { return ldexpf(1.f, SUNERF_GSCALE_LOG2 - sunerf_gscale_exponent(absmax_bits)); }
__device__ __forceinline__ float sunerf_gscale_inv(unsigned absmax_bits) { return ldexpf(1.f, sunerf_gscale_exponent(absmax_bits) - SUNERF_GSCALE_LOG2); }
// Per-layer power-of-two BOOST of the backward chain (round 2).  The data gradient shrinks (or grows) by the layer's gain
// -- sqrt(sum W^2 / fan-in) * rms(cos) ~ 0.41 for a default-initialised layer -- every time it passes one: after seven
// layers it is 2^-9 of g_raw and its small entries reach fp16's subnormals, where the per-tensor relative error of the weight
// gradients is no longer 2^-12 (fuzz sweep: 2e-2 on a 7-layer net with hidden weights x 0.25).  sunerf_pack_mlp_t folds
// 2^boost(l) into W_l^T so that the chain keeps the scale of g_raw (exact: a power of two), sunerf_mlp_wgrad's reduce kernel
// divides layer l's sums by the product of the boosts above it.  A pure function of the current weights: no state, no host
// read.  `sumsq`: sum of squares of the layer's nn.Linear weight, `cols` its fan-in.
__host__ __device__ __forceinline__ int sunerf_bwd_boost(float sumsq, int cols) {
  const float g = sqrtf(sumsq / (float)cols) * 0.70710678f;
  if (!(g > 0.f) || !(g < 3.0e38f)) return 0;
  const int s = (int)rintf(-log2f(g));
  return s < -4 ? -4 : (s > 6 ? 6 : s);
}
// fp32 -> fp16 pair that SATURATES at +-65504 instead of overflowing to infinity (a few saturated elements bend a gradient
// that the clip will rescale anyway; an infinity becomes NaN in the weight gradients and costs the whole step)
__device__ __forceinline__ float sunerf_sat16(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
#endif

#if defined(__HIPCC__)
// Stash and scratch traffic goes through BUFFER instructions: wave-uniform base in a scalar resource descriptor, the lane
// part (lane * 16) as the one VGPR offset, the many constant fragment offsets as scalar offsets / immediates.  With per-lane
// 64-bit pointers the compiler materialises one VGPR pair per fragment address, hoists them all out of the chunk loop and
// spills them (200 spilled VGPRs at d = 512).
using Rsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lane_off() { return (threadIdx.x & 63u) * 16u; }
__device__ __forceinline__ Rsrc make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_store(half8 v, Rsrc r, int off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)lane_off(), off, 0);
}
#ifndef SUNERF_NT_STORE_AUX
#define SUNERF_NT_STORE_AUX 2      // cache-policy bits of the write-once streams (activation / dZ stash): 2 = nt; experiment: 0, 18 (sc1 nt), 19 (sc0 sc1 nt)
#endif
__device__ __forceinline__ void buf_store_nt(half8 v, Rsrc r, int off) {   // non-temporal: written once, read by another kernel
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)lane_off(), off, SUNERF_NT_STORE_AUX);
}
__device__ __forceinline__ half8 buf_load(Rsrc r, int off) {
  return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off(), off, 0));
}
__device__ __forceinline__ half8 buf_load_nt(Rsrc r, int off) {   // non-temporal: read once (another kernel's output stream)
  return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off(), off, 2));
}
#endif
