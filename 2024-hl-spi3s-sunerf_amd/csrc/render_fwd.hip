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
#include "weight_ring.h"
#include "../../include/sunerf_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

using sunerf_ring::NSLOT;
using sunerf_ring::Ring;
using sunerf_ring::WAVES;
constexpr int THREADS = WAVES * 64;

struct RenderArgs {
  const char* packed;
  const float* rays_o;
  const float* rays_d;
  const float* times;
  const float* z_vals;
  const float* points;   // optional [n_rays * S][4]: explicit query points (x, y, z, t) -- NeRF.forward on free-standing points;
                         // rays_o / rays_d / times / z_vals are then unused and the integral outputs may be null
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
  int stash_phase;   // 1: the stash takes 16-bit phases (SUNERF_STASH_PHASE) instead of fp16 sin + cos fragments
  char* scratch;   // d_filter = 512: per-wave activation scratch, gridDim.x * 4 waves * (D/16) * 2 KiB
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

// Register plan (512 unified registers per lane, one wave per SIMD): the two activation fragment sets (layer input
// and layer output, 2 x 128 registers at D = 256) live in the ACCUMULATION half of the register file -- MFMA reads
// its B operand from AGPRs directly -- which leaves the 256 architectural VGPRs for the fp32 accumulators (two
// tiles in flight), a multi-k-step ring of A fragments and the epilogue temporaries.  `pin_agpr` is an empty asm
// that makes a freshly produced fragment live in AGPRs (the allocator then keeps it there for the MFMAs).
__device__ __forceinline__ void pin_agpr(half8& f) { asm volatile("" : "+a"(f)); }

// Epilogue of one accumulator tile = 8 "pair" micro-ops (elements 2p, 2p+1 -> one dword of a hi and of a lo fragment),
// each cut into three stages that are issued behind the three MFMAs of a k-step of the NEXT tile:
//   A: sin (argument in revolutions: the 1/(2 pi) is folded into the packed weights) [training: and cos]
//   B: hi = fp16(x) (packed), remainder x - hi
//   C: lo = fp16(remainder) (packed), insert into the fragments [training: stash fp16(sin), fp16(cos) fragments]
// The empty asm statements anchor every stage where it is written: without them the optimiser sinks the whole
// epilogue to its only consumer (the end of the tile), where it would run unoverlapped with matrix work.
struct PairTmp {
  float s0, s1, r0, r1, c0, c1;
  half2v hi, cpk;
};
#ifndef SUNERF_ABL_NO_TRANS
#define SUNERF_ABL_NO_TRANS 0
#endif
constexpr int XL_SHIFT = 17;      // log2 of the factor on the fp8 remainder operand (folded into its conversion, undone by the block scale)
#ifndef SUNERF_ABL_NO_L8
#define SUNERF_ABL_NO_L8 0        // the remainder's scale + fp8 conversion (3 vector instructions per pair micro-op) not issued
#endif
template <int STASH>
__device__ __forceinline__ void epi_stage_a(const f32x16& acc, int p, PairTmp& t) {
#if SUNERF_ABL_NO_TRANS
  t.s0 = __builtin_amdgcn_fractf(acc[2 * p]);
  t.s1 = __builtin_amdgcn_fractf(acc[2 * p + 1]);
  if (STASH) { t.c0 = t.s0; t.c1 = t.s1; }
  asm volatile("" : "+v"(t.s0), "+v"(t.s1));
  return;
#endif
  t.s0 = __builtin_amdgcn_sinf(acc[2 * p]);
  t.s1 = __builtin_amdgcn_sinf(acc[2 * p + 1]);
  asm volatile("" : "+v"(t.s0), "+v"(t.s1));
  if (STASH == 1) {   // d sin(z)/dz for the backward pass
    t.c0 = __builtin_amdgcn_cosf(acc[2 * p]);
    t.c1 = __builtin_amdgcn_cosf(acc[2 * p + 1]);
    asm volatile("" : "+v"(t.c0), "+v"(t.c1));
  }
  if (STASH == 2) {   // phase of the pre-activation (revolutions, [0, 1)): the pipelined backward recovers sin AND cos from it
    t.c0 = __builtin_amdgcn_fractf(acc[2 * p]);
    t.c1 = __builtin_amdgcn_fractf(acc[2 * p + 1]);
    asm volatile("" : "+v"(t.c0), "+v"(t.c1));
  }
}
template <int STASH, bool HALF = false>
__device__ __forceinline__ void epi_stage_b(PairTmp& t) {
  const f32x2 sv = {t.s0, t.s1};
  t.hi = __builtin_convertvector(sv, half2v);   // one v_cvt_pk_f16_f32 (round to nearest even)
  asm volatile("" : "+v"(t.hi));                 // ... and convert back from the packed word, not from two scalar casts
  // remainder x - (float)hi in ONE instruction per element: v_fma_mix_f32 reads the fp16 half of the packed word directly
  // (fma(hi, -1, x): exact, bit-identical to convert-then-subtract -- tools/probes/probe_fma_mix.hip); saves the two
  // v_cvt_f32_f16 per pair in an epilogue that is bound by VALU issue slots
  if (!HALF) {   // (HALF: single fp16 operands, no remainders)
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t.r0) : "v"(t.hi), "v"(t.s0));
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t.r1) : "v"(t.hi), "v"(t.s1));
  }
  if (STASH == 1) {
    const f32x2 cv = {t.c0, t.c1};
    t.cpk = __builtin_convertvector(cv, half2v);
    asm volatile("" : "+v"(t.cpk));
  }
  if (STASH == 2) {   // two 16-bit phase codes in one dword (v_cvt_pknorm_u16_f32: round(65535 f))
    t.cpk = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pknorm_u16(t.c0, t.c1));
    asm volatile("" : "+v"(t.cpk));
  }
}
// `st`: this lane's address of the tile's first H fragment in the stash (training only); `cos_delta` = byte distance
// from an H fragment to the matching cos fragment.  SPILL_OUT (d_filter = 512): the two activation sets do not fit the
// register file together, so the finished hi / lo fragments go to this wave's global scratch at `sc`
// ([fragment][hi 1 KiB | lo 1 KiB]) instead of staying in registers; they come back as the next layer's input.
template <int STASH, bool SPILL_OUT, bool HALF = false>
__device__ __forceinline__ void epi_stage_c(const PairTmp& t, int p, half8& hi0, half8& lo0, half8& hi1, half8& lo1,
                                            half8& ch0, half8& ch1, Rsrc st, int st_off, int cos_delta, Rsrc sc, int sc_off) {
  if (p < 4) { hi0[2 * p] = t.hi[0]; hi0[2 * p + 1] = t.hi[1]; }
  else { hi1[2 * p - 8] = t.hi[0]; hi1[2 * p - 7] = t.hi[1]; }
  if (!HALF) {
    half2v lo;
    lo[0] = (_Float16)t.r0; lo[1] = (_Float16)t.r1;
    if (p < 4) { lo0[2 * p] = lo[0]; lo0[2 * p + 1] = lo[1]; }
    else { lo1[2 * p - 8] = lo[0]; lo1[2 * p - 7] = lo[1]; }
  }
  if (STASH) {
    if (p < 4) { ch0[2 * p] = t.cpk[0]; ch0[2 * p + 1] = t.cpk[1]; }
    else { ch1[2 * p - 8] = t.cpk[0]; ch1[2 * p - 7] = t.cpk[1]; }
  }
  // (pinning after every insertion instead makes hipcc rewrite the whole 4-dword tuple each time: measured worse)
  if (p == 3) {
    if (SPILL_OUT) { buf_store(hi0, sc, sc_off); buf_store(lo0, sc, sc_off + 1024); }
    else { pin_agpr(hi0); if (!HALF) pin_agpr(lo0); }
    if (STASH == 1) { buf_store_nt(hi0, st, st_off); buf_store_nt(ch0, st, st_off + cos_delta); }
    if (STASH == 2) buf_store_nt(ch0, st, st_off);          // one phase fragment instead of a sin and a cos fragment
  }
  if (p == 7) {
    if (SPILL_OUT) { buf_store(hi1, sc, sc_off + 2048); buf_store(lo1, sc, sc_off + 3072); }
    else { pin_agpr(hi1); if (!HALF) pin_agpr(lo1); }
    if (STASH == 1) { buf_store_nt(hi1, st, st_off + 1024); buf_store_nt(ch1, st, st_off + 1024 + cos_delta); }
    if (STASH == 2) buf_store_nt(ch1, st, st_off + 1024);
  }
}

__device__ __forceinline__ f32x16 bias_tile(const float* bias, int h) {
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *(const f32x4*)(bias + 8 * j + 4 * h);
    acc[4 * j + 0] = b[0]; acc[4 * j + 1] = b[1]; acc[4 * j + 2] = b[2]; acc[4 * j + 3] = b[3];
  }
  return acc;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// The MLP as ONE software pipeline over k-steps that runs across tiles, layers, chunks and rays: the A fragments
// (hi, lo: one ds_read_b128 each) of k-step f + PF are requested while the three MFMAs of k-step f issue, so LDS
// latency is always covered by matrix work, and the VALU epilogue of tile U-1 (sin, split, pack) is dealt out over
// the first k-steps of tile U.  One wave issues at most one instruction per ~4 cycles, so the per-k-step order is
// pinned with sched_group_barrier: MFMA, a few VALU, a DS read, ... instead of leaving the epilogue as one clump
// between two tiles.
// HALF (SUNERF_PRECISION_HALF): single fp16 operands -- only the head product of every k-step is issued, the lo fragments
// of the weight stream and of the activations are never read or produced (the "bf16 weights on MFMA" class of BASELINE
// config 3, with fp16's three extra mantissa bits); d <= 256 only.
template <int D, bool HALF = false>
struct Mlp {
  static constexpr int NT = D / 32;
  static constexpr int KS = D / 16;
  static constexpr int XK = KS > SUNERF_KS0 ? KS : SUNERF_KS0;  // fragments per register set
  static constexpr int PF = 4;                                   // prefetch distance in k-steps
  static constexpr int PAGE_STEPS = Ring<D>::PAGE_STEPS;         // k-steps per DMA page
  static constexpr int RING_STEPS = NSLOT * PAGE_STEPS;
  // one activation set in registers, layer outputs via global scratch (HALF has no lo fragments: both sets fit at 512)
  static constexpr bool SPILL = D > 256 && !HALF;
  static_assert(KS >= PF && SUNERF_KS0 >= PF, "prefetch distance exceeds a tile");
  static_assert((NT * SUNERF_KS0) % PF == 0 && KS % PF == 0, "fragment ring phase must be 0 at every layer start");
  static_assert((NT * SUNERF_KS0) % PAGE_STEPS == 0 && KS % PAGE_STEPS == 0, "layers must end on page boundaries");
  // D >= 128: a hidden layer is a whole number of ring revolutions and a pass ends where it began, so the ring
  // position of every k-step is a compile-time constant (immediate ds_read offsets).  D = 64: tracked at run time.
  static constexpr bool STATIC_RING = (NT * KS) % RING_STEPS == 0 && (NT * SUNERF_KS0 + KS) % RING_STEPS == 0;
  static constexpr int RS_HIDDEN = STATIC_RING ? (NT * SUNERF_KS0) % RING_STEPS : -1;   // ring step where hidden layers start
  static constexpr int RS_IN = STATIC_RING ? 0 : -1;

  struct Pipe {
    half8 ahi[PF], alo[PF];   // fragments of the next PF k-steps
    const char* frag;         // LDS address (lane-adjusted) of ring offset 0
    int rstep;                // ring position (in k-steps, 0..RING_STEPS-1) of the NEXT k-step to execute
  };

  static __device__ __forceinline__ void issue_piece_dyn(Ring<D>& ring, int j) {   // j is a constant after unrolling
    if (j == 0) ring.template issue_piece<0>();
    if (j == 1) ring.template issue_piece<1>();
    if constexpr (Ring<D>::PIECES > 2) {
      if (j == 2) ring.template issue_piece<2>();
      if (j == 3) ring.template issue_piece<3>();
    }
    if constexpr (Ring<D>::PIECES > 4) {
      if (j == 4) ring.template issue_piece<4>();
      if (j == 5) ring.template issue_piece<5>();
      if (j == 6) ring.template issue_piece<6>();
      if (j == 7) ring.template issue_piece<7>();
    }
  }

  static __device__ __forceinline__ void load_frag(Pipe& p, int r, int ring_step) {
    const char* q = p.frag + ring_step * 2048;
    p.ahi[r] = *(const half8*)(q);
    if (!HALF) p.alo[r] = *(const half8*)(q + 1024);
  }

  // before the first k-step of the first chunk: pages 0 and 1 in flight, page 0 acquired, first PF k-steps requested
  static __device__ __forceinline__ void start(Ring<D>& ring, Pipe& p) {
    ring.template issue_page<0>();
    ring.template issue_page<0>();
#if SUNERF_ABL_NO_DMA
    ring.template issue_page<0>();
    ring.template issue_page<0>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring.priming = false;
#endif
    ring.acquire();
    // the pass starts at page phase 0, i.e. in the middle of a trickle cycle that began at the acquire phase: deal out
    // the pieces of page 2 that the k-steps between the acquire phase and the page end would have issued
    constexpr int ACQ = (PAGE_STEPS - PF) % PAGE_STEPS;
#pragma unroll
    for (int j = 0; j < Ring<D>::PIECES; ++j)
      if (ACQ + 1 + 2 * j < PAGE_STEPS) issue_piece_dyn(ring, j);
    p.rstep = 0;
#pragma unroll
    for (int s = 0; s < PF; ++s) load_frag(p, s, s);
  }

  // k-steps of one tile.  T0 = stream position (in k-steps) of the tile's first k-step relative to a page boundary
  // (tiles of the in layer are 6 k-steps and straddle pages; all other tiles start on a page boundary).
  // `prev`: accumulator of the previous tile whose epilogue is interleaved here (HAS_PREV), writing y*.
  // STASH (training): the epilogue also stores fp16 sin / cos fragments at `st` (4 stores of 1 KiB per tile); in a
  // steady-state page cycle those 4 stores are younger than the page being acquired, hence acquire<4>.
  // SPILL_OUT: the epilogue's fragments go to scratch at `sc_out` (see epi_stage_c).  RELOAD (last tile of a d = 512
  // hidden layer): k-step s is the last reader of input fragment s, so it is refilled right there from `sc_in` with
  // fragment s of the layer's OUTPUT (written to scratch at least one tile earlier) = the next layer's input.
  template <int KIN, int T0, bool HAS_PREV, int RS0, int STASH, bool SPILL_OUT = false, bool RELOAD = false>
  static __device__ __forceinline__ f32x16 tile(Ring<D>& ring, Pipe& p, f32x16 acc, half8* xhi, half8* xlo,
                                                const f32x16& prev, half8& yh0, half8& yl0, half8& yh1, half8& yl1,
                                                Rsrc st, int st_off, int cos_delta, Rsrc sc, int sc_out_off = 0) {
    // k-steps that carry epilogue micro-ops.  For the first tile of a layer the epilogue produces the layer's own last
    // two input fragments (read by k-steps KIN-2 and KIN-1), so it must be complete before k-step KIN-2.
    constexpr int EPI_STEPS = (KIN - 2) >= 8 ? 8 : (KIN - 2);
    static_assert(EPI_STEPS >= 1, "tile too short to hide the previous tile's epilogue");
    constexpr int PER = (8 + EPI_STEPS - 1) / EPI_STEPS;   // pair micro-ops per k-step
    constexpr int ACQ = (PAGE_STEPS - PF) % PAGE_STEPS;    // page phase at which the next page is acquired
    half8 ch0, ch1;                                        // fp16 cos fragments (training)
    constexpr int RL = 3;                                  // RELOAD: k-steps between the request of a fragment and its use
    half8 rl_hi[RL], rl_lo[RL];
#pragma unroll
    for (int s = 0; s < KIN; ++s) {
      const int r = (T0 + s) % PF;
      PairTmp t[PER];
      // --- MFMA 1 | stage A --------------------------------------------------------------------------------
      if (!HALF) acc = mfma16(p.alo[r], xhi[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (s * PER + q < 8) epi_stage_a<STASH>(prev, s * PER + q, t[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // --- MFMA 2 | stage B --------------------------------------------------------------------------------
      if (!HALF) acc = mfma16(p.ahi[r], xlo[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (s * PER + q < 8) epi_stage_b<STASH, HALF>(t[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // --- MFMA 3 | stage C | next page | A-fragment reads of k-step s + PF --------------------------------------
      acc = mfma16(p.ahi[r], xhi[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (s * PER + q < 8) epi_stage_c<STASH, SPILL_OUT, HALF>(t[q], s * PER + q, yh0, yl0, yh1, yl1, ch0, ch1, st, st_off, cos_delta, sc, sc_out_off);
      }
      if (RELOAD) {   // fragments KIN-2, KIN-1 arrive through the carry epilogue of the next layer's first tile
        // requested at k-step s, moved into the operand registers RL k-steps later: committing right away would park the
        // wave for a whole L2 round trip in every k-step of the tile
        if (s >= RL && s - RL < KIN - 2) {
          xhi[s - RL] = rl_hi[(s - RL) % RL];
          xlo[s - RL] = rl_lo[(s - RL) % RL];
          pin_agpr(xhi[s - RL]); pin_agpr(xlo[s - RL]);
        }
        if (s < KIN - 2) {
          rl_hi[s % RL] = buf_load(sc, s * 2048);
          rl_lo[s % RL] = buf_load(sc, s * 2048 + 1024);
        }
      }
      {
        const int phase = (T0 + s) % PAGE_STEPS;
        if (phase == ACQ) {                                          // the reads below cross into the next page
          // vector-memory operations certainly issued after the last piece of the page being acquired (besides the
          // pieces of the following page): the epilogue's stores, which fall into the first page cycle of a tile
          constexpr int STORES = (HAS_PREV && KIN >= 16 && PER == 1) ? ((SPILL_OUT ? 4 : 0) + (STASH == 1 ? 4 : STASH == 2 ? 2 : 0)) : 0;
          if (s < 16) ring.template acquire<STORES>();
          else ring.template acquire<0>();
        }
        const int rel = (phase - ACQ - 1 + PAGE_STEPS) % PAGE_STEPS; // k-steps since the acquire, minus one
        if (rel % 2 == 0 && rel / 2 < Ring<D>::PIECES) issue_piece_dyn(ring, rel / 2);
      }
      if (RS0 >= 0) {
        load_frag(p, r, (RS0 + s + PF) % RING_STEPS);
      } else {
        int rs = p.rstep + s + PF;
        rs = rs >= RING_STEPS ? rs - RING_STEPS : rs;
        load_frag(p, r, rs);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (RELOAD) {   // the last RL requests
#pragma unroll
      for (int s = KIN; s < KIN + RL; ++s)
        if (s - RL >= 0 && s - RL < KIN - 2) {
          xhi[s - RL] = rl_hi[(s - RL) % RL];
          xlo[s - RL] = rl_lo[(s - RL) % RL];
          pin_agpr(xhi[s - RL]); pin_agpr(xlo[s - RL]);
        }
    }
    if (RS0 < 0) {
      p.rstep += KIN;
      if (p.rstep >= RING_STEPS) p.rstep -= RING_STEPS;
    }
    return acc;
  }

  // one layer: X (KIN k-steps) -> Y (KS k-steps).  `carry` is the accumulator of the previous layer's last tile,
  // whose epilogue (into that layer's output set = our X, fragments 2*NT-2, 2*NT-1) overlaps our first tile: those
  // fragments are only read by our last two k-steps.  Returns our own last accumulator the same way.
  // st_prev / st_own: this lane's stash address of fragment 0 of the previous / of this layer's H block (training).
  template <int KIN, bool HAS_CARRY, int RSL0, int STASH>
  static __device__ __forceinline__ f32x16 layer(Ring<D>& ring, Pipe& p, const float* bias, int h, half8* xhi, half8* xlo,
                                                 half8* yhi, half8* ylo, const f32x16& carry, Rsrc st, int st_prev, int st_own,
                                                 Rsrc scratch) {
    f32x16 prev = carry;
    constexpr int CD = KS * 1024;
    constexpr int XL = 2 * NT - 2;   // previous layer's last tile -> our X fragments
    half8 th0, tl0, th1, tl1;        // SPILL: staging of the fragments that go to scratch
    if constexpr (SPILL) {
      // 16 tiles: a `#pragma unroll` loop over a 16-way `if (U == UU)` dispatch exceeds the unroller's size limit and would
      // stay a run-time loop; compile-time recursion instead
      static_for<0, NT>([&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        constexpr int RS = RSL0 < 0 ? -1 : (RSL0 + UU * KIN) % RING_STEPS;
        constexpr int T0 = (UU * KIN) % PAGE_STEPS;
        constexpr bool LAST = (UU == NT - 1) && (KIN == KS);
        f32x16 acc = bias_tile(bias + 32 * UU, h);
        if constexpr (UU == 0)
          acc = tile<KIN, T0, HAS_CARRY, RS, STASH, false, false>(ring, p, acc, xhi, xlo, prev, xhi[XL], xlo[XL], xhi[XL + 1],
                                                                  xlo[XL + 1], st, st_prev + XL * 1024, CD, scratch);
        else
          acc = tile<KIN, T0, true, RS, STASH, true, LAST>(ring, p, acc, xhi, xlo, prev, th0, tl0, th1, tl1, st,
                                                           st_own + (2 * UU - 2) * 1024, CD, scratch, (2 * UU - 2) * 2048);
        prev = acc;
      });
      return prev;
    }
    if constexpr (NT > 8) {   // (d = 512 in HALF mode: 16 tiles, both sets in registers; see the SPILL branch for static_for)
      static_for<0, NT>([&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        constexpr int RS = RSL0 < 0 ? -1 : (RSL0 + UU * KIN) % RING_STEPS;
        constexpr int T0 = (UU * KIN) % PAGE_STEPS;
        f32x16 acc = bias_tile(bias + 32 * UU, h);
        if constexpr (UU == 0)
          acc = tile<KIN, T0, HAS_CARRY, RS, STASH>(ring, p, acc, xhi, xlo, prev, xhi[XL], xlo[XL], xhi[XL + 1], xlo[XL + 1], st,
                                                    st_prev + XL * 1024, CD, scratch);
        else
          acc = tile<KIN, T0, true, RS, STASH>(ring, p, acc, xhi, xlo, prev, yhi[2 * UU - 2], ylo[2 * UU - 2], yhi[2 * UU - 1],
                                               ylo[2 * UU - 1], st, st_own + (2 * UU - 2) * 1024, CD, scratch);
        prev = acc;
      });
      return prev;
    }
#pragma unroll
    for (int U = 0; U < NT; ++U) {
      f32x16 acc = bias_tile(bias + 32 * U, h);
#define SUNERF_TILE(UU)                                                                                          \
      if (U == UU) {                                                                                               \
        constexpr int RS = RSL0 < 0 ? -1 : (RSL0 + UU * KIN) % RING_STEPS;                                         \
        constexpr int T0 = (UU * KIN) % PAGE_STEPS;                                                               \
        if (UU == 0) acc = tile<KIN, T0, HAS_CARRY, RS, STASH>(ring, p, acc, xhi, xlo, prev, xhi[XL], xlo[XL], xhi[XL + 1], xlo[XL + 1], st, st_prev + XL * 1024, CD, scratch); \
        else acc = tile<KIN, T0, true, RS, STASH>(ring, p, acc, xhi, xlo, prev, yhi[2 * UU - 2], ylo[2 * UU - 2], yhi[2 * UU - 1], ylo[2 * UU - 1], st, st_own + (2 * UU - 2) * 1024, CD, scratch); \
      }
      SUNERF_TILE(0) SUNERF_TILE(1) SUNERF_TILE(2) SUNERF_TILE(3) SUNERF_TILE(4) SUNERF_TILE(5) SUNERF_TILE(6) SUNERF_TILE(7)
#undef SUNERF_TILE
      prev = acc;
    }
    return prev;
  }

  template <int STASH>
  static __device__ __forceinline__ f32x16 out_layer(Ring<D>& ring, Pipe& p, const float* bias, int h, half8* xhi,
                                                     half8* xlo, const f32x16& carry, Rsrc st, int st_prev, Rsrc scratch) {
    constexpr int XL = 2 * NT - 2;
    return tile<KS, 0, true, RS_HIDDEN, STASH, false, false>(ring, p, bias_tile(bias, h), xhi, xlo, carry, xhi[XL], xlo[XL],
                                                             xhi[XL + 1], xlo[XL + 1], st, st_prev + XL * 1024, KS * 1024, scratch);
  }
};

#ifndef SUNERF_PIN_MFMA
#define SUNERF_PIN_MFMA 1
#endif
// Energy ablations (tools/energy_ablation.sh; results in DESIGN.md section 10): each removes ONE kind of work from the hidden
// tiles of the fp8c forward while every operand stays finite -- the outputs are wrong, the time tells what that work costs.
#ifndef SUNERF_ABL_NO_AREAD
#define SUNERF_ABL_NO_AREAD 0     // weight operands are not re-read from LDS (the first group's stay in registers)
#endif
#ifndef SUNERF_ABL_NO_TRANS
#define SUNERF_ABL_NO_TRANS 0     // v_fract instead of v_sin (/ v_cos) in the epilogue
#endif
#ifndef SUNERF_BATCH_READS
#define SUNERF_BATCH_READS 1
#endif
// ---- d <= 256: fp16 head product + two block-scaled fp8 correction products (format: sunerf_common.h, "fp8c") ----------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pin_agpr8(v8i& f) { asm volatile("" : "+a"(f)); }

// Stage C of a pair micro-op, fp8c operand format: the finished values go into fragment FP + (p >> 2) of the output set as
// fp16 head (yhi) and as two fp8 words -- the head itself and the remainder scaled by 2^11 -- of the 64-deep fp8 operands
// (group = fragment / 4; dword 2 (fragment % 4) + (p % 4) / 2, 16-bit word (p % 4) % 2).
// SPILL_OUT (d = 512: one set of fp16 heads, two sets of fp8 operands): the finished fp16 fragment goes to this wave's
// scratch (1 KiB per fragment, fragment order) instead of registers, the fp8 group stays in registers like at d <= 256;
// `th` collects the fragment under construction.
template <int STASH, bool SPILL_OUT = false>
__device__ __forceinline__ void epi_stage_c8(const PairTmp& t, int p, int FP, half8* yhi, v8i* yh8, v8i* yl8, v8i& w8h,
                                             v8i& w8l, half8& ch0, half8& ch1, Rsrc st, int st_off, int cos_delta,
                                             Rsrc sc = Rsrc(), half8* th = nullptr) {
  const int f = FP + (p >> 2), dq = p & 3;                 // literals after unrolling
  half8& hf = SPILL_OUT ? th[p >> 2] : yhi[f];
  hf[2 * dq] = t.hi[0];
  hf[2 * dq + 1] = t.hi[1];
  const int g = f >> 2, d = 2 * (f & 3) + (dq >> 1);
  // the 64-deep fp8 operands of a group (4 fragments = 2 tiles) are collected in the VGPR tuples w8h / w8l and moved to
  // the (AGPR-resident) operand set in one piece when the group is complete: partial writes to an 8-register AGPR tuple
  // make the allocator copy the tuple around
  // The remainder's power of two: round 1 multiplied (two v_mul_f32 -- v_pk_mul_f32 measured 2 % slower) and converted with
  // v_cvt_pk_fp8_f32, because the scaled conversion "measured 10x the output error".  That was not the instruction (bit-identical to
  // multiply + convert: tools/probes/probe_cvt_scale2.hip) but its BUILTIN: hipcc 7.2 feeds the second half's `old` operand with
  // a neighbouring dword when the two halves of a dword are written by different micro-ops.  Round 2: the instruction through
  // inline asm with the dword tied in and out -- 16 vector instructions fewer per tile in a kernel bound by its one wave's issue
  // slots (a vector instruction costs that wave ~4.8 cycles, tools/probes/probe_valu_cost.hip).
#if SUNERF_ABL_NO_L8   // issue-slot ablation: 3 of the ~17 vector instructions of a pair micro-op are not issued
  if (dq & 1) w8h[d] = __builtin_amdgcn_cvt_pk_fp8_f32(t.s0, t.s1, w8h[d], true);
  else w8h[d] = __builtin_amdgcn_cvt_pk_fp8_f32(t.s0, t.s1, w8h[d], false);
#else
  // (the scaled conversion divides by its scale operand; 2^XL_SHIFT puts |remainder| <= 2^-12 at <= 32, i.e. every remainder
  // above 2^-23 into e4m3's normal range; the matrix instruction's block scale of this operand is 127 - XL_SHIFT.  The head's
  // conversion merges into the dword's stale content: the other word is rewritten by the neighbouring micro-op anyway)
  const float cvt_scale = 1.f / (float)(1 << XL_SHIFT);
  if (dq & 1) {   // the word selectors must be literals
    w8h[d] = __builtin_amdgcn_cvt_pk_fp8_f32(t.s0, t.s1, w8h[d], true);
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(w8l[d]) : "v"(t.r0), "v"(t.r1), "s"(cvt_scale));
  } else {
    w8h[d] = __builtin_amdgcn_cvt_pk_fp8_f32(t.s0, t.s1, w8h[d], false);
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(w8l[d]) : "v"(t.r0), "v"(t.r1), "s"(cvt_scale));
  }
#endif
  if (STASH) {
    if (p < 4) { ch0[2 * p] = t.cpk[0]; ch0[2 * p + 1] = t.cpk[1]; }
    else { ch1[2 * p - 8] = t.cpk[0]; ch1[2 * p - 7] = t.cpk[1]; }
  }
  if (dq == 3) {                                            // fragment f is complete
    if (SPILL_OUT) {
      buf_store(hf, sc, f * 1024);
      if ((f & 3) == 3) {
        yh8[g] = w8h; yl8[g] = w8l;
        pin_agpr8(yh8[g]); pin_agpr8(yl8[g]);
      }
    } else {
      pin_agpr(hf);
      if ((f & 3) == 3) {
        yh8[g] = w8h; yl8[g] = w8l;
        pin_agpr8(yh8[g]); pin_agpr8(yl8[g]);
      }
    }
    if (STASH == 1) {
      buf_store_nt(hf, st, st_off + (p >> 2) * 1024);
      buf_store_nt(p < 4 ? ch0 : ch1, st, st_off + (p >> 2) * 1024 + cos_delta);
    }
    if (STASH == 2) buf_store_nt(p < 4 ? ch0 : ch1, st, st_off + (p >> 2) * 1024);
  }
}
__device__ __forceinline__ v8i buf_load8(Rsrc r, int off) {   // two 16-byte halves of an fp8 operand (1 KiB apart)
  const half8 a = buf_load(r, off), b = buf_load(r, off + 1024);
  const v4i x = __builtin_bit_cast(v4i, a), y = __builtin_bit_cast(v4i, b);
  v8i v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return v;
}

template <int D>
struct Mlp8 : Mlp<D> {
  using M = Mlp<D>;
  using M::KS; using M::NT; using M::PF; using M::PAGE_STEPS; using M::RING_STEPS; using M::RS_HIDDEN; using M::RS_IN;
  using Pipe = typename M::Pipe;
  static constexpr int G = KS / 4;                          // 64-deep groups per hidden tile
  static constexpr int GP = PAGE_STEPS / 4;                 // ... per weight page (d = 512: a tile is two pages)
  static constexpr int PIECES = Ring<D>::PIECES;
  static constexpr bool SPILL = M::SPILL;
  static_assert(KS % PAGE_STEPS == 0 && PAGE_STEPS % 4 == 0, "fp8c: a hidden tile is whole pages of whole groups");

  // A operands of the next DIST groups (in-place refill behind the consuming instruction)
  static constexpr int DIST = 1;   // (2 was measured: no faster at d = 256, and it costs 32 registers)
  static_assert(GP % DIST == 0, "operand slot of a group must be a compile-time constant");
  struct Pipe8 {
    half8 a16[DIST][4];
    v8i al8[DIST], ah8[DIST];
  };
  struct Scales { int a_lo, a_hi; };   // E8M0 block scales of the two fp8 A operands of a layer: 127 - (sh + 11), 127 - sh

  static __device__ __forceinline__ const char* group_ptr(const Pipe& p, int ring_step) { return p.frag + ring_step * 2048; }
  // LDS addresses of a tile's operand reads as (one of three bases) + 16-bit immediate.  The bases are made opaque once per
  // tile: otherwise the compiler treats every `frag + constant` as a loop invariant of the chunk loop, keeps ~30 of them
  // in VGPRs and evicts operand tuples from the AGPRs to make room.
  typedef const __attribute__((address_space(3))) char* lds_cptr;   // 32-bit LDS pointer: keeps the reads ds_read, not flat
  struct LdsBases {
    lds_cptr b[3];
    __device__ __forceinline__ explicit LdsBases(const Pipe& p) {
      b[0] = (lds_cptr)(uintptr_t)(unsigned)(uintptr_t)p.frag;   // low 32 bits of a generic LDS address = the LDS address
      b[1] = b[0] + 49152;
      b[2] = b[0] + 98304;
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
    }
    // byte offset (a literal when the ring position is static) -> address
    __device__ __forceinline__ lds_cptr at(int off) const {
      return off < 49152 ? b[0] + off : (off < 98304 ? b[1] + (off - 49152) : b[2] + (off - 98304));
    }
  };
  static __device__ __forceinline__ half8 lds_half8(lds_cptr q) { return *(const __attribute__((address_space(3))) half8*)q; }
  static __device__ __forceinline__ v8i lds_v8i(lds_cptr q0, lds_cptr q1) {
    const v4i lo = *(const __attribute__((address_space(3))) v4i*)q0, hi = *(const __attribute__((address_space(3))) v4i*)q1;
    v8i r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
  }
  static __device__ __forceinline__ v8i load8(const char* q) {
    const v4i lo = *(const v4i*)q, hi = *(const v4i*)(q + 1024);
    v8i r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
  }
  // all operands of the first DIST groups from ring position `ring_step` on (a layer stack's start: nothing prefetched them)
  static __device__ __forceinline__ void preload(const Pipe& p, Pipe8& q, int ring_step) {
#pragma unroll
    for (int d = 0; d < DIST; ++d) {
      int rs = ring_step + 4 * d;
      rs = rs >= RING_STEPS ? rs - RING_STEPS : rs;
      const char* b = group_ptr(p, rs);
#pragma unroll
      for (int i = 0; i < 4; ++i) q.a16[d][i] = *(const half8*)(b + i * 1024);
      q.al8[d] = load8(b + 4096);
      q.ah8[d] = load8(b + 6144);
    }
  }

  // One hidden / out tile (KS k-steps = G groups = one page).  Per group six matrix instructions on one accumulator: four
  // fp16 head products, then fp8(xh) * fp8(wl) and fp8(xl) * fp8(wh).  Each instruction's A operand is refilled with the
  // next group's right behind it (in-place prefetch one group = 256+ matrix cycles ahead).  The previous tile's epilogue
  // (8 pair micro-ops, writing fragments FP, FP + 1 of the y set) is dealt out over the first segments; it must be done
  // before k-step KS - 2, whose operands it produces when the previous tile closed the layer above.
  // RS0: ring k-step of the tile's first group (compile-time) or -1 (run-time, p.rstep).
  // SPILL_OUT / RELOAD (d = 512): the pending epilogue writes to scratch `scr`; in the LAST tile of a layer every operand
  // of the single register set is refilled, right behind its last use, with the layer's own output from scratch (requests
  // are committed to the operand registers a few instructions later so the wave never waits for the round trip).
  template <bool HAS_PREV, int RS0, int STASH, bool SPILL_OUT = false, bool RELOAD = false>
  static __device__ __forceinline__ f32x16 tile8(Ring<D>& ring, Pipe& p, Pipe8& q, f32x16 acc, half8* xhi,
                                                 v8i* xh8, v8i* xl8, const Scales sc, const f32x16& prev,
                                                 f32x16& pc /* in: correction accumulator of prev; out: ours */, int FP,
                                                 half8* yhi, v8i* yh8, v8i* yl8, v8i& w8h, v8i& w8l, Rsrc st,
                                                 int st_off, int cos_delta, Rsrc scr = Rsrc()) {
    constexpr int SEGS = 6 * ((KS - 2) / 4) + ((KS - 2) % 4);        // segments before the deadline
    constexpr int PER = SEGS >= 16 ? 0 : (8 + SEGS - 1) / SEGS;      // 0: one micro-op per two segments (A+B | C)
    half8 ch0, ch1;
    PairTmp t[PER == 0 ? 2 : PER];
    static_assert(PER != 0 || 2 + 2 * 8 < SEGS, "shifted epilogue schedule must still meet the deadline");
    const LdsBases lb(p);
    f32x16 pv;
    if (PER != 0) pv = prev + pc;        // short tiles (d <= 128): summed up front
    half8 th[2];                         // SPILL_OUT: fragments under construction
    constexpr int RL = 3;                // RELOAD: segments between the request of an operand and its commit
    half8 rl_hi[RL];
    // the block-scaled fp8 instruction sums with ~17 bits (probe: 8e-6 relative on a 64-deep sum): harmless for the
    // corrections themselves (2^-12 of the result) but not for a running sum of order one passed through it, so they
    // get their own accumulator, added once per tile
    f32x16 accc = {0};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // ring position (k-steps) of the operands requested while this group executes
      const int slot8 = g % DIST;
      int nrs;
      if (RS0 >= 0) nrs = (RS0 + 4 * (g + DIST)) % RING_STEPS;
      else { nrs = p.rstep + 4 * (g + DIST); nrs = nrs >= RING_STEPS ? nrs - RING_STEPS : nrs; }
      const char* nb = RS0 >= 0 ? nullptr : group_ptr(p, nrs);      // run-time ring position (d = 64): one add per group
      const int nbo = nrs * 2048;                                   // static ring position: base + immediate
      if (g % GP == GP - DIST) {
        // the refills from here on read the next page: acquire it.  Of the page after it, 2 + 2 (GP - DIST) pieces have
        // been issued by now (two per group since the previous page's last group); anything else in flight is younger.
        ring.template acquire<-2 * (DIST - 1)>();
      }
#pragma unroll
      for (int sg = 0; sg < 6; ++sg) {
        const int seg = 6 * g + sg;
        if (sg < 4) {
          acc = mfma16(q.a16[slot8][sg], xhi[4 * g + sg], acc);
        } else if (sg == 4) {
          accc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q.al8[slot8], xh8[g], accc, 0, 0, 0, sc.a_lo, 0, 127);
        } else {
          accc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q.ah8[slot8], xl8[g], accc, 0, 0, 0, sc.a_hi, 0, 127 - XL_SHIFT);
        }
#if SUNERF_PIN_MFMA
        // the matrix instruction opens its segment: what follows runs in ITS shadow (without this the scheduler is free to
        // move it to the end of the segment, i.e. behind work that was sized for the previous instruction)
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (HAS_PREV) {
          if (PER == 0) {
            // The previous tile's two accumulators are summed in segment 2, behind two of this tile's matrix instructions
            // (summing at the end of their own tile idles the wave through the latency of its last 16-pass instruction).
            // Micro-op i: stage A in segment 2 + 2 i, B in 3 + 2 i, C in 4 + 2 i (beside stage A of op i + 1): every
            // segment holds independent pieces, so the transcendental / conversion result latencies are covered
            if (seg == 2) pv = prev + pc;
            const int i = (seg - 2) >> 1;
            if (seg >= 2 && ((seg - 2) & 1) == 0) {
              if (i < 8) epi_stage_a<STASH>(pv, i, t[i & 1]);
              if (i >= 1 && i <= 8)
                epi_stage_c8<STASH, SPILL_OUT>(t[(i - 1) & 1], i - 1, FP, yhi, yh8, yl8, w8h, w8l, ch0, ch1, st, st_off, cos_delta, scr, th);
            } else if (seg >= 2 && i < 8) {
              epi_stage_b<STASH>(t[i & 1]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (seg * PER + e < 8) epi_stage_a<STASH>(pv, seg * PER + e, t[e]);
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (seg * PER + e < 8) epi_stage_b<STASH>(t[e]);
#pragma unroll
            for (int e = 0; e < PER; ++e)
              if (seg * PER + e < 8)
                epi_stage_c8<STASH, SPILL_OUT>(t[e], seg * PER + e, FP, yhi, yh8, yl8, w8h, w8l, ch0, ch1, st, st_off, cos_delta, scr, th);
          }
        }
        // weight stream: two pieces per group behind the acquire (page + 3 into the slot everyone left)
        if (sg == 1 || sg == 3) {
          const int piece = 2 * (((g % GP) + 1) % GP) + (sg == 3 ? 1 : 0);
          if (piece < PIECES) M::issue_piece_dyn(ring, piece);
        }
        if (RELOAD) {
          // k-step ks = 4 g + sg (fp16 segments) was the last reader of xhi[ks].  Fragments KS-2, KS-1 arrive through the
          // carry epilogue instead; the fp8 operands of the next layer are the other register set (nothing to reload).
          const int ks = 4 * g + sg;
          if (sg < 4) {
            if (ks >= RL && ks - RL < KS - 2) { xhi[ks - RL] = rl_hi[(ks - RL) % RL]; pin_agpr(xhi[ks - RL]); }
            if (ks < KS - 2) rl_hi[ks % RL] = buf_load(scr, ks * 1024);
          }
        }
        // in-place prefetch of the next group's operand
        if (SUNERF_ABL_NO_AREAD) {
          // (opaque to the optimiser, or it folds the matrix instructions that now see the same operand again)
          if (sg == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(q.a16[slot8][i]));
            asm volatile("" : "+v"(q.al8[slot8]), "+v"(q.ah8[slot8]));
          }
        } else if (RS0 >= 0) {
#if SUNERF_BATCH_READS
          // all reads of the next group behind the two long (84-cycle) fp8 instructions: the four fp16 operands after the
          // first, the two fp8 operands after the second.  One s_waitcnt per batch instead of one per matrix instruction,
          // and the 32-cycle fp16 segments keep their few issue slots for the epilogue.
          if (sg == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) q.a16[slot8][i] = lds_half8(lb.at(nbo + i * 1024));
          } else if (sg == 5) {
            q.al8[slot8] = lds_v8i(lb.at(nbo + 4096), lb.at(nbo + 5120));
            q.ah8[slot8] = lds_v8i(lb.at(nbo + 6144), lb.at(nbo + 7168));
          }
#else
          if (sg < 4) q.a16[slot8][sg] = lds_half8(lb.at(nbo + sg * 1024));
          else if (sg == 4) q.al8[slot8] = lds_v8i(lb.at(nbo + 4096), lb.at(nbo + 5120));
          else q.ah8[slot8] = lds_v8i(lb.at(nbo + 6144), lb.at(nbo + 7168));
#endif
        } else {
          if (sg < 4) q.a16[slot8][sg] = *(const half8*)(nb + sg * 1024);
          else if (sg == 4) q.al8[slot8] = load8(nb + 4096);
          else q.ah8[slot8] = load8(nb + 6144);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (RELOAD) {   // the last RL fp16 requests
#pragma unroll
      for (int ks = KS; ks < KS + RL; ++ks)
        if (ks - RL >= 0 && ks - RL < KS - 2) { xhi[ks - RL] = rl_hi[(ks - RL) % RL]; pin_agpr(xhi[ks - RL]); }
    }
    if (RS0 < 0) {
      p.rstep += KS;
      if (p.rstep >= RING_STEPS) p.rstep -= RING_STEPS;
    }
    pc = accc;
    return acc;
  }

  // in-layer tile (6 k-steps, classic hi | lo weights and encoding operands) with the fp8c epilogue format
  template <int T0, bool HAS_PREV, int RS0, int STASH, bool SPILL_OUT = false>
  static __device__ __forceinline__ f32x16 tile_in(Ring<D>& ring, Pipe& p, f32x16 acc, const half8* xhi, const half8* xlo,
                                                   const f32x16& prev, int FP, half8* yhi, v8i* yh8, v8i* yl8, v8i& w8h,
                                                   v8i& w8l, Rsrc st, int st_off, int cos_delta, Rsrc scr = Rsrc()) {
    constexpr int KIN = SUNERF_KS0;
    constexpr int ACQ = (PAGE_STEPS - PF) % PAGE_STEPS;
    half8 ch0, ch1;
    half8 th[2];
    // The pending epilogue writes the y set, which nothing reads before the in layer is over: no deadline, so its 8
    // micro-ops are spread over ALL k-steps (micro-op i behind k-step i KIN / 8: 2 1 1 2 1 1) -- with two per k-step on the
    // first four, as the hidden tiles' deadline would demand, these short tiles were issue-bound at twice their matrix time
#pragma unroll
    for (int s = 0; s < KIN; ++s) {
      const int r = (T0 + s) % PF;
      PairTmp t[2];
      acc = mfma16(p.alo[r], xhi[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((i * KIN) / 8 == s) epi_stage_a<STASH>(prev, i, t[i & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma16(p.ahi[r], xlo[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((i * KIN) / 8 == s) epi_stage_b<STASH>(t[i & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma16(p.ahi[r], xhi[s], acc);
      if (HAS_PREV) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((i * KIN) / 8 == s)
            epi_stage_c8<STASH, SPILL_OUT>(t[i & 1], i, FP, yhi, yh8, yl8, w8h, w8l, ch0, ch1, st, st_off, cos_delta, scr, th);
      }
      {
        const int phase = (T0 + s) % PAGE_STEPS;
        if (phase == ACQ) ring.template acquire<0>();
        const int rel = (phase - ACQ - 1 + PAGE_STEPS) % PAGE_STEPS;
        if (rel % 2 == 0 && rel / 2 < PIECES) M::issue_piece_dyn(ring, rel / 2);
      }
      if (RS0 >= 0) {
        M::load_frag(p, r, (RS0 + s + PF) % RING_STEPS);
      } else {
        int rs = p.rstep + s + PF;
        rs = rs >= RING_STEPS ? rs - RING_STEPS : rs;
        M::load_frag(p, r, rs);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (RS0 < 0) {
      p.rstep += KIN;
      if (p.rstep >= RING_STEPS) p.rstep -= RING_STEPS;
    }
    return acc;
  }

  // in layer: encoding (6 k-steps, hi | lo) -> y set (fp8c format); returns the last tile's accumulator (its epilogue is
  // carried into the first hidden tile)
  template <int STASH>
  static __device__ __forceinline__ f32x16 in_layer(Ring<D>& ring, Pipe& p, const float* bias, int h, const half8* ehi,
                                                    const half8* elo, half8* yhi, v8i* yh8, v8i* yl8, v8i& w8h, v8i& w8l,
                                                    Rsrc st, int st_own, Rsrc scr = Rsrc()) {
    f32x16 prev = {0};
    constexpr int CD = KS * 1024;
    if constexpr (SPILL) {   // 16 tiles, output to scratch
      static_for<0, NT>([&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        constexpr int RS = RS_IN < 0 ? -1 : (RS_IN + UU * SUNERF_KS0) % RING_STEPS;
        constexpr int T0 = (UU * SUNERF_KS0) % PAGE_STEPS;
        f32x16 acc = bias_tile(bias + 32 * UU, h);
        acc = tile_in<T0, (UU > 0), RS, STASH, true>(ring, p, acc, ehi, elo, prev, 2 * UU - 2, yhi, yh8, yl8, w8h, w8l, st,
                                                     st_own + (2 * UU - 2) * 1024, CD, scr);
        prev = acc;
      });
      return prev;
    }
#pragma unroll
    for (int U = 0; U < NT; ++U) {
      f32x16 acc = bias_tile(bias + 32 * U, h);
#define SUNERF_TILE_IN(UU)                                                                                           \
      if (U == UU) {                                                                                                   \
        constexpr int RS = RS_IN < 0 ? -1 : (RS_IN + UU * SUNERF_KS0) % RING_STEPS;                                    \
        constexpr int T0 = (UU * SUNERF_KS0) % PAGE_STEPS;                                                             \
        acc = tile_in<T0, (UU > 0), RS, STASH>(ring, p, acc, ehi, elo, prev, 2 * UU - 2, yhi, yh8, yl8, w8h, w8l, st,  \
                                               st_own + (2 * UU - 2) * 1024, CD);                                     \
      }
      SUNERF_TILE_IN(0) SUNERF_TILE_IN(1) SUNERF_TILE_IN(2) SUNERF_TILE_IN(3)
      SUNERF_TILE_IN(4) SUNERF_TILE_IN(5) SUNERF_TILE_IN(6) SUNERF_TILE_IN(7)
#undef SUNERF_TILE_IN
      prev = acc;
    }
    return prev;
  }

  // hidden layer: x set -> y set; `carry` = accumulator of the layer above's last tile (its epilogue completes x while our
  // first tile runs: fragments 2 NT - 2, 2 NT - 1 = k-steps KS - 2, KS - 1)
  template <int STASH>
  static __device__ __forceinline__ f32x16 hidden_layer(Ring<D>& ring, Pipe& p, Pipe8& q, const float* bias, int h,
                                                        const Scales sc, half8* xhi, v8i* xh8, v8i* xl8, half8* yhi,
                                                        v8i* yh8, v8i* yl8, v8i& w8h, v8i& w8l, const f32x16& carry, f32x16& pc,
                                                        Rsrc st, int st_prev, int st_own, Rsrc scr = Rsrc()) {
    f32x16 prev = carry;
    constexpr int CD = KS * 1024;
    constexpr int XL = 2 * NT - 2;
    if constexpr (SPILL) {
      // one set of fp16 heads: tile 0 finishes it (carry epilogue into registers), tiles 1 .. NT-1 send their fp16 output
      // to scratch, the last tile pulls it back in as the next layer's input; the fp8 operands alternate between two
      // register sets (x8 -> y8), the carry epilogue completes the x8 set in place
      static_for<0, NT>([&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        constexpr int RS = RS_HIDDEN < 0 ? -1 : (RS_HIDDEN + UU * KS) % RING_STEPS;
        f32x16 acc = bias_tile(bias + 32 * UU, h);
        if constexpr (UU == 0)
          acc = tile8<true, RS, STASH, false, false>(ring, p, q, acc, xhi, xh8, xl8, sc, prev, pc, XL, xhi, xh8, xl8, w8h, w8l,
                                                     st, st_prev + XL * 1024, CD, scr);
        else
          acc = tile8<true, RS, STASH, true, (UU == NT - 1)>(ring, p, q, acc, xhi, xh8, xl8, sc, prev, pc, 2 * UU - 2, xhi, yh8,
                                                             yl8, w8h, w8l, st, st_own + (2 * UU - 2) * 1024, CD, scr);
        prev = acc;
      });
      return prev;
    }
#pragma unroll
    for (int U = 0; U < NT; ++U) {
      f32x16 acc = bias_tile(bias + 32 * U, h);
#define SUNERF_TILE8(UU)                                                                                             \
      if (U == UU) {                                                                                                   \
        constexpr int RS = RS_HIDDEN < 0 ? -1 : (RS_HIDDEN + UU * KS) % RING_STEPS;                                    \
        ring.mark_begin(UU == 0 ? 1 : (UU % 2 ? 2 : 3));   /* debug builds: 1 carry tile, 2 / 3 odd / even tile */      \
        if (UU == 0) acc = tile8<true, RS, STASH>(ring, p, q, acc, xhi, xh8, xl8, sc, prev, pc, XL, xhi, xh8, xl8, w8h, w8l, st, \
                                                  st_prev + XL * 1024, CD);                                           \
        else acc = tile8<true, RS, STASH>(ring, p, q, acc, xhi, xh8, xl8, sc, prev, pc, 2 * UU - 2, yhi, yh8, yl8, w8h, w8l, st, \
                                          st_own + (2 * UU - 2) * 1024, CD);                                          \
        ring.mark_end(UU == 0 ? 1 : (UU % 2 ? 2 : 3));                                                                  \
      }
      SUNERF_TILE8(0) SUNERF_TILE8(1) SUNERF_TILE8(2) SUNERF_TILE8(3) SUNERF_TILE8(4) SUNERF_TILE8(5) SUNERF_TILE8(6) SUNERF_TILE8(7)
#undef SUNERF_TILE8
      prev = acc;
    }
    return prev;
  }

  template <int STASH>
  static __device__ __forceinline__ f32x16 out_layer(Ring<D>& ring, Pipe& p, Pipe8& q, const float* bias, int h,
                                                     const Scales sc, half8* xhi, v8i* xh8, v8i* xl8, v8i& w8h, v8i& w8l,
                                                     const f32x16& carry, f32x16& pc, Rsrc st, int st_prev) {
    constexpr int XL = 2 * NT - 2;
    const f32x16 r = tile8<true, RS_HIDDEN, STASH>(ring, p, q, bias_tile(bias, h), xhi, xh8, xl8, sc, carry, pc, XL, xhi, xh8,
                                                   xl8, w8h, w8l, st, st_prev + XL * 1024, KS * 1024);
    return r + pc;
  }
};

template <int D, int STASH, bool FP8C, bool HALF = false>
__global__ __launch_bounds__(THREADS, 1) void render_fwd_kernel(RenderArgs a) {
  static_assert(!HALF || !FP8C, "HALF: classic stream format");
  using M = Mlp<D, HALF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PackedLayout L(D, a.n_linear);
  const StashLayout SL(D, a.n_linear, STASH == 2);
  char* slot = smem;                                            // ring of NSLOT weight pages
  float* bias = (float*)(smem + (size_t)Ring<D>::RING);          // n_bias floats
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;

  {  // biases -> LDS once
    const float* gb = (const float*)(a.packed + L.bias_off());
    for (int i = tid; i < (int)L.n_bias(); i += THREADS) bias[i] = gb[i];
  }
  __syncthreads();
  Ring<D> ring;
  ring.init(a.packed, L.stream_bytes(), (unsigned)(uintptr_t)slot, wave, lane);
  typename M::Pipe pipe;
  pipe.frag = slot + lane * 16;
  M::start(ring, pipe);
  ring.mark_begin(6);   // debug builds: 6 = the whole kernel after start-up
  const int S = a.S;
  const int n_chunks = (S + 31) >> 5;
  const int64_t n_groups = (a.n_rays + WAVES - 1) / WAVES;

  for (int64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
    const int64_t ray_raw = group * WAVES + wave;
    const bool ray_ok = ray_raw < a.n_rays;
    const int64_t ray = ray_ok ? ray_raw : a.n_rays - 1;
    const bool free_points = a.points != nullptr;          // kernel-uniform
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f, tm = 0.f;
    if (!free_points) {
      ox = a.rays_o[ray * 3 + 0]; oy = a.rays_o[ray * 3 + 1]; oz = a.rays_o[ray * 3 + 2];
      dx = a.rays_d[ray * 3 + 0]; dy = a.rays_d[ray * 3 + 1]; dz = a.rays_d[ray * 3 + 2];
      tm = a.times[ray];
    }
    const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);   // torch.norm(rays_d), emission.py:26
    const float* zrow = free_points ? nullptr : a.z_vals + ray * S;

    float carry_T = 1.f;      // product of (absorption + 1e-10) over all previous samples of the ray
    float carry_z = 0.f;      // z of the previous chunk's last sample
    float sum_em = 0.f, sum_abs = 0.f, sum_h = 0.f;

    for (int c = 0; c < n_chunks; ++c) {
      const int i = 32 * c + n;
      const bool valid = i < S;
      const float z = free_points ? (float)i : zrow[valid ? i : S - 1];
      // sampling.py:100 -- product and sum rounded separately
      float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z, pt = tm;
      if (free_points) {   // the query point itself (model.py:44-57 on arbitrary points); the integral below runs on, unread
        const f32x4 q = *(const f32x4*)(a.points + ((size_t)ray * S + (valid ? i : S - 1)) * 4);
        px = q[0]; py = q[1]; pz = q[2]; pt = q[3];
      }
      const float v[4] = {px, py, pz, pt};

      // training: this chunk's slice of the activation stash (lane-adjusted); enc fragments first
      Rsrc st = make_rsrc(nullptr, 0);
      if (STASH) {
        // waves of a ragged last group (no ray) run the same instruction stream: they write to one spare chunk
        const size_t chunk_id = ray_ok ? (size_t)ray_raw * n_chunks + c : (size_t)a.n_rays * n_chunks;
        st = make_rsrc(a.stash + chunk_id * SL.chunk_bytes(), (unsigned)SL.chunk_bytes());   // wave-uniform base
      }
      f32x16 out;
      const float* obias = bias + (size_t)(a.n_linear - 1) * D;
      if constexpr (FP8C && M::SPILL) {
        // d_filter = 512, fp8c arithmetic: ONE set of fp16 heads (128 AGPRs) -- every layer writes its fp16 output to this
        // wave's scratch and its last tile pulls it back in as the next layer's input -- and TWO sets of the 64-deep fp8
        // operands (2 x 64 AGPRs) that alternate as input / output from layer to layer.  (With the fp8 operands in the
        // scratch as well the wave moved 64 KiB per layer each way; 4 waves x 64 KiB is more than this CU's share of the
        // XCD's L2, so all of it went over the fabric: 140 GB per 4.19 M samples, measured -- profiles/hbm_traffic_d512.)
        using M8 = Mlp8<D>;
        constexpr int G = M8::G;
        const Rsrc scratch = make_rsrc(a.scratch + ((size_t)blockIdx.x * WAVES + wave) * ((size_t)M::KS * 2048), M::KS * 2048);
        half8 x_hi[M::KS];
        v8i a_h8[G], a_l8[G], b_h8[G], b_l8[G];
        half8 e_hi[SUNERF_KS0], e_lo[SUNERF_KS0];
        encode_point(v, h, [&](int q, float val) {
          const _Float16 hi = (_Float16)val;
          e_hi[q >> 3][q & 7] = hi;
          e_lo[q >> 3][q & 7] = (_Float16)(val - (float)hi);
        });
        if (STASH) {
#pragma unroll
          for (int s = 0; s < SUNERF_KS0; ++s) buf_store(e_hi[s], st, s * 1024);
        }
        const int* shp = (const int*)(a.packed + L.scale_off());
        if (c != 0 || group != (int64_t)blockIdx.x) {
#pragma unroll
          for (int s = 0; s < M::PF; ++s) M::load_frag(pipe, s, (M::RS_IN + s) % M::RING_STEPS);
        }
        v8i w8h = {0}, w8l = {0};
        f32x16 carry = M8::template in_layer<STASH>(ring, pipe, bias, h, e_hi, e_lo, x_hi, a_h8, a_l8, w8h, w8l, st,
                                                    (int)SL.h_off(0), scratch);
        // the in layer's tiles are too short to pull the next input in piece by piece: fetch all fp16 fragments but the
        // last two (they arrive through the carry epilogue), one group at a time
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int sl = 0; sl < 4; ++sl)
            if (4 * g + sl < M::KS - 2) x_hi[4 * g + sl] = buf_load(scratch, (4 * g + sl) * 1024);
#pragma unroll
          for (int sl = 0; sl < 4; ++sl)
            if (4 * g + sl < M::KS - 2) pin_agpr(x_hi[4 * g + sl]);
          __builtin_amdgcn_sched_barrier(0);
        }
        typename M8::Pipe8 q8;
        f32x16 pc = {0};
        M8::preload(pipe, q8, M::RS_HIDDEN);
        auto scales = [&](int l) {
          const int sh = shp[l];
          typename M8::Scales sc = {127 - (sh + 11), 127 - sh};
          return sc;
        };
        int l = 1;
        for (; l + 1 < a.n_linear - 1; l += 2) {
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)l * D, h, scales(l), x_hi, a_h8, a_l8, x_hi, b_h8,
                                                   b_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l - 1), (int)SL.h_off(l), scratch);
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)(l + 1) * D, h, scales(l + 1), x_hi, b_h8, b_l8,
                                                   x_hi, a_h8, a_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l), (int)SL.h_off(l + 1),
                                                   scratch);
        }
        if (l < a.n_linear - 1) {
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)l * D, h, scales(l), x_hi, a_h8, a_l8, x_hi, b_h8,
                                                   b_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l - 1), (int)SL.h_off(l), scratch);
          out = M8::template out_layer<STASH>(ring, pipe, q8, obias, h, scales(a.n_linear - 1), x_hi, b_h8, b_l8, w8h, w8l, carry, pc,
                                              st, (int)SL.h_off(a.n_linear - 2));
        } else {
          out = M8::template out_layer<STASH>(ring, pipe, q8, obias, h, scales(a.n_linear - 1), x_hi, a_h8, a_l8, w8h, w8l, carry, pc,
                                              st, (int)SL.h_off(a.n_linear - 2));
        }
      } else if constexpr (FP8C) {
        using M8 = Mlp8<D>;
        constexpr int G = M8::G;
        // two operand sets in the fp8c format: fp16 heads + the two 64-deep fp8 operands (head, scaled remainder)
        half8 xa_hi[M::KS], xb_hi[M::KS];
        v8i xa_h8[G], xa_l8[G], xb_h8[G], xb_l8[G];
        half8 e_hi[SUNERF_KS0], e_lo[SUNERF_KS0];            // encoding, classic hi | lo operands of the in layer
        encode_point(v, h, [&](int q, float val) {
          const _Float16 hi = (_Float16)val;
          e_hi[q >> 3][q & 7] = hi;
          e_lo[q >> 3][q & 7] = (_Float16)(val - (float)hi);
        });
        if (STASH) {
#pragma unroll
          for (int s = 0; s < SUNERF_KS0; ++s) buf_store(e_hi[s], st, s * 1024);
        }
        const int* shp = (const int*)(a.packed + L.scale_off());
        // the in layer's first k-steps: nothing prefetched them (the previous chunk ended with an fp8c tile)
        if (c != 0 || group != (int64_t)blockIdx.x) {
#pragma unroll
          for (int s = 0; s < M::PF; ++s) {
            int rs = M::RS_IN >= 0 ? (M::RS_IN + s) % M::RING_STEPS : pipe.rstep + s;
            if (M::RS_IN < 0 && rs >= M::RING_STEPS) rs -= M::RING_STEPS;
            M::load_frag(pipe, s, rs);
          }
        }
        v8i w8h = {0}, w8l = {0};                              // fp8 operands of the group under construction
#if SUNERF_DBG_BARRIER
        ring.dbg_kind = 0;
#endif
        ring.mark_end(5);
        ring.mark_begin(0);
        f32x16 carry = M8::template in_layer<STASH>(ring, pipe, bias, h, e_hi, e_lo, xa_hi, xa_h8, xa_l8, w8h, w8l, st, (int)SL.h_off(0));
        ring.mark_end(0);
#if SUNERF_DBG_BARRIER
        ring.dbg_kind = 1;
#endif
        typename M8::Pipe8 q8;
        f32x16 pc = {0};                                       // correction accumulator of the pending tile (in layer: none)
        M8::preload(pipe, q8, M::RS_HIDDEN >= 0 ? M::RS_HIDDEN : pipe.rstep);
        auto scales = [&](int l) {
          const int sh = shp[l];
          typename M8::Scales sc = {127 - (sh + 11), 127 - sh};
          return sc;
        };
        int l = 1;
        for (; l + 1 < a.n_linear - 1; l += 2) {
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)l * D, h, scales(l), xa_hi, xa_h8, xa_l8, xb_hi,
                                                   xb_h8, xb_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l - 1), (int)SL.h_off(l));
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)(l + 1) * D, h, scales(l + 1), xb_hi, xb_h8,
                                                   xb_l8, xa_hi, xa_h8, xa_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l), (int)SL.h_off(l + 1));
        }
        if (l < a.n_linear - 1) {
#if SUNERF_DBG_BARRIER
          ring.dbg_kind = 2;
#endif
          carry = M8::template hidden_layer<STASH>(ring, pipe, q8, bias + (size_t)l * D, h, scales(l), xa_hi, xa_h8, xa_l8, xb_hi,
                                                   xb_h8, xb_l8, w8h, w8l, carry, pc, st, (int)SL.h_off(l - 1), (int)SL.h_off(l));
#if SUNERF_DBG_BARRIER
          ring.dbg_kind = 3;
#endif
          ring.mark_begin(4);
          out = M8::template out_layer<STASH>(ring, pipe, q8, obias, h, scales(a.n_linear - 1), xb_hi, xb_h8, xb_l8, w8h, w8l, carry, pc, st,
                                              (int)SL.h_off(l));
          ring.mark_end(4);
          ring.mark_begin(5);
        } else {
          out = M8::template out_layer<STASH>(ring, pipe, q8, obias, h, scales(a.n_linear - 1), xa_hi, xa_h8, xa_l8, w8h, w8l, carry, pc, st,
                                              (int)SL.h_off(l - 1));
        }
      } else if constexpr (!M::SPILL) {   // exact mode: three fp16 products per term (hi*hi + hi*lo + lo*hi)
        half8 xa_hi[M::XK], xa_lo[M::XK], xb_hi[M::XK], xb_lo[M::XK];
        {  // positional encoding straight into the in-layer B fragments (k-steps 0..5 of xb)
          encode_point(v, h, [&](int q, float val) {
            const _Float16 hi = (_Float16)val;
            xb_hi[q >> 3][q & 7] = hi;
            xb_lo[q >> 3][q & 7] = (_Float16)(val - (float)hi);
          });
#pragma unroll
          for (int s = 0; s < SUNERF_KS0; ++s) { pin_agpr(xb_hi[s]); pin_agpr(xb_lo[s]); }
        }
        if (STASH) {
#pragma unroll
          for (int s = 0; s < SUNERF_KS0; ++s) buf_store(xb_hi[s], st, s * 1024);
        }
        // in layer: 84(96) -> D
        f32x16 carry = {0};
        const Rsrc none = make_rsrc(nullptr, 0);
        carry = M::template layer<SUNERF_KS0, false, M::RS_IN, STASH>(ring, pipe, bias, h, xb_hi, xb_lo, xa_hi, xa_lo, carry,
                                                                      st, 0, (int)SL.h_off(0), none);
        // hidden layers, ping-pong between the two register sets
        int l = 1;
        for (; l + 1 < a.n_linear - 1; l += 2) {
          carry = M::template layer<M::KS, true, M::RS_HIDDEN, STASH>(ring, pipe, bias + (size_t)l * D, h, xa_hi, xa_lo, xb_hi, xb_lo,
                                                                      carry, st, (int)SL.h_off(l - 1), (int)SL.h_off(l), none);
          carry = M::template layer<M::KS, true, M::RS_HIDDEN, STASH>(ring, pipe, bias + (size_t)(l + 1) * D, h, xb_hi, xb_lo, xa_hi,
                                                                      xa_lo, carry, st, (int)SL.h_off(l), (int)SL.h_off(l + 1), none);
        }
        if (l < a.n_linear - 1) {
          carry = M::template layer<M::KS, true, M::RS_HIDDEN, STASH>(ring, pipe, bias + (size_t)l * D, h, xa_hi, xa_lo, xb_hi, xb_lo,
                                                                      carry, st, (int)SL.h_off(l - 1), (int)SL.h_off(l), none);
          out = M::template out_layer<STASH>(ring, pipe, obias, h, xb_hi, xb_lo, carry, st, (int)SL.h_off(l), none);
        } else {
          out = M::template out_layer<STASH>(ring, pipe, obias, h, xa_hi, xa_lo, carry, st, (int)SL.h_off(l - 1), none);
        }
      } else {
        // d_filter = 512: ONE activation set in registers (256 AGPRs); every layer writes its output fragments to this
        // wave's scratch and the last tile of the layer pulls them back in as the next layer's input
        const Rsrc scratch = make_rsrc(a.scratch + ((size_t)blockIdx.x * WAVES + wave) * ((size_t)M::KS * 2048), M::KS * 2048);
        half8 x_hi[M::KS], x_lo[M::KS];
        encode_point(v, h, [&](int q, float val) {
          const _Float16 hi = (_Float16)val;
          x_hi[q >> 3][q & 7] = hi;
          x_lo[q >> 3][q & 7] = (_Float16)(val - (float)hi);
        });
#pragma unroll
        for (int s = 0; s < SUNERF_KS0; ++s) { pin_agpr(x_hi[s]); pin_agpr(x_lo[s]); }
        if (STASH) {
#pragma unroll
          for (int s = 0; s < SUNERF_KS0; ++s) buf_store(x_hi[s], st, s * 1024);
        }
        f32x16 carry = {0};
        carry = M::template layer<SUNERF_KS0, false, M::RS_IN, STASH>(ring, pipe, bias, h, x_hi, x_lo, nullptr, nullptr, carry,
                                                                      st, 0, (int)SL.h_off(0), scratch);
        // the in layer's tiles are too short to pull the next input in one by one: fetch fragments 0 .. KS-3 now (the
        // last two arrive through the carry epilogue)
        // (in groups of 4 k-steps: all 60 loads in flight at once would need 240 transit VGPRs on their way to the AGPRs)
#pragma unroll
        for (int s0 = 0; s0 < M::KS - 2; s0 += 4) {
#pragma unroll
          for (int s = s0; s < s0 + 4 && s < M::KS - 2; ++s) {
            x_hi[s] = buf_load(scratch, s * 2048);
            x_lo[s] = buf_load(scratch, s * 2048 + 1024);
          }
#pragma unroll
          for (int s = s0; s < s0 + 4 && s < M::KS - 2; ++s) { pin_agpr(x_hi[s]); pin_agpr(x_lo[s]); }
          __builtin_amdgcn_sched_barrier(0);
        }
        for (int l = 1; l < a.n_linear - 1; ++l)
          carry = M::template layer<M::KS, true, M::RS_HIDDEN, STASH>(ring, pipe, bias + (size_t)l * D, h, x_hi, x_lo, nullptr, nullptr,
                                                                      carry, st, (int)SL.h_off(l - 1), (int)SL.h_off(l), scratch);
        out = M::template out_layer<STASH>(ring, pipe, obias, h, x_hi, x_lo, carry, st, (int)SL.h_off(a.n_linear - 2), scratch);
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
        if (a.weights) {
          a.weights[o] = em;  // un-normalised; finalised below by the same lane
          a.absorption[o] = absn;
        }
        if (a.raw) { a.raw[o * 2 + 0] = r0; a.raw[o * 2 + 1] = r1; }
        if (a.regularization) a.regularization[o] = fmaxf(pdist - a.reg_radius, 0.f) * (1.f - absn);
      }
    }
    // ---- per-ray finalisation: normalise weights (emission.py:49-50), per-ray outputs ----
    if (ray_ok && h == 0 && a.weights) {
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
#if SUNERF_DBG_BARRIER
  if (lane == 0 && blockIdx.x < 64) {
    float* o = a.weights + ((size_t)blockIdx.x * 4 + wave) * 16;
    o[0] = (float)ring.dbg_vm; o[1] = (float)ring.dbg_bar; o[2] = (float)ring.dbg_n;
  }
#endif
  ring.mark_end(6);
#if SUNERF_DBG_TILES
  if (lane == 0 && blockIdx.x < 64) {
    float* o = a.weights + ((size_t)blockIdx.x * 4 + wave) * 16;
    o[0] = (float)ring.dbg_cyc; o[1] = 0.f; o[2] = (float)ring.dbg_cnt;
  }
#endif
}

template <int D, int STASH, bool FP8C, bool HALF = false>
int launch_render_t(const RenderArgs& a, hipStream_t stream) {
  const PackedLayout L(D, a.n_linear);
  const size_t lds = (size_t)Ring<D>::RING + L.n_bias() * 4;
  if (lds > 160 * 1024) return SUNERF_E_UNSUPPORTED;
  hipError_t e = hipFuncSetAttribute((const void*)render_fwd_kernel<D, STASH, FP8C, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int64_t n_groups = (a.n_rays + WAVES - 1) / WAVES;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus > 1024) cus = 1024;   // sunerf_render_workspace_bytes is sized for at most 1024 workgroups
  cus = sunerf_grid_cap("SUNERF_GRID_CAP_FWD", cus);
  const unsigned grid = (unsigned)(n_groups < cus ? n_groups : cus);
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL((render_fwd_kernel<D, STASH, FP8C, HALF>), dim3(grid), dim3(THREADS), lds, stream, a);
  SUNERF_CHECK_LAUNCH();
  return 0;
}
template <int D>
int launch_render(const RenderArgs& a, int precision, hipStream_t stream) {
  if constexpr (D == 256) {        // the 16-bit phase stash is what the pipelined backward reads, and that exists at this width only
    if (a.stash && a.stash_phase) {
      if (precision == SUNERF_PRECISION_FAST) return launch_render_t<D, 2, true>(a, stream);
      if (precision == SUNERF_PRECISION_HALF) return launch_render_t<D, 2, false, true>(a, stream);
      return launch_render_t<D, 2, false>(a, stream);
    }
  }
  if (a.stash && a.stash_phase) return SUNERF_E_UNSUPPORTED;
  if (precision == SUNERF_PRECISION_FAST)
    return a.stash ? launch_render_t<D, 1, true>(a, stream) : launch_render_t<D, 0, true>(a, stream);
  if (precision == SUNERF_PRECISION_HALF)
    return a.stash ? launch_render_t<D, 1, false, true>(a, stream) : launch_render_t<D, 0, false, true>(a, stream);
  return a.stash ? launch_render_t<D, 1, false>(a, stream) : launch_render_t<D, 0, false>(a, stream);
}

}  // namespace

// d_filter = 512 keeps one activation set in registers and spills layer outputs to a per-wave scratch: at most 256
// workgroups (one per CU) x 4 waves x (D/16) fragments x 2 KiB
extern "C" size_t sunerf_render_workspace_bytes(int d_filter) {
  return d_filter > 256 ? (size_t)1024 * 4 * (size_t)(d_filter / 16) * 2048 : 0;
}

extern "C" size_t sunerf_act_stash_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear, int stash_format) {
  if (n_rays < 0 || n_samples < 1 || d_filter < 32 || d_filter % 32 || n_linear < 2) return 0;
  if (stash_format != SUNERF_STASH_FP16 && !(stash_format == SUNERF_STASH_PHASE && d_filter == 256)) return 0;
  const int64_t chunks = n_rays * ((n_samples + 31) / 32) + 1;   // + one spare chunk for ragged groups
  return (size_t)chunks * StashLayout(d_filter, n_linear, stash_format == SUNERF_STASH_PHASE).chunk_bytes();
}

extern "C" int sunerf_emission_render_fwd(const void* packed, int d_filter, int n_linear, int precision, const float* rays_o,
                                          const float* rays_d, const float* times, const float* z_vals,
                                          int64_t n_rays, int n_samples, float* image, float* weights,
                                          float* absorption, float* raw, float* height_map, float* absorption_map,
                                          float* regularization, float reg_radius, void* act_stash, int stash_format,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rays < 0 || n_samples < 2) return SUNERF_E_BADARG;
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return SUNERF_E_UNSUPPORTED;
  if (precision != SUNERF_PRECISION_FAST && precision != SUNERF_PRECISION_EXACT && precision != SUNERF_PRECISION_HALF)
    return SUNERF_E_BADARG;
  if (n_rays == 0) return 0;      // an empty batch is valid (its tensors have null data pointers)
  if (!packed || !rays_o || !rays_d || !times || !z_vals || !image || !weights || !absorption) return SUNERF_E_BADARG;
  RenderArgs a;
  a.packed = (const char*)packed; a.rays_o = rays_o; a.rays_d = rays_d; a.times = times; a.z_vals = z_vals; a.points = nullptr;
  a.n_rays = n_rays; a.S = n_samples; a.n_linear = n_linear; a.image = image; a.weights = weights;
  a.absorption = absorption; a.raw = raw; a.height_map = height_map; a.absorption_map = absorption_map;
  a.regularization = regularization; a.reg_radius = reg_radius; a.stash = (char*)act_stash;
  if (stash_format != SUNERF_STASH_FP16 && stash_format != SUNERF_STASH_PHASE) return SUNERF_E_BADARG;
  a.stash_phase = stash_format == SUNERF_STASH_PHASE;
  a.scratch = (char*)workspace;
  if (workspace_bytes < sunerf_render_workspace_bytes(d_filter)) return SUNERF_E_WORKSPACE;
  switch (d_filter) {
    case 64: return launch_render<64>(a, precision, (hipStream_t)stream);
    case 128: return launch_render<128>(a, precision, (hipStream_t)stream);
    case 256: return launch_render<256>(a, precision, (hipStream_t)stream);
    case 512: return launch_render<512>(a, precision, (hipStream_t)stream);
    default: return SUNERF_E_UNSUPPORTED;
  }
}

extern "C" int sunerf_mlp_points_fwd(const void* packed, int d_filter, int n_linear, int precision, const float* points,
                                     int64_t n_points, float* raw, void* act_stash, int stash_format, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (n_points < 0 || n_points % 32) return SUNERF_E_BADARG;       // whole 32-point chunks (callers pad)
  if (n_linear < 2 || n_linear > SUNERF_MAX_LAYERS) return SUNERF_E_UNSUPPORTED;
  if (precision != SUNERF_PRECISION_FAST && precision != SUNERF_PRECISION_EXACT && precision != SUNERF_PRECISION_HALF)
    return SUNERF_E_BADARG;
  if (n_points == 0) return 0;
  if (!packed || !points || !raw) return SUNERF_E_BADARG;
  RenderArgs a;
  a.packed = (const char*)packed; a.rays_o = nullptr; a.rays_d = nullptr; a.times = nullptr; a.z_vals = nullptr; a.points = points;
  a.n_rays = n_points / 32; a.S = 32; a.n_linear = n_linear; a.image = nullptr; a.weights = nullptr; a.absorption = nullptr;
  a.raw = raw; a.height_map = nullptr; a.absorption_map = nullptr; a.regularization = nullptr; a.reg_radius = 0.f;
  a.stash = (char*)act_stash; a.scratch = (char*)workspace;
  if (stash_format != SUNERF_STASH_FP16 && stash_format != SUNERF_STASH_PHASE) return SUNERF_E_BADARG;
  a.stash_phase = stash_format == SUNERF_STASH_PHASE;
  if (workspace_bytes < sunerf_render_workspace_bytes(d_filter)) return SUNERF_E_WORKSPACE;
  switch (d_filter) {
    case 64: return launch_render<64>(a, precision, (hipStream_t)stream);
    case 128: return launch_render<128>(a, precision, (hipStream_t)stream);
    case 256: return launch_render<256>(a, precision, (hipStream_t)stream);
    case 512: return launch_render<512>(a, precision, (hipStream_t)stream);
    default: return SUNERF_E_UNSUPPORTED;
  }
}
