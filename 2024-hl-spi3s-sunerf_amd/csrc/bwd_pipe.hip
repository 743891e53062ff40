// Layer-pipelined backward of the sine MLP (gfx950, d_filter = 256): data AND weight gradients of every Linear layer from
// the activation stash, without ever writing the hidden layers' dZ to HBM.
//
// Replaces what torch.autograd derives from sunerf/model/model.py:44-57 (reference root), like sunerf_mlp_dgrad +
// sunerf_mlp_wgrad, whose two kernels move 16.4 KB per sample of an 8 x 256 network (cos in, dZ out; H and dZ in).  Here the
// workgroup that produces dZ_{l-1} hands it to the workgroup that consumes it through the XCD's L2, and the forward leaves
// one 16-bit PHASE per pre-activation (SUNERF_STASH_PHASE) from which this kernel recovers sin and cos: the backward reads
// 2 bytes per activation once (4.2 KB per sample with the encoding) plus the top layer's dZ (0.5 KB, written by the prologue).
//
//   dW_l[j][k]  = sum_n dZ_l[j][n] H_{l-1}[k][n]        db_l[j] = sum_n dZ_l[j][n]
//   dZ_{l-1}[k][n] = (sum_j W_l[j][k] dZ_l[j][n]) cos(Z_{l-1})[k][n]
//
// STAGES.  A CU cannot hold a whole layer (dW_l alone is all 256 AGPRs of its four waves), two can: stage l is a PAIR of
// workgroups, workgroup j of it owning the 128 features J_j = [128 j, 128 j + 128) of layer l-1 -- the columns J_j of dW_l
// (32 accumulator tiles = 128 registers per wave), the rows J_j of W_l^T (each data-gradient wave keeps ITS output tile's 16
// A fragments in registers for the whole launch: 64 registers, + 64 for the low parts unless HI_ONLY) and the fragments J_j of
// the phases of layer l-1 and of dZ_{l-1}.  Per 32-sample chunk it takes in dZ_l (16 KB, from the two workgroups of stage l+1)
// and P_{l-1}[J] (8 KB, HBM, every byte read once chip-wide) by LDS-DMA, and puts out its half of dZ_{l-1} (8 KB) and, at the
// end of the launch, its share of db_{l-1} (fp32 sums of dH cos kept by the wave that forms the tile).  A pipeline is
// the stages l = n_act-1 ... 1 plus the in-layer stage (dW_0 from dZ_0 and the encoding stash): 2 (n_linear - 1) workgroups,
// all on ONE XCD so that the hand-off never leaves that XCD's L2 (plain stores; the consumer's LDS-DMA reads are `sc1`, i.e.
// served by L2 past its own stale L1).  8 x 256 network: 16 workgroups per pipeline, two pipelines per XCD, 16 in all, each
// working through 1/16 of the chunks.
//
// The top stage's input, dZ of the last activation layer, that layer's db and the out layer's own dW / db come from a
// streaming prologue kernel (g_raw and the phases of the last activation layer -> dZ_top, 512 B per sample).
//
// HAND-OFF.  Per link (pipeline, layer) a ring of RING chunk slots (16 KB each) in device memory and four monotonic
// counters, each with ONE writer: prod[j] = chunks whose half producer j has stored AND drained (in-order vmcnt: known
// DRAIN iterations later), cons[j] = chunks consumer j has landed in its LDS.  A consumer may fetch chunk c once
// min(prod) > c; a producer may overwrite slot c % RING once min(cons) > c - RING.  The counters are polled AHEAD by LDS-DMA
// (a 4-byte-per-lane DMA into a small LDS slot, issued PIPE_POLL_LAG iterations before its value is read), so that a wave
// never waits for a flag round trip unless the partner really is late; only then it falls back to a bounded spin.  The ring's
// 16 slots are shared by the age of the polled counters, the distance at which dZ is requested (PIPE_ZD) and the publication
// delay of the outputs (DRAIN): each iteration saved there is an iteration of slack between two stages (round 4: 4 + 4 + 4 ->
// 2 + 2 + 3; before, every stage spent 15 % of the launch on round trips for counters that were merely old).
// Placement is CHECKED, not assumed: every workgroup publishes its XCC_ID,
// and a class (blockIdx % 8) that does not sit on one XCD makes the whole launch give up (status word; the reduce kernel then
// writes NaN gradients, which the optimiser's non-finite guard skips, raises the STICKY status word in front of the workspace,
// and the host falls back to the two-kernel backward).  Every spin is bounded by s_memrealtime and watches the status word,
// so the grid always drains.
#include "grad_common.h"
#include <mutex>
#include <vector>

namespace {

constexpr int PD = 256, PKS = PD / 16, PNT = PD / 32, PT = 8;   // d_filter, fragments / tiles per layer, workspace tiles
constexpr int WG = 512;                 // pipelined kernel: eight waves, two per SIMD
constexpr int WG_PRE = 256;             // prologue kernel
constexpr int NBUF = 4;                 // staging buffers of the in-layer stage and of the prologue (chunks in flight: NBUF - 1)
constexpr int NBUF_H = 5;               // ... of a hidden stage: chunk it + 4 is being fetched, chunk it + 1's phases are decoded, chunk it is consumed
#ifndef PIPE_DMA_ON_WEIGHT
#define PIPE_DMA_ON_WEIGHT 2            // who issues the LDS-DMA pieces of a hidden stage: see "WHO ISSUES THE PIECES" below (0: 11.48, 1: 11.38, 2: 11.04 ms, r4_pipe_ab11).
                                        // Three more variants (3 / 7 / 7 / 7 pieces, protocol split over two waves, the data waves' tile in VGPRs) were measured slower
                                        // and removed again: commit cafd360 has them, tools/experiments/README.md the numbers
#endif
#ifndef PIPE_ZD
#define PIPE_ZD 2                       // hidden stages below the top: dZ_l is requested this many iterations before it is consumed
#endif
#ifndef PIPE_RING
#define PIPE_RING 16
#endif
#ifndef PIPE_POLL_LAG
#define PIPE_POLL_LAG 2                 // iterations between the issue of a counter poll and the gate that reads it (1 .. NBUF_H - 1); 4 -> 2: kernel 12.2 -> 11.8 ms (r4_pipe_ab7)
#endif
constexpr int RING = PIPE_RING;                // chunk slots per hand-off ring
constexpr int SLOT = PKS * 1024;        // one chunk of dZ: 16 fragments
constexpr int BUF_HID = 24 * 1024;      // hidden stage staging: dZ_l 16 | P[J] 8 KiB (16-bit phases, decoded IN PLACE to fp16 sin = H[J])
constexpr int BUF_IN = 16 * 1024;       // in-layer stage staging: dZ_0[J] 8 | enc 6 (| 2 unused) KiB
constexpr int BUF_PRE = 19 * 1024;      // prologue staging: P 16 (decoded in place to H) | dZ_out operand tile 2 KiB | g_raw of the chunk 256 B
constexpr unsigned SPIN_LIMIT = 50000000u;   // s_memrealtime ticks (100 MHz): 0.5 s

// status codes (word 0 of the control block)
constexpr unsigned ST_OK = 0, ST_START_TIMEOUT = 1, ST_PLACEMENT = 2, ST_WAIT_TIMEOUT = 3;

struct PipeArgs {
  const char* packedT;
  const char* act_stash;
  const char* dz_top;       // [chunk][16 fragments] fp16, written by the prologue kernel
  char* rings;              // [pipeline][link n_act-1][RING][SLOT] + scratch for the dummy stores
  unsigned* ctrl;           // control block (zeroed before every launch)
  float* partial;           // [layer 0 .. n_act-1][pipeline][72 tiles][16][64]
  const float* g_raw;
  const unsigned* g_absmax_bits;
  float* partial_out;       // prologue: [workgroup][10 tiles][16][64]: dW_out (8), db_out, db of the last activation layer
  int64_t n_chunks_total;
  int S, n_chunks;          // samples per ray, chunks per ray
  int n_linear, d_out;
  int NP, NPX;              // pipelines in all / per XCD class
  int hi_only;
  int inject;               // test hook (flags bit 8): workgroup class 0 reports a second XCD -> the launch gives up (status 2)
  unsigned* dbg;            // optional [8 regions][workgroup][8] (regions 4, 5: one iteration's timeline of pipeline 0): loop ticks, fallback spins / ticks of the input and output link, chunks, layer, pipeline
};

// control block: word 0 status, 1 arrived, 8..15 XCC masks per class; counters from word 64, each on a 128-byte line
__host__ __device__ inline size_t ctrl_counter_word(int n_links, int P, int link, int which) {
  return 64 + (((size_t)P * n_links + link) * 4 + which) * 32;
}
__host__ __device__ inline size_t ctrl_bytes(int NP, int n_links) { return (64 + (size_t)NP * n_links * 4 * 32) * 4; }

typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) {
  __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS-DMA of one 1 KiB piece (16 bytes per lane); POLICY 0 = default, 1 = nt (read-once HBM stream), 2 = sc1 (handed-off
// bytes: past this CU's L1)
template <int POLICY>
__device__ __forceinline__ void dma_piece(const char* src_lane, unsigned lds_dst) {
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
  if (POLICY == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src_lane), "s"(d) : "memory");
  else if (POLICY == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" :: "v"(src_lane), "s"(d) : "memory");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src_lane), "s"(d) : "memory");
}
// the same with a wave-uniform source (scalar base) and the lane part (lane * 16) as the 32-bit vector offset
template <int POLICY>
__device__ __forceinline__ void dma_piece_s(const char* src_uniform, unsigned voff, unsigned lds_dst) {
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
#ifndef PIPE_STREAM_BITS
#define PIPE_STREAM_BITS "nt"      // cache policy of the read-once HBM streams (phases, top dZ); experiment: "sc0 sc1 nt", "sc1 nt", "sc0 nt"
#endif
  if (POLICY == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 " PIPE_STREAM_BITS :: "v"(voff), "s"(d), "s"(src_uniform) : "memory");
  else if (POLICY == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 sc1" :: "v"(voff), "s"(d), "s"(src_uniform) : "memory");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(d), "s"(src_uniform) : "memory");
}
// counter poll by LDS-DMA: lane i's dword lands at lds_dst + 4 i
__device__ __forceinline__ void dma_poll(const unsigned* src_lane, unsigned lds_dst) {
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1" :: "v"(src_lane), "s"(d) : "memory");
}

// one accumulator tile -> workspace [reg 16][lane 64]: buffer stores (one resource, the lane part as the only address VGPR;
// with plain pointers hipcc forms all 128 addresses of a wave's eight tiles first -- 256 VGPRs -- and spills)
__device__ __forceinline__ void store_tile(const f32x16& t, float* dst) {
  const Rsrc r = make_rsrc(dst, 16 * 64 * 4);
  const int lo4 = (int)(threadIdx.x & 63u) * 4;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    // read out through an asm with an AGPR operand: a plain element read makes hipcc give the whole loop-carried tile a VGPR
    // home and copy it to AGPRs and back around every matrix instruction
    const float te = t[g];
    unsigned tv;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tv) : "a"(te));
    __builtin_amdgcn_raw_buffer_store_b32(tv, r, lo4, g * 256, 0);
    // (keeps the accumulator reads next to their stores: hoisted together, the reads of a wave's eight tiles want 128 VGPRs)
    if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// Accumulate into an AGPR-resident tile.  Written as asm because hipcc (ROCm 7.2) keeps loop-carried accumulators of the
// builtin in VGPRs when they would fit there and copies 16 registers to AGPRs and back around every instruction; the "+a"
// operand pins the tile.  Hazards: the operands come from v_mov / ds_read (hardware-interlocked); nothing reads a tile
// between two of its own accumulations (>= 7 other matrix instructions apart) and the final read-out is preceded by
// drain_matrix_pipe().
__device__ __forceinline__ void mfma_agpr(f32x16& acc, const half8& a, const half8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void drain_matrix_pipe() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

// Transposed operand read straight into FOUR CONSECUTIVE, FIXED registers (the matrix instruction's operand tuple): the two
// ds_read_b64_tr_b16 halves of tr_issue land in separate register pairs and cost 4 v_mov per operand to join -- 48 per chunk
// in a wave whose instruction stream is the kernel's critical path.  R0..R3: VGPR numbers, OFF: immediate byte offset.
#ifndef PIPE_KO_WREADS
#define PIPE_KO_WREADS 0
#endif
#if PIPE_KO_WREADS      // knock-out (results garbage): the weight-gradient waves' transposed operand reads are not issued
#define TR_FIXED(R0, R1, R2, R3, OFF, var, base) asm volatile("" : "={v[" #R0 ":" #R3 "]}"(var) : "v"(base) : "memory")
#else
#define TR_FIXED(R0, R1, R2, R3, OFF, var, base)                                                                        \
  asm volatile("ds_read_b64_tr_b16 v[" #R0 ":" #R1 "], %1 offset:%2\n\tds_read_b64_tr_b16 v[" #R2 ":" #R3 "], %1 offset:%3" \
               : "={v[" #R0 ":" #R3 "]}"(var) : "v"(base), "i"(OFF), "i"((OFF) + 64) : "memory")
#endif

// dZ = dH * cos of one 32 x 32 tile, rounded to (saturating) fp16 for the chain -- and, before that rounding, added to this
// lane's running bias sums `bs` (db = sum over samples of dZ, in fp32: round 4; until then the weight-gradient waves summed
// the ROUNDED dZ with v_dot2_f32_f16).  bs[g] = register g of the tile: row acc_row(g, h), this lane's sample column.
__device__ __forceinline__ void dz_tile(const f32x16& acc, const float* c0, const float* c1, half8& d0, half8& d1, float* bs) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float p0 = acc[j] * c0[j], p1 = acc[8 + j] * c1[j];
    bs[j] += p0;
    bs[8 + j] += p1;
    d0[j] = (_Float16)sunerf_sat16(p0);
    d1[j] = (_Float16)sunerf_sat16(p1);
  }
}
#ifndef PIPE_TRIM
#define PIPE_TRIM 1      // 1: the data-gradient waves' epilogue and decoder on explicit register PAIRS (g, g + 1): packed fp32 multiplies / adds without
                         // the 13 v_mov per chunk that hipcc's own pairing (j, 8 + j) needed, the phase scaling as 8 packed multiplies
#endif
#ifndef PIPE_FP16_OVFL
#define PIPE_FP16_OVFL 1  // 1: MODE.FP16_OVFL set in the data-gradient waves (an overflowing f16 conversion clamps to +-65504, true infinities stay: probed,
                          // tools/probes/probe_fp16_ovfl.hip) instead of 16 v_med3 per chunk: kernel 10.90 -> 10.76 ms (r4_pipe_ab20)
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
// dz_tile on register pairs: p = acc * cos pairwise, fp32 bias sums, (saturating) fp16
__device__ __forceinline__ void dz_tile_pairs(const f32x16& acc, const v2f* cs2, half8& d0, half8& d1, v2f* bs2) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const v2f a = {acc[2 * q], acc[2 * q + 1]};
    const v2f p = a * cs2[q];
    bs2[q] += p;
    const float x = PIPE_FP16_OVFL ? p.x : sunerf_sat16(p.x), y = PIPE_FP16_OVFL ? p.y : sunerf_sat16(p.y);
    if (q < 4) { d0[2 * q] = (_Float16)x; d0[2 * q + 1] = (_Float16)y; }
    else { d1[2 * q - 8] = (_Float16)x; d1[2 * q - 7] = (_Float16)y; }
  }
}
// 8 phases of one fragment -> fp16 sin (in `h`) and cos pairs
__device__ __forceinline__ half8 decode_phases_pairs(const v4u& p, v2f* cs2) {
  half8 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v2f r = {(float)(p[q] & 0xffffu), (float)(p[q] >> 16)};
    r = r * (v2f){SUNERF_PHASE_SCALE, SUNERF_PHASE_SCALE};
    h[2 * q] = (_Float16)__builtin_amdgcn_sinf(r.x);
    h[2 * q + 1] = (_Float16)__builtin_amdgcn_sinf(r.y);
    cs2[q] = (v2f){__builtin_amdgcn_cosf(r.x), __builtin_amdgcn_cosf(r.y)};
  }
  return h;
}
// the 32 bias sums of a tile from the 16 x 64 per-lane sums: register g on lane half h is fragment-order index
// 16 (g >> 3) + 8 h + (g & 7) of the tile (grad_common.h: reduce_grads_kernel reads bias slot `lane` as that index)
__device__ __forceinline__ void store_bias_sums(const float* bs, float* dst32) {
  const int lane = threadIdx.x & 63, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float v = bs[g];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 32);
    if ((lane & 31) == 0) dst32[16 * (g >> 3) + 8 * h + (g & 7)] = v;
  }
}

// 8 phases (one lane's 16 bytes of a phase fragment, sunerf_common.h: SUNERF_STASH_PHASE) -> sin and cos of the pre-activations
__device__ __forceinline__ void decode_phases(const v4u& p, float* sn, float* cs) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned w = p[e >> 1];
    const float rev = (float)((e & 1) ? (w >> 16) : (w & 0xffffu)) * SUNERF_PHASE_SCALE;
    sn[e] = __builtin_amdgcn_sinf(rev);
    cs[e] = __builtin_amdgcn_cosf(rev);
  }
}
__device__ __forceinline__ half8 to_half8(const float* v) {
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (_Float16)v[e];
  return r;
}

// 4-byte LDS words by LDS address: a `volatile` access through a generic pointer becomes a FLAT load, which hipcc follows
// with s_waitcnt vmcnt(0) -- draining the whole LDS-DMA queue every iteration
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned lds_ld(unsigned addr) { return *(volatile lds_u32*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { *(volatile lds_u32*)(uintptr_t)addr = v; }

__device__ __forceinline__ void barrier_mem() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// bounded wait of one lane for min(*a0, *a1) >= need; returns false (and raises the status word) on timeout / foreign abort
__device__ __forceinline__ bool spin_until(const unsigned* a0, const unsigned* a1, unsigned need, unsigned* status, unsigned& have, unsigned* stat) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  stat[0] += 1;
  for (;;) {
    const unsigned v0 = ld_agent(a0), v1 = ld_agent(a1);
    have = v0 < v1 ? v0 : v1;
    if ((int)(have - need) >= 0) { stat[1] += (unsigned)(__builtin_amdgcn_s_memrealtime() - t0); return true; }
    if (ld_agent(status) != ST_OK) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
      st_agent(status, ST_WAIT_TIMEOUT);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// hidden stage l (n_act-1 >= l >= 1), workgroup j of the pair
// ------------------------------------------------------------------------------------------------------------------
template <bool HI_ONLY, bool TOP>
__device__ __forceinline__ void hidden_stage(const PipeArgs& a, char* smem, int P, int l, int j, int64_t cbeg, int n_my) {
  // EIGHT waves, two per SIMD, with different jobs (one register budget of 256 per wave, two simple loops instead of one that
  // needs 400 registers -- hipcc rotated whole accumulator tiles between the register-file halves in that one):
  //   waves 0..3 ("data"):   dH tile 4 j + w = W_l^T dZ_l (W^T fragments resident in AGPRs), dZ_{l-1} = dH * cos -> ring, db_{l-1};
  //                          ALL of the workgroup's DMA; the decoding of the phase stash (below)
  //   waves 4..7 ("weight"): the 4 x 2 accumulator tiles of dW_l[:, J_j]; the hand-off protocol (wave 4)
  // The two waves of a SIMD share its matrix pipe; while one issues LDS reads, DMA pieces or the epilogue, the other's
  // matrix instructions run.
  //
  // PHASE STASH [r4].  The forward leaves ONE 16-bit phase per pre-activation (sunerf_common.h) instead of an fp16 sin and an fp16
  // cos: per chunk this workgroup takes in dZ_l (16 KiB) and the phases of ITS 128 features of layer l-1 (8 KiB, was 16).  Data wave
  // w fetches the two phase fragments of its own output tile itself, and one iteration before the chunk is consumed it decodes
  // them: sin -> fp16, written back IN PLACE (the LDS image the weight-gradient waves read with ds_read_b64_tr_b16 is the same as
  // before), cos -> 16 registers for its own epilogue of the next iteration.  That work (2 LDS reads, 48 vector instructions
  // 16 of them transcendental, 2 LDS writes) sits where these waves used to wait at the barrier; the weight-gradient waves -- the
  // workgroup's critical path once W^T is a single fp16 image -- issue no DMA at all any more.
  // Per iteration `it`: chunk it + NBUF_H - 1 is requested, chunk it + 1's phases (requested three iterations ago by this very
  // wave: its own counted wait covers them) are decoded, chunk it is consumed.
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_act = a.n_linear - 1, n_links = n_act - 1;
  const StashLayout SL(PD, a.n_linear, true);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned lds_dummy = lds0 + NBUF_H * BUF_HID;
  const unsigned lds_poll = lds_dummy + 1024;
  const unsigned lds_abort = lds_poll + NBUF_H * 256;      // [2]
  unsigned* status = a.ctrl;
  const unsigned voff = lane * 16;

  char* ring_in = TOP ? nullptr : a.rings + ((size_t)P * n_links + l) * RING * SLOT;
  char* ring_out = a.rings + ((size_t)P * n_links + (l - 1)) * RING * SLOT;
  const size_t act_chunk = SL.chunk_bytes();
  const char* srcP0 = a.act_stash + SL.h_off(l - 1) + (size_t)(8 * j) * 1024;
  const int64_t safe = cbeg < a.n_chunks_total ? cbeg : a.n_chunks_total - 1;     // a chunk inside the stashes for surplus DMA

  // Fetch distances.  The phases come from the stash -- nothing gates them -- and are wanted one iteration before their chunk is
  // consumed: requested NBUF_H-1 iterations ahead.  dZ_l comes out of the hand-off ring: every iteration of distance is an
  // iteration of lead the stage above must have, and one more before this stage's own outputs count as published (in-order
  // vmcnt).  ZD = 2 iterations (~6000 clocks, an L2 read takes ~1000) leaves 9 of the ring's 16 slots as slack, 4 gave 6.
  // The top stage reads dZ_top from HBM, ungated: it keeps the long distance.
  constexpr int ZD = TOP ? NBUF_H - 1 : PIPE_ZD;
  // wave-uniform sources / LDS destinations of the pieces being fetched, set at the top of an iteration: dZ of chunk it + ZD,
  // phases of chunk it + NBUF_H - 1 (both into the staging buffer of their chunk, chunk % NBUF_H)
  const char *nz = nullptr, *np = nullptr;
  unsigned zdst = 0, pdst = 0;
  bool zreal = false, preal = false;
  // (called with nxt = 0, 1, 2, ... in turn: the addresses advance by additions -- 64-bit multiplications here cost every wave
  // several hundred cycles per chunk)
  const char* const safe_p = srcP0 + (size_t)safe * act_chunk;
  const char* const safe_z = TOP ? a.dz_top + (size_t)safe * SLOT : ring_in;
  const char* run_p = srcP0 + (size_t)cbeg * act_chunk;
  const char* run_z = TOP ? a.dz_top + (size_t)cbeg * SLOT : ring_in;
  int run_slot = 0, run_zbuf = 0, run_pbuf = 0;
  auto next_p = [&](int nxt) __attribute__((always_inline)) {
    preal = nxt < n_my;
    pdst = lds0 + run_pbuf * BUF_HID + 16 * 1024;
    run_pbuf = run_pbuf + 1 == NBUF_H ? 0 : run_pbuf + 1;
    np = preal ? run_p : safe_p;
    run_p += act_chunk;
  };
  auto next_z = [&](int nxt) __attribute__((always_inline)) {
    if (nxt < 0) { zreal = false; nz = safe_z; return; }      // (prologue iterations ahead of chunk 0: surplus pieces)
    zreal = nxt < n_my;
    zdst = lds0 + run_zbuf * BUF_HID;
    run_zbuf = run_zbuf + 1 == NBUF_H ? 0 : run_zbuf + 1;
    nz = zreal ? run_z : safe_z;
    if (TOP) run_z += SLOT;
    else {
      run_slot = run_slot + 1 == RING ? 0 : run_slot + 1;
      run_z = run_slot == 0 ? ring_in : run_z + SLOT;
    }
  };
  // LDS image of a chunk: pieces 0..15 dZ fragments, 16..23 P[J] (decoded in place to H[J])
  auto piece_z = [&](int f) __attribute__((always_inline)) {
    const unsigned dst = zreal ? zdst + f * 1024 : lds_dummy;
    if (TOP) dma_piece_s<1>(nz + (size_t)f * 1024, voff, dst);
    else dma_piece_s<2>(nz + (size_t)f * 1024, voff, dst);
  };
  auto piece_p = [&](int f) __attribute__((always_inline)) { dma_piece_s<1>(np + (size_t)f * 1024, voff, preal ? pdst + f * 1024 : lds_dummy); };
  // WHO ISSUES THE PIECES (PIPE_DMA_ON_WEIGHT): 0 = the data-gradient waves, between their matrix instructions (4 dZ fragments +
  // the 2 phase fragments of the wave's own tile each); 1 = the phase fragments by the weight-gradient waves (2 each, between
  // THEIR matrix instructions); 2 = everything by the weight-gradient waves (4 + 2 each).  A piece holds the issuing wave for
  // 100 - 200 clocks; behind the barrier the weight-gradient waves have ~1000 clocks to spare per chunk, the data-gradient waves
  // -- the workgroup's long ones since they decode the phases -- none.
  // Counted waits (vmcnt counts in issue order).  Data-gradient waves, per iteration z z z z p p s s (mode 0), z z z z s s (1),
  // s s (2): at the top of iteration `it` dZ of chunk `it` (issued in it - ZD) and the phases of chunk it + 1 (issued in it - 3)
  // must have landed if the wave fetched them itself: at most WAIT_LEFT younger operations outstanding; the stores of iteration
  // it - DRAIN come before all of those: chunk it - DRAIN of this stage's output is complete, which is what wave 4 publishes.
  // Weight-gradient waves, per iteration [flag store, poll (wave 4)] then their pieces: W_WAIT younger operations (below).
  constexpr int NO = 2;
  constexpr int NP_D = PIPE_DMA_ON_WEIGHT == 0 ? 6 : PIPE_DMA_ON_WEIGHT == 1 ? 4 : 0;      // (mode 2: none)      // pieces per data-gradient wave and iteration
  constexpr int NP_W = 6 - NP_D;                                                           // ... per weight-gradient wave
  constexpr int PER_D = NP_D + NO;
  auto min3 = [](int x, int y, int z) constexpr { return x < y ? (x < z ? x : z) : (y < z ? y : z); };
  // z's are the oldest four operations of their iteration, p's the next two, the stores the youngest two
  constexpr int WAIT_LEFT = NP_D == 0 ? NO : min3((ZD - 1) * PER_D + (NP_D - 4) + NO, NP_D == 6 ? 2 * PER_D + NO : 63, 2 * PER_D);
  constexpr int DRAIN = (WAIT_LEFT + PER_D - 1) / PER_D + 1;      // smallest k with WAIT_LEFT <= (k - 1) PER_D
  // weight-gradient wave v, per iteration [flag store, poll: wave 4 only] z z z z p p (mode 2) / p p (mode 1): dZ of chunk `it`
  // was issued in it - ZD, the phases of chunk it + 1 in it - 3, the poll read now in it - PIPE_POLL_LAG
  auto w_wait = [min3](int per_w, bool gate_wave) constexpr {
    return min3(NP_W == 6 ? (ZD - 1) * per_w + 2 : 63, NP_W >= 2 ? 2 * per_w : 63, gate_wave ? PIPE_POLL_LAG * per_w - 2 : 63);
  };
  constexpr int W_WAIT_GATE = w_wait(NP_W + 2, true), W_WAIT_REST = w_wait(NP_W, false);

  static_assert(W_WAIT_GATE >= 0 && W_WAIT_GATE < 64 && WAIT_LEFT < 64, "vmcnt is a 6-bit counter");
  static_assert(ZD >= 1 && ZD <= NBUF_H - 1, "dZ fetch distance");

  if (wave < 4) {
    // =============================================== data-gradient waves ===============================================
#if defined(PIPE_PRIO) && PIPE_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
    // output stores / DMA pieces (dZ fragments w, 4 + w, 8 + w, 12 + w; phase fragments 2 w, 2 w + 1 of this wave's own tile) per iteration
#ifndef PIPE_PF
#define PIPE_PF 4
#endif
    constexpr int PF = PIPE_PF;                      // B fragments requested PF k-steps ahead of their matrix instructions
    const int U = 4 * j + wave;                      // output tile: features 32 U .. 32 U + 31 of layer l-1
    const int li = (a.n_linear - 2) - l;
    half8 wt_hi[PKS], wt_lo[HI_ONLY ? 1 : PKS];
    {
      const char* blk = a.packedT + (size_t)PNT * 1024 + ((size_t)li * PNT + U) * PKS * 2048 + lane * 16;
#pragma unroll
      for (int s = 0; s < PKS; ++s) {
        wt_hi[s] = *(const half8*)(blk + s * 2048);
        if constexpr (!HI_ONLY) wt_lo[s] = *(const half8*)(blk + s * 2048 + 1024);
      }
#pragma unroll
      for (int s = 0; s < PKS; ++s) {
        asm volatile("" : "+a"(wt_hi[s]));
        if constexpr (!HI_ONLY) asm volatile("" : "+a"(wt_lo[s]));
      }
    }
    char* scratch = a.rings + (size_t)a.NP * n_links * RING * SLOT + ((size_t)blockIdx.x * 4 + wave) * 2048;
    // cos of this wave's tile for the chunk consumed NEXT (decoded one iteration ahead), and the decoder: the two phase fragments
    // of the tile (this wave's OWN DMA pieces: its counted wait covers them) -> fp16 sin in place (the H image of the
    // weight-gradient waves, who see it behind the next barrier) + cos here.  It runs behind the epilogue, where these waves used
    // to wait ~900 clocks at the barrier once W^T is a single fp16 image; the LDS reads are issued a few k-steps earlier.
    // Measured alternatives (tools/experiments/README.md, round 4; this one: kernel 12.0 ms): decoding dealt out over the k-steps
    // 12.2 (k-steps 1050 -> 2120 clocks), the sin half on the weight-gradient waves 12.9 (900 clocks there for half the work), all
    // DMA on the weight-gradient waves 13.5 (seven pieces in a row cost them 250 clocks each).
    float cosn[16];
    v2f cosn2[8];      // (PIPE_TRIM) the same as pairs (g, g + 1)
    typedef __attribute__((address_space(3))) v4u lds_v4u;
    typedef __attribute__((address_space(3))) half8 lds_half8;
    v4u dp[2];
    auto read_phases = [&](int b) __attribute__((always_inline)) {
      const unsigned at = lds0 + b * BUF_HID + (16 + 2 * wave) * 1024 + voff;
      dp[0] = *(const lds_v4u*)(uintptr_t)at;
      dp[1] = *(const lds_v4u*)(uintptr_t)(at + 1024);
    };
    auto decode = [&](int b) __attribute__((always_inline)) {
      const unsigned at = lds0 + b * BUF_HID + (16 + 2 * wave) * 1024 + voff;
      if (PIPE_TRIM) {
        const half8 h0 = decode_phases_pairs(dp[0], cosn2), h1 = decode_phases_pairs(dp[1], cosn2 + 4);
        *(lds_half8*)(uintptr_t)at = h0;
        *(lds_half8*)(uintptr_t)(at + 1024) = h1;
        return;
      }
      float sn[16];
      decode_phases(dp[0], sn, cosn);
      decode_phases(dp[1], sn + 8, cosn + 8);
      *(lds_half8*)(uintptr_t)at = to_half8(sn);
      *(lds_half8*)(uintptr_t)(at + 1024) = to_half8(sn + 8);
    };
    // prologue iterations -(NBUF_H-1) .. -1: DMA only (+ two dummy stores: the same vmcnt accounting as a full iteration); the
    // last of them decodes chunk 0
    bool stop = false;
    for (int it = -(NBUF_H - 1); it < 0; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "i"(WAIT_LEFT) : "memory");
      barrier_mem();
      const unsigned ab = lds_ld(lds_abort + (it & 1) * 4);
      if (NP_D >= 4) next_z(it + ZD);
      if (NP_D == 6) next_p(it + NBUF_H - 1);
      const Rsrc sc = make_rsrc(scratch, 2048);
      const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      if (NP_D >= 4) { piece_z(wave); piece_z(4 + wave); piece_z(8 + wave); piece_z(12 + wave); }
      if (NP_D == 6) { piece_p(2 * wave); piece_p(2 * wave + 1); }
      buf_store(zero, sc, 0);
      buf_store(zero, sc, 1024);
      if (ab) { stop = true; break; }
      if (it == -1 && n_my > 0) { read_phases(0); decode(0); }
    }
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};      // SUNERF_PIPE_DEBUG: shader clocks in wait / barrier / k-steps / epilogue + decode; [4] products + conversion, [5] .. + stores
    unsigned long long tl[5] = {0, 0, 0, 0, 0};   //   and the stamps of iteration n_my / 2 themselves
    const bool stamp = a.dbg != nullptr;
    char* out_z = ring_out;                        // ring slot of this iteration's output
    int out_slot = 0, buf = 0;                     // buf = it % NBUF_H
    float bs[16];                                  // this lane's share of db_{l-1}[32 U ..]: fp32 sums of dH * cos
    v2f bs2[8];
#pragma unroll
    for (int g = 0; g < 16; ++g) bs[g] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) bs2[q] = (v2f){0.f, 0.f};
    if (PIPE_FP16_OVFL) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
    for (int it = 0; it < (stop ? 0 : n_my); ++it) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      if (stamp) s0 = __builtin_amdgcn_s_memtime();
      // dZ of chunk `it` and this wave's phase pieces of chunk it + 1 have landed, the stores of chunk it - DRAIN are complete
      asm volatile("s_waitcnt vmcnt(%0)" :: "i"(WAIT_LEFT) : "memory");
      const int nbuf = buf + 1 == NBUF_H ? 0 : buf + 1;
      const bool decode_next = it + 1 < n_my;
      if (stamp) s1 = __builtin_amdgcn_s_memtime();
      barrier_mem();
      if (stamp) s2 = __builtin_amdgcn_s_memtime();
      // abort word of this iteration: requested now, looked at when the iteration's work is done (every wave leaves in the same
      // iteration, so the barrier counts still agree; what a doomed iteration computes and stores is garbage either way)
      const unsigned ab = lds_ld(lds_abort + (it & 1) * 4);
      if (NP_D >= 4) next_z(it + ZD);
      if (NP_D == 6) next_p(it + NBUF_H - 1);
      const char* B = smem + (size_t)buf * BUF_HID;
      f32x16 dacc = {0};
      half8 bf[PF + 1];
      float cosc[16];                              // cos of the chunk consumed now (decoded in the previous iteration)
      v2f cosc2[8];
#pragma unroll
      for (int g = 0; g < 16; ++g) cosc[g] = cosn[g];
#pragma unroll
      for (int q = 0; q < 8; ++q) cosc2[q] = cosn2[q];
#pragma unroll
      for (int s = 0; s < PF; ++s) bf[s] = *(const half8*)(B + s * 1024 + lane * 16);
#pragma unroll
      for (int ks = 0; ks < PKS; ++ks) {
        if (ks + PF < PKS) bf[(ks + PF) % (PF + 1)] = *(const half8*)(B + (ks + PF) * 1024 + lane * 16);
        if constexpr (!HI_ONLY) dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_lo[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_hi[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        if (NP_D >= 4) {
          if (ks == 0) piece_z(wave);
          if (ks == 2) piece_z(4 + wave);
          if (ks == 4) piece_z(8 + wave);
          if (ks == 6) piece_z(12 + wave);
        }
        if (NP_D == 6) {
          if (ks == 8) piece_p(2 * wave);
          if (ks == 10) piece_p(2 * wave + 1);
        }
        if (ks == 12 && decode_next) read_phases(nbuf);      // the next chunk's phases, decoded behind the epilogue
        __builtin_amdgcn_sched_barrier(0);
      }
      if (stamp) { asm volatile("" :: "v"(dacc)); s3 = __builtin_amdgcn_s_memtime(); }
      // dZ_{l-1} = dH * cos (fp16, saturating) -> ring slot, fragments 2 U, 2 U + 1 of the chunk
      half8 d0, d1;
      const Rsrc ro = make_rsrc(out_z, SLOT);
      if (PIPE_TRIM) dz_tile_pairs(dacc, cosc2, d0, d1, bs2);
      else dz_tile(dacc, cosc, cosc + 8, d0, d1, bs);
      unsigned long long s3a = 0, s3b = 0;
      if (stamp) { asm volatile("" :: "v"(d0), "v"(d1)); s3a = __builtin_amdgcn_s_memtime(); }
      buf_store(d0, ro, (2 * U) * 1024);
      buf_store(d1, ro, (2 * U + 1) * 1024);
      if (stamp) s3b = __builtin_amdgcn_s_memtime();
      out_slot = out_slot + 1 == RING ? 0 : out_slot + 1;
      out_z = out_slot == 0 ? ring_out : out_z + SLOT;
      buf = nbuf;
      if (decode_next) decode(buf);
      if (stamp) {
        const unsigned long long s4 = __builtin_amdgcn_s_memtime();
        ph[0] += s1 - s0; ph[1] += s2 - s1; ph[2] += s3 - s2; ph[3] += s4 - s3; ph[4] += s3a - s3; ph[5] += s3b - s3;
        if (it == n_my / 2) { tl[0] = s0; tl[1] = s1; tl[2] = s2; tl[3] = s3; tl[4] = s4; }
      }
      if (ab) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_mem();
    // db_{l-1}, row tile U, of this pipeline: the bias column of layer l-1's partial-sum slot (every slot is written exactly once:
    // a workgroup without chunks writes its zeros)
    if (PIPE_TRIM) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { bs[2 * q] = bs2[q].x; bs[2 * q + 1] = bs2[q].y; }
    }
    store_bias_sums(bs, a.partial + ((size_t)(l - 1) * a.NP + P) * (PT * (PT + 1)) * 1024 + ((size_t)U * (PT + 1) + PT) * 1024);
    if (stamp && P == 0 && lane == 0) {      // timeline of iteration n_my / 2: [workgroup of the pipeline][wave][8] 64-bit shader clocks
      unsigned long long* t = (unsigned long long*)(a.dbg + 256 * 8 * 4) + (((size_t)(blockIdx.x >> 3) * 8 + wave) * 8);
      for (int k = 0; k < 5; ++k) t[k] = tl[k];
    }
    if (stamp && wave == 1 && lane == 0) {
      unsigned* d = a.dbg + 256 * 8 + (size_t)blockIdx.x * 8;
      for (int k = 0; k < 6; ++k) d[k] = (unsigned)(ph[k] >> 4);
    }
    return;
  }

  // ================================================= weight-gradient waves =================================================
#if defined(PIPE_PRIO) && PIPE_PRIO == 2
  __builtin_amdgcn_s_setprio(1);
#endif
  const int v = wave - 4;
  const bool gatew = v == 0;      // the protocol wave: publishes this workgroup's counters, polls the partners', gates the next iteration
  // block of this wave: row tiles 4 rq .. +3 (features of dZ_l), column tiles 2 cq, 2 cq + 1 of J_j (db_l is summed where dZ_l is
  // formed: by the data-gradient waves of the stage above, or by the prologue)
  const int rq = v >> 1, cq = v & 1;
  const int r0 = 4 * rq, c0 = 2 * cq;
  f32x16 acc[4][2];
  {
    // zero tiles DEFINED in AGPRs (0 * 0 + 0 by the matrix pipe): a `= {0}` gives the loop-carried tiles a VGPR home and hipcc
    // then copies 16 registers into AGPRs in front of every matrix instruction and back behind it
    half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_nop 7" : "+v"(hz));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[i][0]) : "v"(hz));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[i][1]) : "v"(hz));
    }
  }
  // counters: input link l (producers 0, 1; my consumer count), output link l-1 (my producer count; consumers 0, 1 -- only
  // consumer j on link 0, whose in-layer stage reads its own half)
  unsigned* in_prod0 = a.ctrl + ctrl_counter_word(n_links, P, TOP ? 0 : l, 0);
  unsigned* in_prod1 = a.ctrl + ctrl_counter_word(n_links, P, TOP ? 0 : l, 1);
  unsigned* my_cons = a.ctrl + ctrl_counter_word(n_links, P, TOP ? 0 : l, 2 + j);
  unsigned* my_prod = a.ctrl + ctrl_counter_word(n_links, P, l - 1, j);
  unsigned* out_cons0 = a.ctrl + ctrl_counter_word(n_links, P, l - 1, 2 + (l == 1 ? j : 0));
  unsigned* out_cons1 = a.ctrl + ctrl_counter_word(n_links, P, l - 1, 2 + (l == 1 ? j : 1));
  const unsigned* poll_src = (lane & 3) == 0 ? in_prod0 : (lane & 3) == 1 ? in_prod1 : (lane & 3) == 2 ? out_cons0 : out_cons1;
  unsigned have_in = 0, have_out = 0;      // wave 4: newest known min(prod) of the input link / min(cons) of the output link
  unsigned st_in[2] = {0, 0}, st_out[2] = {0, 0};       // (lane 0 of wave 4) fallback spins and the 100 MHz ticks they took
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  const unsigned laneoff = tr_lane_offset(lane);

  // gate of iteration `itn` (wave 4): its DMA (chunk itn + NBUF_H - 1) needs that chunk published by both producers, its output
  // (slot itn % RING) needs chunk itn - RING landed in both consumers.  Normally the polled counters already say so.
  auto gate = [&](int itn) __attribute__((always_inline)) {
    const int nx = itn + ZD;      // (the chunk whose dZ the data waves request in iteration itn)
    const bool need_in = !TOP && nx >= 0 && nx < n_my && (int)(have_in - (unsigned)(nx + 1)) < 0;
    const bool need_out = itn >= RING && (int)(have_out - (unsigned)(itn - RING + 1)) < 0;
    if (need_in || need_out) {
      bool ok = true;
      if (lane == 0) {
        if (need_in) ok = spin_until(in_prod0, in_prod1, (unsigned)(nx + 1), status, have_in, st_in);
        if (ok && need_out) ok = spin_until(out_cons0, out_cons1, (unsigned)(itn - RING + 1), status, have_out, st_out);
        if (!ok) lds_st(lds_abort + (itn & 1) * 4, 1u);
      }
      have_in = __builtin_amdgcn_readfirstlane(have_in);
      have_out = __builtin_amdgcn_readfirstlane(have_out);
    }
  };
  if (gatew) {
    if (lane == 0) { lds_st(lds_abort, 0u); lds_st(lds_abort + 4, 0u); }      // the abort words are only ever SET (by a failed gate)
    gate(-(NBUF_H - 1));
  }

  bool aborted = false;
  // These waves issue no DMA (round 4: every piece of the workgroup goes through the data-gradient waves); wave 4 has the two
  // vector-memory operations of the protocol per iteration (flag store, poll).  The two waves of a SIMD share its matrix pipe:
  // the data wave starts its matrix instructions right behind the barrier, this wave's 16 follow its operand reads.
  // top(): counted wait (wave 4), barrier, abort word, wave 4's publication and poll.
  unsigned long long tw = 0, tb = 0;       // SUNERF_PIPE_DEBUG: shader clocks in the counted wait / the barrier
  unsigned long long t_wait = 0, t_bar = 0, tl[6] = {0, 0, 0, 0, 0, 0};
  const bool stamp = a.dbg != nullptr;
  int pslot = PIPE_POLL_LAG;               // the poll issued in iteration `it` lands in LDS slot (it + PIPE_POLL_LAG) % NBUF_H, read in iteration it + PIPE_POLL_LAG
  auto top = [&](int it) __attribute__((always_inline)) {
    unsigned long long sa = 0, sb = 0;
    if (stamp) sa = __builtin_amdgcn_s_memtime();
    // (the poll issued PIPE_POLL_LAG iterations ago and this wave's pieces of chunk `it` / phases of chunk it + 1 have landed)
    if (gatew) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(W_WAIT_GATE) : "memory");
    else if (NP_W) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(W_WAIT_REST) : "memory");
    if (stamp) sb = __builtin_amdgcn_s_memtime();
    barrier_mem();
    if (stamp) { t_wait = sb; t_bar = __builtin_amdgcn_s_memtime(); tw += sb - sa; tb += t_bar - sb; }
    const unsigned ab = lds_ld(lds_abort + (it & 1) * 4);     // looked at by the caller when the iteration's work is done
    if (gatew) {
      // publish: chunk `it` has landed in this workgroup (its ring slot may be overwritten); the outputs of chunk
      // it - DRAIN are in L2 (the data waves' counted waits in front of the barrier).  One store instruction, lanes 0 and 1.
      if (lane < 2) {
        const int pv = lane == 0 ? it + 1 : it - DRAIN + 1;
        st_agent(lane == 0 ? my_cons : my_prod, (unsigned)(pv > 0 ? pv : 0));
      }
      // ... then the poll whose value is read PIPE_POLL_LAG iterations on
      pslot = pslot + 1 == NBUF_H ? 0 : pslot + 1;
      dma_poll(poll_src, lds_poll + pslot * 256);
    }
    return ab;
  };
  // this wave's pieces of an iteration (modes 1, 2): dZ fragments v, 4 + v, 8 + v, 12 + v of chunk it + ZD, phase fragments 2 v, 2 v + 1
  // of chunk it + NBUF_H - 1; `q` = 0 .. 5 in the order z z z z p p
  auto w_piece = [&](int q) __attribute__((always_inline)) {
    if (q < 4) { if (NP_W == 6) piece_z(4 * q + v); }
    else if (NP_W >= 2) piece_p(2 * v + q - 4);
  };
  for (int it = -(NBUF_H - 1); it < 0; ++it) {      // prologue iterations: protocol (+ this wave's pieces of the first chunks)
    const unsigned ab = top(it);
    if (NP_W == 6) next_z(it + ZD);
    if (NP_W >= 2) next_p(it + NBUF_H - 1);
#pragma unroll
    for (int q = 0; q < 6; ++q) w_piece(q);
    if (ab) { aborted = true; break; }
    if (gatew && it + 1 < n_my) gate(it + 1);
  }
  unsigned long long ph[4] = {0, 0, 0, 0};      // SUNERF_PIPE_DEBUG: shader clocks in top (wait + barrier + publication) / operand reads + gate / matrix
  tw = 0; tb = 0;
  int buf = 0;                                   // it % NBUF_H
  for (int it = 0; it < (aborted ? 0 : n_my); ++it) {
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (stamp) s0 = __builtin_amdgcn_s_memtime();
    const unsigned ab = top(it);
    if (NP_W == 6) next_z(it + ZD);
    if (NP_W >= 2) next_p(it + NBUF_H - 1);
    if (stamp) s1 = __builtin_amdgcn_s_memtime();
    // operand bases of this wave's block: row tiles r0.. of the dZ fragments, column tiles c0.. of the H fragments (fp16 sin,
    // decoded in place by the data-gradient waves one iteration ago); the tile and k-step parts of the addresses are immediates
    const unsigned bufA = lds0 + buf * BUF_HID + laneoff + r0 * 2048, bufB = lds0 + buf * BUF_HID + laneoff + 16 * 1024 + c0 * 2048;
    // operands of the two k-steps: A = dZ^T row tiles r0 .. r0 + 3, B = H column tiles c0, c0 + 1; registers v80 .. v127
    half8 A0[4], B0[2], A1[4], B1[2];
    TR_FIXED(80, 81, 82, 83, 0 * 2048, A0[0], bufA); TR_FIXED(84, 85, 86, 87, 1 * 2048, A0[1], bufA);
    TR_FIXED(88, 89, 90, 91, 2 * 2048, A0[2], bufA); TR_FIXED(92, 93, 94, 95, 3 * 2048, A0[3], bufA);
    TR_FIXED(96, 97, 98, 99, 0 * 2048, B0[0], bufB); TR_FIXED(100, 101, 102, 103, 1 * 2048, B0[1], bufB);
    typedef __attribute__((address_space(3))) v4u lds_v4u;
    v4u pw = {0, 0, 0, 0};
    if (gatew) pw = *(const lds_v4u*)(uintptr_t)(lds_poll + buf * 256);     // the poll issued NBUF_H-1 iterations ago
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (gatew) {     // -> gate of the next iteration (in the shadow of the operand reads' wait)
      asm volatile("" : "+v"(pw));
      const unsigned pi = pw[0] < pw[1] ? pw[0] : pw[1], qo = pw[2] < pw[3] ? pw[2] : pw[3];
      if ((int)(pi - have_in) > 0) have_in = pi;
      if ((int)(qo - have_out) > 0) have_out = qo;
      have_in = __builtin_amdgcn_readfirstlane(have_in);
      have_out = __builtin_amdgcn_readfirstlane(have_out);
      if (it + 1 < n_my) gate(it + 1);
    }
    if (stamp) s2 = __builtin_amdgcn_s_memtime();
    asm volatile("" : "+v"(A0[0]), "+v"(A0[1]), "+v"(A0[2]), "+v"(A0[3]), "+v"(B0[0]), "+v"(B0[1]));
    TR_FIXED(104, 105, 106, 107, 0 * 2048 + 256, A1[0], bufA); TR_FIXED(108, 109, 110, 111, 1 * 2048 + 256, A1[1], bufA);
    TR_FIXED(112, 113, 114, 115, 2 * 2048 + 256, A1[2], bufA); TR_FIXED(116, 117, 118, 119, 3 * 2048 + 256, A1[3], bufA);
    TR_FIXED(120, 121, 122, 123, 0 * 2048 + 256, B1[0], bufB); TR_FIXED(124, 125, 126, 127, 1 * 2048 + 256, B1[1], bufB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(A1[0]), "+v"(A1[1]), "+v"(A1[2]), "+v"(A1[3]), "+v"(B1[0]), "+v"(B1[1]));
      }
      // (the asm matrix instructions get no hazard handling from hipcc: with the earlier v_mov joins an operand written
      // immediately in front of the instruction that reads it was read stale -- the first tile of every k-step came out wrong.
      // Now the operands are written by the LDS reads and covered by the lgkmcnt wait; two idle cycles stay as a margin.)
      const half8 bf0 = ks ? B1[0] : B0[0], bf1 = ks ? B1[1] : B0[1];
      asm volatile("s_nop 1" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const half8 af = ks ? A1[i] : A0[i];
        mfma_agpr(acc[i][0], af, bf0);
        mfma_agpr(acc[i][1], af, bf1);
        if (4 * ks + i < NP_W) w_piece(6 - NP_W + 4 * ks + i);      // one piece behind each of the first pairs of matrix instructions
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf = buf + 1 == NBUF_H ? 0 : buf + 1;
    if (stamp) {
      const unsigned long long s3 = __builtin_amdgcn_s_memtime();
      ph[0] += s1 - s0; ph[1] += s2 - s1; ph[2] += s3 - s2;
      if (it == n_my / 2) { tl[0] = s0; tl[1] = t_wait; tl[2] = t_bar; tl[3] = s1; tl[4] = s2; tl[5] = s3; }
    }
    if (ab) { aborted = true; break; }
  }
  drain_matrix_pipe();
  if (stamp && (v == 0 || v == 1) && lane == 0) {
    unsigned* d = a.dbg + 256 * 8 * (2 + v) + (size_t)blockIdx.x * 8;
    for (int k = 0; k < 3; ++k) d[k] = (unsigned)(ph[k] >> 4);
    d[3] = (unsigned)(tw >> 4); d[4] = (unsigned)(tb >> 4);
  }
  if (stamp && P == 0 && lane == 0) {
    unsigned long long* t = (unsigned long long*)(a.dbg + 256 * 8 * 4) + (((size_t)(blockIdx.x >> 3) * 8 + wave) * 8);
    for (int k = 0; k < 6; ++k) t[k] = tl[k];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier_mem();
  if (!aborted && gatew && lane < 2) st_agent(lane == 0 ? my_cons : my_prod, (unsigned)n_my);
  if (a.dbg && gatew && lane == 0) {
    unsigned* d = a.dbg + (size_t)blockIdx.x * 8;
    d[0] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t_begin); d[1] = st_in[0]; d[2] = st_in[1]; d[3] = st_out[0]; d[4] = st_out[1];
    d[5] = (unsigned)n_my; d[6] = (unsigned)l; d[7] = (unsigned)P;
  }

  // ---- partial sums -> workspace [layer l][pipeline P][tr][tc 9][reg][lane] ----
  float* out = a.partial + ((size_t)l * a.NP + P) * (PT * (PT + 1)) * 1024;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      store_tile(acc[i][jj], out + ((size_t)(r0 + i) * (PT + 1) + (4 * j + c0 + jj)) * 1024);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// in-layer stage: dW_0[J_j rows][84] and db_0[J_j] from dZ_0[J_j] (link 0, producer j) and the encoding stash
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void in_stage(const PipeArgs& a, char* smem, int P, int j, int64_t cbeg, int n_my) {
  constexpr int PW = 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= 4) {      // the workgroup has eight waves for the hidden stages' sake; this light stage uses four
    for (int it = -(NBUF - 1); it < n_my; ++it) {
      barrier_mem();
      if (lds_ld((unsigned)(uintptr_t)smem + NBUF * BUF_IN + 1024 + NBUF * 256 + (it & 1) * 4)) break;
    }
    barrier_mem();
    return;
  }
  const int n_act = a.n_linear - 1, n_links = n_act - 1;
  const StashLayout SL(PD, a.n_linear, true);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned lds_dummy = lds0 + NBUF * BUF_IN;
  const unsigned lds_poll = lds_dummy + 1024;
  const unsigned lds_abort = lds_poll + NBUF * 256;
  unsigned* status = a.ctrl;

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = (f32x16){0};

  const char* ring_in = a.rings + ((size_t)P * n_links + 0) * RING * SLOT;
  unsigned* in_prod = a.ctrl + ctrl_counter_word(n_links, P, 0, j);
  unsigned* my_cons = a.ctrl + ctrl_counter_word(n_links, P, 0, 2 + j);
  unsigned have_in = 0;
  unsigned st_in[2] = {0, 0};
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  const size_t act_chunk = SL.chunk_bytes();
  const unsigned laneoff = tr_lane_offset(lane);
  const int64_t safe = cbeg < a.n_chunks_total ? cbeg : a.n_chunks_total - 1;

  // 14 pieces: dZ_0 fragments 8 j + p (p < 8), encoding fragments p - 8 (8 <= p < 14); waves 2, 3 issue one surplus piece
  auto issue_chunk = [&](int c, int buf) __attribute__((always_inline)) {
    const bool real = c < n_my;
    const int64_t g = real ? cbeg + c : safe;
    const unsigned dst0 = lds0 + buf * BUF_IN;
    const char* z = ring_in + (size_t)((real ? c : 0) % RING) * SLOT + (size_t)(8 * j) * 1024 + lane * 16;
    const char* e = a.act_stash + (size_t)g * act_chunk + lane * 16;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int p = 4 * k + wave;
      if (k < 2) dma_piece<2>(z + (size_t)p * 1024, real ? dst0 + p * 1024 : lds_dummy);
      else {
        const bool use = real && p < 14;
        dma_piece<1>(e + (size_t)(p < 14 ? p - 8 : 0) * 1024, use ? dst0 + p * 1024 : lds_dummy);
      }
    }
  };

  bool aborted = false;
  for (int it = -(NBUF - 1); it < n_my; ++it) {
    const int buf = it & (NBUF - 1);
    const int nxt = it + NBUF - 1;
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (PW + 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * PW) : "memory");
    if (wave == 0) {
      bool ok = true;
      if (it >= 0) {
        const unsigned p0 = lds_ld(lds_poll + buf * 256);
        if ((int)(p0 - have_in) > 0) have_in = p0;
      }
      if (lane == 0) {
        if (nxt < n_my && (int)(have_in - (unsigned)(nxt + 1)) < 0) ok = spin_until(in_prod, in_prod, (unsigned)(nxt + 1), status, have_in, st_in);
        lds_st(lds_abort + (it & 1) * 4, ok ? 0u : 1u);
      }
      have_in = __builtin_amdgcn_readfirstlane(have_in);
    }
    barrier_mem();
    if (lds_ld(lds_abort + (it & 1) * 4)) { aborted = true; break; }
    if (wave == 0) {
      if (lane == 0) st_agent(my_cons, (unsigned)(it + 1 > 0 ? it + 1 : 0));
      dma_poll(in_prod, lds_poll + (nxt & (NBUF - 1)) * 256);
    }
    issue_chunk(nxt, nxt & (NBUF - 1));
    if (it < 0) continue;
#if defined(PIPE_ABL_IN_STAGE) && PIPE_ABL_IN_STAGE
    continue;       // timing-only knock-out (round 4): the in-layer stage keeps its protocol and its DMA, computes nothing
#endif
    const unsigned bufA = lds0 + buf * BUF_IN + laneoff, bufB = bufA + 8 * 1024;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half4 alo, ahi, blo[3], bhi[3];
      tr_issue(bufA + (2 * wave) * 1024 + ks * 256, alo, ahi);
#pragma unroll
      for (int c = 0; c < 3; ++c) tr_issue(bufB + (2 * c) * 1024 + ks * 256, blo[c], bhi[c]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(alo), "+v"(ahi));
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(blo[c]), "+v"(bhi[c]));
      const half8 af = join(alo, ahi);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, join(blo[c], bhi[c]), acc[c], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier_mem();
  if (!aborted && wave == 0 && lane == 0) st_agent(my_cons, (unsigned)n_my);
  if (a.dbg && tid == 0) {
    unsigned* d = a.dbg + (size_t)blockIdx.x * 8;
    d[0] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t_begin); d[1] = st_in[0]; d[2] = st_in[1]; d[3] = 0; d[4] = 0;
    d[5] = (unsigned)n_my; d[6] = 0; d[7] = (unsigned)P;
  }

  float* out = a.partial + ((size_t)0 * a.NP + P) * (PT * (PT + 1)) * 1024;
  const int tr = 4 * j + wave;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    store_tile(acc[c], out + ((size_t)tr * (PT + 1) + c) * 1024);
  }
  // (db_0 is summed by the data-gradient waves of stage 1, which form dZ_0)
}

template <bool HI_ONLY>
__global__ __launch_bounds__(WG, 1) void bwd_pipe_kernel(PipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int n_act = a.n_linear - 1;
  const int per_pipe = 2 * n_act;
  const int b = blockIdx.x, x = b & 7, i = b >> 3;
  unsigned* status = a.ctrl;

  // ---- start-up: everybody resident, every class (blockIdx % 8) on one XCD ----
  // (no static __shared__: it would shift the dynamic region off its 16-byte alignment)
  const unsigned start_ok = (unsigned)(uintptr_t)smem + NBUF_H * BUF_HID + 1024 + NBUF_H * 256 + 32;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    const unsigned mine = (1u << xcc) | ((a.inject && x == 0) ? 1u << ((xcc + 1) & 7) : 0u);
    const unsigned old = __hip_atomic_fetch_or((gu32*)(a.ctrl + 8 + x), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)old;
    __hip_atomic_fetch_add((gu32*)(a.ctrl + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (ld_agent(a.ctrl + 1) < gridDim.x) {
      if (ld_agent(status) != ST_OK) { ok = false; break; }
      if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) { st_agent(status, ST_START_TIMEOUT); ok = false; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (ok) {
      const unsigned m = ld_agent(a.ctrl + 8 + x);
      if (__builtin_popcount(m) != 1) { st_agent(status, ST_PLACEMENT); ok = false; }
    }
    lds_st(start_ok, ok ? 1u : 0u);
  }
  __syncthreads();
  const bool ok = lds_ld(start_ok) != 0;

  const int q = i / per_pipe;
  if (q >= a.NPX) return;                       // workgroups beyond the last whole pipeline of their class
  const int r = i % per_pipe, si = r >> 1, j = r & 1;
  const int l = n_act - 1 - si;
  const int P = q * 8 + x;
  const int64_t per = (a.n_chunks_total + a.NP - 1) / a.NP;
  const int64_t cbeg = (int64_t)P * per;
  int64_t cend = cbeg + per;
  if (cend > a.n_chunks_total) cend = a.n_chunks_total;
  // a launch that failed its start-up check does no work: zero partial sums, NaN gradients via the status word
  const int n_my = (ok && cend > cbeg) ? (int)(cend - cbeg) : 0;
  if (l == 0) in_stage(a, smem, P, j, cbeg, n_my);
  else if (l == n_act - 1) hidden_stage<HI_ONLY, true>(a, smem, P, l, j, cbeg, n_my);
  else hidden_stage<HI_ONLY, false>(a, smem, P, l, j, cbeg, n_my);
}

// ------------------------------------------------------------------------------------------------------------------
// prologue: dZ of the last activation layer (-> dz_top), its db, and dW / db of the out layer: one streaming pass over g_raw and
// the PHASE fragments of the last activation layer (16 KiB per chunk in, 16 KiB out).  Wave w owns tiles 2 w, 2 w + 1 of that
// layer: it fetches their four phase fragments itself and decodes them -- cos -> registers for dZ = (W_out^T g) cos, sin -> fp16
// written back in place, which is the LDS image its own transposed reads for dW_out take.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_PRE, 1) void bwd_prologue_kernel(PipeArgs a) {
  constexpr int PW = 4, NO = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int n_act = a.n_linear - 1;
  const StashLayout SL(PD, a.n_linear, true);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned lds_dummy = lds0 + NBUF * BUF_PRE;
  const float gscale = sunerf_gscale(*a.g_absmax_bits);

  const int64_t per = (a.n_chunks_total + gridDim.x - 1) / gridDim.x;
  const int64_t cbeg = (int64_t)blockIdx.x * per;
  int64_t cend = cbeg + per;
  if (cend > a.n_chunks_total) cend = a.n_chunks_total;
  const int n_my = cend > cbeg ? (int)(cend - cbeg) : 0;
  const int64_t safe = cbeg < a.n_chunks_total ? cbeg : a.n_chunks_total - 1;

  // W_out^T A fragments of this wave's two tiles (one k-step: rows = features 32 U + m, k = output index)
  half8 aT[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) aT[t] = *(const half8*)(a.packedT + (size_t)(2 * wave + t) * 1024 + lane * 16);

  f32x16 acc[2] = {(f32x16){0}, (f32x16){0}};
  float bs[2][16];          // fp32 bias sums of the last activation layer, tiles 2 w and 2 w + 1 (this lane's samples)
#pragma unroll
  for (int g = 0; g < 16; ++g) { bs[0][g] = 0.f; bs[1][g] = 0.f; }
  float gsum0 = 0.f, gsum1 = 0.f;       // wave 0: db of the out layer = sum of the (scaled) fp32 g_raw itself
  const size_t act_chunk = SL.chunk_bytes();
  const char* srcP0 = a.act_stash + SL.h_off(n_act - 1) + (size_t)(4 * wave) * 1024 + lane * 16;     // this wave's four fragments
  const unsigned laneoff = tr_lane_offset(lane);
  char* scratch = a.rings + (size_t)a.NP * (n_act - 1) * RING * SLOT + ((size_t)blockIdx.x * 4 + wave) * 2048;
  typedef __attribute__((address_space(3))) v4u lds_v4u;
  typedef __attribute__((address_space(3))) half8 lds_half8;

  // LDS image of a chunk: [16 phase fragments -> fp16 sin][dZ_out operand tile, 2 fragments][g_raw, 256 B]
  auto issue_chunk = [&](int c, int buf) __attribute__((always_inline)) {
    const bool real = c < n_my;
    const int64_t g = real ? cbeg + c : safe;
    const unsigned dst0 = lds0 + buf * BUF_PRE;
    if (wave == 0) {
      // g_raw of the chunk's 32 samples (2 floats each, contiguous inside a ray): lane t <- float t.  Samples past the end of
      // the ray are not loaded (the first sample of a chunk always exists, so the instruction is always issued: the counted
      // vmcnt waits rely on that).  A compiler-visible load here would drain the whole DMA queue with vmcnt(0) every chunk.
      const int64_t ray = g / a.n_chunks;
      const int c32 = (int)(g % a.n_chunks) * 32;
      if (c32 + (lane >> 1) < a.S) dma_poll((const unsigned*)(a.g_raw + ((size_t)ray * a.S + c32) * 2 + lane), real ? dst0 + 18 * 1024 : lds_dummy);
    }
#pragma unroll
    for (int k = 0; k < PW; ++k)
      dma_piece<1>(srcP0 + (size_t)g * act_chunk + (size_t)k * 1024, real ? dst0 + (4 * wave + k) * 1024 : lds_dummy);
  };

  for (int it = -(NBUF - 1); it < n_my; ++it) {
    const int buf = it & (NBUF - 1);
    const int nxt = it + NBUF - 1;
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NO + 2 * (PW + NO + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NO + 2 * (PW + NO)) : "memory");
    half8 dzo = {0, 0, 0, 0, 0, 0, 0, 0};
    float cs[2][16];
    if (it >= 0) {
      // this wave's own phase pieces of the chunk have landed (counted wait above): decode -- cos stays here, fp16 sin goes back
      // to the same 16 bytes
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const unsigned at = lds0 + buf * BUF_PRE + (4 * wave + f) * 1024 + lane * 16;
        const v4u p = *(const lds_v4u*)(uintptr_t)at;
        float sn[8];
        decode_phases(p, sn, cs[f >> 1] + 8 * (f & 1));
        *(lds_half8*)(uintptr_t)at = to_half8(sn);
      }
      // g_raw of this lane's sample -> B fragment of the out layer's dZ (features 0 / 1 = d loss / d raw[..., 0 / 1]) and the
      // A-operand tile of the out layer's weight gradient (fragment-order index 16 s + 8 h + e -> s = 0, h = 0, e = 0 / 1)
      const int64_t chunk = cbeg + it;
      const int c = (int)(chunk % a.n_chunks);
      const int i = 32 * c + n;
      if (h == 0 && i < a.S) {
        const f32x2 g = *(const f32x2*)(smem + (size_t)buf * BUF_PRE + 18 * 1024 + n * 8);
        const float g0 = g[0] * gscale, g1 = a.d_out > 1 ? g[1] * gscale : 0.f;
        dzo[0] = (_Float16)g0;
        dzo[1] = (_Float16)g1;
        gsum0 += g0;
        gsum1 += g1;
      }
      if (wave == 0) *(half8*)(smem + (size_t)buf * BUF_PRE + 16 * 1024 + lane * 16) = dzo;
      else if (wave == 1) *(half8*)(smem + (size_t)buf * BUF_PRE + 17 * 1024 + lane * 16) = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    barrier_mem();
    issue_chunk(nxt, nxt & (NBUF - 1));
    if (it < 0) {
      const Rsrc sc = make_rsrc(scratch, 2048);
      const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NO; ++k) buf_store(zero, sc, (k & 1) * 1024);
      continue;
    }
    // ---- dZ_top tiles 2 w, 2 w + 1: (W_out^T g) * cos ----
    const Rsrc ro = make_rsrc(a.dz_top + (size_t)(cbeg + it) * SLOT, SLOT);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int U = 2 * wave + t;
      f32x16 d = {0};
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(aT[t], dzo, d, 0, 0, 0);
      half8 d0, d1;
      dz_tile(d, cs[t], cs[t] + 8, d0, d1, bs[t]);
      buf_store(d0, ro, (2 * U) * 1024);
      buf_store(d1, ro, (2 * U + 1) * 1024);
    }
    // ---- out layer weight gradient: row tile 0 (outputs 0 / 1), column tiles 2 w, 2 w + 1 of H_top (decoded above) ----
    const unsigned bufH = lds0 + buf * BUF_PRE + laneoff, bufA = bufH + 16 * 1024;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half4 alo, ahi, blo[2], bhi[2];
      tr_issue(bufA + ks * 256, alo, ahi);
#pragma unroll
      for (int t = 0; t < 2; ++t) tr_issue(bufH + (2 * (2 * wave + t)) * 1024 + ks * 256, blo[t], bhi[t]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(alo), "+v"(ahi));
#pragma unroll
      for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(blo[t]), "+v"(bhi[t]));
      const half8 af = join(alo, ahi);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, join(blo[t], bhi[t]), acc[t], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // per workgroup: [8 tiles dW_out][tile PT: db_out, 32 floats][tile PT + 1: db of the last activation layer, 8 row tiles x 32]
  float* out = a.partial_out + (size_t)blockIdx.x * (PT + 2) * 1024;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    store_tile(acc[t], out + (size_t)(2 * wave + t) * 1024);
    store_bias_sums(bs[t], out + (size_t)(PT + 1) * 1024 + (2 * wave + t) * 32);
  }
  if (wave == 0) {
    // g_raw sits on the lanes of half 0 (one sample each); bias slot lane i = output index i (fragment-order index i: s = 0,
    // h = 0, e = i for i < 8)
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { gsum0 += __shfl_xor(gsum0, d, 32); gsum1 += __shfl_xor(gsum1, d, 32); }
    if (lane < 32) out[(size_t)PT * 1024 + lane] = lane == 0 ? gsum0 : (lane == 1 ? gsum1 : 0.f);
  }
}

// Workspace: [sticky status block 256 B][debug counters][control block][rings + dummy-store scratch][dz_top][partial sums].
// The first two have FIXED offsets (SUNERF_PIPE_WS_STICKY / SUNERF_PIPE_WS_DEBUG in the header) whatever the batch size, so a
// caller that keeps one workspace for launches of different sizes always finds them.
struct PipeLayout {
  int n_act, n_links, NPX, NP, grid;
  size_t sticky, dbg, ctrl, rings, dz_top, partial, partial_out, total;
  PipeLayout(int64_t n_chunks_total, int n_linear, int cus) {
    n_act = n_linear - 1;
    n_links = n_act - 1;
    grid = cus;
    NPX = (cus / 8) / (2 * n_act);
    NP = 8 * NPX;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    sticky = off; off += SUNERF_PIPE_WS_DEBUG;
    dbg = off; off += up((size_t)cus * 64 * sizeof(unsigned));
    ctrl = off; off += up(ctrl_bytes(NP, n_links));
    rings = off; off += up((size_t)NP * n_links * RING * SLOT + (size_t)cus * 4 * 2048);
    dz_top = off; off += up((size_t)(n_chunks_total > 0 ? n_chunks_total : 1) * SLOT);
    partial = off; off += up((size_t)n_act * NP * PT * (PT + 1) * 1024 * sizeof(float));
    partial_out = off; off += up((size_t)cus * (PT + 2) * 1024 * sizeof(float));
    total = off;
  }
};
static_assert(SUNERF_PIPE_WS_STICKY == 0 && SUNERF_PIPE_WS_DEBUG == 256, "fixed offsets of the workspace header");

// flags bit 7: the pipelined kernel of every call is bracketed by library-owned HIP events on the launch stream; read (and
// cleared) by sunerf_bwd_pipe_kernel_time.  The product's call sequence stays one C-ABI call per backward.
struct TimedLaunch { hipEvent_t e0, e1; };
std::mutex g_timed_mutex;
std::vector<TimedLaunch> g_timed;

bool pipe_supported(int d_filter, int n_linear, int d_out, int cus) {
  return d_filter == PD && n_linear >= 3 && n_linear <= SUNERF_MAX_LAYERS && d_out >= 1 && d_out <= 2 && cus == 256 &&
         2 * (n_linear - 1) <= cus / 8;
}

int device_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus;
}

}  // namespace

extern "C" size_t sunerf_bwd_pipe_workspace_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear) {
  const int cus = device_cus();
  if (n_rays < 0 || n_samples < 2 || !pipe_supported(d_filter, n_linear, 1, cus)) return 0;
  const int64_t n_chunks = (n_samples + 31) / 32;
  return PipeLayout(n_rays * n_chunks, n_linear, cus).total;
}

extern "C" int sunerf_mlp_backward_pipe(int d_filter, int n_linear, int d_out, const void* packedT, const void* act_stash,
                                        const float* g_raw, const void* g_absmax, int64_t n_rays, int n_samples,
                                        void* workspace, size_t workspace_bytes, float* const* grad_weights_host,
                                        float* const* grad_biases_host, int accumulate, int flags, void* stream) {
  if (!grad_weights_host || !grad_biases_host) return SUNERF_E_BADARG;
  if (n_rays <= 0 || n_samples < 2) return SUNERF_E_BADARG;
  if (!packedT || !act_stash || !g_raw || !g_absmax || !workspace) return SUNERF_E_BADARG;
  const int cus = device_cus();
  if (!pipe_supported(d_filter, n_linear, d_out, cus)) return SUNERF_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (n_samples + 31) / 32;
  const PipeLayout L(n_rays * n_chunks, n_linear, cus);
  if (workspace_bytes < L.total) return SUNERF_E_WORKSPACE;
  char* ws = (char*)workspace;
  PipeArgs a;
  a.packedT = (const char*)packedT; a.act_stash = (const char*)act_stash; a.dz_top = ws + L.dz_top; a.rings = ws + L.rings;
  a.ctrl = (unsigned*)(ws + L.ctrl); a.partial = (float*)(ws + L.partial); a.g_raw = g_raw;
  a.g_absmax_bits = (const unsigned*)g_absmax; a.partial_out = (float*)(ws + L.partial_out);
  a.n_chunks_total = n_rays * n_chunks; a.S = n_samples; a.n_chunks = n_chunks; a.n_linear = n_linear; a.d_out = d_out;
  a.NP = L.NP; a.NPX = L.NPX; a.hi_only = flags & 1; a.inject = (flags >> 8) & 1;
  a.dbg = (flags & 2) ? (unsigned*)(ws + L.dbg) : nullptr;
  hipError_t e;
  // the control block (launch status, arrival counter, XCC masks, hand-off counters) starts from zero in every launch; the
  // sticky block in front of it is only ever OR-ed into (reduce_grads_kernel) and belongs to the caller
  if ((e = hipMemsetAsync(ws + L.ctrl, 0, L.rings - L.ctrl, st)) != hipSuccess) return (int)e;
  const size_t lds_pre = (size_t)NBUF * BUF_PRE + 1024;
  const size_t lds_pipe = (size_t)NBUF_H * BUF_HID + 1024 + NBUF_H * 256 + 64;
  if ((e = hipFuncSetAttribute((const void*)bwd_prologue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pre)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)bwd_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)bwd_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe)) != hipSuccess) return (int)e;
  TimedLaunch tl = {nullptr, nullptr};
  if (flags & 0x80) {
    if ((e = hipEventCreate(&tl.e0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&tl.e1)) != hipSuccess) { (void)hipEventDestroy(tl.e0); return (int)e; }
  }
  SUNERF_CLEAR_ERROR();
  hipLaunchKernelGGL(bwd_prologue_kernel, dim3((unsigned)cus), dim3(WG_PRE), lds_pre, st, a);
  SUNERF_CHECK_LAUNCH();
  if (tl.e0) (void)hipEventRecord(tl.e0, st);
  if (a.hi_only) hipLaunchKernelGGL(bwd_pipe_kernel<true>, dim3((unsigned)cus), dim3(WG), lds_pipe, st, a);
  else hipLaunchKernelGGL(bwd_pipe_kernel<false>, dim3((unsigned)cus), dim3(WG), lds_pipe, st, a);
  if (tl.e0) {
    (void)hipEventRecord(tl.e1, st);
    std::lock_guard<std::mutex> lock(g_timed_mutex);
    g_timed.push_back(tl);
  }
  SUNERF_CHECK_LAUNCH();
  ReduceArgs r;
  const size_t slot = (size_t)PT * (PT + 1) * 1024;
  for (int i = 0; i < n_linear; ++i) {
    if (!grad_weights_host[i] || !grad_biases_host[i]) return SUNERF_E_BADARG;
    r.gW[i] = grad_weights_host[i];
    r.gb[i] = grad_biases_host[i];
    if (i < n_linear - 1) {
      r.partial[i] = a.partial + (size_t)i * L.NP * slot;
      r.split[i] = L.NP;
      r.slot[i] = slot;
    } else {
      r.partial[i] = a.partial_out;
      r.split[i] = cus;
      r.slot[i] = (size_t)(PT + 2) * 1024;
    }
    if (i < n_linear - 2) {                 // db_i: summed by the data-gradient waves of stage i + 1 into layer i's own slots
      r.bias[i] = r.partial[i] + (size_t)PT * 1024;
      r.bias_split[i] = L.NP;
      r.bias_slot[i] = slot;
      r.bias_tr[i] = (size_t)(PT + 1) * 1024;
    } else {                                // last activation layer and out layer: summed by the prologue's workgroups
      r.bias[i] = a.partial_out + (size_t)(i == n_linear - 1 ? PT : PT + 1) * 1024;
      r.bias_split[i] = cus;
      r.bias_slot[i] = (size_t)(PT + 2) * 1024;
      r.bias_tr[i] = 32;
    }
  }
  r.g_absmax_bits = (const unsigned*)g_absmax; r.n_linear = n_linear; r.D = d_filter; r.d_out = d_out; r.accumulate = accumulate;
  r.sumsq = (const float*)((const char*)packedT + sunerf_packed_mlp_t_bytes(d_filter, n_linear) - SUNERF_MAX_LAYERS * sizeof(float));
  r.status = a.ctrl;
  r.sticky = (unsigned*)(ws + L.sticky);
  hipLaunchKernelGGL(reduce_grads_kernel, dim3(PT * (PT + 1) * 1024 / 256, n_linear), dim3(256), 0, st, r);
  SUNERF_CHECK_LAUNCH();
  return 0;
}

extern "C" int sunerf_bwd_pipe_kernel_time(double* total_ms, int* launches) {
  std::vector<TimedLaunch> mine;
  {
    std::lock_guard<std::mutex> lock(g_timed_mutex);
    mine.swap(g_timed);
  }
  double sum = 0.0;
  int n = 0, rc = 0;
  for (const TimedLaunch& t : mine) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(t.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.e0, t.e1);
    if (e == hipSuccess) { sum += ms; ++n; } else rc = (int)e;
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = n;
  return rc;
}
