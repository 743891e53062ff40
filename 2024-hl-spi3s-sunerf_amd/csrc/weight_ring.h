// LDS ring of weight pages fed by asynchronous LDS-DMA (gfx950), shared by the forward render kernel and the data-gradient
// kernel: both walk a packed fp16 (hi | lo) A-fragment stream that all four waves of a workgroup consume in lock-step.
#pragma once
#include "sunerf_common.h"

#ifndef SUNERF_DBG_BARRIER
#define SUNERF_DBG_BARRIER 0
#endif
#ifndef SUNERF_DBG_TILES
#define SUNERF_DBG_TILES 0
#endif
#ifndef SUNERF_DBG_SHIFT
#define SUNERF_DBG_SHIFT 0
#endif
#ifndef SUNERF_DBG_KIND
#define SUNERF_DBG_KIND 1
#endif
// SUNERF_ABL_NO_DMA (round 4, tools/experiments/r4_fwd_dma.sh): timing-only ablation -- the ring is filled once with real
// pages and never refilled: what the weight stream (its LDS-DMA instructions, its L2 -> LDS traffic) costs the forward.
// Outputs are wrong, operands stay finite.
#ifndef SUNERF_ABL_NO_DMA
#define SUNERF_ABL_NO_DMA 0
#endif
namespace sunerf_ring {

constexpr int WAVES = 4;
constexpr int NSLOT = 4;  // LDS ring slots for weight pages

// ---- weight stream: global (L2-resident packed image) -> LDS ring, asynchronous LDS-DMA ------------------------
// The packed image is a byte FIFO in consumption order, 2048 B per k-step (sunerf_common.h).  It is DMA'd page by
// page (PAGE = one hidden tile = KS k-steps) into an LDS ring of 4 pages by `global_load_lds_dwordx4` (no VGPR
// staging); each of the 4 waves moves a quarter of a page.  The stream length is a multiple of the ring size, so the
// LDS address of k-step f of a pass is simply (2048 f) mod RING.  Page p+1 is ACQUIRED while page p is still being
// read (PF k-steps before the reads cross into it):
//     s_waitcnt vmcnt(pieces of one page)  -> this wave's pieces of page p+1 have landed (page p+2 may be in flight)
//     s_barrier                            -> everyone's pieces landed; every wave is past page p-1
//     then, one piece per k-step: the DMA of page p+3 into the slot page p-1 occupied
// Ring occupancy: p (read), p+1 (landed), p+2 (in flight), p+3 (being issued) = 4 pages.
// The DMA and its waits are inline asm on purpose: hipcc does not count them, so it neither drains them with
// vmcnt(0) at barriers nor in front of unrelated LDS reads, and our counted waits stay valid when compiler-issued
// loads/stores interleave (extra younger operations only make `vmcnt(N)` stricter).
template <int D>
struct Ring {
  static constexpr int KS = D / 16;
  static constexpr int PAGE_STEPS = KS < 16 ? KS : 16;   // k-steps per page (a d = 512 tile is two pages)
  static constexpr int PAGE = PAGE_STEPS * 2048;
  static constexpr int RING = NSLOT * PAGE;
  static constexpr int PIECES = PAGE / 1024 / WAVES;     // 1 KiB DMA instructions per wave and page
  static_assert(NSLOT == 4 && PAGE % (1024 * WAVES) == 0, "page must split evenly over the waves");
  const char* src;       // this wave's quarter of the next page to prefetch (wave-uniform)
  const char* src_first; // ... of page 0
  const char* src_end;   // ... one past the last page
  unsigned dst;          // LDS byte address of this wave's quarter of the ring slot to fill next
  unsigned dst_first;    // ... of slot 0
  unsigned voff;         // lane * 16
#if SUNERF_ABL_NO_DMA
  bool priming = true;
#endif

  __device__ __forceinline__ void init(const char* packed, size_t stream_bytes, unsigned lds_base, int wave, int lane) {
    const unsigned quarter = __builtin_amdgcn_readfirstlane(wave) * (PAGE / WAVES);
    src_first = packed + quarter;
    src_end = src_first + stream_bytes;
    src = src_first;
    dst_first = lds_base + quarter;
    dst = dst_first;
    voff = lane * 16;
  }
  // DMA of piece j (1 KiB) of this wave's quarter of the page being prefetched.  The immediate offset of
  // global_load_lds applies to the global AND the LDS address (measured), so src/dst only move in 4 KiB strides.
  // The pieces of a page are dealt out one per k-step between two acquires instead of being issued as a burst right
  // behind the barrier: 4 waves x 8 KiB arriving together would collide with the fragment reads of all four waves.
  template <int J>
  __device__ __forceinline__ void issue_piece() {
    static_assert(J >= 0 && J < PIECES, "piece index");
#if SUNERF_ABL_NO_DMA
    if (priming)
#endif
    {
      const char* sg = src + (J / 4) * 4096;
      const unsigned dg = dst + (J / 4) * 4096;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%3"
                   :: "v"(voff), "s"(dg), "s"(sg), "i"((J % 4) * 1024) : "memory");
    }
    if (J == PIECES - 1) {   // page complete: advance to the next page / ring slot
      src += PAGE;
      if (src == src_end) src = src_first;
      dst += PAGE;
      if (dst == dst_first + RING) dst = dst_first;
    }
  }
  template <int J = 0>
  __device__ __forceinline__ void issue_page() {
    issue_piece<J>();
    if constexpr (J + 1 < PIECES) issue_page<J + 1>();
  }
  // makes the next page readable: this wave's pieces of it have landed (the PIECES younger ones, of the page after
  // it, may still be in flight), then everyone's
  // EXTRA = vector-memory operations (stash stores) that are guaranteed to have been issued after the last piece of
  // the page being acquired, besides the PIECES of the following page
  template <int EXTRA = 0>
  __device__ __forceinline__ void acquire() {
#if SUNERF_DBG_BARRIER
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PIECES + EXTRA) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (dbg_kind == SUNERF_DBG_KIND) { dbg_vm += (unsigned)(t1 - t0); dbg_bar += (unsigned)(t2 - t1); dbg_n += 1; }
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PIECES + EXTRA) : "memory");
    __builtin_amdgcn_s_barrier();
#endif
  }
#if SUNERF_DBG_BARRIER
  unsigned dbg_vm = 0, dbg_bar = 0, dbg_n = 0;
  int dbg_kind = 0;
#endif
#if SUNERF_DBG_TILES
  // cycles between mark_begin() and mark_end() of the regions tagged SUNERF_DBG_KIND (one kind per build: the counters
  // live in SGPRs, of which the DMA addressing leaves few)
  unsigned long long dbg_t0 = 0;
  unsigned dbg_cyc = 0, dbg_cnt = 0;
  __device__ __forceinline__ void mark_begin(int kind) { if (kind == SUNERF_DBG_KIND) dbg_t0 = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void mark_end(int kind) {
    if (kind == SUNERF_DBG_KIND && dbg_t0 != 0) { dbg_cyc += (unsigned)((__builtin_amdgcn_s_memtime() - dbg_t0) >> SUNERF_DBG_SHIFT); dbg_cnt += 1; }
  }
#else
  __device__ __forceinline__ void mark_begin(int) {}
  __device__ __forceinline__ void mark_end(int) {}
#endif
};

}  // namespace sunerf_ring
