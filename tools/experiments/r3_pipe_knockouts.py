"""Knock-out instrumentation of csrc/bwd_pipe.hip for timing experiments (round 3).

    python tools/experiments/r3_pipe_knockouts.py <bwd_pipe.hip> <out.hip>

writes a copy of the kernel source in which -DPIPE_KO=<bits> removes parts of the hidden stages' per-chunk work (the results are
then garbage; the hand-off protocol still runs unless bits 512 / 2048 are set):
   1 data waves' matrix instructions      2 data waves' B-fragment reads       4 epilogue arithmetic (dZ = dH * cos)
   8 weight waves' matrix instructions   16 weight waves' transposed reads    32 DMA pieces of waves 5 - 7
  64 DMA pieces of the data waves       256 wave 4's DMA piece               512 the gate never waits (stages free-run)
2048 no publication / poll (except stage 1, which the in-layer stage waits for)
1024 the data waves' output stores
Build each variant as its own library (hipcc -DPIPE_KO=n -c; link with the other objects of csrc/build.sh) and select it with
SUNERF_HIP_LIB=... python tools/pipe_check.py 181 128 5.  Measured table: tools/experiments/README.md.
"""
import sys

src, dst = sys.argv[1:3]
s = open(src).read()


def rep(a, b):
    global s
    assert s.count(a) == 1, (a, s.count(a))
    s = s.replace(a, b)


rep('constexpr int WG = 512; ', '#ifndef PIPE_KO\n#define PIPE_KO 0\n#endif\nconstexpr int WG = 512; ')
rep('''      for (int s = 0; s < PF; ++s) bf[s] = *(const half8*)(B + s * 1024 + lane * 16);
#pragma unroll
      for (int ks = 0; ks < PKS; ++ks) {
        if (ks + PF < PKS) bf[(ks + PF) % (PF + 1)] = *(const half8*)(B + (ks + PF) * 1024 + lane * 16);
        if constexpr (!HI_ONLY) dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_lo[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_hi[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        if (ks == 1) piece_z(wave);
        if (ks == 4) piece_z(4 + wave);
        if (ks == 7) piece_z(8 + wave);
        if (ks == 10) piece_z(12 + wave);''', '''      for (int s = 0; s < PF; ++s) bf[s] = (PIPE_KO & 2) ? (half8){0, 0, 0, 0, 0, 0, 0, 0} : *(const half8*)(B + s * 1024 + lane * 16);
#pragma unroll
      for (int ks = 0; ks < PKS; ++ks) {
        if (ks + PF < PKS && !(PIPE_KO & 2)) bf[(ks + PF) % (PF + 1)] = *(const half8*)(B + (ks + PF) * 1024 + lane * 16);
        if (!(PIPE_KO & 1)) {
        if constexpr (!HI_ONLY) dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_lo[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt_hi[ks], bf[ks % (PF + 1)], dacc, 0, 0, 0);
        } else { asm volatile("" : "+v"(bf[ks % (PF + 1)])); }
        if (!(PIPE_KO & 64)) {
        if (ks == 1) piece_z(wave);
        if (ks == 4) piece_z(4 + wave);
        if (ks == 7) piece_z(8 + wave);
        if (ks == 10) piece_z(12 + wave);
        }''')
rep('''      dz_tile(dacc, c0f, c1f, d0, d1);
      const Rsrc ro = make_rsrc(out_z, SLOT);
      buf_store(d0, ro, (2 * U) * 1024);
      buf_store(d1, ro, (2 * U + 1) * 1024);''', '''      if (PIPE_KO & 4) { d0 = c0f; d1 = c1f; asm volatile("" :: "v"(dacc)); }
      else dz_tile(dacc, c0f, c1f, d0, d1);
      const Rsrc ro = make_rsrc(out_z, SLOT);
      if (!(PIPE_KO & 1024)) {
      buf_store(d0, ro, (2 * U) * 1024);
      buf_store(d1, ro, (2 * U + 1) * 1024);
      } else { asm volatile("" :: "v"(d0), "v"(d1)); }''')
rep('''    if (gatew) { piece_h(0); return; }
#pragma unroll
    for (int q = 0; q < 5; ++q) {''', '''    if (gatew) { if (!(PIPE_KO & 256)) piece_h(0); return; }
#pragma unroll
    for (int q = 0; q < ((PIPE_KO & 32) ? 0 : 5); ++q) {''')
rep('''        mfma_agpr(acc[i][0], af, bf0);
        mfma_agpr(acc[i][1], af, bf1);''', '''        if (!(PIPE_KO & 8)) {
        mfma_agpr(acc[i][0], af, bf0);
        mfma_agpr(acc[i][1], af, bf1);
        }''')
rep('''#define TR_FIXED(R0, R1, R2, R3, OFF, var, base)                                                                        \\
  asm volatile(''', '''#define TR_FIXED(R0, R1, R2, R3, OFF, var, base)                                                                        \\
  if (PIPE_KO & 16) asm volatile("" : "={v[" #R0 ":" #R3 "]}"(var)); else                                                \\
  asm volatile(''')
# protocol off: no gate, no publication, no poll
rep('''    if (need_in || need_out) {
      bool ok = true;''', '''    if (!(PIPE_KO & 512) && (need_in || need_out)) {
      bool ok = true;''')
rep('''      if (lane < 2) {
        const int pv = lane == 0 ? it + 1 : it - NBUF + 1;
        st_agent(lane == 0 ? my_cons : my_prod, (unsigned)(pv > 0 ? pv : 0));
      }
      dma_poll(poll_src, lds_poll + (nxt & (NBUF - 1)) * 256);''', '''      if (!(PIPE_KO & 2048) || l == 1) {
      if (lane < 2) {
        const int pv = lane == 0 ? it + 1 : it - NBUF + 1;
        st_agent(lane == 0 ? my_cons : my_prod, (unsigned)(pv > 0 ? pv : 0));
      }
      dma_poll(poll_src, lds_poll + (nxt & (NBUF - 1)) * 256);
      }''')
open(dst, 'w').write(s)
