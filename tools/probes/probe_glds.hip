// Hardware probe: semantics of global_load_lds_dwordx4 with SGPR base (saddr), VGPR offset and an immediate offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned voff = threadIdx.x * 16;
  unsigned m0v = 1024;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:512\n\ts_waitcnt vmcnt(0)"
               :: "v"(voff), "s"(m0v), "s"(src) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(8192); for (int i = 0; i < 8192; ++i) h[i] = i;   // word index as value
  unsigned *d, *o; hipMalloc(&d, 8192 * 4); hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  k<<<1, 64, 16384>>>(d, o);
  std::vector<unsigned> r(4096); hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 4096; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("LDS words written: [%d, %d] (bytes [%d, %d]); value at first = %u (source word index => source byte %u)\n", first, last, first * 4, last * 4 + 3, r[first], r[first] * 4);
  printf("expect if imm offset applies to BOTH: LDS bytes [1536, 2559], source byte 512\n");
  return 0;
}
