// Is the read ceiling of hbm_streams.hip (6.2 - 6.6 TB/s, ~2.7 KB per shader clock) HBM's or the L2 <-> fabric path's?
// The same 16-B-per-lane read loop over working sets that fit the memory-side cache (256 MiB Infinity Cache) and ones that do
// not, each read repeatedly (every pass misses the 8 x 4 MiB L2s), non-temporal and plain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void rd(const f4* __restrict__ in, size_t n, int passes, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main() {
  const size_t cap = (size_t)8 << 30;
  f4* a; float* sink;
  if (hipMalloc(&a, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(a, 0, cap);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t mb : {32, 64, 128, 192, 256, 512, 2048, 8192}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    const int passes = (int)(((size_t)16 << 30) / bytes);            // 16 GiB of reads per run
    for (int nt = 0; nt < 2; ++nt) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (nt) rd<true><<<1024, 256>>>(a, n, passes, sink); else rd<false><<<1024, 256>>>(a, n, passes, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("working set %5zu MiB x %4d passes  %-12s %8.2f ms  %.2f TB/s\n", mb, passes, nt ? "non-temporal" : "plain", best,
             (double)bytes * passes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
