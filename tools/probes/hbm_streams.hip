// Practical HBM ceilings of an MI355X for the three stream shapes of the training step: read-only (wgrad: H and dZ in),
// write-only (the forward's stash), and copy = one byte written per byte read (dgrad: cos in, dZ out).  8 GiB per buffer
// (far beyond the 256 MiB memory-side cache), 16 bytes per lane and iteration, grid-stride, plain and non-temporal accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>   // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ in, f4* __restrict__ out, size_t n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    if (MODE != 1) v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (MODE == 0) acc += v;
    else if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
  const size_t bytes = (size_t)8 << 30, n = bytes / 16;
  f4 *a, *b; float* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[3] = {"read-only", "write-only", "copy (1 B out per B in)"};
  for (int grid : {256 * 4, 256 * 16})
    for (int mode = 0; mode < 3; ++mode)
      for (int nt = 0; nt < 2; ++nt) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          (void)hipEventRecord(e0);
          if (mode == 0) { if (nt) stream<0, true><<<grid, 256>>>(a, b, n, sink); else stream<0, false><<<grid, 256>>>(a, b, n, sink); }
          if (mode == 1) { if (nt) stream<1, true><<<grid, 256>>>(a, b, n, sink); else stream<1, false><<<grid, 256>>>(a, b, n, sink); }
          if (mode == 2) { if (nt) stream<2, true><<<grid, 256>>>(a, b, n, sink); else stream<2, false><<<grid, 256>>>(a, b, n, sink); }
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        const double moved = (mode == 2 ? 2.0 : 1.0) * bytes;
        printf("grid %5d  %-26s %-12s %7.2f ms  %.2f TB/s\n", grid, names[mode], nt ? "non-temporal" : "plain", best, moved / (best * 1e-3) / 1e12);
      }
  return 0;
}
