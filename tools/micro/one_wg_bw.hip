// build: hipcc --offload-arch=gfx950 -O3 tools/micro/one_wg_bw.hip -o tools/micro/one_wg_bw   (result: profiles/r02_one_wg_bw.txt)
// How fast can ONE workgroup (1024 threads) stream a weight set?  (sizing of a single-workgroup decode-step kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ p, long n16, unsigned* out, int unroll8) {
  unsigned acc = 0;
  const int tid = threadIdx.x;
  for (long i = tid; i + 7 * 1024 < n16; i += 8 * 1024) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[i + j * 1024];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x12345678) out[0] = acc;
}
int main() {
  const long bytes = 24l << 20;
  void* d; unsigned* o;
  hipMalloc(&d, bytes); hipMalloc(&o, 4);
  hipMemset(d, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int nwg : {1, 2, 4, 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      stream_kernel<<<nwg, 1024>>>((const u32x4*)d, bytes / 16 / nwg, o, 1);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep == 2) printf("%d workgroup(s) x %.1f MB each: %.1f us -> %.1f GB/s per workgroup\n", nwg, bytes / 1e6 / nwg, ms * 1e3, bytes / nwg / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
