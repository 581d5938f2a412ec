// build: hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o tools/micro/clock_probe   (result: profiles/r02_clock_probe.txt)
// Does the shader clock depend on how much of the chip is busy?  One wave per workgroup runs a chain of N dependent v_fma_f32
// (issue-to-issue latency of a dependent VALU op is a fixed number of cycles), for 1 .. 1024 workgroups; s_memtime alongside.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void chain_kernel(float* out, int n, unsigned long long* ticks) {
  float v = threadIdx.x * 1e-3f, a = 1.0000001f, b = 1e-7f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, a, b);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (v == 123.f) out[0] = v;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 4); hipMalloc(&ticks, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int n = 1 << 20;
  for (int G : {1, 32, 256, 1024, 1}) {
    float best = 1e9f; unsigned long long tk = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      chain_kernel<<<G, 64>>>(out, n, ticks);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) { best = ms; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost); }
    }
    printf("%4d workgroup(s): %d dependent FMAs in %.3f ms = %.2f ns each; s_memtime ticks %.0f = %.1f MHz\n", G, n, best, best * 1e6 / n,
           (double)tk, tk / (best * 1e3));
  }
  return 0;
}
