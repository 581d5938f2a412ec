// build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier   (result: profiles/r02_grid_barrier.txt)
// What does a stage boundary cost INSIDE one launch?  G co-resident workgroups run `nstage` stages; in every stage each workgroup
// reads the 16 x 256 bf16 activation rows all workgroups wrote in the previous stage (8 KB, device-coherent loads), writes its own
// slice, and meets the others at a grid barrier (one device-scope atomic add on a monotonic counter + a polling load).  Compared
// with the ~5 us per dependent launch of the decode-step chain (csrc/decode_rt.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release this workgroup's stores device-wide
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
    __threadfence();  // acquire
  }
  __syncthreads();
}

// mode 0: barriers only; 1: + activation exchange through global memory
__global__ __launch_bounds__(256) void stages_kernel(unsigned* counter, unsigned short* act, int nstage, int mode, unsigned* sink) {
  const int G = gridDim.x, tid = threadIdx.x;
  unsigned base;
  if (tid == 0) {
    const unsigned v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = v - v % (unsigned)G;
  }
  base = __shfl(base, 0);
  __shared__ unsigned sbase;
  if (tid == 0) sbase = base;
  __syncthreads();
  base = sbase;
  unsigned acc = 0;
  for (int s = 0; s < nstage; ++s) {
    if (mode) {
      // read all 16 x 256 values of the previous stage (two buffers, alternating), 16 bytes per thread
      const unsigned short* src = act + (s & 1) * 4096;
      const uint4 v = reinterpret_cast<const uint4*>(src)[tid];        // 256 threads x 16 B = 4 KB ...
      const uint4 w = reinterpret_cast<const uint4*>(src)[256 + tid];  // ... x 2 = 8 KB (plain loads: the barrier's acquire fence dropped stale lines)
      acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
      // write this workgroup's slice of the next stage's rows: 4096 values / G workgroups
      unsigned short* dst = act + ((s + 1) & 1) * 4096;
      const int per = 4096 / G;
      if (tid < per) dst[blockIdx.x * per + tid] = (unsigned short)(acc + s);
    }
    grid_barrier(counter, base + (unsigned)G * (unsigned)(s + 1));
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  unsigned* counter; unsigned short* act; unsigned* sink;
  hipMalloc(&counter, 4); hipMalloc(&act, 2 * 4096 * 2); hipMalloc(&sink, 4);
  hipMemset(counter, 0, 4); hipMemset(act, 0, 2 * 4096 * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nstage = 512;
  for (int mode : {0, 1})
    for (int G : {8, 16, 32, 64, 128}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        stages_kernel<<<G, 256>>>(counter, act, nstage, mode, sink);
        hipEventRecord(b);
        if (hipEventSynchronize(b) != hipSuccess) { printf("failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("%s G = %3d workgroups: %.2f us per stage\n", mode ? "barrier + 8 KB exchange," : "barrier only,           ", G, best * 1e3 / nstage);
    }
  return 0;
}
