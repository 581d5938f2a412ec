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

// the same boundary with one FLAG per workgroup instead of one shared counter: arriving is a plain device-scope store of the epoch
// into the workgroup's own word (no read-modify-write, no contention), waiting is wave 0 polling the G words (one 128-byte line
// for G <= 32) until all have reached the epoch
__device__ __forceinline__ void flag_barrier(unsigned* flags, int G, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int lane = threadIdx.x;
    for (;;) {
      bool ok = true;
      for (int i = lane; i < G; i += 64) ok = ok && (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// mode 0: barriers only; 1: + activation exchange through global memory; 2 / 3: the same with the flag barrier
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
  const unsigned fbase = counter[64 + blockIdx.x];   // this workgroup's own flag: every flag holds the same epoch between launches
  for (int s = 0; s < nstage; ++s) {
    if (mode & 1) {
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
    if (mode >= 2) flag_barrier(counter + 64, G, fbase + (unsigned)(s + 1));
    else grid_barrier(counter, base + (unsigned)G * (unsigned)(s + 1));
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned* counter; unsigned short* act; unsigned* sink;
  hipMalloc(&counter, 4096); hipMalloc(&act, 2 * 4096 * 2); hipMalloc(&sink, 4);
  hipMemset(counter, 0, 4096); hipMemset(act, 0, 2 * 4096 * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nstage = 512;
  for (int mode : {0, 1, 2, 3})
    for (int G : {8, 16, 32, 64, 128}) {
      float best = 1e9f;
      hipMemset(counter, 0, 4096);   // (the flags of all G workgroups must hold one epoch when a launch starts)
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        stages_kernel<<<G, 256>>>(counter, act, nstage, mode, sink);
        hipEventRecord(b);
        if (hipEventSynchronize(b) != hipSuccess) { printf("failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("%s %s G = %3d workgroups: %.2f us per stage\n", mode >= 2 ? "flags:  " : "counter:", (mode & 1) ? "barrier + 8 KB exchange," : "barrier only,           ", G, best * 1e3 / nstage);
    }
  return 0;
}
