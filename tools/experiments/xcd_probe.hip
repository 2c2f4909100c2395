// (a) which XCD does workgroup b run on?  (b) how fast is a software barrier among the 32 workgroups of ONE XCD
// (atomic counter in that XCD's L2, agent-scope relaxed atomics, no L2 write-back / invalidate)?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__global__ void where(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

__global__ void barrier_bench(unsigned* counter, double* data, int rounds, unsigned* info) {
  if ((blockIdx.x & 7) != 0) return;                 // only the workgroups the dispatcher places on XCD 0
  const unsigned nwg = gridDim.x / 8, me = blockIdx.x / 8;
  unsigned target = 0;
  double acc = 0.;
  for (int r = 0; r < rounds; ++r) {
    // every workgroup publishes a value, then everybody reads everybody's
    if (threadIdx.x == 0) __hip_atomic_store(&data[me], (double)(r * 1000 + me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    target += nwg;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 50000000) __builtin_amdgcn_s_sleep(1);
      if (spins >= 50000000) info[1] = 1;           // timeout (never hangs)
    }
    __syncthreads();
    if (threadIdx.x < nwg) {
      const double v = __hip_atomic_load(&data[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (double)(r * 1000 + threadIdx.x)) atomicAdd(&info[0], 1u);   // stale value seen
      acc += v;
    }
    // second barrier so that nobody overwrites data[] before all have read it
    __syncthreads();
    target += nwg;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 50000000) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (acc == -1.) info[2] = 1;
}
int main() {
  unsigned *out, *counter, *info; double* data;
  hipMalloc(&out, 512 * 4); hipMalloc(&counter, 4); hipMalloc(&info, 16); hipMalloc(&data, 64 * 8);
  hipLaunchKernelGGL(where, dim3(512), dim3(64), 0, 0, out);
  unsigned h[512]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0; for (int b = 0; b < 512; ++b) bad += h[b] != (unsigned)(b & 7);
  printf("XCC_ID of workgroups 0..15:"); for (int b = 0; b < 16; ++b) printf(" %u", h[b]);
  printf("\nworkgroup b on XCD b %% 8: %s (%d of 512 differ)\n", bad ? "NO" : "yes", bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwg : {4, 16, 32}) {
    hipMemset(counter, 0, 4); hipMemset(info, 0, 16);
    const int rounds = 2000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(barrier_bench, dim3(nwg * 8), dim3(256), 0, 0, counter, data, rounds, info);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned hi[4]; hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost);
    printf("%2d workgroups on XCD 0: %.2f us per barrier (2 per round), stale reads %u, timeout %u\n", nwg, ms * 1e3 / (2 * rounds), hi[0], hi[1]);
  }
  return 0;
}
