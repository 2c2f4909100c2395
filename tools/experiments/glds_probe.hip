// Probe of two gfx950 primitives the SP16 conv path relies on:
//   global_load_lds_dwordx4: LDS destination = uniform base + lane * 16, per-lane global source
//   v_permlane32_swap: which halves are exchanged
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* g, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __builtin_amdgcn_global_load_lds(g + (63 - lane) + wave * 64, (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x];
}
__global__ void k2(unsigned* out) {
  unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  u32x4 *g, *o; unsigned* o2;
  hipMalloc(&g, 128 * 16); hipMalloc(&o, 128 * 16); hipMalloc(&o2, 64 * 8);
  unsigned h[128 * 4];
  for (int i = 0; i < 128; ++i) for (int j = 0; j < 4; ++j) h[i * 4 + j] = i * 10 + j;
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 128 * 16, 0, g, o);
  hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) { const int w = i / 64, l = i % 64; const unsigned exp = ((63 - l) + w * 64) * 10; for (int j = 0; j < 4; ++j) bad += h[i * 4 + j] != exp + j; }
  printf("glds: %s (lds[0]=%u lds[1]=%u lds[64]=%u)\n", bad ? "MISMATCH" : "lane-linear OK", h[0], h[4], h[256]);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, o2);
  unsigned r[128];
  hipMemcpy(r, o2, sizeof r, hipMemcpyDeviceToHost);
  printf("permlane32_swap(a=1000+l, b=2000+l): lane0 -> (%u,%u) lane31 -> (%u,%u) lane32 -> (%u,%u) lane63 -> (%u,%u)\n", r[0], r[1], r[62], r[63], r[64], r[65], r[126], r[127]);
  return 0;
}
