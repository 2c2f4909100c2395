// cost of a chain of small dependent kernels on one stream (the Newton-Schulz launch schedule's building block)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void load_k(const double* a, double* out, int n) {   // 64 dependent-free loads per lane + a store
  double s = 0.;
  const double* p = a + (blockIdx.x * 256 + threadIdx.x) % 1024;
#pragma unroll
  for (int u = 0; u < 64; ++u) s += p[u * 1024];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double *a, *o; hipMalloc(&a, 1 << 20); hipMalloc(&o, 1 << 20); hipMemset(a, 0, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 32, 128}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(256), 0, 0, (int*)nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("empty kernel, grid %3d: %.2f us per launch\n", grid, ms);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(load_k, dim3(32), dim3(256), 0, 0, a, o, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("64 loads per lane, grid 32: %.2f us per launch\n", ms);
  }
  return 0;
}
