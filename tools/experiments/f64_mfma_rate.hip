// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (1, 2, 4 independent accumulators) and of v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = x[i] * a + b;
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; hipMalloc(&out, 1024 * 1024 * 8);
  const int iters = 20000;
  for (int wpb : {1, 2, 4}) {
    const int blocks = 256 * 4, threads = 64 * wpb;  // wpb waves per block; 4 blocks per CU
    float m1 = timeit([&] { k_mfma<1><<<blocks, threads>>>(out, iters); });
    float m2 = timeit([&] { k_mfma<2><<<blocks, threads>>>(out, iters); });
    float m4 = timeit([&] { k_mfma<4><<<blocks, threads>>>(out, iters); });
    const double waves = (double)blocks * wpb;
    auto tf = [&](float ms, int nacc) { return waves * iters * nacc * 2048.0 / (ms * 1e-3) / 1e12; };
    auto cyc = [&](float ms, int nacc) { return ms * 1e-3 * 2.4e9 / (iters * nacc) / (waves / 1024.0); };
    printf("waves/CU=%2d  nacc=1: %.1f TF (%.0f cyc/mfma/SIMD)  nacc=2: %.1f TF (%.0f)  nacc=4: %.1f TF (%.0f)\n", 4 * wpb,
           tf(m1, 1), cyc(m1, 1), tf(m2, 2), cyc(m2, 2), tf(m4, 4), cyc(m4, 4));
  }
  float mf = timeit([&] { k_fma<<<1024, 256>>>(out, iters); });
  printf("v_fma_f64: %.1f TF\n", 1024.0 * 256 * iters * 8 * 2 / (mf * 1e-3) / 1e12);
  return 0;
}
