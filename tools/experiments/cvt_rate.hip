// issue rate of the VALU instructions in the moments kernels' inner loop: v_cvt_f64_f32, v_add_f64, v_fma_f64, and the fp64 MFMA
//   hipcc --offload-arch=gfx950 -O3 -o cvt_rate cvt_rate.hip && ./cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  double d0 = f0, d1 = f1, d2 = f2, d3 = f3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
      if (OP == 1) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4), "v"(d5));
      if (OP == 3) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d4, d5, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d4, d5, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(d6, d7, c2, 0, 0, 0); }
      if (OP == 4) {   // the moments loop's mix: 6 cvt + 3 add + 3 mfma
        asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3" : "=v"(d4), "=v"(d5) : "v"(f0), "v"(f1));
        asm volatile("v_add_f64 %0, %0, %3\n v_add_f64 %1, %1, %3\n v_add_f64 %2, %2, %3" : "+v"(d6), "+v"(d7), "+v"(d3) : "v"(d0));
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d4, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(d5, d1, c2, 0, 0, 0);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3 + d6 + d7 + c0[0] + c1[1] + c2[2];
}
template <int OP> void run(const char* name, double* out, int n_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wpc : {4, 8}) {       // waves per CU: 1 or 2 per SIMD
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256 * wpc / 4), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wpc / 4 waves, each iters * 8 * n_per_iter instructions
    printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction per SIMD (at 2.1 GHz)\n", name, wpc / 4, ms * 1e-3 * 2.1e9 / ((double)iters * 8 * n_per_iter * (wpc / 4)));
  }
}
int main() {
  double* out; hipMalloc(&out, 256 * 8 * 256 * 8);
  run<0>("v_cvt_f64_f32", out, 4); run<1>("v_add_f64", out, 4); run<2>("v_fma_f64", out, 4); run<3>("v_mfma_f64_16x16x4", out, 3);
  run<4>("moments mix (6 cvt + 3 add + 3 mfma)", out, 12);
  return 0;
}
