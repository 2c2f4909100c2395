// Can VALU work of one wave overlap MFMA work of ANOTHER wave on the same SIMD (gfx950)?  The fused full-resolution ends fit an
// ADDITIVE model (VALU cycles + MFMA cycles ~ 82 % of their time); this probe runs, on every SIMD, waves that only issue
// v_fma_f32 and waves that only issue v_mfma_f32_16x16x32_f16 -- alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o valu_mfma_overlap valu_mfma_overlap.hip && ./valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// role of a wave: bit 0 = VALU loop, bit 1 = MFMA loop (both: the same wave alternates 3 VALU : 1 MFMA, independent chains)
__global__ __launch_bounds__(768) void k(float* out, int iters, int role_even, int role_odd, int nv, int nm) {
  const int wave = threadIdx.x >> 6;
  const int role = (wave & 1) ? role_odd : role_even;
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f16x8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (threadIdx.x + j)); y[j] = (_Float16)(0.02f * (threadIdx.x - j)); }
  const float s = 1.0001f, t = 0.5f;
  for (int i = 0; i < iters; ++i) {
    if (role & 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {   // nv = 8 * 8 independent-chain fmas per iteration
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t));
      }
    }
    if (role & 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // nm = 4 * 4 MFMAs per iteration, four independent accumulators
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c3, 0, 0, 0);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  float* out; hipMalloc(&out, 256 * 768 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  struct { const char* name; int even, odd; } cases[] = {
    {"VALU only  (6 of 12 waves: 64 fma / iter each)          ", 1, 0},
    {"MFMA only  (6 of 12 waves: 16 mfma / iter each)         ", 0, 2},
    {"VALU waves + MFMA waves (6 + 6, different waves)        ", 1, 2},
    {"all 12 waves VALU                                       ", 1, 1},
    {"all 12 waves MFMA                                       ", 2, 2},
    {"all 12 waves both (64 fma then 16 mfma, same wave)      ", 3, 3},
  };
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, out, iters, c.even, c.odd, 64, 16);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s %8.3f ms  = %6.1f ns per iteration\n", c.name, ms, ms * 1e6 / iters);
  }
  return 0;
}
