// Why does a Newton-Schulz stage kernel at Cp = 128 (32 workgroups, one 16x16 tile product of K = 128 per wave) take ~9 us?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int Cp = 128;
template <int V>
__global__ __launch_bounds__(256) void stage(const double* P, const double* Q, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
  const int li = lane & 15, kk = lane >> 4;
  const double* pp = P + (size_t)(i0 + li) * Cp + kk;
  const double* qq = Q + (size_t)kk * Cp + j0 + li;
  f64x4 acc = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};
  if (V == 0) {          // as in solve.hip: 8 batches of 8 loads + 4 MFMAs
    for (int k0 = 0; k0 < Cp; k0 += 16) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = pp[k0 + 4 * u]; b[u] = qq[(size_t)(k0 + 4 * u) * Cp]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
  } else if (V == 1) {   // all 64 loads first
    double a[32], b[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { a[u] = pp[4 * u]; b[u] = qq[(size_t)(4 * u) * Cp]; }
#pragma unroll
    for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  } else if (V == 2) {   // all loads first, two accumulator chains
    double a[32], b[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { a[u] = pp[4 * u]; b[u] = qq[(size_t)(4 * u) * Cp]; }
#pragma unroll
    for (int u = 0; u < 32; u += 2) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u + 1], b[u + 1], acc2, 0, 0, 0);
    }
    acc += acc2;
  } else if (V == 3) {   // loads only (no MFMA)
    double s = 0.;
#pragma unroll
    for (int u = 0; u < 32; ++u) s += pp[4 * u] * qq[(size_t)(4 * u) * Cp];
    acc[0] = s;
  } else {               // MFMA only
#pragma unroll
    for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)lane, (double)u, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(size_t)(i0 + kk + 4 * r) * Cp + j0 + li] = acc[r];
}
int main() {
  double *P, *Q, *O; hipMalloc(&P, Cp * Cp * 8); hipMalloc(&Q, Cp * Cp * 8); hipMalloc(&O, Cp * Cp * 8);
  hipMemset(P, 0, Cp * Cp * 8); hipMemset(Q, 0, Cp * Cp * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto k, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(k, dim3(4, 4), dim3(256), 0, 0, P, Q, O); hipLaunchKernelGGL(k, dim3(4, 4), dim3(256), 0, 0, O, P, Q); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-44s %.2f us per launch\n", name, ms);
    }
  };
  run(stage<0>, "V0 as shipped (8 dependent batches)");
  run(stage<1>, "V1 all operands preloaded");
  run(stage<2>, "V2 preloaded, two accumulator chains");
  run(stage<3>, "V3 loads only");
  run(stage<4>, "V4 MFMAs only");
  return 0;
}
