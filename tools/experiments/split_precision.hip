// Experiment: accuracy of split-precision MFMA dot products (K = 1152, the 128->128 3x3 conv) against fp64,
// and whether f16 MFMA preserves subnormal operands.  One wave computes a 32x32 tile straight from global memory.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int M = 32, N = 32, K = 1152;

__device__ inline unsigned short bf16_rn(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// mode 0: fp32 mfma 32x32x2 ; 1: f16 x3 ; 2: f16 x4 ; 3: bf16 x3 ; 4: bf16 x6 ; 5: f16 x1
__global__ void tile(const float* A, const float* B, float* D, int mode, float sa, float sb) {
  const int l = threadIdx.x, i = l & 31, kh = l >> 5;
  f32x16 acc = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + kh], B[(k + kh) * N + i], acc, 0, 0, 0);
  } else if (mode == 1 || mode == 2 || mode == 5) {
    for (int k = 0; k < K; k += 16) {
      f16x8 ah, al, bh, bl;
      for (int j = 0; j < 8; ++j) {
        float a = A[i * K + k + kh * 8 + j] * sa, b = B[(k + kh * 8 + j) * N + i] * sb;
        _Float16 h = (_Float16)a; ah[j] = h; al[j] = (_Float16)(a - (float)h);
        h = (_Float16)b; bh[j] = h; bl[j] = (_Float16)(b - (float)h);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      if (mode != 5) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      }
      if (mode == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      s16x8 a1, a2, a3, b1, b2, b3;
      for (int j = 0; j < 8; ++j) {
        float a = A[i * K + k + kh * 8 + j], b = B[(k + kh * 8 + j) * N + i];
        unsigned short h = bf16_rn(a); a1[j] = h; float r = a - bf16_f(h); h = bf16_rn(r); a2[j] = h; r -= bf16_f(h); a3[j] = bf16_rn(r);
        h = bf16_rn(b); b1[j] = h; r = b - bf16_f(h); h = bf16_rn(r); b2[j] = h; r -= bf16_f(h); b3[j] = bf16_rn(r);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a1), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b1), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a1), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b2), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a2), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b1), acc, 0, 0, 0);
      if (mode == 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a1), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b3), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a3), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a2), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b2), acc, 0, 0, 0);
      }
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    D[row * N + i] = acc[r] / ((mode == 1 || mode == 2 || mode == 5) ? sa * sb : 1.f);
  }
}

int main() {
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(M * K), B(K * N), D(M * N);
  for (auto& v : A) v = 0.05f * nd(rng);                                   // weights
  for (auto& v : B) { float x = nd(rng) * 40.f + 10.f; v = x > 0 ? x : 0; if (rng() % 97 == 0) v *= 30.f; }  // post-ReLU activations with outliers
  std::vector<double> ref(M * N, 0.), mag(M * N, 0.);
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0, m = 0; for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * N + j]; m += fabs((double)A[i * K + k] * B[k * N + j]); } ref[i * N + j] = s; mag[i * N + j] = m; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  const char* names[] = {"fp32 mfma", "f16 x3", "f16 x4", "bf16 x3", "bf16 x6", "f16 x1"};
  struct { int mode; float sa, sb; } runs[] = {{0, 1, 1}, {1, 1, 1}, {2, 1, 1}, {3, 1, 1}, {4, 1, 1}, {5, 1, 1},
      {1, 16.f, 1.f / 4}, {1, 1.f / 64, 1.f / 1024}, {1, 1.f / 1024, 1.f / 65536}, {1, 1.f, 1.f / (1 << 22)}, {2, 1.f, 1.f / (1 << 22)}};
  for (auto& r : runs) {
    tile<<<1, 64>>>(dA, dB, dD, r.mode, r.sa, r.sb);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double e_abs = 0, e_rel_mag = 0, rmax = 0;
    for (int i = 0; i < M * N; ++i) { e_abs = fmax(e_abs, fabs(D[i] - ref[i])); e_rel_mag = fmax(e_rel_mag, fabs(D[i] - ref[i]) / mag[i]); rmax = fmax(rmax, fabs(ref[i])); }
    printf("%-10s sa=%-10g sb=%-12g max|err|/max|ref| %.3e   max |err|/sum|a*b| %.3e\n", names[r.mode], r.sa, r.sb, e_abs / rmax, e_rel_mag);
  }
  return 0;
}
