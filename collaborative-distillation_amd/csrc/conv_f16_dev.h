// Device-side helpers shared by the split-f16 convolution kernels (conv3x3_f16.hip, conv3x3_sp.hip).  gfx950 only.
#pragma once
#include "wct_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// native 16-byte vector for register staging: HIP's u32x4 is a struct of unions, arrays of which are not promoted to
// registers (they land in scratch)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FTW = 32;
constexpr int FHW = FTW + 2;                  // halo tile is 34 x (TH + 2), TH = 8 (4 waves) or 16 (8 waves)
constexpr int nph(int TH) { return FHW * (TH + 2); }                 // 340 / 612 halo pixels
constexpr int npp(int TH) { return (nph(TH) + 15) / 16 * 16; }       // plane stride in 16-B units: 352 / 624

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int reflect_clamp(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = j < 4 ? a[j] : b[j - 4];
    x = fminf(fmaxf(x, -65504.f), 65504.f);
    const _Float16 h = (_Float16)x;
    hi[j] = h;
    lo[j] = (_Float16)(x - (float)h);
  }
}

// ---- "SP16" activation format (intermediate activations of the f16x3 path):
// per pixel, per group of 8 channels: [8 hi halfs][8 lo halfs] (32 B), x = hi + lo; C * 4 bytes per pixel like fp32.
// The split (with its +-65504 clamp) is done ONCE by the producing kernel's epilogue instead of by every consumer on
// every halo re-read, and consumers move the 16-byte groups global -> LDS without touching them
// (global_load_lds_dwordx4: no VGPR staging, no VALU).  hi / lo are exactly what split8() computes from the fp32 value.

// 4 consecutive channels -> (hi, lo) as 2 + 2 dwords
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
  f16x4 h, l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float x = fminf(fmaxf(v[r], -65504.f), 65504.f);
    h[r] = (_Float16)x;
    l[r] = (_Float16)(x - (float)h[r]);
  }
  hi = __builtin_bit_cast(u32x2, h);
  lo = __builtin_bit_cast(u32x2, l);
}

// 32x32 MFMA accumulator layout: lanes l and l + 32 hold channels 8q + {0..3} and 8q + {4..7} of the SAME pixel.
// v_permlane32_swap hands the low lane both hi halves and the high lane both lo halves, so each lane writes one
// 16-byte group: returns the 16 bytes this lane stores at  record + group * 32 + (lane >> 5) * 16.
// Must be executed by all 64 lanes (no divergence).
__device__ __forceinline__ u32x4 sp16_pair_exchange(const f32x4& v) {
  u32x2 hi, lo;
  split4(v, hi, lo);
  const auto ra = __builtin_amdgcn_permlane32_swap(hi[0], lo[0], false, false);
  const auto rb = __builtin_amdgcn_permlane32_swap(hi[1], lo[1], false, false);
  return u32x4{ra[0], rb[0], ra[1], rb[1]};
}

// 16x16 MFMA accumulator layout: lane (pixel, kq) holds channels 4 kq .. 4 kq + 3 -> two 8-byte stores into the
// pixel's record (group kq >> 1, half kq & 1)
__device__ __forceinline__ void sp16_store4(char* record, int kq, const f32x4& v) {
  u32x2 hi, lo;
  split4(v, hi, lo);
  char* g = record + (kq >> 1) * 32 + (kq & 1) * 8;
  *reinterpret_cast<u32x2*>(g) = hi;
  *reinterpret_cast<u32x2*>(g + 16) = lo;
}

inline int num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    return v;
  }();
  return n;
}

}  // namespace
