// Small helper kernels: layout changes at the API boundary and weight preparation for the fused apply.
#include "wct_common.h"

namespace {

// NHWC [npix][C] <-> NCHW [C][npix] through a 32x33 LDS tile (both sides coalesced)
__global__ void transpose_kernel(const float* in, float* out, int rows, int cols) {
  // in: [rows][cols] -> out: [cols][rows]
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const long r = (long)by + j, c = bx + tx;
    if (r < rows && c < cols) tile[j][tx] = in[r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long r = (long)bx + j, c = by + tx;  // out row = in col
    if (r < cols && c < rows) out[r * rows + c] = tile[tx][j];
  }
}

// W'[o][i][t] = sum_c W[o][c][t] M[c][i]  written straight into the conv kernel's packed layout
// [chunk][tap][kq][cout_pad][4]  (i = chunk*16 + kq*4 + r); accumulated in fp64.
__global__ void fold_affine_kernel(const float* w, int cout, int cin, int cout_pad, const double* M, float* wpk) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = (cin + 15) / 16;
  const long total = (long)chunks * 36 * cout_pad * 4;
  if (e >= total) return;
  const int r = (int)(e & 3);
  long t = e >> 2;
  const int o = (int)(t % cout_pad); t /= cout_pad;
  const int kq = (int)(t & 3); t >>= 2;
  const int tap = (int)(t % 9);
  const int chunk = (int)(t / 9);
  const int i = chunk * 16 + kq * 4 + r;
  double s = 0.;
  if (o < cout && i < cin) {
    for (int c = 0; c < cin; ++c) s += (double)w[((size_t)o * cin + c) * 9 + tap] * M[(size_t)c * cin + i];
  }
  wpk[e] = (float)s;
}

__global__ void fold_bias_kernel(const float* w, const float* bias, int cout, int cin, int cout_pad, const double* b, float* bias_out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= cout_pad) return;
  double s = 0.;
  if (o < cout) {
    s = bias[o];
    for (int c = 0; c < cin; ++c) {
      double ws = 0.;
      for (int t = 0; t < 9; ++t) ws += (double)w[((size_t)o * cin + c) * 9 + t];
      s += ws * b[c];
    }
  }
  bias_out[o] = (float)s;
}

// 1x1 affine as a centre-tap-only 3x3: used by wct_apply / wct_transform (the un-fused drop-in surface)
__global__ void pack_center_kernel(const double* M, const double* b, int C, int cout_pad, float* wpk, float* bias_out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = (C + 15) / 16;
  const long total = (long)chunks * 36 * cout_pad * 4;
  if (e < cout_pad) bias_out[e] = e < C ? (float)b[e] : 0.f;
  if (e >= total) return;
  const int r = (int)(e & 3);
  long t = e >> 2;
  const int o = (int)(t % cout_pad); t /= cout_pad;
  const int kq = (int)(t & 3); t >>= 2;
  const int tap = (int)(t % 9);
  const int chunk = (int)(t / 9);
  const int i = chunk * 16 + kq * 4 + r;
  wpk[e] = (tap == 4 && o < C && i < C) ? (float)M[(size_t)o * C + i] : 0.f;
}

}  // namespace

hipError_t launch_nhwc_to_nchw(const float* in, float* out, int C, int npix, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (npix + 31) / 32), dim3(256), 0, s, in, out, npix, C);
  return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int C, int npix, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((npix + 31) / 32, (C + 31) / 32), dim3(256), 0, s, in, out, C, npix);
  return hipGetLastError();
}

hipError_t launch_fold_affine(const float* w, const float* bias, int cout, int cin, int cout_pad, const double* M,
                              const double* b, float* wpk_out, float* bias_out, hipStream_t s) {
  const int chunks = (cin + 15) / 16;
  const long total = (long)chunks * 36 * cout_pad * 4;
  hipLaunchKernelGGL(fold_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, cout, cin, cout_pad, M, wpk_out);
  hipLaunchKernelGGL(fold_bias_kernel, dim3((cout_pad + 63) / 64), dim3(64), 0, s, w, bias, cout, cin, cout_pad, b, bias_out);
  return hipGetLastError();
}

hipError_t launch_pack_center_tap(const double* M, const double* b, int C, int cout_pad, float* wpk_out, float* bias_out, hipStream_t s) {
  const int chunks = (C + 15) / 16;
  const long total = (long)chunks * 36 * cout_pad * 4;
  hipLaunchKernelGGL(pack_center_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, M, b, C, cout_pad, wpk_out, bias_out);
  return hipGetLastError();
}
