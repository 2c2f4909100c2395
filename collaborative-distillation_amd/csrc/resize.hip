// transforms.Resize of the reference's harness on the device.
//
// PytorchWCT/data_loader.py:52-56 resizes the decoded PIL image with torchvision 0.2.1's transforms.Resize(size), i.e.
// Image.resize((ow, oh), Image.BILINEAR).  The arithmetic lives in Pillow (requirements.txt pins Pillow==8.2.0; source not under
// /root/reference), whose resampler -- libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc -- is restated here from its published algorithm:
//   * separable, horizontal pass first, uint8 between the passes
//   * per output index: centre = (xx + 0.5) * scale, the triangle filter stretched by max(scale, 1) (antialiasing when
//     shrinking), taps [xmin, xmin + n) clipped to the image, weights normalised to sum 1 in double precision
//   * weights rounded to 22 fractional bits; a pixel = clip8((2^21 + sum_k pix * w_k) >> 22), 32-bit integer accumulation
// Bit-exact against Pillow (tests/test_resize.py: the golden set made by tools/make_goldens.py gen_g12 with Pillow 12.2.0, and Pillow
// itself where it is importable).  The O(W + H) weight tables are host work (doubles; the host compiler does not contract
// a * b + c, device code would); every per-pixel operation runs on the device.
#include "wct_common.h"
#include <cmath>
#include <vector>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// Pillow's bilinear_filter (support 1.0)
inline double triangle(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

__device__ __forceinline__ unsigned clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per output pixel (three channels); consecutive threads walk neighbouring spans of the input row
__global__ void resize_h_kernel(const uint8_t* in, int W, int row0, int rows, int oW, int ksize, const int* bounds, const int* kk,
                                uint8_t* out) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (xx >= oW || y >= rows) return;
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const uint8_t* src = in + ((size_t)(row0 + y) * W + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int w = k[x];
    s0 += (int)src[3 * x] * w;
    s1 += (int)src[3 * x + 1] * w;
    s2 += (int)src[3 * x + 2] * w;
  }
  uint8_t* dst = out + ((size_t)y * oW + xx) * 3;
  dst[0] = (uint8_t)clip8(s0); dst[1] = (uint8_t)clip8(s1); dst[2] = (uint8_t)clip8(s2);
}

// one thread per output byte: the taps of a column are `rowbytes` apart, consecutive threads read consecutive bytes.
// PLANAR: the result goes out as planar fp32 / 255 (ToTensor, data_loader.py:57) instead of uint8 HWC.
// bounds are relative to the first row of `in` (already shifted by the caller)
template <bool PLANAR>
__global__ void resize_v_kernel(const uint8_t* in, int rowbytes, int oH, int ksize, const int* bounds, const int* kk, uint8_t* out,
                                float* planar) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
  if (j >= rowbytes) return;
  const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  const uint8_t* src = in + (size_t)ymin * rowbytes + j;
  int s = 1 << (PRECISION_BITS - 1);
  for (int y = 0; y < n; ++y) s += (int)src[(size_t)y * rowbytes] * k[y];
  const unsigned v = clip8(s);
  if constexpr (PLANAR) {
    const int px = j / 3, c = j - px * 3, oW = rowbytes / 3;
    planar[((size_t)c * oH + yy) * oW + px] = (float)v / 255.0f;
  } else {
    out[(size_t)yy * rowbytes + j] = (uint8_t)v;
  }
}

}  // namespace

// Weight tables of one axis (Pillow's precompute_coeffs + normalize_coeffs_8bpc for the box [0, in_size))
void resize_axis_tables(int in_size, int out_size, int& ksize, std::vector<int>& bounds, std::vector<int>& kk) {
  const float in0 = 0.f, in1 = (float)in_size;
  double scale = (double)(in1 - in0) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> pre((size_t)ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = triangle((x + xmin - center + 0.5) * ss);
      pre[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) pre[x] /= ww;
      const double p = pre[x];
      kk[(size_t)xx * ksize + x] = p < 0 ? (int)(-0.5 + p * (1 << PRECISION_BITS)) : (int)(0.5 + p * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

// tmp: at least (rows used by the vertical pass) * oW * 3 bytes when both passes run.  Exactly one of out / planar is written.
hipError_t launch_resize_u8(const uint8_t* in, int H, int W, int oH, int oW, const int* bounds_h, const int* kk_h, int ksize_h,
                            const int* bounds_v_shifted, const int* kk_v, int ksize_v, int row0, int rows, uint8_t* tmp, uint8_t* out,
                            float* planar, hipStream_t s) {
  const bool need_h = oW != W, need_v = oH != H;
  const uint8_t* vsrc = in + (size_t)row0 * W * 3;   // input of the vertical pass when there is no horizontal one
  if (need_h) {
    uint8_t* hdst = need_v || planar ? tmp : out;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((oW + 255) / 256), (unsigned)rows), dim3(256), 0, s, in, W, row0, rows, oW, ksize_h,
                       bounds_h, kk_h, hdst);
    vsrc = hdst;
  }
  const int rowbytes = oW * 3;
  if (need_v) {
    const dim3 grid((unsigned)((rowbytes + 255) / 256), (unsigned)oH);
    if (planar) hipLaunchKernelGGL(resize_v_kernel<true>, grid, dim3(256), 0, s, vsrc, rowbytes, oH, ksize_v, bounds_v_shifted, kk_v, out, planar);
    else hipLaunchKernelGGL(resize_v_kernel<false>, grid, dim3(256), 0, s, vsrc, rowbytes, oH, ksize_v, bounds_v_shifted, kk_v, out, planar);
  } else if (planar) {
    hipError_t e = launch_u8_to_planar(vsrc, (long)oH * oW, planar, s);
    if (e != hipSuccess) return e;
  } else if (!need_h) {
    hipError_t e = hipMemcpyAsync(out, in, (size_t)H * W * 3, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}
