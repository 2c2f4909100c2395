// Small helper kernels: layout changes at the API boundary and weight preparation for the fused apply.
#include "wct_common.h"
#include <type_traits>
#include <algorithm>

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

// Fold csF = M x + b into the decoder's first conv, one workgroup per output channel o (fp64 accumulate):
//   W'[o][i][t] = sum_c W[o][c][t] M[c][i]  -> the conv kernel's packed layout [chunk][tap][kq][cout_pad][4]
//   b'[o]       = bias[o] + sum_c (sum_t W[o][c][t]) b[c]
//   maxbits     = max |W'| as float bits (atomicMax), the input of the split-f16 scale (split_pack_kernel)
// The weight row W[o][:][:] (cin x 9 floats) is staged in LDS; the M reads are coalesced over i.
__global__ __launch_bounds__(256) void fold_row_kernel(const float* w, const float* bias, int cout, int cin, int cout_pad,
                                                         const double* M, const double* b, float* wpk, float* bias_out, unsigned* maxbits) {
  extern __shared__ float wrow[];          // [cin * 9]
  __shared__ double red[256];
  const int o = blockIdx.x, tid = threadIdx.x;
  const int chunks = (cin + 15) / 16, ipad = chunks * 16;
  const bool live = o < cout;
  if (live) for (int e = tid; e < cin * 9; e += 256) wrow[e] = w[(size_t)o * cin * 9 + e];
  __syncthreads();
  float mx = 0.f;
  if (ipad <= 256 && (256 % ipad) == 0) {
    // a thread owns ONE input column i and the taps tg, tg + G, tg + 2G, ... (G = 256 / ipad column groups): one pass over c
    // with NT = ceil(9 / G) independent accumulators and one M element per step, instead of 9 * ipad / 256 dependent passes.
    // Every (tap, i) still sums over c in the same order: bit-identical results.
    const int G = 256 / ipad, i = tid % ipad, tg = tid / ipad;
    auto run = [&](auto nt_tag) {
      constexpr int NT = decltype(nt_tag)::value;
      double s[NT];
      int tp[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) { s[k] = 0.; const int t = tg + k * G; tp[k] = t < 9 ? t : 8; }   // clamped: a surplus slot repeats tap 8
      if (live && i < cin)
        for (int c = 0; c < cin; ++c) {
          const double m = M[(size_t)c * cin + i];
#pragma unroll
          for (int k = 0; k < NT; ++k) s[k] += (double)wrow[c * 9 + tp[k]] * m;
        }
      const int chunk = i >> 4, kq = (i >> 2) & 3, r = i & 3;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int tap = tg + k * G;
        if (tap < 9) {
          const float v = (float)s[k];
          mx = fmaxf(mx, fabsf(v));
          wpk[((((size_t)chunk * 9 + tap) * 4 + kq) * cout_pad + o) * 4 + r] = v;
        }
      }
    };
    if (G == 1) run(std::integral_constant<int, 9>{});
    else if (G == 2) run(std::integral_constant<int, 5>{});
    else if (G == 4) run(std::integral_constant<int, 3>{});
    else if (G == 8) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
  } else {
    for (int e = tid; e < 9 * ipad; e += 256) {
      const int tap = e / ipad, i = e - tap * ipad;
      double s = 0.;
      if (live && i < cin)
        for (int c = 0; c < cin; ++c) s += (double)wrow[c * 9 + tap] * M[(size_t)c * cin + i];
      const float v = (float)s;
      mx = fmaxf(mx, fabsf(v));
      const int chunk = i >> 4, kq = (i >> 2) & 3, r = i & 3;
      wpk[((((size_t)chunk * 9 + tap) * 4 + kq) * cout_pad + o) * 4 + r] = v;
    }
  }
  double part = 0.;
  if (live)
    for (int c = tid; c < cin; c += 256) {
      double ws = 0.;
      for (int t = 0; t < 9; ++t) ws += (double)wrow[c * 9 + t];
      part += ws * b[c];
    }
  red[tid] = part;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
  if (tid == 0) bias_out[o] = live ? (float)((double)bias[o] + red[0]) : 0.f;
  for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
  if (maxbits && (tid & 63) == 0) atomicMax(maxbits, __float_as_uint(mx));
}

// cin >= 256 (--mode original): the same fold with OB output channels per workgroup and one input column i per thread.  The
// row-per-workgroup kernel above makes every one of its 512 workgroups stream all of M (2 MB at C = 512) through a
// dependent 512-step loop -- 0.58 ms per launch; here an M element is loaded once per OB * 9 products, the weight rows sit in
// LDS and are read wave-uniformly.  Same summation order over c for every (o, tap, i): bit-identical results.
template <int OB>
__global__ __launch_bounds__(256) void fold_block_kernel(const float* w, const float* bias, int cout, int cin, int cout_pad,
                                                           const double* M, const double* b, float* wpk, float* bias_out, unsigned* maxbits) {
  extern __shared__ float wrow[];          // [OB][cin * 9]
  __shared__ double red[256];
  const int o0 = blockIdx.x * OB, tid = threadIdx.x, i = blockIdx.y * 256 + tid;
  const int chunks = (cin + 15) / 16, ipad = chunks * 16, row = cin * 9;
  for (int e = tid; e < OB * row; e += 256) {
    const int o = o0 + e / row;
    wrow[e] = o < cout ? w[(size_t)o * row + (e - (e / row) * row)] : 0.f;
  }
  __syncthreads();
  double s[OB][9];
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t) s[o][t] = 0.;
  if (i < cin) {
    for (int c = 0; c < cin; ++c) {
      const double m = M[(size_t)c * cin + i];
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) s[o][t] += (double)wrow[o * row + c * 9 + t] * m;
    }
  }
  float mx = 0.f;
  if (i < ipad) {
    const int chunk = i >> 4, kq = (i >> 2) & 3, r = i & 3;
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      if (o0 + o >= cout_pad) break;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = (float)s[o][t];
        mx = fmaxf(mx, fabsf(v));
        wpk[((((size_t)chunk * 9 + t) * 4 + kq) * cout_pad + o0 + o) * 4 + r] = v;
      }
    }
  }
  if (blockIdx.y == 0) {   // b'[o] = bias[o] + sum_c (sum_t W[o][c][t]) b[c]
    for (int o = 0; o < OB && o0 + o < cout_pad; ++o) {
      double part = 0.;
      for (int c = tid; c < cin; c += 256) {
        double ws = 0.;
        for (int t = 0; t < 9; ++t) ws += (double)wrow[o * row + c * 9 + t];
        part += ws * b[c];
      }
      __syncthreads();
      red[tid] = part;
      __syncthreads();
      for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
      if (tid == 0) bias_out[o0 + o] = (o0 + o < cout) ? (float)((double)bias[o0 + o] + red[0]) : 0.f;
    }
  }
  for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
  if (maxbits && (tid & 63) == 0) atomicMax(maxbits, __float_as_uint(mx));
}

// ---- the fold WITHOUT the C x C products on the content side's critical path (16x: cin <= 128).
// csF = M cF + b with M = alpha Ss Wc + (1 - alpha) I (Ss = cov_s^(1/2): style side; Wc = cov_c^(-1/2): content side), so
//   W'[o][i][t] = alpha SUM_k Ws[(o,t)][k] Wc[k][i] + (1 - alpha) W[o][i][t],     Ws[(o,t)][k] = SUM_c W[o][c][t] Ss[c][k]
//   b'[o]       = bias[o] + alpha (Wmus[o] - SUM_k WsumS[o][k] v[k]),             v = Wc mu_c,  WsumS = SUM_t Ws,  Wmus = Wsum mu_s
// Ws / WsumS / Wmus depend on the style only: fold_style_kernel forms them once per style on the SIDE stream, and the content
// side goes from Wc straight to the folded weights in ONE launch (fold_fast_kernel) -- no T = Ss Wc, no M, no b: launch_assemble
// (9 us + a launch gap per level) leaves the critical path.  Everything in fp64 like the M-based fold; the association differs
// ((W Ss) Wc instead of W (Ss Wc)), i.e. fp64 round-off, then the same rounding to fp32.
// One workgroup per output channel o; a thread owns input column i and the taps tg, tg + G, ... (fold_row_kernel's scheme).
// FOLD_NT threads = 256 column/tap owners x FOLD_CS slices of the contraction index: a thread's dependent chain is cin / FOLD_CS
// steps (32 at cin = 128) instead of cin, the slices are summed through LDS in a fixed order.  NROW = 9 taps (+ 1: a tenth
// left-hand row riding in the same pass, the bias row of fold_fast_kernel).
constexpr int FOLD_CS = 4, FOLD_NT = 256 * FOLD_CS;

template <int NROW, typename OUT>
__device__ __forceinline__ void fold_columns(const double* lhs, int lhs_row_stride, int lhs_c_stride, const double* rhs, int cin, int ipad, bool live,
                                             double* part /* LDS [FOLD_CS - 1][16 * ipad / ... see below] */, OUT&& emit) {
  const int tid = threadIdx.x & 255, cs = threadIdx.x >> 8;
  const int G = 256 / ipad, i = tid % ipad, tg = tid / ipad;
  const int c0 = (cin * cs) / FOLD_CS, c1 = (cin * (cs + 1)) / FOLD_CS;
  auto run = [&](auto nt_tag) {
    constexpr int NT = decltype(nt_tag)::value;
    double s[NT];
    int tp[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { s[k] = 0.; const int t = tg + k * G; tp[k] = t < NROW ? t : NROW - 1; }   // clamped: a surplus slot repeats the last row
    if (live && i < cin)
#pragma unroll 4
      for (int c = c0; c < c1; ++c) {
        const double m = rhs[(size_t)c * cin + i];
#pragma unroll
        for (int k = 0; k < NT; ++k) s[k] += lhs[tp[k] * lhs_row_stride + c * lhs_c_stride] * m;
      }
    // slices 1.. park their partial sums in LDS ([slice - 1][k][256 owners]); slice 0 adds them in slice order
    if (cs > 0) {
#pragma unroll
      for (int k = 0; k < NT; ++k) part[((cs - 1) * NT + k) * 256 + tid] = s[k];
    }
    __syncthreads();
    if (cs == 0) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
#pragma unroll
        for (int q = 1; q < FOLD_CS; ++q) s[k] += part[((q - 1) * NT + k) * 256 + tid];
        const int row = tg + k * G;
        if (row < NROW) emit(row, i, s[k]);
      }
    }
  };
  constexpr int R = NROW;
  if (G == 1) run(std::integral_constant<int, R>{});
  else if (G == 2) run(std::integral_constant<int, (R + 1) / 2>{});
  else if (G == 4) run(std::integral_constant<int, (R + 3) / 4>{});
  else if (G == 8) run(std::integral_constant<int, (R + 7) / 8>{});
  else run(std::integral_constant<int, 1>{});
}
// cin <= 128 (fold_fast_capable) means ipad <= 128, i.e. G >= 2 column groups and NT <= (nrow + 1) / 2 rows per thread
__host__ __device__ inline size_t fold_part_doubles(int nrow) { return (size_t)(FOLD_CS - 1) * ((nrow + 1) / 2) * 256; }

// style side: Ws [cout][9][cin] | WsumS [cout][cin] | Wmus [cout]   (doubles, one buffer)
__global__ __launch_bounds__(FOLD_NT) void fold_style_kernel(const float* w, int cout, int cin, const double* Ss, const double* mu_s, double* Ws,
                                                               double* WsumS, double* Wmus) {
  extern __shared__ double fsm[];          // wrow [cin * 9] (as doubles, [c][t]) | ws [9 * ipad] | part
  __shared__ double red[256];
  const int o = blockIdx.x, tid = threadIdx.x;
  const int ipad = (cin + 15) / 16 * 16;
  double* wrow = fsm;
  double* ws = fsm + cin * 9;
  double* part = ws + 9 * ipad;
  for (int e = tid; e < cin * 9; e += FOLD_NT) wrow[e] = (double)w[(size_t)o * cin * 9 + e];
  for (int e = tid; e < 9 * ipad; e += FOLD_NT) ws[e] = 0.;
  __syncthreads();
  fold_columns<9>(wrow, 1, 9, Ss, cin, ipad, true, part, [&](int tap, int i, double v) { ws[tap * ipad + i] = v; });
  __syncthreads();
  for (int e = tid; e < 9 * cin; e += FOLD_NT) { const int tap = e / cin, i = e - tap * cin; Ws[((size_t)o * 9 + tap) * cin + i] = ws[tap * ipad + i]; }
  for (int i = tid; i < cin; i += FOLD_NT) {
    double a = 0.;
    for (int t = 0; t < 9; ++t) a += ws[t * ipad + i];
    WsumS[(size_t)o * cin + i] = a;
  }
  double p = 0.;
  if (tid < 256)
    for (int c = tid; c < cin; c += 256) {
      double a = 0.;
      for (int t = 0; t < 9; ++t) a += wrow[c * 9 + t];
      p += a * mu_s[c];
    }
  if (tid < 256) red[tid] = p;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
  if (tid == 0) Wmus[o] = red[0];
}

// content side: Wc (= F of the content's EigResult), mu_c -> packed fp32 weights + bias + per-row max |W'| (float bits; rows >=
// cout are zero).  rowmax replaces the atomicMax on one word (and the memset in front of it): launch_split_pack reduces it.
// The bias rides in the same pass as a TENTH left-hand row: (WsumS[o][:] Wc)[i], then a dot product with mu_c over i.
__global__ __launch_bounds__(FOLD_NT) void fold_fast_kernel(const float* w, const float* bias, int cout, int cin, int cout_pad, const double* Ws,
                                                              const double* WsumS, const double* Wmus, const double* Wc, const double* mu_c, double alpha,
                                                              float* wpk, float* bias_out, unsigned* rowmax) {
  extern __shared__ double fsm[];          // lhs [10 * cin] ([t][k]; row 9 = WsumS[o][:]) | part
  __shared__ double red[256];
  __shared__ float redf[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const int ipad = (cin + 15) / 16 * 16;
  const bool live = o < cout;
  double* lhs = fsm;
  double* part = fsm + 10 * cin;
  if (live) {
    for (int e = tid; e < 9 * cin; e += FOLD_NT) lhs[e] = Ws[(size_t)o * 9 * cin + e];
    for (int e = tid; e < cin; e += FOLD_NT) lhs[9 * cin + e] = WsumS[(size_t)o * cin + e];
  }
  if (tid < 256) red[tid] = 0.;
  __syncthreads();
  float mx = 0.f;
  fold_columns<10>(lhs, cin, 1, Wc, cin, ipad, live, part, [&](int row, int i, double sacc) {
    if (row == 9) { red[i] = (live && i < cin) ? sacc * mu_c[i] : 0.; return; }      // ipad <= 256 columns: one slot each
    double x = alpha * sacc;
    if (live && i < cin && alpha != 1.0) x += (1.0 - alpha) * (double)w[((size_t)o * cin + i) * 9 + row];
    const float f = (float)x;
    mx = fmaxf(mx, fabsf(f));
    const int chunk = i >> 4, kq = (i >> 2) & 3, r = i & 3;
    wpk[((((size_t)chunk * 9 + row) * 4 + kq) * cout_pad + o) * 4 + r] = f;
  });
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
  if (tid == 0) bias_out[o] = live ? (float)((double)bias[o] + alpha * (Wmus[o] - red[0])) : 0.f;
  for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
  if (tid < 256 && (tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) rowmax[o] = __float_as_uint(fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3])));
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

// ---- image edge: the reference's ToTensor (data_loader.py:57-58: uint8 HWC -> float CHW / 255) and
// save_image (WCT.py:128 -> torchvision 0.2.1: mul(255).clamp(0,255).byte(), i.e. truncation) on the device, so that
// a frame crosses PCIe as 3 B/px instead of 12.  4 pixels per thread: 12 contiguous bytes in, three 16-byte stores out.
__global__ void u8_to_planar_kernel(const uint8_t* in, long npix, float* out) {
  const long p0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (p0 >= npix) return;
  if (p0 + 4 <= npix) {
    const uint8_t* src = in + p0 * 3;   // 4-byte aligned: p0 is a multiple of 4
    const unsigned w0 = *reinterpret_cast<const unsigned*>(src), w1 = *reinterpret_cast<const unsigned*>(src + 4),
                   w2 = *reinterpret_cast<const unsigned*>(src + 8);
    const unsigned b[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u,
                            (w1 >> 16) & 255u, w1 >> 24, w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)b[k * 3 + c] / 255.0f;   // IEEE division, as torch's .div(255)
      float* dst = out + (size_t)c * npix + p0;
      if ((reinterpret_cast<size_t>(dst) & 15) == 0) *reinterpret_cast<f32x4*>(dst) = v;
      else { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    }
  } else {
    for (long p = p0; p < npix; ++p)
      for (int c = 0; c < 3; ++c) out[(size_t)c * npix + p] = (float)in[p * 3 + c] / 255.0f;
  }
}

// round_mode 0: truncation (torchvision 0.2.1, the reference's pin); 1: add 0.5 first (torchvision >= 0.4)
__global__ void planar_to_u8_kernel(const float* in, long npix, uint8_t* out, int round_mode) {
  const long p0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (p0 >= npix) return;
  const int n = npix - p0 < 4 ? (int)(npix - p0) : 4;
  unsigned b[12];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = k < n ? in[(size_t)c * npix + p0 + k] * 255.0f : 0.f;
      if (round_mode) x += 0.5f;
      x = fminf(fmaxf(x, 0.f), 255.f);     // NaN -> 0 (fmaxf), like clamp on the host
      b[k * 3 + c] = (unsigned)x;          // truncation toward zero
    }
  if (n == 4) {
    unsigned* dst = reinterpret_cast<unsigned*>(out + p0 * 3);
    dst[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    dst[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    dst[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
  } else {
    for (int k = 0; k < n * 3; ++k) out[p0 * 3 + k] = (uint8_t)b[k];
  }
}

}  // namespace

hipError_t launch_u8_to_planar(const uint8_t* in, long npix, float* out, hipStream_t s) {
  if ((reinterpret_cast<size_t>(in) & 3) != 0) return hipErrorInvalidValue;
  const long nt = (npix + 3) / 4;
  hipLaunchKernelGGL(u8_to_planar_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, in, npix, out);
  return hipGetLastError();
}

hipError_t launch_planar_to_u8(const float* in, long npix, uint8_t* out, int round_mode, hipStream_t s) {
  if ((reinterpret_cast<size_t>(out) & 3) != 0) return hipErrorInvalidValue;
  const long nt = (npix + 3) / 4;
  hipLaunchKernelGGL(planar_to_u8_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, in, npix, out, round_mode);
  return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const float* in, float* out, int C, int npix, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (npix + 31) / 32), dim3(256), 0, s, in, out, npix, C);
  return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int C, int npix, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((npix + 31) / 32, (C + 31) / 32), dim3(256), 0, s, in, out, C, npix);
  return hipGetLastError();
}

hipError_t launch_fold_affine(const float* w, const float* bias, int cout, int cin, int cout_pad, const double* M,
                              const double* b, float* wpk_out, float* bias_out, unsigned* maxbits_dev, hipStream_t s) {
  if (maxbits_dev) {
    hipError_t e = hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
  }
  if (cin >= 256) {
    constexpr int OB = 4;
    const size_t lds = (size_t)OB * cin * 9 * sizeof(float);   // 73.7 KB at cin = 512
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fold_block_kernel<OB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int ipad = (cin + 15) / 16 * 16;
    hipLaunchKernelGGL(fold_block_kernel<OB>, dim3((unsigned)((cout_pad + OB - 1) / OB), (unsigned)((ipad + 255) / 256)), dim3(256), lds, s, w, bias,
                       cout, cin, cout_pad, M, b, wpk_out, bias_out, maxbits_dev);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(fold_row_kernel, dim3((unsigned)cout_pad), dim3(256), (size_t)cin * 9 * sizeof(float), s, w, bias, cout, cin, cout_pad, M, b,
                     wpk_out, bias_out, maxbits_dev);
  return hipGetLastError();
}

bool fold_fast_capable(int cin) {
  const int ipad = (cin + 15) / 16 * 16;
  return cin <= 128 && ipad <= 128 && (256 % ipad) == 0;
}
size_t fold_style_doubles(int cout, int cin) { return (size_t)cout * 9 * cin + (size_t)cout * cin + (size_t)cout; }

hipError_t launch_fold_style(const float* w, int cout, int cin, const double* Ss, const double* mu_s, double* buf, hipStream_t s) {
  if (!fold_fast_capable(cin)) return hipErrorInvalidValue;
  const int ipad = (cin + 15) / 16 * 16;
  double* Ws = buf;
  double* WsumS = Ws + (size_t)cout * 9 * cin;
  double* Wmus = WsumS + (size_t)cout * cin;
  hipLaunchKernelGGL(fold_style_kernel, dim3((unsigned)cout), dim3(FOLD_NT), (size_t)(cin * 9 + 9 * ipad + fold_part_doubles(9)) * sizeof(double), s, w,
                     cout, cin, Ss, mu_s, Ws, WsumS, Wmus);
  return hipGetLastError();
}

hipError_t launch_fold_fast(const float* w, const float* bias, int cout, int cin, int cout_pad, const double* style_buf, const double* Wc,
                            const double* mu_c, double alpha, float* wpk_out, float* bias_out, unsigned* rowmax_dev, hipStream_t s) {
  if (!fold_fast_capable(cin)) return hipErrorInvalidValue;
  const double* Ws = style_buf;
  const double* WsumS = Ws + (size_t)cout * 9 * cin;
  const double* Wmus = WsumS + (size_t)cout * cin;
  hipLaunchKernelGGL(fold_fast_kernel, dim3((unsigned)cout_pad), dim3(FOLD_NT), (size_t)(10 * cin + fold_part_doubles(10)) * sizeof(double), s, w, bias,
                     cout, cin, cout_pad, Ws, WsumS, Wmus, Wc, mu_c, alpha, wpk_out, bias_out, rowmax_dev);
  return hipGetLastError();
}

hipError_t launch_pack_center_tap(const double* M, const double* b, int C, int cout_pad, float* wpk_out, float* bias_out, hipStream_t s) {
  const int chunks = (C + 15) / 16;
  const long total = (long)chunks * 36 * cout_pad * 4;
  hipLaunchKernelGGL(pack_center_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, M, b, C, cout_pad, wpk_out, bias_out);
  return hipGetLastError();
}

namespace {
__global__ void counter_to_f64_kernel(const unsigned* counter, double* dst) { *dst = (double)*counter; }
}  // namespace

hipError_t launch_counter_to_f64(const unsigned* counter, double* dst, hipStream_t s) {
  hipLaunchKernelGGL(counter_to_f64_kernel, dim3(1), dim3(1), 0, s, counter, dst);
  return hipGetLastError();
}

// ---- pitched block copy (wct_api.hip wct_stylize_sharded: level crops, halo packing and assembly of planar images; 3 planes of H rows
//      are 3H rows of one pitch).  One float per thread and grid-stride: the blocks are 2 ... 1600 columns wide, HBM-bound, a few us.
namespace {
__global__ __launch_bounds__(256) void copy_block_kernel(const float* __restrict__ src, long src_pitch, float* __restrict__ dst, long dst_pitch,
                                                         long rows, int width) {
  const long total = rows * width;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / width;
    const int c = (int)(i - r * width);
    dst[r * dst_pitch + c] = src[r * src_pitch + c];
  }
}
}  // namespace

// up to three such blocks of the same height in ONE launch (the neighbour exchange: both edge blocks packed together; received left margin | owned
// columns | received right margin assembled together): the blocks are a few columns wide, so the launches, not the bytes, are what they cost
namespace {
__global__ __launch_bounds__(256) void copy_blocks_kernel(CopySegs g, long rows) {
  const long w0 = g.width[0], w1 = g.width[1], w2 = g.width[2], wsum = w0 + w1 + w2, total = rows * wsum;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / wsum;
    long c = i - r * wsum;
    const int k = c < w0 ? 0 : (c < w0 + w1 ? 1 : 2);
    c -= k == 0 ? 0 : (k == 1 ? w0 : w0 + w1);
    g.dst[k][r * g.dst_pitch[k] + c] = g.src[k][r * g.src_pitch[k] + c];
  }
}
}  // namespace

hipError_t launch_copy_blocks(const CopySegs& g, long rows, hipStream_t s) {
  const long wsum = (long)g.width[0] + g.width[1] + g.width[2];
  if (rows <= 0 || wsum <= 0) return hipSuccess;
  const long total = rows * wsum;
  const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 256L * 64);
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(blocks), dim3(256), 0, s, g, rows);
  return hipGetLastError();
}

hipError_t launch_copy_block(const float* src, long src_pitch, float* dst, long dst_pitch, long rows, int width, hipStream_t s) {
  if (rows <= 0 || width <= 0) return hipSuccess;
  const long total = rows * width;
  const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 256L * 64);
  hipLaunchKernelGGL(copy_block_kernel, dim3(blocks), dim3(256), 0, s, src, src_pitch, dst, dst_pitch, rows, width);
  return hipGetLastError();
}
