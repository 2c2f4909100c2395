// reflect-pad(1) + conv3x3 + bias + ReLU as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces every `self.relu(self.convXY(self.pad(y)))` of the reference's encoder/decoder stacks
// (model/model_cd.py:726-742, 277-293; model/model_original.py:492-511, 581-599), with
//   * MaxPool2d(2,2)            (model_cd.py:728,731,736,741) fused into the producer's epilogue,
//   * UpsamplingNearest2d(x2)   (model_cd.py:278,283,288,291) fused into the consumer's tile load,
//   * conv0 (1x1 colour affine, model_cd.py:725) folded into conv11's weights on the host,
//   * the WCT affine map csF = M cF + b folded into the decoder's first conv (fold_affine.hip).
//
// GEMM view: D[cout][pixel] = sum_{tap, cin} Wt[cout][tap, cin] * X[tap, cin][pixel]
//   A operand = weights (16 couts x 4 k), B operand = activations (4 k x 16 pixels),
//   v_mfma_f32_16x16x4_f32: exact fp32 (one rounding per product, fmaf chain), 64 FLOP/clk/SIMD.
// Activations are NHWC fp32 in HBM; a workgroup (4 waves) owns a 16x16 pixel tile x up to 128 couts:
//   wave w -> tile rows 4w..4w+3 (PT = 4 pixel tiles of 16), all CT cout tiles.
// K is walked in chunks of 16 input channels; per chunk the 18x18 halo tile and the 9x16xCout weight
// slab are staged in LDS:
//   ldsIn[q][pix]      float4 = channels 4q..4q+3 of halo pixel `pix`   (q = 0..3, plane stride 336)
//   ldsW [tap][q][co]  float4 = W[co][4q..4q+3][tap]
// so that one ds_read_b128 per operand feeds FOUR MFMAs (k-slot q of MFMA r is channel 4q+r for both
// operands) and both reads are bank-conflict free (16 consecutive 16-B slots per plane, plane stride a
// multiple of 256 B -- MI355X_MICROARCH "LDS" table).
#include "wct_common.h"
#include "conv_f16_dev.h"

namespace {

constexpr int TW = 16, TH = 16;          // output tile
constexpr int HW_ = TW + 2, HH_ = TH + 2; // halo tile 18 x 18
constexpr int NPIX_HALO = HW_ * HH_;     // 324
constexpr int NPIX_PAD = 336;            // plane stride (multiple of 16 float4 = 256 B)
constexpr int PT = 4;                    // pixel tiles (rows) per wave

struct ConvArgs {
  const float* in;
  float* out;
  const float* wpk;
  const float* bias;
  int H, W;          // conv spatial size
  int inH, inW;      // stored input size (H/2, W/2 when up_in)
  int cin, cout;     // logical
  int cin_chunks;
  int cout_pad;      // total padded couts in wpk
  int tiles_x, tiles_y;
  int up_in, relu;
  int out_sp;        // SP16 output (conv_f16_dev.h): the consumer is an f16x3 layer and takes the split as it is
  unsigned* sat;     // sticky saturation counter of the context (SP16 output only), may be null
};

// reflect_clamp (ReflectionPad2d(1): -1 -> 1, n -> n-2; tiles hanging over the image edge are clamped, never stored) and
// xcd_swizzle (each XCD gets a contiguous run of tiles so neighbouring halos share its L2) come from conv_f16_dev.h

template <int CT, bool IN3, bool POOL, bool OUT3>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COW = CT * 16;  // couts handled by this workgroup
  f32x4* ldsIn = reinterpret_cast<f32x4*>(smem);
  // IN3: one plane of NPIX_PAD float4 (r,g,b,0); else 4 planes
  constexpr int IN_F4 = IN3 ? NPIX_PAD : 4 * NPIX_PAD;
  f32x4* ldsW = ldsIn + IN_F4;  // IN3: floats [9][4][COW]; else float4 [9][4][COW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const int tile = xcd_swizzle(blockIdx.x, ntiles);
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int co0 = blockIdx.y * COW;
  SatTrack sat;

  f32x4 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (IN3) {
    // ---- stage the 3-channel halo tile from the planar image: float4 (c0,c1,c2,0) per pixel
    for (int e = tid; e < NPIX_HALO; e += 256) {
      const int py = e / HW_, px = e - py * HW_;
      const int gy = reflect_clamp(ty0 - 1 + py, a.H), gx = reflect_clamp(tx0 - 1 + px, a.W);
      const size_t plane = (size_t)a.H * a.W, off = (size_t)gy * a.W + gx;
      ldsIn[e] = f32x4{a.in[off], a.in[plane + off], a.in[2 * plane + off], 0.f};
    }
    float* ldsWf = reinterpret_cast<float*>(ldsW);
    for (int e = tid; e < 36 * COW; e += 256) {
      const int seg = e / COW, j = e - seg * COW;
      ldsWf[e] = a.wpk[(size_t)seg * a.cout_pad + co0 + j];
    }
    __syncthreads();
    const float* ldsInF = reinterpret_cast<const float*>(ldsIn);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      float bv[PT], av[CT];
#pragma unroll
      for (int p = 0; p < PT; ++p) bv[p] = ldsInF[((wave * PT + p + kh) * HW_ + li + kw) * 4 + kq];
#pragma unroll
      for (int c = 0; c < CT; ++c) av[c] = ldsWf[(tap * 4 + kq) * COW + c * 16 + li];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[p], acc[c][p], 0, 0, 0);
    }
  } else {
    for (int ch = 0; ch < a.cin_chunks; ++ch) {
      if (ch) __syncthreads();  // previous chunk fully consumed
      // ---- stage activations: e -> (pixel, plane q), q fastest so a pixel's 64 B are read by 4 lanes
      const int cbase = ch * 16;
      for (int e = tid; e < NPIX_HALO * 4; e += 256) {
        const int q = e & 3, pix = e >> 2;
        const int py = pix / HW_, px = pix - py * HW_;
        int gy = reflect_clamp(ty0 - 1 + py, a.H), gx = reflect_clamp(tx0 - 1 + px, a.W);
        if (a.up_in) { gy >>= 1; gx >>= 1; }
        const int c = cbase + q * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < a.cin) v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)gy * a.inW + gx) * a.cin + c);
        ldsIn[q * NPIX_PAD + pix] = v;
      }
      // ---- stage the weight slab of this chunk: [tap*4+q][cout_pad][4] -> [tap*4+q][COW][4]
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)ch * 36 * a.cout_pad;
      for (int e = tid; e < 36 * COW; e += 256) {
        const int seg = e / COW, j = e - seg * COW;
        ldsW[e] = wsrc[(size_t)seg * a.cout_pad + co0 + j];
      }
      __syncthreads();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        f32x4 bv[PT], av[CT];
#pragma unroll
        for (int p = 0; p < PT; ++p) bv[p] = ldsIn[kq * NPIX_PAD + (wave * PT + p + kh) * HW_ + li + kw];
#pragma unroll
        for (int c = 0; c < CT; ++c) av[c] = ldsW[(tap * 4 + kq) * COW + c * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p)
              acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][r], bv[p][r], acc[c][p], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds couts co0 + c*16 + kq*4 + {0..3} of pixel (row wave*4+p, col li)
  const int gx = tx0 + li;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int co = co0 + c * 16 + kq * 4;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + co);
    if constexpr (POOL) {
      const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
      for (int p = 0; p < PT; p += 2) {
        f32x4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaxf(acc[c][p][r], acc[c][p + 1][r]);
          v = fmaxf(v, __shfl_xor(v, 1));
          v += bias[r];
          m[r] = a.relu ? fmaxf(v, 0.f) : v;
        }
        const int oy = (ty0 + wave * PT + p) >> 1, ox = gx >> 1;
        if (!(li & 1) && oy < Hp && ox < Wp && co < a.cout)
          *reinterpret_cast<f32x4*>(a.out + ((size_t)oy * Wp + ox) * a.cout + co) = m;
      }
    } else {
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const int gy = ty0 + wave * PT + p;
        f32x4 v = acc[c][p] + bias;
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (gy < a.H && gx < a.W) {
          if constexpr (OUT3) {
            if (kq == 0) {
              const size_t plane = (size_t)a.H * a.W, off = (size_t)gy * a.W + gx;
              a.out[off] = v[0];
              a.out[plane + off] = v[1];
              a.out[2 * plane + off] = v[2];
            }
          } else if (co < a.cout) {
            if (a.out_sp) sp16_store4(reinterpret_cast<char*>(a.out) + (size_t)(co >> 4) * sp16_plane_bytes(a.H, a.W) + ((size_t)gy * a.W + gx) * 64, kq, v, sat, a.relu != 0);
            else *reinterpret_cast<f32x4*>(a.out + ((size_t)gy * a.W + gx) * a.cout + co) = v;
          }
        }
      }
    }
  }
  sat.commit(a.sat);
}

template <int CT, bool IN3, bool POOL, bool OUT3>
hipError_t launch_t(const ConvArgs& a, int cout_groups, hipStream_t s) {
  constexpr int COW = CT * 16;
  const size_t lds = IN3 ? (size_t)NPIX_PAD * 16 + (size_t)36 * COW * 4 : (size_t)4 * NPIX_PAD * 16 + (size_t)36 * COW * 16;
  auto k = conv3x3_kernel<CT, IN3, POOL, OUT3>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid(a.tiles_x * a.tiles_y, cout_groups);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3(const ConvDesc& d, const float* in, float* out, int H, int W, hipStream_t s) {
  ConvArgs a;
  a.in = in; a.out = out; a.wpk = d.wpk; a.bias = d.bias;
  a.H = H; a.W = W;
  a.up_in = (d.flags & CONV_UP_IN) ? 1 : 0;
  a.inH = a.up_in ? H / 2 : H; a.inW = a.up_in ? W / 2 : W;
  a.cin = d.cin; a.cout = d.cout; a.cin_chunks = d.cin_chunks; a.cout_pad = d.cout_pad;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.relu = (d.flags & CONV_NO_RELU) ? 0 : 1;
  a.out_sp = (d.flags & CONV_OUT_SP16) ? 1 : 0;
  a.sat = d.sat;
  const bool in3 = d.flags & CONV_IN_NCHW3, pool = d.flags & CONV_POOL_OUT, out3 = d.flags & CONV_OUT_NCHW3;
  if (a.out_sp && (pool || out3 || (d.cout & 15))) return hipErrorInvalidValue;   // SP16 from the plain epilogue only, whole 16-channel chunks
  if (H < 2 || W < 2) return hipErrorInvalidValue;  // reflect pad needs >= 2 samples
  int ct = d.cout_pad / 16, groups = 1;
  if (ct > 8) {
    if (d.cout_pad % 128) return hipErrorInvalidValue;
    groups = d.cout_pad / 128; ct = 8;
  }
  if (out3) return (ct == 1 && !in3 && !pool) ? launch_t<1, false, false, true>(a, 1, s) : hipErrorInvalidValue;
  if (in3) {
    if (pool) return hipErrorInvalidValue;
    switch (ct) {
      case 1: return launch_t<1, true, false, false>(a, groups, s);
      case 2: return launch_t<2, true, false, false>(a, groups, s);
      case 4: return launch_t<4, true, false, false>(a, groups, s);
      default: return hipErrorInvalidValue;
    }
  }
#define WCT_CASE(CTV)                                                    \
  case CTV:                                                              \
    return pool ? launch_t<CTV, false, true, false>(a, groups, s)        \
                : launch_t<CTV, false, false, false>(a, groups, s);
  switch (ct) {
    WCT_CASE(1) WCT_CASE(2) WCT_CASE(4) WCT_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef WCT_CASE
}
