// Level 1 of the 16x cascade without materialising relu1_1.
//
// The level-1 encoder is ONE convolution (conv0 folded into conv11: 3 -> 24 channels, model_cd.py:724-727) and the
// level-1 decoder is ONE convolution (conv11: 24 -> 3, model_cd.py:291-293), both at full image resolution.  Run layer
// by layer the 24-channel feature map (96 B/px: 796 MB at 4K) is written by the encoder, read by the moments kernel and
// read again by the decoder -- 2.4 GB of HBM traffic around 1.5 kFLOP/px of arithmetic.  Here relu1_1 only ever exists
// as a tile in LDS:
//     l1_moments_kernel (moments.hip)   image -> conv11 -> fp64 moments               12 B/px read
//     l1_decode_kernel                  image -> conv11 -> folded conv11' -> image    12 B/px read + 12 B/px written
//     l1_encode_kernel                  image -> conv11 -> relu1_1 (the API's wct_encode(1); same conv11 arithmetic)
// conv11 runs as f16x3 with the K = 27 products in the RGB0 slot layout of enc_head_kernel (conv3x3_f16.hip); the
// three kernels share l1_conv_group(), so their relu1_1 values are bit-identical.
#include "wct_common.h"
#include "conv_f16_dev.h"
#include <cstdlib>

namespace {

struct L1DecArgs {
  const float* img; float* out;
  L1Conv c;                                  // conv11 of the encoder (f16x3 slot layout, 2 cout tiles)
  const u32x4* w2; const float* b2;          // folded decoder conv, block-packed (c3_block_compute): [2 chunks][PH_WSLOTS] x 16 B, bias [16]
  const float* inv2_ptr; float inv2;
  int H, W, tiles_x, tiles_y;
  unsigned* sat;
};

struct L1EncArgs {
  const float* img; float* out;              // out: NHWC fp32 [H*W][C]
  L1Conv c;
  int C, H, W, tiles_x, tiles_y;
  unsigned* sat;
};

// image -> relu1_1 (NHWC fp32)
__global__ __launch_bounds__(256, 3) void l1_encode_kernel(L1EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x2* imgH = reinterpret_cast<u32x2*>(smem);
  u32x2* imgL = imgH + IMG_E;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);   // once per workgroup; tile_rc() then stays on the scalar unit
  if (tid < 4) { imgH[NPI2 + tid] = u32x2{0u, 0u}; imgL[NPI2 + tid] = u32x2{0u, 0u}; }
  L1Weights w;
  l1_load_weights(a.c, li, kq, IMG_E, w);
  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + 256 * k;
    e = e < NPI2 ? e : NPI2 - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  float pxr[2][3];
  SatTrack sat;
  int v = blockIdx.x;
  if (v < ntiles) {
    head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(v, ntiles), tid);
    head_commit(pxr, imgH, imgL, tid, sat);
  }
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * 8, tx0 = tcol_ * FTW;
    __syncthreads();
    const int vn = v + gridDim.x;
    if (vn < ntiles) head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(vn, ntiles), tid);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int py = wave * 2 + r, px = h * 16 + li;
        const int base = (py + 1) * I2W + px + 1;   // top-left of the 3x3 window in the 36 x 12 tile (origin -2, -2)
        const int gy = ty0 + py, gx = tx0 + px;
        f32x4 xs[2];
        l1_conv_pair(imgH, base, w, xs[0], xs[1]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int co = ct * 16 + 4 * kq;
          if (gy < a.H && gx < a.W && co < a.C) *reinterpret_cast<f32x4*>(a.out + ((size_t)gy * a.W + gx) * a.C + co) = xs[ct];
        }
      }
    __syncthreads();
    if (vn < ntiles) { head_pin(pxr); head_commit(pxr, imgH, imgL, tid, sat); }
  }
  sat.commit(a.sat);
}

// ---- the 3 -> 64 first convolution of the un-pruned encoders (--mode original: model_original.py conv1_1 with conv0 folded in), f16x3.
// It used to run as exact-fp32 MFMA (conv3x3.hip: 9 x 32 issue cycles per 16 pixels and cout tile, 197 us per 1920x1080 launch =
// 2.7 TB/s of its 256 B/px output); here it is l1_encode_kernel's structure with FOUR cout tiles from one set of operand reads
// (K = 27 singles concatenated: 4 MFMAs of 16 cycles per 16 pixels and cout tile), fp32 NHWC or SP16 out -- a write stream.
struct In3WideArgs {
  const float* img; void* out;               // out: NHWC fp32 [H*W][64], or SP16 (four 16-channel chunk planes)
  const u32x4* w; const float* b; float inv; // [4 cout tiles][4 K-steps][4 kq][16] x 8 halfs; bias [64]
  int H, W, tiles_x, tiles_y, out_sp;
  unsigned* sat;
};

__global__ __launch_bounds__(256, 2) void in3_wide_kernel(In3WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x2* imgH = reinterpret_cast<u32x2*>(smem);
  u32x2* imgL = imgH + IMG_E;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);
  if (tid < 4) { imgH[NPI2 + tid] = u32x2{0u, 0u}; imgL[NPI2 + tid] = u32x2{0u, 0u}; }
  f16x8 wa[4][4];
  f32x4 bias[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[ct][s] = __builtin_bit_cast(f16x8, a.w[((ct * 4 + s) * 4 + kq) * 16 + li]);
    bias[ct] = *reinterpret_cast<const f32x4*>(a.b + ct * 16 + 4 * kq);
  }
  int boff[4][2];
  l1_lane_offsets(kq, IMG_E, boff);
  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + 256 * k;
    e = e < NPI2 ? e : NPI2 - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  float pxr[2][3];
  SatTrack sat;
  const size_t plane = sp16_plane_bytes(a.H, a.W);
  int v = blockIdx.x;
  if (v < ntiles) {
    head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(v, ntiles), tid);
    head_commit(pxr, imgH, imgL, tid, sat);
  }
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * 8, tx0 = tcol_ * FTW;
    __syncthreads();
    const int vn = v + gridDim.x;
    if (vn < ntiles) head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(vn, ntiles), tid);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int py = wave * 2 + r, px = h * 16 + li;
        const int base = (py + 1) * I2W + px + 1;   // top-left of the 3x3 window in the 36 x 12 tile (origin -2, -2)
        const int gy = ty0 + py, gx = tx0 + px;
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* bp = reinterpret_cast<const char*>(imgH) + base * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x2 r0 = *reinterpret_cast<const u32x2*>(bp + boff[s][0]), r1 = *reinterpret_cast<const u32x2*>(bp + boff[s][1]);
          const f16x8 b = __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ct][s], b, acc[ct], 0, 0, 0);
        }
        const bool ok = gy < a.H && gx < a.W;
        const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x4 x = fma4(acc[ct], a.inv, bias[ct]);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
          if (a.out_sp) {
            u32x2 hi, lo;
            split4(x, hi, lo, sat, true);
            if (ok) {
              char* g = reinterpret_cast<char*>(a.out) + ct * plane + pix * 64 + (kq >> 1) * 32 + (kq & 1) * 8;
              *reinterpret_cast<u32x2*>(g) = hi;
              *reinterpret_cast<u32x2*>(g + 16) = lo;
            }
          } else if (ok) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + pix * 64 + ct * 16 + 4 * kq) = x;
          }
        }
      }
    __syncthreads();
    if (vn < ntiles) { head_pin(pxr); head_commit(pxr, imgH, imgL, tid, sat); }
  }
  sat.commit(a.sat);
}

// image -> relu1_1 on the 34 x (TH + 2) halo (LDS, split f16) -> folded decoder conv (24 -> 3) + ReLU -> planar image.
// TH = 8: 4 waves, two workgroups per CU; TH = 16: 8 waves, one workgroup per CU, halo recompute 1.20 instead of 1.33.
// Round 6: the folded conv's weight operands live in REGISTERS (96 VGPRs: sixteen for the 16-channel chunk, eight for the 8 real channels
// of the second, whose K-steps are whole window rows -- conv_f16_dev.h c3_block_compute_w / _half) instead of a 32 KB slab in LDS that every
// wave re-read per tile, and the second chunk keeps two planes instead of four: per tile and wave 24 operand reads and 36 MFMAs in the second
// phase instead of 64 and 48, LDS 42 KB (TH = 8: two or three workgroups per CU, whose phases overlap) / 75 KB instead of 87 / 128.
template <int TH>
struct L1DecGeo {
  static constexpr int NT = 32 * TH, NWV = TH / 2, HROWS = TH + 2, NPH = FHW * HROWS, NGRP = (NPH + 15) / 16, NG = (NGRP + NWV - 1) / NWV;
  static constexpr int NPI = I2W * (TH + 4), IMGE = NPI + 4, NPX = (HROWS * PH_W + 15) / 16 * 16;
  static constexpr size_t lds = (size_t)2 * IMGE * 8 + (size_t)6 * NPX * 16;   // 42.3 KB (TH = 8) / 74.6 KB (TH = 16)
};

template <int TH>
__global__ __launch_bounds__(32 * TH, 2) void l1_decode_kernel(L1DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = L1DecGeo<TH>;
  constexpr int NT = G::NT, NWV = G::NWV, NPH = G::NPH, NG = G::NG, NPI = G::NPI, NPX = G::NPX;
  u32x2* imgH = reinterpret_cast<u32x2*>(smem);
  u32x2* imgL = imgH + G::IMGE;
  u32x4* act = reinterpret_cast<u32x4*>(imgL + G::IMGE);   // [4][NPX] channels 0..15 (hl, channel half) | [2][NPX] channels 16..23 (hl): relu1_1 on the halo, pair-major slots (ph_slot)
  u32x4* act2 = act + 4 * NPX;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);   // once per workgroup; tile_rc() then stays on the scalar unit
  if (tid < 4) { imgH[NPI + tid] = u32x2{0u, 0u}; imgL[NPI + tid] = u32x2{0u, 0u}; }
  L1Weights w;
  l1_load_weights(a.c, li, kq, G::IMGE, w);
  C3Weights w2a;            // the folded decoder conv's operands: registers, loaded once per workgroup from the block-packed slab
  C3HalfWeights w2b;
  c3_load_weights(a.w2, li, kq, w2a);
  c3_load_weights_half(a.w2 + PH_WSLOTS, li, kq, w2b);
  const float inv2 = a.inv2_ptr ? *a.inv2_ptr : a.inv2;
  const f32x4 bias2 = *reinterpret_cast<const f32x4*>(a.b2);
  const size_t plane = (size_t)a.H * a.W;
  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + NT * k;
    e = e < NPI ? e : NPI - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  int gpix[NG], gpy[NG], gpx[NG], gslot[NG];
  bool gok[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int pixr = (wave + NWV * u) * 16 + li;
    gok[u] = pixr < NPH;
    gpix[u] = gok[u] ? pixr : NPH - 1;
    gpy[u] = gpix[u] / FHW;
    gpx[u] = gpix[u] - gpy[u] * FHW;
    gslot[u] = gok[u] ? ph_slot(gpy[u], gpx[u]) : NPX - 1;   // lanes without a pixel store to a slot nobody reads (no exec masking)
  }
  float pxr[2][3];
  SatTrack sat;
  int v = blockIdx.x;
  if (v < ntiles) {
    head_fetch<TH>(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(v, ntiles), tid);
    head_commit<TH>(pxr, imgH, imgL, tid, sat);
  }
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * TH, tx0 = tcol_ * FTW;
    __syncthreads();   // image window in LDS; previous tile's planes consumed
    const int vn = v + gridDim.x;
    if (vn < ntiles) head_fetch<TH>(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(vn, ntiles), tid);
    const bool interior = tile_interior_h(ty0, tx0, a.H, a.W, TH);
    // ---- relu1_1 on the halo pixels, each evaluated at its REFLECTED image coordinate (the decoder's own padding)
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      int base;
      if (interior) {
        base = gpy[u] * I2W + gpx[u];
      } else {
        const int iy = reflect_clamp(ty0 - 1 + gpy[u], a.H) - (ty0 - 2), ix = reflect_clamp(tx0 - 1 + gpx[u], a.W) - (tx0 - 2);
        base = (iy - 1) * I2W + ix - 1;
      }
      f32x4 xs[2];
      l1_conv_pair<false>(imgH, base, w, xs[0], xs[1]);
      store_split4<true>(act, NPX, gslot[u], kq, xs[0], sat);
      store_split4_half(act2, NPX, gslot[u], kq, xs[1], sat);
    }
    __syncthreads();
    // ---- folded decoder conv on the two 16-channel chunks, block-packed (conv_f16_dev.h): every lane ends with one output pixel
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    c3_block_compute_w<NPX>(act, w2a, wave, li, kq, acc);
    c3_block_compute_half<NPX>(act2, w2b, wave, li, kq, acc);
    c3_block_store(acc, inv2, bias2, a.out, plane, ty0, tx0, wave, li, kq, a.H, a.W);
    if (vn < ntiles) { head_pin(pxr); head_commit<TH>(pxr, imgH, imgL, tid, sat); }
  }
  sat.commit(a.sat);
}

}  // namespace

bool l1_capable(const ConvDesc& enc0) {
  return (enc0.flags & CONV_IN_NCHW3) && !(enc0.flags & (CONV_POOL_OUT | CONV_NO_RELU)) && enc0.l1w16 && enc0.cout <= 32 && (enc0.cout % 4) == 0;
}

hipError_t launch_l1_encode(const ConvDesc& e, const float* img, float* out, int H, int W, hipStream_t s) {
  if (!l1_capable(e) || H < 2 || W < 2) return hipErrorInvalidValue;
  L1EncArgs a;
  a.img = img; a.out = out;
  a.c.w = reinterpret_cast<const u32x4*>(e.l1w16); a.c.b = e.l1bias; a.c.inv = e.l1inv;
  a.C = e.cout; a.H = H; a.W = W; a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + 7) / 8;
  a.sat = e.sat;
  const size_t lds = (size_t)2 * IMG_E * 8;
  const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < 3 * num_cus() ? ntiles : 3 * num_cus();
  hipLaunchKernelGGL(l1_encode_kernel, dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

// The same layer in EXACT fp32 products (v_mfma_f32_16x16x4_f32, one tap = one K = 4 step over the RGB0 slots).  Why it exists
// (round 4, tools/experiments/first_conv_ab.py): conv0 folds `255 x - mean` into this layer, so its sums cancel from O(255 |w|) to
// O(|w| sigma), and a split-f16 operand carries 22-23 bits against fp32's 24 -- the f16x3 form of THIS layer is measurably further
// from the reference than fp32 (G15 end to end 3.4e-4 against 2.3e-4, G14 2.29e-3 against 1.58e-3), and it feeds every level of
// both lanes.  The layer is a write stream (256 B/px out, 12 B/px in): 9 x 4 MFMAs of 32 cycles per 16 pixels still fit under the
// HBM time of the stores, which the generic fp32 kernel (conv3x3.hip, one cout tile per operand read) did not (197 us per 1080p
// launch against 128 us for the f16x3 form).  LDS: four fp32 planes R, G, B, 0 of the 36 x 12 window, plane stride 432 = 16 (mod 32)
// dwords: the 32 lanes of a ds_read_b32 group (16 pixels x 2 planes) hit 32 distinct banks.
constexpr int IMGF_PLANE = NPI2;          // 432 floats
static_assert(IMGF_PLANE % 32 == 16, "plane stride must be 16 mod 32 dwords (conflict-free operand reads)");

__global__ __launch_bounds__(256, 2) void in3_wide_f32_kernel(In3WideArgs a) {
  __shared__ float imgF[4 * IMGF_PLANE + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);
  for (int e = tid; e < IMGF_PLANE + 4; e += 256) imgF[3 * IMGF_PLANE + e] = 0.f;   // the "0" slot of RGB0 (zero weights, finite data)
  float wa[4][9];
  f32x4 bias[4];
  const float* wf = reinterpret_cast<const float*>(a.w);     // [tap][k = 4][64] fp32 (wct_api.hip pack_weights, in3)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wa[ct][t] = wf[(t * 4 + kq) * 64 + ct * 16 + li];
    bias[ct] = *reinterpret_cast<const f32x4*>(a.b + ct * 16 + 4 * kq);
  }
  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + 256 * k;
    e = e < NPI2 ? e : NPI2 - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  auto commit = [&](const float (&r)[2][3]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      if (e < NPI2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) imgF[c * IMGF_PLANE + e] = r[k][c];
      }
    }
  };
  float pxr[2][3];
  SatTrack sat;
  const size_t plane = sp16_plane_bytes(a.H, a.W);
  int v = blockIdx.x;
  if (v < ntiles) {
    head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(v, ntiles), tid);
    commit(pxr);
  }
  settle_preloop_loads();
  const float* plane_k = imgF + kq * IMGF_PLANE;
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * 8, tx0 = tcol_ * FTW;
    __syncthreads();
    const int vn = v + gridDim.x;
    if (vn < ntiles) head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(vn, ntiles), tid);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int py = wave * 2 + r, px = h * 16 + li;
        const int base = (py + 1) * I2W + px + 1;   // top-left of the 3x3 window in the 36 x 12 tile (origin -2, -2)
        const int gy = ty0 + py, gx = tx0 + px;
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        float b[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[t] = plane_k[base + (t / 3) * I2W + t % 3];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ct][t], b[t], acc[ct], 0, 0, 0);
        const bool ok = gy < a.H && gx < a.W;
        const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x4 x = acc[ct] + bias[ct];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
          if (a.out_sp) {
            u32x2 hi, lo;
            split4(x, hi, lo, sat, true);
            if (ok) {
              char* g = reinterpret_cast<char*>(a.out) + ct * plane + pix * 64 + (kq >> 1) * 32 + (kq & 1) * 8;
              *reinterpret_cast<u32x2*>(g) = hi;
              *reinterpret_cast<u32x2*>(g + 16) = lo;
            }
          } else if (ok) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + pix * 64 + ct * 16 + 4 * kq) = x;
          }
        }
      }
    __syncthreads();
    if (vn < ntiles) { head_pin(pxr); commit(pxr); }
  }
  sat.commit(a.sat);
}

bool in3_wide_capable(const ConvDesc& enc0) {
  return (enc0.flags & CONV_IN_NCHW3) && !(enc0.flags & (CONV_POOL_OUT | CONV_NO_RELU)) && enc0.l1w16 && enc0.cout == 64 && enc0.cout_pad == 64;
}

hipError_t launch_in3_wide(const ConvDesc& e, const float* img, void* out, int H, int W, bool out_sp, bool exact_fp32, hipStream_t s) {
  if (!in3_wide_capable(e) || H < 2 || W < 2) return hipErrorInvalidValue;
  In3WideArgs a;
  a.img = img; a.out = out;
  a.w = reinterpret_cast<const u32x4*>(e.l1w16); a.b = e.l1bias; a.inv = e.l1inv;
  if (exact_fp32) {
    a.w = reinterpret_cast<const u32x4*>(e.wpk); a.b = e.bias; a.inv = 1.f;
    a.H = H; a.W = W; a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + 7) / 8; a.out_sp = out_sp ? 1 : 0;
    a.sat = e.sat;
    const int nt = a.tiles_x * a.tiles_y, g = nt < 2 * num_cus() ? nt : 2 * num_cus();
    hipLaunchKernelGGL(in3_wide_f32_kernel, dim3(g), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  a.H = H; a.W = W; a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + 7) / 8; a.out_sp = out_sp ? 1 : 0;
  a.sat = e.sat;
  const size_t lds = (size_t)2 * IMG_E * 8;
  const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < 2 * num_cus() ? ntiles : 2 * num_cus();
  hipLaunchKernelGGL(in3_wide_kernel, dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

// dec0: the decoder's only conv with the WCT map folded in (split-f16 packed, 10 taps, cout_pad 16)
hipError_t launch_l1_decode(const ConvDesc& e, const ConvDesc& dec0, const float* img, float* out, int H, int W, hipStream_t s) {
  if (!l1_capable(e) || H < 2 || W < 2 || !dec0.wpk16 || !dec0.wph16 || dec0.cout != 3 || dec0.cout_pad != 16 || dec0.cin != e.cout ||
      dec0.cin_chunks != 2 || !(dec0.flags & CONV_OUT_NCHW3) || (dec0.flags & (CONV_UP_IN | CONV_NO_RELU | CONV_POOL_OUT)))
    return hipErrorInvalidValue;
  L1DecArgs a;
  a.img = img; a.out = out;
  a.c.w = reinterpret_cast<const u32x4*>(e.l1w16); a.c.b = e.l1bias; a.c.inv = e.l1inv;
  a.w2 = reinterpret_cast<const u32x4*>(dec0.wph16); a.b2 = dec0.bias; a.inv2_ptr = dec0.inv_scale_ptr; a.inv2 = dec0.inv_scale;
  a.H = H; a.W = W; a.tiles_x = (W + FTW - 1) / FTW;
  a.sat = e.sat;
  static const int th_env = [] { const char* v = wct_debug_env("WCT_L1DEC_TH"); return v ? atoi(v) : 0; }();   // experiment: force 8 / 16
  auto go = [&](auto kern, auto geo, int th, int per_cu) -> hipError_t {
    using G = decltype(geo);
    a.tiles_y = (H + th - 1) / th;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds);
    if (err != hipSuccess) return err;
    const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < per_cu * num_cus() ? ntiles : per_cu * num_cus();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::lds, s, a);
    return hipGetLastError();
  };
  // 32 x 16 tiles (less halo recompute, one workgroup of eight waves per CU) once they still fill the chip, else 32 x 8 tiles, two workgroups of
  // four waves per CU; results do not depend on the tile shape.  Measured at 4K (round 6, same box, profiles/r06_l1_decode_ab.txt): 32 x 16 one
  // per CU 0.229 ms, 32 x 8 two per CU 0.246, 32 x 8 one per CU 0.374 (round 5's kernel, weights in LDS: 0.255-0.262).  WCT_L1DEC_TH forces one.
  const bool tall = th_env ? th_env == 16 : ((H + 15) / 16) * a.tiles_x >= 2 * num_cus();
  return tall ? go(l1_decode_kernel<16>, L1DecGeo<16>{}, 16, 1) : go(l1_decode_kernel<8>, L1DecGeo<8>{}, 8, 2);
}
