// reflect-pad(1) + conv3x3 + bias + ReLU on the f16 matrix cores with SPLIT operands ("f16x3"):
//     x = hi + lo,  hi = f16(x), lo = f16(x - hi)          (two-term split: 11 + 11 bits + sign of the residual
//     w.x  ~=  hi_w.hi_x + hi_w.lo_x + lo_w.hi_x            ~ 23-24 significant bits, i.e. fp32-class)
// accumulated in fp32 by v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16.  The dropped lo.lo term is 2^-22
// relative.  Measured on the GPU against fp64 (tools/experiments/split_precision.hip, K = 1152, post-ReLU
// activations with 30x outliers): f16x3 6.2e-7 vs exact-fp32 MFMA 7.7e-7 of max|ref| -- the same accuracy class,
// at 3/16 of the fp32-MFMA issue time (bf16x3: 9.4e-6, plain f16: 4.1e-4, both rejected by the parity gate).
// Range: activations of this path are <= ~250 and weights <= ~2 in magnitude (measured on the shipped images);
// weights are pre-scaled by a per-layer power of two so that their lo parts stay normal f16 numbers, and the
// epilogue multiplies by the exact inverse.  Activations beyond +-65504 saturate (never inf).
//
// Same fusions and the same NHWC/planar layouts as conv3x3.hip (reflect pad, 2x2 max-pool epilogue, nearest-x2
// in the tile load, folded conv0 / WCT affine).  These are the REGISTER-STAGED kernels: fp32 NHWC or SP16 input
// (conv_f16_dev.h), fp32 or SP16 output.  They serve the fp32-input layers (first decoder conv with the folded WCT map, API
// entry points), the 16-cout layers and the pooled 32-cout layers; SP16-input layers with >= 32 couts otherwise take the
// DMA-staged kernel of conv3x3_sp.hip.  A stand-alone 3-channel first conv (original mode) stays on the fp32 kernel.
//
// Tiling: workgroup = 4 waves = 32 x 8 output pixels x all couts (<= 128); wave w owns rows 2w, 2w+1.
// K is walked in 16-channel chunks; per chunk LDS holds, as 16-byte groups of 8 halfs,
//   act[hl][kh][pix]      hl = hi/lo plane, kh = channel half (8kh..8kh+7), pix = 34 x 10 halo pixel
//   wgt[tap][hl][kh][co]
// so every MFMA operand is ONE ds_read_b128 and consecutive lanes (consecutive pixels / couts) read consecutive
// 16-B slots with plane strides that are multiples of 256 B: bank-conflict free.
//   Cout >= 32: 32x32x16, A = 32 couts x 16 channels of one tap, B = 16 channels x 32 pixels of one tile row.
//   Cout == 16 (and the 3-channel last conv, padded): 16x16x32, the 32-deep K holds TWO taps x 16 channels.
#include "wct_common.h"
#include "conv_f16_dev.h"
#include <cstdlib>

namespace {

struct F16Args {
  const float* in;
  float* out;
  const u32x4* wpk;        // [chunk][taps][hl][kh][cout_pad] x 16 B
  const float* bias;
  const float* inv_scale_ptr;  // device scalar (folded layers) or null
  float inv_scale;
  int H, W, inH, inW;
  int cin, cout, cin_chunks, cout_pad, taps;  // taps = 9 (32x32 path) or 10 (16x16x32 path, tap 9 = zeros)
  int tiles_x, tiles_y;
  int up_in, relu;
  int in_sp, out_sp;   // SP16 input (16-byte groups taken as they are) / SP16 output (split in the epilogue)
  unsigned* sat;       // sticky saturation counter of the context (may be null)
};

// stage one 16-channel chunk of the halo tile, converting fp32 -> (hi, lo) f16 planes.
// All of a thread's global loads (<= 3 slots x 2 float4) are issued back to back and UNCONDITIONALLY (addresses
// clamped; out-of-range channels zeroed afterwards) before the first conversion, so their latencies overlap: a load
// under a per-lane condition makes hipcc branch around it and wait on the spot (cdna_hip_programming.md traps (c)).
constexpr int ACT_SLOTS = 3;  // ceil(2 * nph(TH) / (32 * TH)) for TH = 8 and 16
struct ActRegs { f32x4 v0[ACT_SLOTS], v1[ACT_SLOTS]; };

template <int TH>
__device__ __forceinline__ void fetch_act(const F16Args& a, ActRegs& r, int ch, int ty0, int tx0, int tid) {
  constexpr int NPH = nph(TH), NT = 32 * TH;
  static_assert((NPH * 2 + NT - 1) / NT <= ACT_SLOTS, "slots");
  const int cbase = ch * 16;
#pragma unroll
  for (int k = 0; k < ACT_SLOTS; ++k) {
    int e = tid + NT * k;
    e = e < NPH * 2 ? e : NPH * 2 - 1;
    const int kh = e & 1, pix = e >> 1;
    const int py = pix / FHW, px = pix - py * FHW;
    int gy = reflect_clamp(ty0 - 1 + py, a.H), gx = reflect_clamp(tx0 - 1 + px, a.W);
    if (a.up_in) { gy >>= 1; gx >>= 1; }
    int c = cbase + kh * 8;
    c = c + 8 <= a.cin ? c : 0;   // cin is a multiple of 8 on this path (16, 24, 32, 64, 128, ...)
    // SP16 input (conv_f16_dev.h): chunk plane `ch`, 64-byte pixel record, group kh -> the same 32 bytes as [8 hi | 8 lo]
    const float* src = a.in_sp ? a.in + (size_t)ch * ((size_t)a.inH * a.inW * 16) + ((size_t)gy * a.inW + gx) * 16 + kh * 8
                               : a.in + ((size_t)gy * a.inW + gx) * a.cin + c;
    r.v0[k] = *reinterpret_cast<const f32x4*>(src);
    r.v1[k] = *reinterpret_cast<const f32x4*>(src + 4);
  }
}

template <int TH>
__device__ __forceinline__ void commit_act(const F16Args& a, const ActRegs& r, u32x4* act, int ch, int tid, SatTrack& sat) {
  constexpr int NPH = nph(TH), NPP = npp(TH), NT = 32 * TH;
  const int cbase = ch * 16;
#pragma unroll
  for (int k = 0; k < ACT_SLOTS; ++k) {
    const int e = tid + NT * k;
    if (e < NPH * 2) {
      const int kh = e & 1, pix = e >> 1;
      const bool ok = cbase + kh * 8 + 8 <= a.cin;
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f}, w0 = ok ? r.v0[k] : z, w1 = ok ? r.v1[k] : z;
      f16x8 hi, lo;
      if (a.in_sp) { hi = __builtin_bit_cast(f16x8, w0); lo = __builtin_bit_cast(f16x8, w1); }   // the same 32 bytes hold [8 hi | 8 lo]
      else split8(w0, w1, hi, lo, sat);
      act[(0 * 2 + kh) * NPP + pix] = __builtin_bit_cast(u32x4, hi);
      act[(1 * 2 + kh) * NPP + pix] = __builtin_bit_cast(u32x4, lo);
    }
  }
}

// weight slab of one chunk: NW 16-B groups, global -> registers -> LDS
template <int NW, int NT = 256>
struct WRegs { u32x4 w[(NW + NT - 1) / NT]; };

template <int NW, int COW, int NT = 256>
__device__ __forceinline__ void fetch_w(const F16Args& a, WRegs<NW, NT>& r, int ch, int co0, int tid) {
  const u32x4* wsrc = a.wpk + (size_t)ch * (NW / COW) * a.cout_pad;
#pragma unroll
  for (int k = 0; k < (NW + NT - 1) / NT; ++k) {
    int e = tid + NT * k;
    e = e < NW ? e : NW - 1;
    const int seg = e / COW, j = e - seg * COW;
    r.w[k] = wsrc[(size_t)seg * a.cout_pad + co0 + j];
  }
}

template <int NW, int NT = 256>
__device__ __forceinline__ void commit_w(const WRegs<NW, NT>& r, u32x4* wgt, int tid) {
#pragma unroll
  for (int k = 0; k < (NW + NT - 1) / NT; ++k) {
    const int e = tid + NT * k;
    if (e < NW) wgt[e] = r.w[k];
  }
}

// ------------------------------------------------------------------------------------------------ Cout >= 32
template <int CT, bool POOL, int TH>
__global__ __launch_bounds__(32 * TH) void conv3x3_f16_kernel(F16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COW = CT * 32, NPP = npp(TH), NT = 32 * TH, FTH = TH;
  u32x4* act = reinterpret_cast<u32x4*>(smem);   // [4][NPP]
  u32x4* wgt = act + 4 * NPP;                    // [9][2][2][COW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int tile = xcd_swizzle(blockIdx.x, a.tiles_x * a.tiles_y);
  const int ty0 = (tile / a.tiles_x) * FTH, tx0 = (tile % a.tiles_x) * FTW;
  const int co0 = blockIdx.y * COW;
  SatTrack sat;

  f32x16 acc[CT][2];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;

  // software pipeline over the 16-channel chunks: the fp32 activations (and, for <= 64 couts, the weight slab) of
  // chunk ch+1 are fetched into registers while chunk ch is on the matrix cores; they are converted / written to
  // LDS after the barrier that retires chunk ch.  (128 couts: the 74 KB weight slab does not fit the register
  // budget next to 128 accumulators -- it is loaded after the barrier, L2-resident.)
  constexpr int NW = 36 * COW;
  constexpr bool PREW = CT <= 2;   // prefetch the weight slab into registers
  constexpr bool PREA = TH == 8;   // prefetch the activations into registers (8-wave tiles: 2 waves/SIMD overlap instead)
  ActRegs ar;
  WRegs<PREW ? NW : NT, NT> wr;
  if constexpr (PREA) fetch_act<TH>(a, ar, 0, ty0, tx0, tid);
  if constexpr (PREW) fetch_w<NW, COW, NT>(a, wr, 0, co0, tid);
  for (int ch = 0; ch < a.cin_chunks; ++ch) {
    if (ch) __syncthreads();
    if constexpr (!PREA) fetch_act<TH>(a, ar, ch, ty0, tx0, tid);
    if constexpr (PREW) {
      commit_w<NW, NT>(wr, wgt, tid);
    } else {
      const u32x4* wsrc = a.wpk + (size_t)ch * 36 * a.cout_pad;
      for (int e = tid; e < NW; e += NT) {
        const int seg = e / COW, j = e - seg * COW;
        wgt[e] = wsrc[(size_t)seg * a.cout_pad + co0 + j];
      }
    }
    commit_act<TH>(a, ar, act, ch, tid, sat);
    __syncthreads();
    if (ch + 1 < a.cin_chunks) {
      if constexpr (PREA) fetch_act<TH>(a, ar, ch + 1, ty0, tx0, tid);
      if constexpr (PREW) fetch_w<NW, COW, NT>(a, wr, ch + 1, co0, tid);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pix = (wave * 2 + p + dy) * FHW + li + dx;
        bh[p] = __builtin_bit_cast(f16x8, act[(0 * 2 + kh) * NPP + pix]);
        bl[p] = __builtin_bit_cast(f16x8, act[(1 * 2 + kh) * NPP + pix]);
      }
      // the three split terms of one accumulator are issued CT * 2 MFMAs apart (independent accumulators in between)
      f16x8 ah[CT], al[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        ah[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 0) * 2 + kh) * COW + c * 32 + li]);
        al[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 1) * 2 + kh) * COW + c * 32 + li]);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[c] : ah[c], term == 1 ? bl[p] : bh[p], acc[c][p], 0, 0, 0);
    }
  }

  // epilogue.  D: col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cout)
  const float inv = a.inv_scale_ptr ? *a.inv_scale_ptr : a.inv_scale;
  const int gx = tx0 + li;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = co0 + c * 32 + 8 * q + 4 * kh;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + co);
      if constexpr (POOL) {
        const int Hp = a.H >> 1, Wp = a.W >> 1;
        f32x4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(acc[c][0][4 * q + r], acc[c][1][4 * q + r]);
          m[r] = fmaxf(v, lane_xor1(v));
        }
        m = fma4(m, inv, bias);
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], 0.f);
        }
        const int oy = (ty0 + wave * 2) >> 1, ox = gx >> 1;
        const bool ok = !(li & 1) && oy < Hp && ox < Wp && co < a.cout;
        if (a.out_sp) {
          const u32x4 w = sp16_pair_exchange(m, sat, a.relu != 0);
          if (ok) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out) + sp16_piece(sp16_plane_bytes(Hp, Wp), (size_t)oy * Wp + ox, co >> 3, kh)) = w;
        } else if (ok) {
          *reinterpret_cast<f32x4*>(a.out + ((size_t)oy * Wp + ox) * a.cout + co) = m;
        }
      } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int gy = ty0 + wave * 2 + p;
          f32x4 v = fma4(f32x4{acc[c][p][4 * q], acc[c][p][4 * q + 1], acc[c][p][4 * q + 2], acc[c][p][4 * q + 3]}, inv, bias);
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          const bool ok = gy < a.H && gx < a.W && co < a.cout;
          if (a.out_sp) {
            const u32x4 w = sp16_pair_exchange(v, sat, a.relu != 0);
            if (ok) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out) + sp16_piece(sp16_plane_bytes(a.H, a.W), (size_t)gy * a.W + gx, co >> 3, kh)) = w;
          } else if (ok) {
            *reinterpret_cast<f32x4*>(a.out + ((size_t)gy * a.W + gx) * a.cout + co) = v;
          }
        }
      }
    }
  }
  sat.commit(a.sat);
}

// ------------------------------------------------------------------------------------------------ Cout <= 16
template <bool POOL, bool OUT3>
__global__ __launch_bounds__(256) void conv3x3_f16_c16_kernel(F16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FTH = 8, NPP = npp(8);
  u32x4* act = reinterpret_cast<u32x4*>(smem);   // [4][NPP]
  u32x4* wgt = act + 4 * NPP;                    // [10][2][2][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4, kh = kq & 1, ts = kq >> 1;
  const int tile = xcd_swizzle(blockIdx.x, a.tiles_x * a.tiles_y);
  const int ty0 = (tile / a.tiles_x) * FTH, tx0 = (tile % a.tiles_x) * FTW;
  SatTrack sat;

  f32x4 acc[2][2];  // [row][half row]
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NW = 40 * 16;
  ActRegs ar;
  WRegs<NW> wr;
  fetch_act<8>(a, ar, 0, ty0, tx0, tid);
  fetch_w<NW, 16>(a, wr, 0, 0, tid);
  for (int ch = 0; ch < a.cin_chunks; ++ch) {
    if (ch) __syncthreads();
    commit_act<8>(a, ar, act, ch, tid, sat);
    commit_w<NW>(wr, wgt, tid);
    __syncthreads();
    if (ch + 1 < a.cin_chunks) {
      fetch_act<8>(a, ar, ch + 1, ty0, tx0, tid);
      fetch_w<NW, 16>(a, wr, ch + 1, 0, tid);
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + ts;               // tap 9 carries zero weights
      const int tc = tap > 8 ? 8 : tap;
      const int dy = tc / 3, dx = tc - dy * 3;
      const f16x8 ah = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 0) * 2 + kh) * 16 + li]);
      const f16x8 al = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 1) * 2 + kh) * 16 + li]);
      f16x8 bh[2][2], bl[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pix = (wave * 2 + r + dy) * FHW + h * 16 + li + dx;
          bh[r][h] = __builtin_bit_cast(f16x8, act[(0 * 2 + kh) * NPP + pix]);
          bl[r][h] = __builtin_bit_cast(f16x8, act[(1 * 2 + kh) * NPP + pix]);
        }
#pragma unroll
      for (int term = 0; term < 3; ++term)   // dependent MFMAs 4 apart
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[r][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? al : ah, term == 1 ? bl[r][h] : bh[r][h], acc[r][h], 0, 0, 0);
    }
  }

  // D: col = lane & 15 (pixel), row = 4 * (lane >> 4) + reg (cout)
  const float inv = a.inv_scale_ptr ? *a.inv_scale_ptr : a.inv_scale;
  const int co = 4 * kq;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int gx = tx0 + h * 16 + li;
    if constexpr (POOL) {
      const int Hp = a.H >> 1, Wp = a.W >> 1;
      f32x4 m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(acc[0][h][r], acc[1][h][r]);
        m[r] = fmaxf(v, lane_xor1(v));
      }
      m = fma4(m, inv, bias);
      if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], 0.f);
      }
      const int oy = (ty0 + wave * 2) >> 1, ox = gx >> 1;
      if (!(li & 1) && oy < Hp && ox < Wp && co < a.cout) {
        if (a.out_sp) sp16_store4(reinterpret_cast<char*>(a.out) + ((size_t)oy * Wp + ox) * a.cout * 4, kq, m, sat, a.relu != 0);
        else *reinterpret_cast<f32x4*>(a.out + ((size_t)oy * Wp + ox) * a.cout + co) = m;
      }
    } else {
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int gy = ty0 + wave * 2 + r2;
        f32x4 v = fma4(acc[r2][h], inv, bias);
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
            if (a.out_sp) sp16_store4(reinterpret_cast<char*>(a.out) + ((size_t)gy * a.W + gx) * a.cout * 4, kq, v, sat, a.relu != 0);
            else *reinterpret_cast<f32x4*>(a.out + ((size_t)gy * a.W + gx) * a.cout + co) = v;
          }
        }
      }
    }
  }
  sat.commit(a.sat);
}


// =====================================================================================================
// Fused full-resolution ends of the 16x networks.  The first two encoder layers (conv11 3->16, conv12 16->16
// + pool: model_cd.py:726-728) and the last two decoder layers (conv12 16->16 after the upsample, conv11 16->3:
// model_cd.py:291-293) run at full image resolution with only 16 channels: unfused they move 156 B per pixel
// (the 64 B/px intermediate written and read back), fused 28 B per pixel.  The intermediate lives only in LDS,
// already split into f16 hi/lo planes.  The tail's conv12 has the arithmetic and summation order of the unfused kernel; its
// final 16 -> 3 conv runs block-packed (conv_f16_dev.h c3_block_compute: another summation order, fp32 round-off agreement
// with the two layers run separately); the head's conv11 is f16x3 here and exact-fp32 MFMA unfused (3e-6).
// The intermediate's own reflect padding: a halo pixel OUTSIDE the image must hold the intermediate value of its
// mirror pixel (not the first conv evaluated outside the image), so every halo pixel is evaluated at its reflected
// image coordinate -- whose 3x3 input window is inside the staged tile -- and stored at the halo position.
struct HeadArgs {   // conv11 (3->16, conv0 folded) + ReLU -> conv12 (16->16) + ReLU -> 2x2 max-pool, both f16x3
  const float* img; float* out;
  const u32x4* w11; const float* b11; float inv11;   // [K-step][kq][16 couts] x 8 halfs (conv_f16_dev.h l1_conv_group's K layout)
  const u32x4* w12; const float* b12; float inv12;
  int H, W, tiles_x, tiles_y;
  int out_sp;
  unsigned tx_magic;   // tile_div_magic(tiles_x)
  unsigned* sat;
};

// Persistent: a workgroup walks tiles v, v + grid, ...; the image window of the NEXT tile is fetched into registers
// while the current tile is on the matrix cores and written to LDS behind conv12, conv12's weights are staged once.
// conv11 on 16x16x32 MFMAs: the 3 x 27 (split term, window position, channel) products are concatenated along K as 4-half
// "singles" (one window pixel's RGB0, hi or lo plane) -- 27 of 32 singles = four K-steps, a lane's B operand two 8-byte LDS
// reads at fixed per-lane offsets (conv_f16_dev.h l1_lane_offsets).  (fp32 MFMA for this layer cost 9 x 32 issue cycles per 16
// pixels; the first f16x3 form -- the three terms kept apart, two half-empty K-steps each -- 6 x 16; this one 4 x 16.)
#ifdef WCT_HEAD_TIMING   // tools/experiments/head_timing.sh: shader-clock cycles per phase, summed over wave 0 of every workgroup
__device__ unsigned long long g_head_t[8];
#define HT_STAMP(i) do { if (tid == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         ht[i] += t_ - tl; tl = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define HT_STAMP(i)
#endif

// TH: tile height.  8: 4 waves, three workgroups per CU (22 halo groups over 4 waves: 6 slots each).  24: 12 waves, one workgroup
// per CU -- the same three waves per SIMD -- with conv11's halo recompute 884 / 768 = 1.15 instead of 340 / 256 = 1.33 and 56
// groups over 12 waves: 5 slots each.  (12 and 16 were measured: 6-wave workgroups land unevenly on the 4 SIMDs, -36 %; 8 waves
// are two per SIMD, -12 %.)
template <int TH>
struct HeadGeo {
  static constexpr int NT = 32 * TH, NWV = TH / 2, NPH = nph(TH), NPP = npp(TH), NGRP = (NPH + 15) / 16, NG = (NGRP + NWV - 1) / NWV;
  static constexpr int NPI = I2W * (TH + 4), IMGE = NPI + 4;
  static constexpr int PER_CU = TH == 8 ? 3 : (TH == 12 ? 2 : 1);
  static constexpr size_t lds = (size_t)2 * IMGE * 8 + (size_t)4 * NPP * 16 + 640 * 16;   // 39.7 KB at TH = 8, 83.8 KB at 24
};

template <int TH>
__global__ __launch_bounds__(32 * TH, HeadGeo<TH>::PER_CU) void enc_head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = HeadGeo<TH>;
  constexpr int NT = G::NT, NWV = G::NWV, NPP = G::NPP, NPH = G::NPH, NG = G::NG, NPI = G::NPI, IMGE = G::IMGE;
  u32x2* imgH = reinterpret_cast<u32x2*>(smem);                // [IMGE] RGB0 hi
  u32x2* imgL = imgH + IMGE;                                   // [IMGE] RGB0 lo
  u32x4* act = reinterpret_cast<u32x4*>(imgL + IMGE);          // [4][NPP]
  u32x4* wgt = act + 4 * NPP;                                  // [10][2][2][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  for (int e = tid; e < 40 * 16; e += NT) wgt[e] = a.w12[e];
  if (tid < 4) { imgH[NPI + tid] = u32x2{0u, 0u}; imgL[NPI + tid] = u32x2{0u, 0u}; }
  f16x8 a11[4];      // the four K-steps of conv11 (conv_f16_dev.h: 27 (term, window position) singles concatenated along K)
#pragma unroll
  for (int s = 0; s < 4; ++s) a11[s] = __builtin_bit_cast(f16x8, a.w11[(s * 4 + kq) * 16 + li]);
  const f32x4 bias11 = *reinterpret_cast<const f32x4*>(a.b11 + 4 * kq);
  const f32x4 bias12 = *reinterpret_cast<const f32x4*>(a.b12 + 4 * kq);
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  int boff[4][2];    // this lane's two B-operand singles per K-step: window position + plane, BYTES relative to the window's top-left
  l1_lane_offsets(kq, IMGE, boff);

  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + NT * k;
    e = e < NPI ? e : NPI - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  // the wave's NG 16-pixel groups of the 34 x (TH + 2) halo (TH = 8: group 5 exists for waves 0 and 1 only, 22 groups)
  int gpix[NG], gpy[NG], gpx[NG];
  bool gok[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int pixr = (wave + NWV * u) * 16 + li;
    gok[u] = pixr < NPH;
    gpix[u] = gok[u] ? pixr : NPH - 1;
    gpy[u] = gpix[u] / FHW;
    gpx[u] = gpix[u] - gpy[u] * FHW;
    if (!gok[u]) gpix[u] = NPP - 1;   // store target of a lane without a pixel: a slot nobody reads, so that the stores need no exec masking
  }
  static_assert(NPP > NPH, "a spare slot behind the halo");

  float pxr[2][3];
  SatTrack sat;
  int v = blockIdx.x;
  int ty0 = 0, tx0 = 0;            // origin of the current tile (uniform; carried from the previous trip's look-ahead)
  if (v < ntiles) {
    int tr, tc;
    tile_rc(xcd_swizzle(v, ntiles), a.tiles_x, a.tx_magic, tr, tc);
    ty0 = tr * TH; tx0 = tc * FTW;
    head_fetch<TH>(a.img, a.H, a.W, pxr, soff, ty0, tx0, tid);
    head_commit<TH>(pxr, imgH, imgL, tid, sat);
  }
  // per-lane part of the pooled output address: pixel li >> 1 of the half-tile, channels 4 kq .. 4 kq + 3
  const int out_lane = a.out_sp ? (li >> 1) * 64 + (kq >> 1) * 32 + (kq & 1) * 8 : ((li >> 1) * 16 + 4 * kq) * 4;
#ifdef WCT_HEAD_TIMING
  unsigned long long ht[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#endif
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    __syncthreads();   // image window of this tile is in LDS; every wave is done with the previous tile's planes
    HT_STAMP(0);
    const int vn = v + gridDim.x;
    int nty0 = 0, ntx0 = 0;
    if (vn < ntiles) {
      int tr, tc;
      tile_rc(xcd_swizzle(vn, ntiles), a.tiles_x, a.tx_magic, tr, tc);
      nty0 = tr * TH; ntx0 = tc * FTW;
      head_fetch<TH>(a.img, a.H, a.W, pxr, soff, nty0, ntx0, tid);
    }
    HT_STAMP(1);
    const bool interior = tile_interior_h(ty0, tx0, a.H, a.W, TH);
    // ---- conv11 on the halo pixels, three 16-pixel groups in flight per wave
#pragma unroll
    for (int i = 0; i < NG; i += 3) {
      f32x4 acc[3];
      f16x8 bs[3][4];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (i + u >= NG) continue;
        int base;
        if (interior) {
          base = gpy[i + u] * I2W + gpx[i + u];
        } else {
          const int iy = reflect_clamp(ty0 - 1 + gpy[i + u], a.H) - (ty0 - 2), ix = reflect_clamp(tx0 - 1 + gpx[i + u], a.W) - (tx0 - 2);
          base = (iy - 1) * I2W + ix - 1;
        }
        acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* bp = reinterpret_cast<const char*>(imgH) + base * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x2 r0 = *reinterpret_cast<const u32x2*>(bp + boff[s][0]), r1 = *reinterpret_cast<const u32x2*>(bp + boff[s][1]);
          bs[u][s] = __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (i + u < NG) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11[s], bs[u][s], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (i + u >= NG) continue;
        store_split4<true>(act, NPP, gpix[i + u], kq, fma4(acc[u], a.inv11, bias11), sat);
      }
    }
    HT_STAMP(2);
    __syncthreads();
    HT_STAMP(3);
    // ---- conv12 + ReLU + 2x2 max-pool (the arithmetic of conv3x3_f16_c16_kernel<POOL>)
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    c16_compute<NPP>(act, wgt, wave, li, kq, acc);
    HT_STAMP(4);
    const int oy = (ty0 >> 1) + wave;                                   // uniform
    char* orow = reinterpret_cast<char*>(a.out) + ((size_t)oy * Wp + (tx0 >> 1)) * 64;   // uniform: SALU address arithmetic
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = fmaxf(acc[0][h][r], acc[1][h][r]);
        m[r] = fmaxf(x, lane_xor1(x));
      }
      m = fma4(m, a.inv12, bias12);
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], 0.f);
      const int ox = (tx0 >> 1) + h * 8 + (li >> 1);
      if (!(li & 1) && oy < Hp && ox < Wp) {
        char* dst = orow + h * 8 * 64 + out_lane;
        if (a.out_sp) {
          u32x2 hi, lo;
          split4(m, hi, lo, sat, true);
          *reinterpret_cast<u32x2*>(dst) = hi;
          *reinterpret_cast<u32x2*>(dst + 16) = lo;
        } else {
          *reinterpret_cast<f32x4*>(dst) = m;
        }
      }
    }
    HT_STAMP(5);
    if (vn < ntiles) { head_pin(pxr); head_commit<TH>(pxr, imgH, imgL, tid, sat); }   // conv11 of this tile is behind the barrier above
    HT_STAMP(6);
    ty0 = nty0; tx0 = ntx0;
  }
  sat.commit(a.sat);
#ifdef WCT_HEAD_TIMING
  if (tid == 0) { for (int i = 0; i < 7; ++i) atomicAdd(&g_head_t[i], ht[i]); atomicAdd(&g_head_t[7], 1ull); }
#endif
}

// ---- the fused head with TWO ROLES per workgroup (round 3).
// enc_head_kernel above fits an ADDITIVE model: per SIMD and tile trip ~3 k cycles of VALU issue + ~4 k cycles of matrix pipe, and
// ~82 % of their SUM is what a trip takes.  A probe (tools/experiments/valu_mfma_overlap.hip) shows that VALU work and MFMA work
// overlap almost perfectly when they come from DIFFERENT waves of a SIMD (388 ns against 247 / 353 ns alone), while waves that
// each alternate a VALU-heavy and an MFMA-heavy section fall into lockstep and get ~10 % (579 against 260 + 386) -- and there two
// barriers per tile put all twelve waves into the same section by construction.  Here the sections are roles:
//   producer waves 0..NWA-1        image window -> conv11 (4 MFMAs per 16 halo pixels) -> scale, bias, ReLU, split (VALU) -> act[t + 1]
//   consumer waves NWA..NWA+TH/2-1 act[t] -> conv12 (60 MFMAs per wave) -> pool, bias, ReLU (+ split) -> global
// on tiles t + 1 and t at the same time: the conv11 intermediate and the split image window are double-buffered in LDS and ONE
// barrier per tile hands a finished intermediate to the consumers and a committed window to the producers.  CFETCH: the consumers
// (matrix-bound, idle VALU) fetch and split the image window instead of the producers.  The arithmetic per pixel is that of
// enc_head_kernel (same device functions): bit-identical results.
// (Also measured: all twelve waves of a 32 x 24 tile doing both sections, double-buffered, one barrier, waves 4-7 taking the
// sections in the opposite order -- every wave's work as in enc_head_kernel<24>: 5 % SLOWER than enc_head_kernel<24>.)
template <int TH, int NWA_>
struct HeadRolesGeo {
  static constexpr int NWA = NWA_, NWB = TH / 2, NT = 64 * (NWA + NWB), NTA = 64 * NWA, NTB = 64 * NWB;
  static constexpr int NPH = nph(TH), NPP = npp(TH), NGRP = (NPH + 15) / 16, NG = (NGRP + NWA - 1) / NWA;
  static constexpr int NPI = I2W * (TH + 4), IMGE = NPI + 4;
  static constexpr size_t lds = (size_t)2 * 2 * IMGE * 8 + (size_t)2 * 4 * NPP * 16 + 640 * 16;             // 113.4 KB at TH = 16
};

template <int TH, int NWA, bool CFETCH, bool WREG = true>
__global__ __launch_bounds__(64 * (NWA + TH / 2), 1) void enc_head_roles_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = HeadRolesGeo<TH, NWA>;
  constexpr int NT = G::NT, NTA = G::NTA, NTB = G::NTB, NPP = G::NPP, NPH = G::NPH, NG = G::NG, NPI = G::NPI, IMGE = G::IMGE;
  constexpr int NTF = CFETCH ? NTB : NTA, SL = (NPI + NTF - 1) / NTF;     // threads that stage the image window, pixels per thread
  u32x2* img0 = reinterpret_cast<u32x2*>(smem);                // [2 buffers][hi | lo][IMGE]
  u32x4* act0 = reinterpret_cast<u32x4*>(img0 + 4 * IMGE);    // [2 buffers][4][NPP]
  u32x4* wgt = act0 + 2 * 4 * NPP;                             // [10][2][2][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  for (int e = tid; e < 40 * 16; e += NT) wgt[e] = a.w12[e];
  if (tid < 16) { const int b = tid >> 3, pl = (tid >> 2) & 1, k = tid & 3; img0[(b * 2 + pl) * IMGE + NPI + k] = u32x2{0u, 0u}; }
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  SatTrack sat;
  int v = blockIdx.x;
  const int grid = gridDim.x;
  const bool producer = wave < NWA;
  // image-window staging (by the producers, or by the consumers with CFETCH)
  const int ftid = CFETCH ? tid - NTA : tid;
  int soff[SL];
#pragma unroll
  for (int k = 0; k < SL; ++k) {
    int e = ftid + NTF * k;
    e = (e >= 0 && e < NPI) ? e : NPI - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  float pxr[SL][3];
  auto origin = [&](int tile, int& ty0, int& tx0) {
    int tr, tc;
    tile_rc(xcd_swizzle(tile, ntiles), a.tiles_x, a.tx_magic, tr, tc);
    ty0 = tr * TH; tx0 = tc * FTW;
  };
  auto fetch = [&](int tile) {
    int ty0, tx0;
    origin(tile, ty0, tx0);
    const size_t plane = (size_t)a.H * a.W;
    if (tile_interior_h(ty0, tx0, a.H, a.W, TH)) {
      const float* base = a.img + (size_t)(ty0 - 2) * a.W + (tx0 - 2);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < SL; ++k) pxr[k][c] = (base + c * plane)[(unsigned)soff[k]];
    } else {
#pragma unroll
      for (int k = 0; k < SL; ++k) {
        int e = ftid + NTF * k;
        e = e < NPI ? e : NPI - 1;
        const int py = e / I2W, px = e - py * I2W;
        const size_t off = (size_t)reflect_clamp(ty0 - 2 + py, a.H) * a.W + reflect_clamp(tx0 - 2 + px, a.W);
#pragma unroll
        for (int c = 0; c < 3; ++c) pxr[k][c] = a.img[c * plane + off];
      }
    }
  };
  auto commit = [&](u32x2* imgH) {     // imgL = imgH + IMGE
#pragma unroll
    for (int k = 0; k < SL; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(pxr[k][c]));      // head_pin: the conversions stay behind this trip's MFMAs
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const int e = ftid + NTF * k;
      if (e < NPI) {
        u32x2 h, l;
        { const HiLo t_ = split2(clamp_pm(pxr[k][0]), clamp_pm(pxr[k][1])); h[0] = t_.hi; l[0] = t_.lo; }
        { const HiLo t_ = split2(clamp_pm(pxr[k][2]), 0.f); h[1] = t_.hi; l[1] = t_.lo; }
        sat.note_hi(h[0], h[1], false);
        imgH[e] = h;
        imgH[IMGE + e] = l;
      }
    }
  };
  const bool stager = CFETCH ? !producer : producer;
  int vn = v + grid;

  if (producer) {
#ifdef WCT_HEAD_PRIO
    __builtin_amdgcn_s_setprio(WCT_HEAD_PRIO);     // experiment: the producer wave of a SIMD wins issue arbitration
#endif
    f16x8 a11[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a11[s] = __builtin_bit_cast(f16x8, a.w11[(s * 4 + kq) * 16 + li]);
    const f32x4 bias11 = *reinterpret_cast<const f32x4*>(a.b11 + 4 * kq);
    int boff[4][2];
    l1_lane_offsets(kq, IMGE, boff);
    // the wave's NG 16-pixel groups of the 34 x (TH + 2) halo: store slot, and the window base of an interior tile
    int gpix[NG], gbase[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int pixr = (wave + NWA * u) * 16 + li;
      const bool ok = pixr < NPH;
      const int pc = ok ? pixr : NPH - 1;
      const int py = pc / FHW, px = pc - py * FHW;
      gbase[u] = py * I2W + px;
      gpix[u] = ok ? pixr : NPP - 1;      // a lane without a pixel stores to a slot nobody reads: no exec masking around the stores
    }
    static_assert(NPP > NPH, "a spare slot behind the halo");
    auto conv11 = [&](int tile, const u32x2* imgH, u32x4* act) {
      int ty0, tx0;
      origin(tile, ty0, tx0);
      const bool interior = tile_interior_h(ty0, tx0, a.H, a.W, TH);
#pragma unroll
      for (int i = 0; i < NG; i += 3) {
        f32x4 acc[3];
        f16x8 bs[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (i + u >= NG) continue;
          int base = gbase[i + u];
          if (!interior) {
            const int gpy = base / I2W, gpx = base - gpy * I2W;
            const int iy = reflect_clamp(ty0 - 1 + gpy, a.H) - (ty0 - 2), ix = reflect_clamp(tx0 - 1 + gpx, a.W) - (tx0 - 2);
            base = (iy - 1) * I2W + ix - 1;
          }
          acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          const char* bp = reinterpret_cast<const char*>(imgH) + base * 8;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const u32x2 r0 = *reinterpret_cast<const u32x2*>(bp + boff[s][0]), r1 = *reinterpret_cast<const u32x2*>(bp + boff[s][1]);
            bs[u][s] = __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (i + u < NG) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11[s], bs[u][s], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (i + u >= NG) continue;
          store_split4<true>(act, NPP, gpix[i + u], kq, fma4(acc[u], a.inv11, bias11), sat);
        }
      }
    };
    // prologue: window of the first tile, its intermediate, window of the second tile
    if (stager && v < ntiles) { fetch(v); commit(img0); }
    __syncthreads();
    if (v < ntiles) conv11(v, img0, act0);
    if (stager && vn < ntiles) { fetch(vn); commit(img0 + 2 * IMGE); }
    settle_preloop_loads();
    __syncthreads();
    for (int it = 0; v < ntiles; v = vn, vn += grid, ++it) {
      // act[it & 1] holds tile v (the consumers are on it), img[(it + 1) & 1] holds the window of tile vn
      const int vnn = vn + grid, cur = it & 1, nxt = cur ^ 1;
      if (stager && vnn < ntiles) fetch(vnn);
      if (vn < ntiles) conv11(vn, img0 + nxt * 2 * IMGE, act0 + nxt * 4 * NPP);
      if (stager && vnn < ntiles) commit(img0 + cur * 2 * IMGE);
      __syncthreads();
    }
  } else {
    const int bw = wave - NWA;
    const f32x4 bias12 = *reinterpret_cast<const f32x4*>(a.b12 + 4 * kq);
    const int out_lane = a.out_sp ? (li >> 1) * 64 + (kq >> 1) * 32 + (kq & 1) * 8 : ((li >> 1) * 16 + 4 * kq) * 4;
    C16Weights cw;       // WREG: conv12's weight operands stay in registers (the consumers have them to spare)
    if constexpr (WREG) c16_load_weights(a.w12, li, kq, cw);
    if (stager && v < ntiles) { fetch(v); commit(img0); }
    __syncthreads();
    if (stager && vn < ntiles) { fetch(vn); commit(img0 + 2 * IMGE); }
    settle_preloop_loads();
    __syncthreads();
    for (int it = 0; v < ntiles; v = vn, vn += grid, ++it) {
      const int vnn = vn + grid, cur = it & 1;
      if (stager && vnn < ntiles) fetch(vnn);
      int ty0, tx0;
      origin(v, ty0, tx0);
      const u32x4* act = act0 + cur * 4 * NPP;
      f32x4 acc[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (WREG) c16_compute_w<NPP>(act, cw, bw, li, kq, acc);
      else c16_compute<NPP>(act, wgt, bw, li, kq, acc);
      const int oy = (ty0 >> 1) + bw;                                   // uniform
      char* orow = reinterpret_cast<char*>(a.out) + ((size_t)oy * Wp + (tx0 >> 1)) * 64;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = fmaxf(acc[0][h][r], acc[1][h][r]);
          m[r] = fmaxf(x, lane_xor1(x));
        }
        m = fma4(m, a.inv12, bias12);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], 0.f);
        const int ox = (tx0 >> 1) + h * 8 + (li >> 1);
        if (!(li & 1) && oy < Hp && ox < Wp) {
          char* dst = orow + h * 8 * 64 + out_lane;
          if (a.out_sp) {
            u32x2 hi, lo;
            split4(m, hi, lo, sat, true);
            *reinterpret_cast<u32x2*>(dst) = hi;
            *reinterpret_cast<u32x2*>(dst + 16) = lo;
          } else {
            *reinterpret_cast<f32x4*>(dst) = m;
          }
        }
      }
      if (stager && vnn < ntiles) commit(img0 + cur * 2 * IMGE);
      __syncthreads();
    }
  }
  sat.commit(a.sat);
}

struct TailArgs {   // conv12 (16->16 on the nearest-x2 upsampled input) + ReLU -> conv11 (16->3) + ReLU -> planar image
  const float* in; float* out;
  const u32x4* w12; const float* b12; float inv12; const float* inv12_ptr;
  const u32x4* w11; const float* b11; float inv11;   // w11: block-packed (c3_block_compute), PH_WSLOTS slots
  int H, W, inW, up_in, tiles_x, tiles_y;
  int in_sp;
  unsigned* sat;
  const u32x4* w12u; float inv12u;   // conv12 behind the upsample as per-parity 2x2 weights (ConvDesc::wup16), dec_tail_up_kernel
};

// Tile geometry of the fused tail for a tile of 32 x TH output pixels (TH / 2 waves, 32 * TH threads): TH = 8 -> 4 waves, two
// workgroups per CU; TH = 16 -> 8 waves, one workgroup per CU, conv12's halo recompute 612 / 512 = 1.20 instead of 340 / 256 = 1.33
template <int TH>
struct TailGeo {
  static constexpr int NT = 32 * TH, NWV = TH / 2;
  static constexpr int HROWS = TH + 2, NPH = FHW * HROWS, NGRP = (NPH + 15) / 16, NG = (NGRP + NWV - 1) / NWV;
  static constexpr int I2HT = TH + 4, NPI = I2W * I2HT;                      // input window 36 x (TH + 4): 432 / 720 pixels
  static constexpr int SL = (NPI * 2 + NT - 1) / NT;                         // register slots of 8 channels per thread: 4 / 3
  static constexpr int NPX = (HROWS * PH_W + 15) / 16 * 16;                  // pair-major slots of the conv12 output: 368 / 656
  static_assert(NPI % 16 == 0 && (NPI * 16) % 256 == 0 && (NPX * 16) % 256 == 0, "plane strides");
};
template <int TH> struct TailRegs { f32x4 v0[TailGeo<TH>::SL], v1[TailGeo<TH>::SL]; };
// thread slot e -> (window pixel, 8-channel half).  The half sits on bit 3, not bit 0 (rounds 2-4): the 8 contiguous lanes of a
// ds_write_b128 group then write 8 consecutive 16-byte slots of ONE plane (128 contiguous bytes, conflict-free) instead of 4 slots in
// each of two planes whose stride is a multiple of the 128-byte bank row (2-way conflict on every window store).  The global side
// still reads whole 64-byte pixel records per wave-instruction (the two halves of a pixel are 8 lanes apart).  Pixel counts are
// multiples of 8 (or padded to one), so this is a bijection onto [0, pixels) x {0, 1}.
__device__ __forceinline__ int tail_slot_pix(int e) { return (e & 7) | ((e >> 4) << 3); }
__device__ __forceinline__ int tail_slot_half(int e) { return (e >> 3) & 1; }

// soff[k]: tile-independent element offset of slot k from the window origin, valid for interior tiles (the window
// origin (ty0 - 2, tx0 - 2) is even, so the nearest-x2 shift distributes over origin + offset)
template <int TH>
__device__ __forceinline__ void tail_fetch(const TailArgs& a, unsigned txm, TailRegs<TH>& r, const int (&soff)[TailGeo<TH>::SL], int tile, int tid) {
  using G = TailGeo<TH>;
  int trow_, tcol_;
  tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
  const int ty0 = trow_ * TH, tx0 = tcol_ * FTW;
  if (tile_interior_h(ty0, tx0, a.H, a.W, TH)) {
    const float* base = a.in + ((size_t)((ty0 - 2) >> a.up_in) * a.inW + ((tx0 - 2) >> a.up_in)) * 16;
#pragma unroll
    for (int k = 0; k < G::SL; ++k) {
      r.v0[k] = *reinterpret_cast<const f32x4*>(base + soff[k]);
      r.v1[k] = *reinterpret_cast<const f32x4*>(base + soff[k] + 4);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < G::SL; ++k) {
    int e = tid + G::NT * k;
    e = e < G::NPI * 2 ? e : G::NPI * 2 - 1;
    const int h2 = tail_slot_half(e), pix = tail_slot_pix(e);
    const int py = pix / I2W, px = pix - py * I2W;
    int gy = reflect_clamp(ty0 - 2 + py, a.H), gx = reflect_clamp(tx0 - 2 + px, a.W);
    if (a.up_in) { gy >>= 1; gx >>= 1; }
    const float* src = a.in + ((size_t)gy * a.inW + gx) * 16 + h2 * 8;
    r.v0[k] = *reinterpret_cast<const f32x4*>(src);
    r.v1[k] = *reinterpret_cast<const f32x4*>(src + 4);
  }
}

template <int TH>
__device__ __forceinline__ void tail_commit(const TailRegs<TH>& r, u32x4* act0, int tid, int in_sp, SatTrack& sat) {
  using G = TailGeo<TH>;
#pragma unroll
  for (int k = 0; k < G::SL; ++k) {
    const int e = tid + G::NT * k;
    if (e < G::NPI * 2) {
      f16x8 hi, lo;
      if (in_sp) { hi = __builtin_bit_cast(f16x8, r.v0[k]); lo = __builtin_bit_cast(f16x8, r.v1[k]); }
      else split8(r.v0[k], r.v1[k], hi, lo, sat);
      act0[(0 * 2 + tail_slot_half(e)) * G::NPI + tail_slot_pix(e)] = __builtin_bit_cast(u32x4, hi);
      act0[(1 * 2 + tail_slot_half(e)) * G::NPI + tail_slot_pix(e)] = __builtin_bit_cast(u32x4, lo);
    }
  }
}

// Persistent like enc_head_kernel: both weight slabs are staged once per workgroup, the 36 x (TH + 4) x 16 input window of
// the NEXT tile is fetched into registers while this tile is on the matrix cores and split into LDS behind conv11.
// Everything that depends only on the thread (slot offsets, halo-group pixel indices) is computed once; only tiles
// that touch the image border recompute their reflected coordinates.
template <int TH>
__global__ __launch_bounds__(32 * TH, TH == 8 ? 2 : 1) void dec_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = TailGeo<TH>;
  constexpr int NT = G::NT, NWV = G::NWV, NPH = G::NPH, NG = G::NG, NPI = G::NPI, NPX = G::NPX, SL = G::SL;
  u32x4* act0 = reinterpret_cast<u32x4*>(smem);   // [4][NPI]  input of conv12 (two halo rings)
  u32x4* wg12 = act0 + 4 * NPI;                   // [640]
  u32x4* wg11 = wg12 + 640;                       // [PH_WSLOTS]  block-packed 16 -> 3 weights
  u32x4* act1 = wg11 + PH_WSLOTS;                 // [4][NPX]  conv12 output on the 34 x (TH + 2) halo, pair-major slots (ph_slot)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4, kh = kq & 1, ts = kq >> 1;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);   // once per workgroup; tile_rc() then stays on the scalar unit
  for (int e = tid; e < 640; e += NT) wg12[e] = a.w12[e];
  for (int e = tid; e < PH_WSLOTS; e += NT) wg11[e] = a.w11[e];
  const float inv12 = a.inv12_ptr ? *a.inv12_ptr : a.inv12;
  const f32x4 bias12 = *reinterpret_cast<const f32x4*>(a.b12 + 4 * kq);
  const f32x4 bias11 = *reinterpret_cast<const f32x4*>(a.b11);
  const size_t plane = (size_t)a.H * a.W;

  int soff[SL];
#pragma unroll
  for (int k = 0; k < SL; ++k) {
    int e = tid + NT * k;
    e = e < NPI * 2 ? e : NPI * 2 - 1;
    const int pix = tail_slot_pix(e), py = pix / I2W, px = pix - py * I2W;
    soff[k] = ((py >> a.up_in) * a.inW + (px >> a.up_in)) * 16 + tail_slot_half(e) * 8;
  }
  // the wave's NG 16-pixel groups of the 34 x (TH + 2) halo (22 groups over 4 waves / 39 over 8: the last ones may not exist)
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
  static_assert(NPX > G::HROWS * PH_W, "a spare slot behind the halo");

  TailRegs<TH> tr;
  SatTrack sat;
  int v = blockIdx.x;
  if (v < ntiles) {
    tail_fetch<TH>(a, txm, tr, soff, xcd_swizzle(v, ntiles), tid);
    tail_commit<TH>(tr, act0, tid, a.in_sp, sat);
  }
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * TH, tx0 = tcol_ * FTW;
    __syncthreads();   // act0 of this tile is in LDS; every wave is done with the previous tile's act1
    const int vn = v + gridDim.x;
    if (vn < ntiles) tail_fetch<TH>(a, txm, tr, soff, xcd_swizzle(vn, ntiles), tid);
    // ---- conv12 on the halo pixels (evaluated at their reflected image coordinates), into act1.
    //      All the wave's groups are in flight together (tap loop outermost): the operand reads of the next tap pair overlap
    //      the MFMAs of this one.
    {
      f32x4 acc[NG];
      int sp0[NG];
      if (tile_interior_h(ty0, tx0, a.H, a.W, TH)) {
#pragma unroll
        for (int u = 0; u < NG; ++u) sp0[u] = gpy[u] * I2W + gpx[u];
      } else {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          const int iy = reflect_clamp(ty0 - 1 + gpy[u], a.H) - (ty0 - 2), ix = reflect_clamp(tx0 - 1 + gpx[u], a.W) - (tx0 - 2);
          sp0[u] = (iy - 1) * I2W + ix - 1;
        }
      }
#pragma unroll
      for (int u = 0; u < NG; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + ts;
        const int tc = tap > 8 ? 8 : tap;
        const int dy = tc / 3, dx = tc - dy * 3;
        const f16x8 ah = __builtin_bit_cast(f16x8, wg12[((tap * 2 + 0) * 2 + kh) * 16 + li]);
        const f16x8 al = __builtin_bit_cast(f16x8, wg12[((tap * 2 + 1) * 2 + kh) * 16 + li]);
        f16x8 bh[NG], bl[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          const int sp = sp0[u] + dy * I2W + dx;
          bh[u] = __builtin_bit_cast(f16x8, act0[(0 * 2 + kh) * NPI + sp]);
          bl[u] = __builtin_bit_cast(f16x8, act0[(1 * 2 + kh) * NPI + sp]);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int u = 0; u < NG; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? al : ah, term == 1 ? bl[u] : bh[u], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        store_split4<true>(act1, NPX, gslot[u], kq, fma4(acc[u], inv12, bias12), sat);
      }
    }
    __syncthreads();
    // ---- conv11 (16 -> 3) + ReLU -> planar output, block-packed (conv_f16_dev.h): lane (li, kq) holds pixel (2 wave + (kq >> 1), 2 li + (kq & 1))
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    c3_block_compute<NPX>(act1, wg11, wave, li, kq, acc);
    c3_block_store(acc, a.inv11, bias11, a.out, plane, ty0, tx0, wave, li, kq, a.H, a.W);
    if (vn < ntiles) tail_commit<TH>(tr, act0, tid, a.in_sp, sat);   // conv12 of this tile is behind the barrier above
  }
  sat.commit(a.sat);
}

// ---- the same tail with conv12 evaluated on the LOW-RESOLUTION input (up_in = 1 only).
// conv12 runs behind the nearest-x2 upsample, so each output parity (a, b) is a 2x2 convolution of the low-resolution map with summed
// taps (wct_api.hip pack_up_phase_f16): K = 2 x 2 x 16 = 64 = two 16x16x32 K-steps (tap row i; kq -> column j = kq >> 1, channels
// 8 (kq & 1) ..) = 6 MFMAs and 4 operand reads per 16 halo pixels instead of 15 and 10, and the staged window is the 18 x (TH / 2 + 2)
// low-resolution pixels the tile touches instead of their 36 x (TH + 4) fourfold copies (16 KB instead of 64.5 KB at TH = 24).
// A 16-pixel group must share its parity: the 34 x (TH + 2) halo is cut into row groups g = 2 row + c (pixels x = 2 li + 1 - c: every
// second column) plus four groups for the two leftover columns (33 - c, rows 2 li + r); with g dealt to wave g mod NWV (NWV and the
// number of row groups are multiples of 4) ALL groups of a wave have the parity a = ((wave >> 1) & 1) ^ 1, b = wave & 1, so the
// wave keeps its phase's weights in 16 registers and conv12 reads no weights from LDS at all.  Reflect padding of the upsampled
// map = clamping of the low-resolution coordinates; a halo pixel outside the image is evaluated at its reflected coordinate,
// which has the same parity.  Sums of taps are formed in double on the host: fp32 round-off agreement with the 9-tap form.
template <int TH>
struct TailUpGeo {
  static constexpr int NT = 32 * TH, NWV = TH / 2;
  static constexpr int HROWS = TH + 2, NROWG = 2 * HROWS, NLEFT = 4 * ((HROWS / 2 + 15) / 16), NGRP = NROWG + NLEFT, NG = (NGRP + NWV - 1) / NWV;
  static constexpr int LW = FTW / 2 + 2, LH = TH / 2 + 2, NPL = (LW * LH + 15) / 16 * 16;     // low-resolution window 18 x (TH / 2 + 2): 256 / 192 / 112 slots
  static constexpr int NPX = (HROWS * PH_W + 15) / 16 * 16;
  static_assert(NWV % 4 == 0 && NROWG % 4 == 0 && 2 * ((LW * LH + 7) / 8 * 8) <= NT, "phase-uniform waves; one (pixel, channel half) slot per thread");
  // (the two leftover columns' rows of one parity: HROWS / 2 pixels per (column, parity) = one group of 16, or two at TH = 32)
  static constexpr size_t lds = ((size_t)4 * NPL + PH_WSLOTS + (size_t)4 * NPX) * 16;           // 93.2 / 66.6 / 47.1 KB
};

template <int TH>
__global__ __launch_bounds__(32 * TH, TH <= 16 ? 2 : 1) void dec_tail_up_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = TailUpGeo<TH>;
  constexpr int NT = G::NT, NWV = G::NWV, NG = G::NG, NPL = G::NPL, NPX = G::NPX, LW = G::LW, LH = G::LH;
  u32x4* act0 = reinterpret_cast<u32x4*>(smem);   // [4][NPL]  low-resolution input window, planes (hl, channel half)
  u32x4* wg11 = act0 + 4 * NPL;                   // [PH_WSLOTS]  block-packed 16 -> 3 weights
  u32x4* act1 = wg11 + PH_WSLOTS;                 // [4][NPX]  conv12 output on the 34 x (TH + 2) halo, pair-major slots (ph_slot)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4, kh = kq & 1, ts = kq >> 1;
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);
  for (int e = tid; e < PH_WSLOTS; e += NT) wg11[e] = a.w11[e];
  const int pa = ((wave >> 1) & 1) ^ 1, pb = wave & 1;          // the parity of every halo group of this wave
  f16x8 wq[2][2];                                                // [tap row][hi / lo] of this phase
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) wq[i][hl] = __builtin_bit_cast(f16x8, a.w12u[((((pa * 2 + pb) * 2 + i) * 2 + hl) * 4 + kq) * 16 + li]);
  const f32x4 bias12 = *reinterpret_cast<const f32x4*>(a.b12 + 4 * kq);
  const f32x4 bias11 = *reinterpret_cast<const f32x4*>(a.b11);
  const size_t plane = (size_t)a.H * a.W;
  const int inH = a.H >> 1;

  // this thread's slot of the low-resolution window: (pixel, channel half) = tail_slot_*(tid) over the pixel count padded to a multiple
  // of 8 (pixels beyond the window: a clamped fetch, stored to spare slots behind the window that nobody reads)
  constexpr int NWPIX = LW * LH, NWPAD = (NWPIX + 7) / 8 * 8;
  static_assert(2 * NWPAD <= NT && NWPAD <= NPL, "one (pixel, channel half) slot per thread, spare slots inside the plane");
  const int fe = tid < 2 * NWPAD ? tid : 2 * NWPAD - 1;
  const int fslot = tail_slot_pix(fe), fpix = fslot < NWPIX ? fslot : NWPIX - 1, fpy = fpix / LW, fpx = fpix - fpy * LW, fh = tail_slot_half(fe);
  const int soff = (fpy * a.inW + fpx) * 16 + fh * 8;
  // the wave's halo groups
  int gpy[NG], gpx[NG], gslot[NG], gbase[NG];
  bool gok[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int g = wave + NWV * u;
    if (g < G::NROWG) { gpy[u] = g >> 1; gpx[u] = 2 * li + 1 - (g & 1); gok[u] = true; }
    else { const int idx = g - G::NROWG, h = idx & 3; gpx[u] = 33 - (h & 1); gpy[u] = 2 * (li + 16 * (idx >> 2)) + (h >> 1); gok[u] = g < G::NGRP && gpy[u] < G::HROWS; }
    if (!gok[u]) { gpy[u] = 1; gpx[u] = 1; }   // never stored; any in-window patch
    gslot[u] = gok[u] ? ph_slot(gpy[u], gpx[u]) : NPX - 1;   // lanes without a pixel store to a slot nobody reads (no exec masking)
    // interior tiles: window slot of the 2x2 patch's top-left = ((q >> 1) + parity) per axis, q = halo coordinate - 1
    gbase[u] = (((gpy[u] - 1) >> 1) + pa) * LW + ((gpx[u] - 1) >> 1) + pb;
  }

  f32x4 r0, r1;
  SatTrack sat;
  auto fetch = [&](int tile) {
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * TH, tx0 = tcol_ * FTW;
    const float* src;
    if (tile_interior_h(ty0, tx0, a.H, a.W, TH)) {
      src = a.in + ((size_t)((ty0 >> 1) - 1) * a.inW + ((tx0 >> 1) - 1)) * 16 + (unsigned)soff;
    } else {
      int gy = (ty0 >> 1) - 1 + fpy, gx = (tx0 >> 1) - 1 + fpx;
      gy = gy < 0 ? 0 : (gy >= inH ? inH - 1 : gy);
      gx = gx < 0 ? 0 : (gx >= a.inW ? a.inW - 1 : gx);
      src = a.in + ((size_t)gy * a.inW + gx) * 16 + fh * 8;
    }
    r0 = *reinterpret_cast<const f32x4*>(src);
    r1 = *reinterpret_cast<const f32x4*>(src + 4);
  };
  auto commit = [&]() {
    if (tid < 2 * NWPAD) {
      f16x8 hi, lo;
      if (a.in_sp) { hi = __builtin_bit_cast(f16x8, r0); lo = __builtin_bit_cast(f16x8, r1); }
      else split8(r0, r1, hi, lo, sat);
      act0[(0 * 2 + fh) * NPL + fslot] = __builtin_bit_cast(u32x4, hi);
      act0[(1 * 2 + fh) * NPL + fslot] = __builtin_bit_cast(u32x4, lo);
    }
  };
  int v = blockIdx.x;
  if (v < ntiles) { fetch(xcd_swizzle(v, ntiles)); commit(); }
  settle_preloop_loads();
  for (; v < ntiles; v += gridDim.x) {
    const int tile = xcd_swizzle(v, ntiles);
    int trow_, tcol_;
    tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
    const int ty0 = trow_ * TH, tx0 = tcol_ * FTW;
    __syncthreads();   // the window of this tile is in LDS; every wave is done with the previous tile's act1
    const int vn = v + gridDim.x;
    if (vn < ntiles) fetch(xcd_swizzle(vn, ntiles));
    // ---- conv12 on the halo pixels, from the low-resolution window
    {
      f32x4 acc[NG];
      int sp0[NG];
      if (tile_interior_h(ty0, tx0, a.H, a.W, TH)) {
#pragma unroll
        for (int u = 0; u < NG; ++u) sp0[u] = gbase[u];
      } else {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          // evaluated at the reflected image coordinate (same parity); rows / columns clamped into the staged window
          const int ey = reflect_clamp(ty0 - 1 + gpy[u], a.H), ex = reflect_clamp(tx0 - 1 + gpx[u], a.W);
          int wr = (ey >> 1) - (ty0 >> 1) + pa, wc = (ex >> 1) - (tx0 >> 1) + pb;
          wr = wr < 0 ? 0 : (wr > LH - 2 ? LH - 2 : wr);
          wc = wc < 0 ? 0 : (wc > LW - 2 ? LW - 2 : wc);
          sp0[u] = wr * LW + wc;
        }
      }
#pragma unroll
      for (int u = 0; u < NG; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f16x8 bh[NG], bl[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          const int sp = sp0[u] + i * LW + ts;
          bh[u] = __builtin_bit_cast(f16x8, act0[(0 * 2 + kh) * NPL + sp]);
          bl[u] = __builtin_bit_cast(f16x8, act0[(1 * 2 + kh) * NPL + sp]);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int u = 0; u < NG; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[i][term == 2], term == 1 ? bl[u] : bh[u], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        store_split4<true>(act1, NPX, gslot[u], kq, fma4(acc[u], a.inv12u, bias12), sat);
      }
    }
    __syncthreads();
    // ---- conv11 (16 -> 3) + ReLU -> planar output, block-packed (conv_f16_dev.h): lane (li, kq) holds pixel (2 wave + (kq >> 1), 2 li + (kq & 1))
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    c3_block_compute<NPX>(act1, wg11, wave, li, kq, acc);
    c3_block_store(acc, a.inv11, bias11, a.out, plane, ty0, tx0, wave, li, kq, a.H, a.W);
    if (vn < ntiles) commit();   // conv12 of this tile is behind the barrier above
  }
  sat.commit(a.sat);
}

template <typename K>
hipError_t launch_k(K k, const F16Args& a, size_t lds, int groups, hipStream_t s, int threads = 256) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(a.tiles_x * a.tiles_y, groups), dim3(threads), lds, s, a);
  return hipGetLastError();
}

// ---- weight preparation on the device (folded first decoder conv): fp32 packed [chunk][tap][kq][cout_pad][4]
//      -> scaled (hi, lo) f16 in this file's layout.  `maxbits` holds max|w| as float bits (atomicMax on uint).
__global__ void absmax_kernel(const float* w, long n, unsigned* maxbits) {
  float m = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[e]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(m));
}

// max |w| from one word, or from nmax per-row maxima (launch_fold_fast: no atomics, no memset) -- non-negative floats order like
// their bit patterns
__device__ __forceinline__ float pack_max(const unsigned* maxbits, int nmax) {
  unsigned m = 0u;
  for (int j = 0; j < nmax; ++j) m = max(m, maxbits[j]);    // wave-uniform addresses: scalar / broadcast loads, <= 128 of them
  return __uint_as_float(m);
}

__global__ void split_pack_kernel(const float* wpk32, int chunks, int cout_pad, int taps, const unsigned* maxbits, int nmax,
                                  u32x4* out, float* inv_scale_out) {
  // scale = 2^e with max|w| * scale in [256, 512)
  const float mx = pack_max(maxbits, nmax);
  int ex = 0;
  if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = 9 - ex; }
  ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
  const float scale = ldexpf(1.f, ex);
  const long total = (long)chunks * taps * 2 * 2 * cout_pad;  // 16-B groups
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) *inv_scale_out = ldexpf(1.f, -ex);
  if (e >= total) return;
  long t = e;
  const int co = (int)(t % cout_pad); t /= cout_pad;
  const int kh = (int)(t & 1); t >>= 1;
  const int hl = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % taps);
  const int chunk = (int)(t / taps);
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = 0.f;
    if (tap < 9) {
      const int ci = kh * 8 + j;  // channel within the chunk -> fp32 layout [chunk][tap][kq = ci/4][cout_pad][ci%4]
      x = wpk32[((((size_t)chunk * 9 + tap) * 4 + (ci >> 2)) * cout_pad + co) * 4 + (ci & 3)] * scale;
    }
    const _Float16 h = (_Float16)x;
    v[j] = hl ? (_Float16)(x - (float)h) : h;
  }
  out[e] = __builtin_bit_cast(u32x4, v);
}

// the block-packed form (conv_f16_dev.h c3_block_compute) of a cout_pad-16 layer with 3 real couts, same scale as above
__global__ void split_pack_phase_kernel(const float* wpk32, int chunks, const unsigned* maxbits, int nmax, u32x4* out) {
  const float mx = pack_max(maxbits, nmax);
  int ex = 0;
  if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = 9 - ex; }
  ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
  const float scale = ldexpf(1.f, ex);
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)chunks * PH_WSLOTS) return;
  long t = e;
  const int m = (int)(t & 15); t >>= 4;
  const int kq = (int)(t & 3); t >>= 2;
  const int hl = (int)(t & 1); t >>= 1;
  const int ks = (int)(t & 7);
  const int chunk = (int)(t >> 3);
  // block-packed layout (conv_f16_dev.h c3_block_compute): m = 4 (2 py + px) + cout, window row ks >> 1, column 2 (ks & 1) + (kq >> 1)
  const int co = m & 3, py = m >> 3, px = (m >> 2) & 1, dy = (ks >> 1) - py, dxr = 2 * (ks & 1) + (kq >> 1) - px;
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = 0.f;
    if (co < 3 && dy >= 0 && dy <= 2 && dxr >= 0 && dxr <= 2) {
      const int ci = (kq & 1) * 8 + j, tap = dy * 3 + dxr;   // fp32 layout [chunk][tap][kq = ci / 4][cout_pad = 16][ci % 4]
      x = wpk32[((((size_t)chunk * 9 + tap) * 4 + (ci >> 2)) * 16 + co) * 4 + (ci & 3)] * scale;
    }
    const _Float16 h = (_Float16)x;
    v[j] = hl ? (_Float16)(x - (float)h) : h;
  }
  out[e] = __builtin_bit_cast(u32x4, v);
}

}  // namespace


bool conv_fusable_head(const ConvDesc& d0, const ConvDesc& d1) {
  return (d0.flags & CONV_IN_NCHW3) && !(d0.flags & (CONV_POOL_OUT | CONV_NO_RELU)) && d0.cout == 16 && d0.cout_pad == 16 && d0.wpk16 &&
         d1.cin == 16 && d1.cout == 16 && d1.cout_pad == 16 && (d1.flags & CONV_POOL_OUT) && !(d1.flags & (CONV_UP_IN | CONV_NO_RELU)) &&
         d1.wpk16 && !d1.inv_scale_ptr;
}

bool conv_fusable_tail(const ConvDesc& d0, const ConvDesc& d1) {
  return d0.cin == 16 && d0.cout == 16 && d0.cout_pad == 16 && !(d0.flags & (CONV_POOL_OUT | CONV_NO_RELU | CONV_IN_NCHW3 | CONV_OUT_NCHW3)) &&
         d0.wpk16 && (d1.flags & CONV_OUT_NCHW3) && d1.cin == 16 && d1.cout == 3 && d1.cout_pad == 16 && d1.wpk16 && d1.wph16 &&
         !(d1.flags & (CONV_UP_IN | CONV_NO_RELU)) && !d1.inv_scale_ptr;
}

hipError_t launch_enc_head(const ConvDesc& d0, const ConvDesc& d1, const float* img, float* out, int H, int W, hipStream_t s) {
  if (!conv_fusable_head(d0, d1) || H < 2 || W < 2) return hipErrorInvalidValue;
  HeadArgs a;
  a.img = img; a.out = out;
  a.w11 = reinterpret_cast<const u32x4*>(d0.wpk16); a.b11 = d0.bias; a.inv11 = d0.inv_scale;
  a.w12 = reinterpret_cast<const u32x4*>(d1.wpk16); a.b12 = d1.bias; a.inv12 = d1.inv_scale;
  a.H = H; a.W = W; a.tiles_x = (W + FTW - 1) / FTW;
  a.tx_magic = tile_div_magic(a.tiles_x);
  a.out_sp = (d1.flags & CONV_OUT_SP16) ? 1 : 0;
  a.sat = d1.sat;
  static const int th_env = [] { const char* e = wct_debug_env("WCT_HEAD_TH"); return e ? atoi(e) : 0; }();   // experiment: force 8 / 24
  auto go = [&](auto kern, auto geo, int th) -> hipError_t {
    using G = decltype(geo);
    a.tiles_y = (H + th - 1) / th;
    if (G::lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds);
      if (e != hipSuccess) return e;
    }
    const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < G::PER_CU * num_cus() ? ntiles : G::PER_CU * num_cus();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::lds, s, a);
    return hipGetLastError();
  };
  // 32 x 24 tiles (12 waves, one workgroup per CU: the same three waves per SIMD, conv11's halo recompute 1.15 instead of 1.33:
  // -2.6 %) once they still give every CU four tiles; results do not depend on the tile shape
  const int th = th_env ? th_env : (((H + 23) / 24) * a.tiles_x >= 4 * num_cus() ? 24 : 8);
  // two roles per workgroup (producer waves: conv11, consumer waves: conv12 + pool; 32 x 16 tiles).  WCT_HEAD_ROLES: 0 = off,
  // 1 = 4 producers staging the window themselves, 2 = 4 producers, consumers stage, 3 / 4 = the same with 8 producers (16 waves)
  static const int roles_env = [] { const char* e = wct_debug_env("WCT_HEAD_ROLES"); return e ? atoi(e) : 1; }();
  if (roles_env && !th_env && ((H + 15) / 16) * a.tiles_x >= 4 * num_cus()) {
    a.tiles_y = (H + 15) / 16;
    const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < num_cus() ? ntiles : num_cus();
    auto gor = [&](auto kern, auto geo) -> hipError_t {
      using G = decltype(geo);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::lds, s, a);
      return hipGetLastError();
    };
    if (roles_env == 5) return gor(enc_head_roles_kernel<16, 4, false, false>, HeadRolesGeo<16, 4>{});     // weights from LDS (the first form)
    if (roles_env == 2) return gor(enc_head_roles_kernel<16, 4, true>, HeadRolesGeo<16, 4>{});
    if (roles_env == 3) return gor(enc_head_roles_kernel<16, 8, false>, HeadRolesGeo<16, 8>{});
    if (roles_env == 4) return gor(enc_head_roles_kernel<16, 8, true>, HeadRolesGeo<16, 8>{});
    return gor(enc_head_roles_kernel<16, 4, false>, HeadRolesGeo<16, 4>{});
  }
  if (th == 24) return go(enc_head_kernel<24>, HeadGeo<24>{}, 24);
  return go(enc_head_kernel<8>, HeadGeo<8>{}, 8);
}

hipError_t launch_dec_tail(const ConvDesc& d0, const ConvDesc& d1, const float* in, float* out, int H, int W, hipStream_t s) {
  if (!conv_fusable_tail(d0, d1) || H < 2 || W < 2) return hipErrorInvalidValue;
  TailArgs a;
  a.in = in; a.out = out;
  a.w12 = reinterpret_cast<const u32x4*>(d0.wpk16); a.b12 = d0.bias; a.inv12 = d0.inv_scale; a.inv12_ptr = d0.inv_scale_ptr;
  a.w11 = reinterpret_cast<const u32x4*>(d1.wph16); a.b11 = d1.bias; a.inv11 = d1.inv_scale;
  a.H = H; a.W = W; a.up_in = (d0.flags & CONV_UP_IN) ? 1 : 0; a.inW = a.up_in ? W / 2 : W;
  a.in_sp = (d0.flags & CONV_IN_SP16) ? 1 : 0;
  a.sat = d1.sat;
  a.w12u = nullptr; a.inv12u = 1.f;
  a.tiles_x = (W + FTW - 1) / FTW;
  static const int th_env = [] { const char* e = wct_debug_env("WCT_TAIL_TH"); return e ? atoi(e) : 0; }();   // experiment: force 8 / 16
  // 32 x 16 tiles (conv12's halo recompute 1.20 instead of 1.33: -10 % at 4K) once they still fill the chip; results do not
  // depend on the tile shape (same arithmetic per pixel)
  // (32 x 24, 12 waves: three waves per SIMD instead of two and recompute 1.15: another -7 %, when every CU still gets four tiles)
  const int th = th_env ? th_env : (((H + 23) / 24) * a.tiles_x >= 4 * num_cus() ? 24 : (((H + 15) / 16) * a.tiles_x >= 2 * num_cus() ? 16 : 8));
  // conv12 on the low-resolution input with per-parity 2x2 weights (dec_tail_up_kernel) whenever the layer sits behind an upsample
  static const int up_env = [] { const char* e = wct_debug_env("WCT_TAIL_UP"); return e ? atoi(e) : 1; }();
  if (up_env && a.up_in && d0.wup16 && !(H & 1) && !(W & 1)) {
    a.w12u = reinterpret_cast<const u32x4*>(d0.wup16); a.inv12u = d0.inv_scale_up;
    auto gou = [&](auto kern, auto geo, int per_cu) -> hipError_t {
      using G = decltype(geo);
      a.tiles_y = (H + G::HROWS - 3) / (G::HROWS - 2);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds);
      if (e != hipSuccess) return e;
      const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < per_cu * num_cus() ? ntiles : per_cu * num_cus();
      hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::lds, s, a);
      return hipGetLastError();
    };
    // This kernel needs 116-124 registers, so FOUR waves per SIMD fit: two 32 x 16 workgroups per CU (8 waves each, independent
    // phases) beat one 32 x 24 workgroup (12 waves) by 5 % despite the larger halo recompute (1.20 vs 1.15).  32 x 32 tiles (16
    // waves in one workgroup) need 128 registers and spill 9: slower (WCT_TAIL_TH=32 / 24 force those shapes).
    const int thu = th_env ? th_env : (th == 24 ? 16 : th);
    if (thu == 32) return gou(dec_tail_up_kernel<32>, TailUpGeo<32>{}, 1);
    if (thu == 24) return gou(dec_tail_up_kernel<24>, TailUpGeo<24>{}, 1);
    return thu == 16 ? gou(dec_tail_up_kernel<16>, TailUpGeo<16>{}, 2) : gou(dec_tail_up_kernel<8>, TailUpGeo<8>{}, 2);
  }
  auto go = [&](auto kern, auto geo, int per_cu) -> hipError_t {
    using G = decltype(geo);
    a.tiles_y = (H + G::HROWS - 3) / (G::HROWS - 2);
    const size_t lds = ((size_t)4 * G::NPI + 640 + PH_WSLOTS + (size_t)4 * G::NPX) * 16;   // 77.8 KB (TH = 8: 2 per CU) / 114.7 KB (16) / 151.6 KB (24)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < per_cu * num_cus() ? ntiles : per_cu * num_cus();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), lds, s, a);
    return hipGetLastError();
  };
  if (th == 24) return go(dec_tail_kernel<24>, TailGeo<24>{}, 1);
  return th == 16 ? go(dec_tail_kernel<16>, TailGeo<16>{}, 1) : go(dec_tail_kernel<8>, TailGeo<8>{}, 2);
}

size_t conv_f16_weight_bytes(int cin, int cout_pad, int taps) {
  return (size_t)((cin + 15) / 16) * taps * 4 * cout_pad * 16;
}

hipError_t launch_split_pack(const float* wpk32, int cin, int cout_pad, int taps, unsigned* maxbits_dev, void* out,
                             float* inv_scale_out, hipStream_t s, bool have_max, int nmax) {
  const int chunks = (cin + 15) / 16;
  if (!have_max) {
    hipError_t e = hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    const long n = (long)chunks * 36 * cout_pad * 4;
    hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(256), 0, s, wpk32, n, maxbits_dev);
  }
  const long total = (long)chunks * taps * 4 * cout_pad;
  hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wpk32, chunks, cout_pad, taps,
                     maxbits_dev, have_max ? nmax : 1, reinterpret_cast<u32x4*>(out), inv_scale_out);
  return hipGetLastError();
}

size_t conv_phase_weight_bytes(int cin) { return (size_t)((cin + 15) / 16) * PH_WSLOTS * 16; }

hipError_t launch_split_pack_phase(const float* wpk32, int cin, const unsigned* maxbits_dev, void* out, hipStream_t s, int nmax) {
  const int chunks = (cin + 15) / 16;
  const long total = (long)chunks * PH_WSLOTS;
  hipLaunchKernelGGL(split_pack_phase_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wpk32, chunks, maxbits_dev, nmax,
                     reinterpret_cast<u32x4*>(out));
  return hipGetLastError();
}

hipError_t launch_conv3x3_f16(const ConvDesc& d, const float* in, float* out, int H, int W, hipStream_t s) {
  if (H < 2 || W < 2 || (d.flags & CONV_IN_NCHW3) || !d.wpk16 || (d.cin & 7)) return hipErrorInvalidValue;
  F16Args a;
  a.in = in; a.out = out; a.wpk = reinterpret_cast<const u32x4*>(d.wpk16); a.bias = d.bias;
  a.inv_scale_ptr = d.inv_scale_ptr; a.inv_scale = d.inv_scale;
  a.H = H; a.W = W;
  a.up_in = (d.flags & CONV_UP_IN) ? 1 : 0;
  a.inH = a.up_in ? H / 2 : H; a.inW = a.up_in ? W / 2 : W;
  a.cin = d.cin; a.cout = d.cout; a.cin_chunks = d.cin_chunks; a.cout_pad = d.cout_pad;
  a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + 7) / 8;
  a.relu = (d.flags & CONV_NO_RELU) ? 0 : 1;
  a.in_sp = (d.flags & CONV_IN_SP16) ? 1 : 0; a.out_sp = (d.flags & CONV_OUT_SP16) ? 1 : 0;
  a.sat = d.sat;
  const bool pool = d.flags & CONV_POOL_OUT, out3 = d.flags & CONV_OUT_NCHW3;
  if (out3 && a.out_sp) return hipErrorInvalidValue;
  const size_t act_b = (size_t)4 * npp(8) * 16;
  if (d.cout_pad == 16) {
    a.taps = 10;
    const size_t lds = act_b + (size_t)40 * 16 * 16;
    if (out3) return pool ? hipErrorInvalidValue : launch_k(conv3x3_f16_c16_kernel<false, true>, a, lds, 1, s);
    return pool ? launch_k(conv3x3_f16_c16_kernel<true, false>, a, lds, 1, s) : launch_k(conv3x3_f16_c16_kernel<false, false>, a, lds, 1, s);
  }
  if (out3) return hipErrorInvalidValue;
  a.taps = 9;
  int ct = d.cout_pad / 32, groups = 1;
  if (ct > 4) {
    if (d.cout_pad % 128) return hipErrorInvalidValue;
    groups = d.cout_pad / 128; ct = 4;
  }
  // small maps (level 5's first decoder conv: 135 x 240 x 128 -> 72 tiles of 32 x 16 on 256 CUs): 32 x 8 tiles and cout groups of
  // 64 instead -- four times the workgroups, the same arithmetic per output (bit-identical)
  static const int small_env = [] { const char* e = wct_debug_env("WCT_F16_SMALL"); return e ? atoi(e) : 1; }();
  if (ct == 4 && small_env && a.tiles_x * ((H + 15) / 16) * groups < num_cus()) {
    ct = 2; groups = d.cout_pad / 64;
  }
  if (ct == 4) {  // 128 couts: 32 x 16 pixel tile, 8 waves (2 per SIMD), the 74 KB weight slab serves 512 pixels
    a.tiles_y = (H + 15) / 16;
    const size_t lds16 = (size_t)4 * npp(16) * 16 + (size_t)36 * 128 * 16;
    return pool ? launch_k(conv3x3_f16_kernel<4, true, 16>, a, lds16, groups, s, 512)
                : launch_k(conv3x3_f16_kernel<4, false, 16>, a, lds16, groups, s, 512);
  }
  const size_t lds = act_b + (size_t)36 * ct * 32 * 16;
#define WCT_F16_CASE(CTV) \
  case CTV: return pool ? launch_k(conv3x3_f16_kernel<CTV, true, 8>, a, lds, groups, s) : launch_k(conv3x3_f16_kernel<CTV, false, 8>, a, lds, groups, s);
  switch (ct) {
    WCT_F16_CASE(1) WCT_F16_CASE(2)
    default: return hipErrorInvalidValue;
  }
#undef WCT_F16_CASE
}

#ifdef WCT_HEAD_TIMING
extern "C" int wct_debug_head_timing(unsigned long long* out8) {   // read and reset
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_head_t), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_head_t), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
