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

// ---- saturation tracking.  The split clamps to the f16 range (+-65504); a clamp that actually changes a value is a
// deviation from the fp32 reference, so every clamp site feeds a per-thread running maximum and the kernel raises the
// context's sticky counter once per thread at its end (wct_saturation_count).  Two forms, both 2 VALU per four values:
//   note()     max |x| of the fp32 values (v_max3_f32 with |.| modifiers): exact (x > 65504), any sign
//   note_hi()  packed-u16 max of the hi halves AFTER the conversion (v_pk_max_u16): needs only registers that are live
//              anyway -- the DMA conv kernel sits at 256 VGPRs and spilled 18 of them with note().  hi == 0x7BFF means
//              x >= 65488 (everything that rounds to the largest f16), i.e. it also fires in the last 0.02 % below the limit.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// x * s + b on four values: the scale + bias step of every epilogue.  WCT_PKFMA builds it from TWO v_pk_fma_f32 (gfx950 issues a
// packed fp32 fma at the rate of a scalar one; the compiler pairs them only when the source says so) -- the same fused
// multiply-add per element, bit-identical results.  Measured (round 3, same-box A/B of three builds, tools/experiments/
// ab_libs.sh): the packed form is NOT faster -- enc_head +3 % (1.28 -> 1.33 ms per step: its register pairs cost a third spilled
// VGPR at the 168-register cap), dec_tail +1 %, level 1 and the DMA convolutions unchanged -- so the scalar form stays the default.
__device__ __forceinline__ f32x4 fma4(const f32x4& x, float s, const f32x4& b) {
#ifdef WCT_PKFMA
  const f32x2 s2 = {s, s};
  const f32x2 lo = __builtin_elementwise_fma(f32x2{x[0], x[1]}, s2, f32x2{b[0], b[1]});
  const f32x2 hi = __builtin_elementwise_fma(f32x2{x[2], x[3]}, s2, f32x2{b[2], b[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
#else
  return f32x4{x[0] * s + b[0], x[1] * s + b[1], x[2] * s + b[2], x[3] * s + b[3]};
#endif
}

struct SatTrack {
  float m = 0.f;
  unsigned mu = 0u;
  // running maximum of fp32 values that were COMPUTED inside the path (finite f16 operands, finite weights -- checked at load --
  // accumulated in fp32: never NaN)
  __device__ __forceinline__ void note(const f32x4& v) {
    m = fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1]));
    m = fmaxf(fmaxf(m, fabsf(v[2])), fabsf(v[3]));
  }
  // hi0 / hi1: two packed pairs of hi halves of values ALREADY clamped with v_med3_f32(x, -65504 | 0, 65504); nonneg (uniform): the
  // values went through a ReLU, so their sign bits are clear.  A value at the clamp bound (0x7BFF) counts as clamped.  This is
  // also the NaN check of EXTERNAL data (images, fp32 feature maps handed in through the API): v_max ignores a NaN operand, but
  // v_med3_f32 returns the minimum of the other two for one, i.e. -65504 -> 0xFBFF -> flagged here (the fp32 reference would
  // have propagated the NaN; this path turns it into finite garbage, so it must at least say so).
  __device__ __forceinline__ void note_hi(unsigned hi0, unsigned hi1, bool nonneg) {
    if (!nonneg) { hi0 &= 0x7fff7fffu; hi1 &= 0x7fff7fffu; }
    u16x2 a = __builtin_bit_cast(u16x2, mu);
    a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2, hi0));
    a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2, hi1));
    mu = __builtin_bit_cast(unsigned, a);
  }
  __device__ __forceinline__ void commit(unsigned* counter) const {
    if (counter && (m > 65504.f || (mu & 0xffffu) >= 0x7BFFu || (mu >> 16) >= 0x7BFFu)) sat_raise(counter);
  }
};

// ---- the split itself.  x (already clamped to the f16 range) -> hi = f16(x) (round to nearest even), lo = f16(x - hi), two
// values at a time as packed pairs.  hi: one v_cvt_pk_f16_f32 per pair.  lo: ONE v_fma_mix{lo,hi}_f16 per value -- it reads hi
// straight out of the packed register as f16, forms x - hi exactly and rounds once to f16.  (x - hi has <= 13 significant bits,
// so the former route -- v_cvt_f32_f16, fp32 subtract, v_cvt_f16_f32: 2.5 instructions per value, the compiler does not
// select the mix form by itself -- rounded the same exact number once: bit-identical results.)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float x0, float x1) {
  const f16x2 h = {(_Float16)x0, (_Float16)x1};
  HiLo r;
  r.hi = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r.lo) : "v"(r.hi), "v"(x0), "v"(x1));
  return r;
}
__device__ __forceinline__ float clamp_pm(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
__device__ __forceinline__ float clamp_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 65504.f); }   // ReLU and the upper clamp in one

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo, SatTrack& sat) {
  u32x4 h, l;
  { const HiLo t_ = split2(clamp_pm(a[0]), clamp_pm(a[1])); h[0] = t_.hi; l[0] = t_.lo; }
  { const HiLo t_ = split2(clamp_pm(a[2]), clamp_pm(a[3])); h[1] = t_.hi; l[1] = t_.lo; }
  { const HiLo t_ = split2(clamp_pm(b[0]), clamp_pm(b[1])); h[2] = t_.hi; l[2] = t_.lo; }
  { const HiLo t_ = split2(clamp_pm(b[2]), clamp_pm(b[3])); h[3] = t_.hi; l[3] = t_.lo; }
  sat.note_hi(h[0], h[1], false); sat.note_hi(h[2], h[3], false);   // external fp32 data: range AND NaN (see SatTrack)
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

// ---- "SP16" activation format (intermediate activations of the f16x3 path):
// the tensor is cut into 16-channel chunks; each chunk is a plane of 64-byte pixel records
//     [chunk k = c >> 4][pixel][group (c >> 3) & 1][hi 8 halfs | lo 8 halfs],      x = hi + lo
// i.e. C * 4 bytes per pixel like fp32, but one chunk's records are CONTIGUOUS over pixels: a convolution walks its K
// dimension chunk by chunk, so the 64 bytes it needs of a pixel are a full half cache line next to its neighbours'
// (pixel-major NHWC records made every 16-channel slice a strided gather).  A 16-channel tensor is its own single plane.
// The split (with its +-65504 clamp) is done ONCE by the producing kernel's epilogue instead of by every consumer on
// every halo re-read, and consumers move the 16-byte groups global -> LDS without touching them
// (global_load_lds_dwordx4: no VGPR staging, no VALU).  hi / lo are exactly what split8() computes from the fp32 value.
__device__ __forceinline__ size_t sp16_plane_bytes(int H, int W) { return (size_t)H * W * 64; }
// byte offset of the 16-byte piece (hi: hl = 0, lo: hl = 1) of channels 8 g .. 8 g + 7 of a pixel
__device__ __forceinline__ size_t sp16_piece(size_t plane_bytes, size_t pixel, int g, int hl) {
  return (size_t)(g >> 1) * plane_bytes + pixel * 64 + (g & 1) * 32 + hl * 16;
}

// 4 consecutive channels -> (hi, lo) as 2 + 2 dwords
template <bool TRACK = true>
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo, SatTrack& sat, bool nonneg) {
  { const HiLo t_ = split2(clamp_pm(v[0]), clamp_pm(v[1])); hi[0] = t_.hi; lo[0] = t_.lo; }
  { const HiLo t_ = split2(clamp_pm(v[2]), clamp_pm(v[3])); hi[1] = t_.hi; lo[1] = t_.lo; }
  if constexpr (TRACK) sat.note_hi(hi[0], hi[1], nonneg);
}

// 32x32 MFMA accumulator layout: lanes l and l + 32 hold channels 8q + {0..3} and 8q + {4..7} of the SAME pixel.
// v_permlane32_swap hands the low lane both hi halves and the high lane both lo halves, so each lane writes one
// 16-byte group: returns the 16 bytes this lane stores at  record + group * 32 + (lane >> 5) * 16.
// Must be executed by all 64 lanes (no divergence).
// the DMA kernel's form: `v` is the PRE-activation, `lob` the (wave-uniform) lower clamp bound -- 0 with ReLU (ReLU and the
// range clamp are then one v_med3_f32), -65504 without; the caller keeps its own saturation record
__device__ __forceinline__ u32x4 sp16_pair_exchange_lob(const f32x4& v, float lob) {
  u32x2 hi, lo;
  { const HiLo t_ = split2(__builtin_amdgcn_fmed3f(v[0], lob, 65504.f), __builtin_amdgcn_fmed3f(v[1], lob, 65504.f)); hi[0] = t_.hi; lo[0] = t_.lo; }
  { const HiLo t_ = split2(__builtin_amdgcn_fmed3f(v[2], lob, 65504.f), __builtin_amdgcn_fmed3f(v[3], lob, 65504.f)); hi[1] = t_.hi; lo[1] = t_.lo; }
  const auto ra = __builtin_amdgcn_permlane32_swap(hi[0], lo[0], false, false);
  const auto rb = __builtin_amdgcn_permlane32_swap(hi[1], lo[1], false, false);
  return u32x4{ra[0], rb[0], ra[1], rb[1]};
}

// TRACK = false: the caller keeps its own saturation record
template <bool TRACK = true>
__device__ __forceinline__ u32x4 sp16_pair_exchange(const f32x4& v, SatTrack& sat, bool nonneg) {
  u32x2 hi, lo;
  split4<TRACK>(v, hi, lo, sat, nonneg);
  const auto ra = __builtin_amdgcn_permlane32_swap(hi[0], lo[0], false, false);
  const auto rb = __builtin_amdgcn_permlane32_swap(hi[1], lo[1], false, false);
  return u32x4{ra[0], rb[0], ra[1], rb[1]};
}

// 16x16 MFMA accumulator layout: lane (pixel, kq) holds channels 4 kq .. 4 kq + 3 -> two 8-byte stores into the
// pixel's record (group kq >> 1, half kq & 1)
__device__ __forceinline__ void sp16_store4(char* record, int kq, const f32x4& v, SatTrack& sat, bool nonneg) {
  u32x2 hi, lo;
  split4(v, hi, lo, sat, nonneg);
  char* g = record + (kq >> 1) * 32 + (kq & 1) * 8;
  *reinterpret_cast<u32x2*>(g) = hi;
  *reinterpret_cast<u32x2*>(g + 16) = lo;
}

// ---- shared by the fused full-resolution kernels (conv3x3_f16.hip: enc_head / dec_tail, level1.hip)
constexpr int I2W = FTW + 4, I2H = 8 + 4;       // 36 x 12: input tile of the first conv (two halo rings)
constexpr int NPI2 = I2W * I2H;                 // 432 (a multiple of 16)

constexpr int IMG_E = NPI2 + 4;   // 8-byte pixels per plane; the zeroed tail absorbs the "4th pixel" over-read of the last row

// next tile's 36 x 12 x 3 image window -> 6 registers per thread (unconditional, clamped).  soff[k]: tile-independent
// offset of the thread's pixel k from the window origin, valid for interior tiles.
// uniform tile index -> (tile row, tile column) on the SCALAR unit: hipcc expands even a uniform integer division through
// v_rcp_iflag_f32 and ~25 VALU instructions, and the persistent fused kernels -- VALU-issue-bound -- did two or three of
// them per tile.  q = mulhi(tile, ceil(2^32 / tiles_x)) is exact while tile * tiles_x < 2^32.
__host__ __device__ inline unsigned tile_div_magic(int tiles_x) { return tiles_x > 1 ? (unsigned)(((1ull << 32) + tiles_x - 1) / tiles_x) : 0u; }
__device__ __forceinline__ void tile_rc(int tile, int tiles_x, unsigned magic, int& row, int& col) {
  const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane(tile);
  row = tiles_x > 1 ? (int)__umulhi(t, magic) : (int)t;
  col = (int)t - row * tiles_x;
}

// horizontal neighbour (lane ^ 1) through DPP quad_perm [1,0,3,2]: one VALU move instead of ds_bpermute + address
__device__ __forceinline__ float lane_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

// (uniform) true when the 36 x (th + 4) input window of a 32 x th tile lies inside the image: no reflection anywhere
__device__ __forceinline__ bool tile_interior_h(int ty0, int tx0, int H, int W, int th) {
  return ty0 >= 2 && ty0 + th + 2 <= H && tx0 >= 2 && tx0 + 34 <= W;
}

// TH: tile height (8: 4 waves / 256 threads, the 36 x 12 window; 16: 8 waves / 512 threads, 36 x 20) -- two pixels per thread either way
template <int TH = 8>
__device__ __forceinline__ void head_fetch(const float* img, int H, int W, float (&r)[2][3], const int (&soff)[2], int ty0, int tx0, int tid) {
  constexpr int NT = 32 * TH, NPI = I2W * (TH + 4);
  static_assert(2 * NT >= NPI, "two pixels per thread cover the window");
  const size_t plane = (size_t)H * W;
  if (tile_interior_h(ty0, tx0, H, W, TH)) {
    // uniform base per plane + the thread's UNSIGNED 32-bit offset: global_load with an SGPR base (a signed index is widened
    // to a 64-bit VGPR pair per load)
    const float* base = img + (size_t)(ty0 - 2) * W + (tx0 - 2);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pc = base + c * plane;
#pragma unroll
      for (int k = 0; k < 2; ++k) r[k][c] = pc[(unsigned)soff[k]];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + NT * k;
    e = e < NPI ? e : NPI - 1;
    const int py = e / I2W, px = e - py * I2W;
    const size_t off = (size_t)reflect_clamp(ty0 - 2 + py, H) * W + reflect_clamp(tx0 - 2 + px, W);
#pragma unroll
    for (int c = 0; c < 3; ++c) r[k][c] = img[c * plane + off];
  }
}

template <int TH = 8>
__device__ __forceinline__ void head_fetch(const float* img, int H, int W, int tiles_x, unsigned tx_magic, float (&r)[2][3], const int (&soff)[2], int tile, int tid) {
  int tr, tc;
  tile_rc(tile, tiles_x, tx_magic, tr, tc);
  head_fetch<TH>(img, H, W, r, soff, tr * TH, tc * FTW, tid);
}

// Call once before a persistent tile loop that keeps weights / biases loaded from global memory in registers AND prefetches the next
// tile with global loads inside the loop.  The compiler's wait-count insertion cannot prove at the loop header that the
// pre-loop loads have landed, so it guards every use of those registers inside the loop with s_waitcnt vmcnt(k) -- and since
// vmcnt retires in order, that makes the first MFMAs of every tile wait for the prefetch that was issued just before them
// (enc_head: vmcnt(5), (4), (3), (2) in front of the conv11 MFMAs = the HBM latency of the next tile's window, every tile).  An
// explicit, compiler-visible wait before the loop settles the scoreboard.
__device__ __forceinline__ void settle_preloop_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0) only (gfx9 encoding)

// The prefetched window of the NEXT tile must not be touched before the end of the current tile: left to itself the scheduler
// hoists this function's conversions (pure VALU on the loaded registers) to right behind the loads, and every wave then waits for
// HBM inside its first MFMA phase (s_waitcnt vmcnt(5..1) in the middle of conv11).  head_pin() makes the registers opaque at the
// point of the call -- a volatile asm with "+v" constraints, which cannot move above the barrier that precedes it.
__device__ __forceinline__ void head_pin(float (&r)[2][3]) {
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(r[k][c]));
}

template <int TH = 8>
__device__ __forceinline__ void head_commit(const float (&r)[2][3], u32x2* imgH, u32x2* imgL, int tid, SatTrack& sat) {
  constexpr int NT = 32 * TH, NPI = I2W * (TH + 4);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + NT * k;
    if (e < NPI) {
      u32x2 h, l;
      { const HiLo t_ = split2(clamp_pm(r[k][0]), clamp_pm(r[k][1])); h[0] = t_.hi; l[0] = t_.lo; }
      { const HiLo t_ = split2(clamp_pm(r[k][2]), 0.f); h[1] = t_.hi; l[1] = t_.lo; }
      sat.note_hi(h[0], h[1], false);   // the image is external data: range AND NaN (see SatTrack)
      imgH[e] = h;
      imgL[e] = l;
    }
  }
}

// 4 consecutive channels 4kq..4kq+3 of halo pixel `pix` -> 8 bytes in the hi plane and 8 in the lo plane.  RELU: `v` is the
// pre-activation; ReLU and the range clamp are one v_med3_f32 (the range record then looks at the positive side only)
template <bool RELU = false>
__device__ __forceinline__ void store_split4(u32x4* planes, int npp, int pix, int kq, const f32x4& v, SatTrack& sat) {
  u32x2 h, l;
  if constexpr (RELU) {
    sat.m = fmaxf(fmaxf(sat.m, v[0]), v[1]);
    sat.m = fmaxf(fmaxf(sat.m, v[2]), v[3]);
    { const HiLo t_ = split2(clamp_relu(v[0]), clamp_relu(v[1])); h[0] = t_.hi; l[0] = t_.lo; }
    { const HiLo t_ = split2(clamp_relu(v[2]), clamp_relu(v[3])); h[1] = t_.hi; l[1] = t_.lo; }
  } else {
    sat.note(v);
    { const HiLo t_ = split2(clamp_pm(v[0]), clamp_pm(v[1])); h[0] = t_.hi; l[0] = t_.lo; }
    { const HiLo t_ = split2(clamp_pm(v[2]), clamp_pm(v[3])); h[1] = t_.hi; l[1] = t_.lo; }
  }
#ifndef WCT_SS4_SWAP
  // two 8-byte stores per lane.  A ds_write_b64 is serviced in contiguous 16-lane groups against 32 banks: 16 lanes writing the SAME
  // half of 16 consecutive 16-byte slots cover every bank twice -- a 2-way conflict on every store of this function, the largest
  // single source of the 23-26 % conflict share in profiles/r04b_sq_counters_fused_ends.txt.  The conflict-free form below was built
  // and measured in round 5 and LOST: it is kept only as the record of that experiment.
  *(reinterpret_cast<u32x2*>(planes + (0 * 2 + (kq >> 1)) * npp + pix) + (kq & 1)) = h;
  *(reinterpret_cast<u32x2*>(planes + (1 * 2 + (kq >> 1)) * npp + pix) + (kq & 1)) = l;
#else
  // round 5 experiment (-DWCT_SS4_SWAP): lanes (li, kq) and (li, kq ^ 1) -- 16 lanes apart, the same pixel -- trade halves through
  // v_permlane16_swap (the 16x16 counterpart of sp16_pair_exchange): the even row keeps both hi halves, the odd row both lo halves, and
  // each lane writes ONE 16-byte slot; ds_write_b128 groups are 8 contiguous lanes = 128 contiguous bytes: conflict-free, half the
  // LDS store instructions, bit-identical results.  Same-box A/B (profiles/r05_lds_conflict_ab.txt): SQ_WAIT_INST_LDS of dec_tail_up
  // 34.6 M -> 12.3 M quad-cycles -- and the kernel 26 % SLOWER (0.464 -> 0.584 ms per step; enc_head 1.034 -> 1.046, l1_decode
  // unchanged): the two cross-lane swaps sit between the MFMA result and the store in every group's dependent chain, and a 16-byte
  // LDS store occupies the VGPR -> LDS path for 13 cycles against 2 x 6.  The conflicts were never what these kernels wait for.
  const auto ra = __builtin_amdgcn_permlane16_swap(h[0], l[0], false, false);
  const auto rb = __builtin_amdgcn_permlane16_swap(h[1], l[1], false, false);
  planes[((kq & 1) * 2 + (kq >> 1)) * npp + pix] = u32x4{ra[0], rb[0], ra[1], rb[1]};
#endif
}

// 16->16 conv on the [4][NPP] planes of a 34 x 10 halo tile for the wave's 2 rows x 32 px (the c16 kernel's body)
template <int NPP = npp(8)>   // plane stride: npp(tile height)
__device__ __forceinline__ void c16_compute(const u32x4* act, const u32x4* wgt, int wave, int li, int kq, f32x4 (&acc)[2][2]) {
  const int kh = kq & 1, ts = kq >> 1;
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + ts;
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

// the same with the wave's ten weight operands resident in registers (40 VGPRs): 40 instead of 50 LDS reads per 60 MFMAs -- for
// kernels whose conv12 waves have the registers to spare (the consumer role of enc_head_roles_kernel).  Same MFMA order:
// bit-identical to c16_compute.
struct C16Weights { f16x8 ah[5], al[5]; };
__device__ __forceinline__ void c16_load_weights(const u32x4* w /* global or LDS, [10][2][2][16] */, int li, int kq, C16Weights& cw) {
  const int kh = kq & 1, ts = kq >> 1;
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + ts;
    cw.ah[s] = __builtin_bit_cast(f16x8, w[((tap * 2 + 0) * 2 + kh) * 16 + li]);
    cw.al[s] = __builtin_bit_cast(f16x8, w[((tap * 2 + 1) * 2 + kh) * 16 + li]);
  }
}
template <int NPP = npp(8)>
__device__ __forceinline__ void c16_compute_w(const u32x4* act, const C16Weights& cw, int wave, int li, int kq, f32x4 (&acc)[2][2]) {
  const int kh = kq & 1, ts = kq >> 1;
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + ts;
    const int tc = tap > 8 ? 8 : tap;
    const int dy = tc / 3, dx = tc - dy * 3;
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
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          acc[r][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? cw.al[s] : cw.ah[s], term == 1 ? bl[r][h] : bh[r][h], acc[r][h], 0, 0, 0);
  }
}

// ---- "block-packed" 16 -> 3 convolution: the LAST decoder conv inside the fused tails (dec_tail kernels, l1_decode_kernel).
// With 3 real couts the 16-row M dimension of a 16x16x32 MFMA is 81 % padding.  Here one N column is a 2 x 2 BLOCK of output
// pixels and M = 4 * phase + cout (phase = 2 py + px: which pixel of the block; 12 of 16 rows carry results).  K walks the 4 x 4
// input window the block shares -- 4 rows x 4 columns x 16 channels = 256 = 8 K-steps of (2 columns x 16 channels); a weight is
// zero where a window position lies outside a phase's own 3 x 3 taps -- so the wave's 2 rows x 32 pixels (16 blocks) cost
// 8 x 3 = 24 MFMAs and 16 operand reads: 60 / 50 with one pixel per column, 36 / 36 with pixel PAIRS (round 2's first form).
//   act planes: [hl][kh][NPX] 16-byte slots, slot(py, px) = py * PH_W + (px & 1) * 17 + (px >> 1) over the 34 x (TH + 2) halo: the
//               16 blocks of a row segment at a fixed window column are 16 CONSECUTIVE slots (conflict-free ds_read_b128)
//   wgt:        [8 ks][hl][kq][16 m] x 16 B;  K-step ks: window row wy = ks >> 1, column wx = 2 (ks & 1) + (kq >> 1), channels
//               8 (kq & 1) .. + 7;  A[m = 4 (2 py + px) + cout][.] = w[cout][ch][wy - py][wx - px] where both offsets are in 0..2
// Result: lane (li, kq) holds the output pixel (row 2 wave + (kq >> 1), column 2 li + (kq & 1)) of the tile, couts 0..2 in
// registers 0..2 of acc[0] + acc[1] + acc[2] + acc[3] (four independent chains) -- every lane has a pixel to store.
constexpr int PH_W = 36, PH_NPX = ((8 + 2) * PH_W + 15) / 16 * 16;   // 368 slots per plane (5888 B == 0 mod 256)
constexpr int PH_WSLOTS = 8 * 2 * 4 * 16;                             // 1024 weight slots per 16-channel chunk
__host__ __device__ inline int ph_slot(int py, int px) { return py * PH_W + (px & 1) * 17 + (px >> 1); }

template <int NPX = PH_NPX>   // slots per plane: PH_NPX for the 34 x 10 halo of a 32 x 8 tile
__device__ __forceinline__ void c3_block_compute(const u32x4* act, const u32x4* wgt, int wave, int li, int kq, f32x4 (&acc)[4]) {
  const int kh = kq & 1, p = kq >> 1;
  const u32x4* ah_ = act + (0 * 2 + kh) * NPX + p * 17 + li;   // per-lane part of the slot: column parity plane + block index
  const u32x4* al_ = act + (1 * 2 + kh) * NPX + p * 17 + li;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int wy = ks >> 1, c1 = ks & 1;          // window column 2 c1 + p
    const f16x8 wh = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 0) * 4 + kq) * 16 + li]);
    const f16x8 wl = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 1) * 4 + kq) * 16 + li]);
    const int off = (wave * 2 + wy) * PH_W + c1;
    const f16x8 bh = __builtin_bit_cast(f16x8, ah_[off]);
    const f16x8 bl = __builtin_bit_cast(f16x8, al_[off]);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, acc[ks & 3], 0, 0, 0);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, acc[ks & 3], 0, 0, 0);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, acc[ks & 3], 0, 0, 0);
  }
}

// The same with the chunk's sixteen weight operands resident in registers (64 VGPRs): 16 instead of 32 LDS reads per 24 MFMAs, and no
// weight slab in LDS -- for kernels that run at two waves per SIMD and have the registers (l1_decode_kernel).  Same MFMA order as
// c3_block_compute: bit-identical.
struct C3Weights { f16x8 wh[8], wl[8]; };
__device__ __forceinline__ void c3_load_weights(const u32x4* wgt /* global, [8 ks][hl][kq][16 m] */, int li, int kq, C3Weights& w) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    w.wh[ks] = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 0) * 4 + kq) * 16 + li]);
    w.wl[ks] = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 1) * 4 + kq) * 16 + li]);
  }
}
template <int NPX = PH_NPX>
__device__ __forceinline__ void c3_block_compute_w(const u32x4* act, const C3Weights& w, int wave, int li, int kq, f32x4 (&acc)[4]) {
  const int kh = kq & 1, p = kq >> 1;
  const u32x4* ah_ = act + (0 * 2 + kh) * NPX + p * 17 + li;
  const u32x4* al_ = act + (1 * 2 + kh) * NPX + p * 17 + li;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int off = (wave * 2 + (ks >> 1)) * PH_W + (ks & 1);
    const f16x8 bh = __builtin_bit_cast(f16x8, ah_[off]);
    const f16x8 bl = __builtin_bit_cast(f16x8, al_[off]);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wh[ks], bh, acc[ks & 3], 0, 0, 0);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wh[ks], bl, acc[ks & 3], 0, 0, 0);
    acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wl[ks], bh, acc[ks & 3], 0, 0, 0);
  }
}

// A chunk of which only the FIRST 8 channels exist (level 1: relu1_1 has 24 = 16 + 8 channels; the second chunk's upper half is padding).
// The K dimension of a 16x16x32 step is then one whole window ROW: lane group kq = window column kq (4 columns x 8 channels) -- 4 K-steps,
// 12 MFMAs and 8 operand reads instead of 8 / 24 / 16, and the chunk needs two LDS planes (hi, lo) instead of four.  Weights come out of the
// SAME block-packed slab (step (wy, column wx) of this form = K-step 2 wy + (wx >> 1), lane group 2 (wx & 1) of the packed one), into 32 VGPRs.
// The products are those of c3_block_compute on the chunk, summed in another order (fp32 round-off level difference, not bit-identical).
// act: [hl][NPX] slots (ph_slot), channel half 0 only.
struct C3HalfWeights { f16x8 wh[4], wl[4]; };
__device__ __forceinline__ void c3_load_weights_half(const u32x4* wgt, int li, int kq, C3HalfWeights& w) {
#pragma unroll
  for (int wy = 0; wy < 4; ++wy) {
    const int ks = 2 * wy + (kq >> 1), kqp = 2 * (kq & 1);
    w.wh[wy] = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 0) * 4 + kqp) * 16 + li]);
    w.wl[wy] = __builtin_bit_cast(f16x8, wgt[((ks * 2 + 1) * 4 + kqp) * 16 + li]);
  }
}
template <int NPX = PH_NPX>
__device__ __forceinline__ void c3_block_compute_half(const u32x4* act, const C3HalfWeights& w, int wave, int li, int kq, f32x4 (&acc)[4]) {
  // halo column 2 li + kq of the block row: parity plane (kq & 1), index li + (kq >> 1)
  const u32x4* ah_ = act + 0 * NPX + (kq & 1) * 17 + (kq >> 1) + li;
  const u32x4* al_ = act + 1 * NPX + (kq & 1) * 17 + (kq >> 1) + li;
#pragma unroll
  for (int wy = 0; wy < 4; ++wy) {
    const int off = (wave * 2 + wy) * PH_W;
    const f16x8 bh = __builtin_bit_cast(f16x8, ah_[off]);
    const f16x8 bl = __builtin_bit_cast(f16x8, al_[off]);
    acc[wy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wh[wy], bh, acc[wy], 0, 0, 0);
    acc[wy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wh[wy], bl, acc[wy], 0, 0, 0);
    acc[wy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.wl[wy], bh, acc[wy], 0, 0, 0);
  }
}
// the producer side of that chunk: channels 4 kq .. 4 kq + 3 of its FIRST 8 (lane groups kq = 0, 1; the others hold padding and store nothing)
// -> 8 bytes in the hi plane and 8 in the lo plane, [hl][npp] slots.  `v` is the pre-activation (ReLU merged with the range clamp).
__device__ __forceinline__ void store_split4_half(u32x4* planes, int npp, int pix, int kq, const f32x4& v, SatTrack& sat) {
  if (kq < 2) {
    u32x2 h, l;
    sat.m = fmaxf(fmaxf(sat.m, v[0]), v[1]);
    sat.m = fmaxf(fmaxf(sat.m, v[2]), v[3]);
    { const HiLo t_ = split2(clamp_relu(v[0]), clamp_relu(v[1])); h[0] = t_.hi; l[0] = t_.lo; }
    { const HiLo t_ = split2(clamp_relu(v[2]), clamp_relu(v[3])); h[1] = t_.hi; l[1] = t_.lo; }
    *(reinterpret_cast<u32x2*>(planes + 0 * npp + pix) + kq) = h;
    *(reinterpret_cast<u32x2*>(planes + 1 * npp + pix) + kq) = l;
  }
}

// the lane's output pixel of a tile after c3_block_compute: scale, bias, ReLU, three planar stores
__device__ __forceinline__ void c3_block_store(const f32x4 (&acc)[4], float inv, const f32x4& bias, float* out, size_t plane, int ty0,
                                               int tx0, int wave, int li, int kq, int H, int W) {
  const int gy = ty0 + wave * 2 + (kq >> 1), gx = tx0 + 2 * li + (kq & 1);
  if (gy < H && gx < W) {
    const size_t off = (size_t)gy * W + gx;
    out[off] = fmaxf(((acc[0][0] + acc[1][0]) + (acc[2][0] + acc[3][0])) * inv + bias[0], 0.f);
    out[plane + off] = fmaxf(((acc[0][1] + acc[1][1]) + (acc[2][1] + acc[3][1])) * inv + bias[1], 0.f);
    out[2 * plane + off] = fmaxf(((acc[0][2] + acc[1][2]) + (acc[2][2] + acc[3][2])) * inv + bias[2], 0.f);
  }
}

// ---- conv11 of the level-1 / head encoders (3 channels in, <= 32 out) as f16x3 on 16x16x32 MFMAs.
// K layout: "singles" of 4 halfs = one window pixel's RGB0; the 27 singles (term 0: w_hi x_hi, 1: w_hi x_lo, 2: w_lo x_hi; pos =
// 3 dy + dx of the 3x3 window) are concatenated along K -- 27 of 32 singles = FOUR K-steps (the first form kept the three terms
// apart, two half-empty K-steps each: six MFMAs).  Lane group kq of K-step s holds the singles l1_single(s, kq, 0 / 1)
// (wct_common.h: ordered so that the reads are free of bank conflicts): two 8-byte LDS reads at per-lane offsets (window position,
// hi or lo plane), fixed for the kernel's lifetime.
// Weights: [cout tile][s][kq][16 couts] x 8 halfs (wct_api.hip pack_head_f16).
struct L1Conv { const u32x4* w; const float* b; float inv; };
struct L1Weights { f16x8 a[2][4]; int off[4][2]; f32x4 bias[2]; float inv; };

// per-lane operand offsets in BYTES from the window's top-left pixel in the hi plane (an operand address is then one add);
// lo_plane = imgL - imgH in 8-byte pixels
__device__ __forceinline__ void l1_lane_offsets(int kq, int lo_plane, int (&off)[4][2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const L1Single t = l1_single(s, kq, u);          // wct_common.h: which (term, window position) this lane group holds
      off[s][u] = ((t.pos / 3) * I2W + t.pos % 3 + (t.term == 1 ? lo_plane : 0)) * 8;
    }
}

__device__ __forceinline__ void l1_load_weights(const L1Conv& c, int li, int kq, int lo_plane, L1Weights& w) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
    for (int s = 0; s < 4; ++s) w.a[ct][s] = __builtin_bit_cast(f16x8, c.w[((ct * 4 + s) * 4 + kq) * 16 + li]);
    w.bias[ct] = *reinterpret_cast<const f32x4*>(c.b + ct * 16 + 4 * kq);
  }
  l1_lane_offsets(kq, lo_plane, w.off);
  w.inv = c.inv;
}

// relu(conv11) of 16 pixels (lane & 15; `base` = index of the pixel's 3x3 window's top-left in the 36-wide RGB0 tile)
// x 16 couts of tile ct:  result rows = couts ct * 16 + 4 kq + {0..3}
// RELU = false: the pre-activation (the caller merges ReLU with the range clamp of its split: store_split4<true>)
template <bool RELU = true>
__device__ __forceinline__ f32x4 l1_conv_group(const u32x2* imgH, int base, const L1Weights& w, int ct) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* p = reinterpret_cast<const char*>(imgH) + base * 8;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32x2 r0 = *reinterpret_cast<const u32x2*>(p + w.off[s][0]), r1 = *reinterpret_cast<const u32x2*>(p + w.off[s][1]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.a[ct][s], __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]}), acc, 0, 0, 0);
  }
  f32x4 x = fma4(acc, w.inv, w.bias[ct]);
  if constexpr (RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
  }
  return x;
}

// both cout tiles of conv11 from ONE set of operand reads (the callers store to LDS between the tiles, so the compiler may not
// merge the reads of two l1_conv_group calls itself)
template <bool RELU = true>
__device__ __forceinline__ void l1_conv_pair(const u32x2* imgH, int base, const L1Weights& w, f32x4& x0, f32x4& x1) {
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
  const char* p = reinterpret_cast<const char*>(imgH) + base * 8;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32x2 r0 = *reinterpret_cast<const u32x2*>(p + w.off[s][0]), r1 = *reinterpret_cast<const u32x2*>(p + w.off[s][1]);
    const f16x8 b = __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.a[0][s], b, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.a[1][s], b, a1, 0, 0, 0);
  }
  x0 = fma4(a0, w.inv, w.bias[0]);
  x1 = fma4(a1, w.inv, w.bias[1]);
  if constexpr (RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x0[r] = fmaxf(x0[r], 0.f); x1[r] = fmaxf(x1[r], 0.f); }
  }
}

inline int num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    // experiment (tools/experiments/ab_cu_reserve.sh): size the persistent grids for fewer CUs than there are
    if (const char* e = wct_debug_env("WCT_CU_RESERVE")) { const int r = atoi(e); if (r > 0 && r < v - 8) v -= r; }
    return v;
  }();
  return n;
}

}  // namespace
