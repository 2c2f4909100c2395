// Shared declarations of libwct_hip (gfx950 only).  Internal -- the public C ABI is include/wct_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Experiment / debug switches read from the environment are honoured ONLY when WCT_DEBUG is set (to anything but "0"): a
// stray WCT_* variable in a production environment must never change results.  Contexts have wct_debug_set() instead.
static inline const char* wct_debug_env(const char* name) {
  const char* g = getenv("WCT_DEBUG");
  if (!g || !g[0] || (g[0] == '0' && !g[1])) return nullptr;
  return getenv(name);
}

// The context's saturation counter (wct_api.hip sat_dev): +1 per thread (or wave) that clamped an activation, SATURATING -- the
// thread that wraps the 32-bit counter sees old == 0xffffffff itself and pins it at 2^31 (so does every add above it).
__device__ __forceinline__ void sat_raise(unsigned* counter) {
  const unsigned old = atomicAdd(counter, 1u);
  if (old >= 0x80000000u) atomicMax(counter, 0x80000000u);
}

// K layout of the 3-channel first conv on 16x16x32 f16 MFMAs (conv_f16_dev.h l1_conv_group, enc_head_kernel; host packing
// wct_api.hip pack_head_f16): 27 "singles" (split term 0: w_hi x_hi, 1: w_hi x_lo, 2: w_lo x_hi; window position pos = 3 dy + dx;
// 4 halfs RGB0 each) in 4 K-steps x 4 lane groups x 2.  One ds_read_b64 serves 32 lanes = two lane groups: their two singles are
// always in the SAME window row and the SAME plane (hi or lo), so the two 128-byte runs overlap on identical addresses instead of
// landing on the same banks 288 bytes apart (the contiguous order 9 term + pos did: 25 LDS cycles per 16 reads instead of 16).
// Pair q = 4 s + 2 u + (kq >> 1) of lane groups (2 h, 2 h + 1); five pairs per window row, the 16th is empty.
struct L1Single { int term, pos; bool zero; };
__host__ __device__ constexpr L1Single l1_single(int s, int kq, int u) {
  const int q = 4 * s + 2 * u + (kq >> 1), j = kq & 1;
  if (q == 15) return L1Single{0, 0, true};
  const int r = q / 5, m = q - 5 * r;
  return m == 0 ? L1Single{0, 3 * r + j, false}
       : m == 1 ? (j == 0 ? L1Single{0, 3 * r + 2, false} : L1Single{2, 3 * r, false})
       : m == 2 ? L1Single{2, 3 * r + 1 + j, false}
       : m == 3 ? L1Single{1, 3 * r + j, false}
                : L1Single{1, 3 * r + 2, j != 0};      // the partner of the row's last lo single reads the same place with zero weights
}

// ---- conv3x3 (reflect-pad + 3x3 conv + bias + ReLU, optional fused nearest-x2 input / 2x2 max-pool output)
enum ConvFlags : int {
  CONV_IN_NCHW3 = 1,   // input is a planar 3xHxW image (first encoder conv, conv0 folded in)
  CONV_UP_IN = 2,      // input tensor is stored at (H/2, W/2): nearest x2 fused into the tile load
  CONV_POOL_OUT = 4,   // write max over 2x2 (floor mode) instead of the full-resolution output
  CONV_OUT_NCHW3 = 8,  // output is a planar 3xHxW image (last decoder conv)
  CONV_NO_RELU = 16,   // affine only (used by wct_apply: centre-tap 1x1)
  CONV_IN_SP16 = 32,   // input is in the split-f16 SP16 format (conv_f16_dev.h) instead of fp32 NHWC
  CONV_OUT_SP16 = 64,  // output is written in SP16
};

struct ConvDesc {
  int cin, cout;       // logical channels
  int cin_chunks;      // ceil(cin/16)  (1 for CONV_IN_NCHW3)
  int cout_pad;        // multiple of 16
  int flags;
  const float* wpk;    // device, packed [chunk][tap][kq][cout_pad][4]  (IN_NCHW3: [tap][4][cout_pad])
  const float* bias;   // device, [cout_pad]
  // split-f16 ("f16x3") form of the same weights, conv3x3_f16.hip: [chunk][taps][hi/lo][kh][cout_pad] x 8 halfs,
  // pre-multiplied by a power of two; the epilogue multiplies by inv_scale (or *inv_scale_ptr when non-null)
  const void* wpk16 = nullptr;
  float inv_scale = 1.f;
  const float* inv_scale_ptr = nullptr;
  // 3-channel first conv with <= 32 couts (level-1 encoder): f16x3 slot packing for level1.hip, [2 cout tiles] (device)
  const void* l1w16 = nullptr;
  const float* l1bias = nullptr;   // [32], zero padded
  float l1inv = 1.f;
  // last decoder conv (16-channel chunks -> 3 couts): the same split-f16 weights (same scale) in the block-packed layout of
  // conv_f16_dev.h c3_block_compute, [chunk][8 ks][hl][kq][16 m] x 16 B, for the fused tails
  const void* wph16 = nullptr;
  // CONV_UP_IN layers with 16 -> 16 channels (the fused tail's first conv): the 3x3 convolution of a nearest-x2 upsampled map is,
  // per output parity (a, b), a 2x2 convolution of the LOW-RESOLUTION map with summed taps (conv3x3_f16.hip dec_tail_kernel) --
  // split-f16 weights [phase 2a + b][tap row][hl][kq][16 couts] x 16 B with their own power-of-two scale
  const void* wup16 = nullptr;
  float inv_scale_up = 1.f;
  // the context's sticky saturation counter (device): raised by the f16x3 kernels when an activation exceeded +-65504 and
  // was clamped (conv_f16_dev.h SatTrack); may be null
  unsigned* sat = nullptr;
};

// Launch one conv layer.  (H, W) = spatial size the convolution runs at (after the fused upsample,
// before the fused pool).  `in` is NHWC [inH*inW][cin] (or planar 3xHxW), `out` NHWC or planar.
hipError_t launch_conv3x3(const ConvDesc& d, const float* in, float* out, int H, int W, hipStream_t s);

hipError_t launch_conv3x3_f16(const ConvDesc& d, const float* in, float* out, int H, int W, hipStream_t s);
size_t conv_f16_weight_bytes(int cin, int cout_pad, int taps);
// SP16-input layers with >= 32 couts: persistent DMA-staged kernel (conv3x3_sp.hip)
bool conv_sp_supported(const ConvDesc& d);
hipError_t launch_conv3x3_sp(const ConvDesc& d, const void* in, void* out, int H, int W, hipStream_t s);
bool conv_sp_up_form(const ConvDesc& d, int H, int W);   // that launch takes the upsample form (4/9 of the products): own profile family
// fused full-resolution ends of the 16x networks (conv11+conv12+pool / conv12+conv11): see conv3x3_f16.hip
bool conv_fusable_head(const ConvDesc& d0, const ConvDesc& d1);
bool conv_fusable_tail(const ConvDesc& d0, const ConvDesc& d1);
hipError_t launch_enc_head(const ConvDesc& d0, const ConvDesc& d1, const float* img, float* out, int H, int W, hipStream_t s);   // d1.flags & CONV_OUT_SP16
hipError_t launch_dec_tail(const ConvDesc& d0, const ConvDesc& d1, const float* in, float* out, int H, int W, hipStream_t s);
// level 1 without materialising relu1_1 (level1.hip, moments.hip)
bool l1_capable(const ConvDesc& enc0);
hipError_t launch_l1_encode(const ConvDesc& enc0, const float* img, float* out, int H, int W, hipStream_t s);
hipError_t launch_l1_decode(const ConvDesc& enc0, const ConvDesc& dec0_folded, const float* img, float* out, int H, int W, hipStream_t s);
// the 3 -> 64 first conv of the un-pruned encoders in f16x3 (level1.hip in3_wide_kernel): fp32 NHWC or SP16 out
bool in3_wide_capable(const ConvDesc& enc0);
hipError_t launch_in3_wide(const ConvDesc& enc0, const float* img, void* out, int H, int W, bool out_sp, bool exact_fp32, hipStream_t s);
size_t l1_moments_workspace_bytes();
hipError_t launch_l1_moments(const ConvDesc& enc0, const float* img, int H, int W, int x0, int x1, double* sum, double* sumsq,
                             void* workspace, size_t workspace_bytes, hipStream_t s, bool f32_products = false);
// fp32 packed weights (device) -> scaled split-f16 packed weights + inverse scale (device scalar)
//   have_max: *maxbits_dev already holds max |w| (written by launch_fold_affine); otherwise it is computed here
//   nmax > 1: maxbits_dev is an ARRAY of nmax partial maxima (launch_fold_fast's rowmax) that the kernel reduces itself
hipError_t launch_split_pack(const float* wpk32, int cin, int cout_pad, int taps, unsigned* maxbits_dev, void* out,
                             float* inv_scale_out, hipStream_t s, bool have_max = false, int nmax = 1);
// the same weights (cout_pad 16, 3 real couts) in the block-packed layout; *maxbits_dev must hold max |w| (same scale as above)
size_t conv_phase_weight_bytes(int cin);
hipError_t launch_split_pack_phase(const float* wpk32, int cin, const unsigned* maxbits_dev, void* out, hipStream_t s, int nmax = 1);

// ---- image edge: uint8 HWC <-> planar fp32 (ToTensor / save_image of the reference's harness)
hipError_t launch_u8_to_planar(const uint8_t* hwc, long npix, float* planar, hipStream_t s);
hipError_t launch_planar_to_u8(const float* planar, long npix, uint8_t* hwc, int round_mode, hipStream_t s);
// *dst = (double)*counter on the stream (wct_range_flag_f64: the saturation counter as a value a sharded run can all-reduce)
hipError_t launch_counter_to_f64(const unsigned* counter, double* dst, hipStream_t s);
// ---- image edge: transforms.Resize = Pillow's bilinear resampler, bit-exact (resize.hip)
void resize_axis_tables(int in_size, int out_size, int& ksize, std::vector<int>& bounds, std::vector<int>& kk);   // host
hipError_t launch_resize_u8(const uint8_t* in, int H, int W, int oH, int oW, const int* bounds_h, const int* kk_h, int ksize_h,
                            const int* bounds_v_shifted, const int* kk_v, int ksize_v, int row0, int rows, uint8_t* tmp, uint8_t* out,
                            float* planar, hipStream_t s);

// ---- pitched block copy of floats (rows x width; the sharded cascade's level crops, halo packing and assembly: a planar 3 x H x W image is
//      3H rows of pitch W)
hipError_t launch_copy_block(const float* src, long src_pitch, float* dst, long dst_pitch, long rows, int width, hipStream_t s);
struct CopySegs { const float* src[3]; float* dst[3]; long src_pitch[3], dst_pitch[3]; int width[3]; };   // width 0: segment unused
hipError_t launch_copy_blocks(const CopySegs& g, long rows, hipStream_t s);

// ---- layout
hipError_t launch_nhwc_to_nchw(const float* in, float* out, int C, int npix, hipStream_t s);
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int C, int npix, hipStream_t s);

// ---- moments: raw sums  sum[c] = SUM_p x[p][c],  sumsq[a][b] = SUM_p x[p][a] x[p][b]   (fp64)
size_t moments_workspace_bytes(int C, long npix);
// window = rows [0,h) x cols [x0,x1) of an NHWC map of width wfull
// f32_products: products and 64-pixel block sums on the fp32 matrix cores, block totals in fp64 (moments.hip F32 variant); false: fp64 throughout
hipError_t launch_moments(const float* feat_nhwc, int C, int h, int wfull, int x0, int x1, double* sum,
                          double* sumsq, void* workspace, size_t workspace_bytes, hipStream_t s, bool f32_products = false);

// ---- solve, in two steps (solve.hip): moments of one map -> EigResult; two EigResults -> M (C x C), b (C)
//      csF = M cF + b   (util_wct.py:62-131, 219).  EigResult = doubles G[C*C] | lam[C] | mu[C] | floor | pad
size_t eig_result_bytes(int C);
size_t eig_result_F_offset(size_t C);   // doubles: where F = cov^(+-1/2) starts inside an EigResult (mu is at C*C + C)
size_t eig_workspace_bytes(int C);
size_t assemble_workspace_bytes(int C);
//   diag_add is added to the covariance's diagonal (1.0 on the content side = the reference's `--numpy` variant)
//   wide_model: the module set has feature maps wider than 128 channels (--mode original); its 128-channel level then takes the
//   deflated iteration too (ill-conditioned there: the 26-iteration budget ran out and the 2 ms LDS Jacobi took over)
//   ok_defer (device int, C > 128 path only -- eig_is_big): instead of reading the iteration's outcome back and synchronising the
//   stream inside the call (to decide about the slow global-memory Jacobi net), the outcome is written there and the CALLER checks
//   it at its own synchronisation point, re-running without deferral if it is 0
bool eig_is_big(int C, bool wide_model);
hipError_t launch_eig(int C, double n, const double* sum, const double* sumsq, int inverse, double* res, int* info_dev,
                      void* workspace, size_t workspace_bytes, hipStream_t s, double diag_add = 0.0, bool wide_model = false,
                      int* ok_defer = nullptr, int coop_xcd = -1, unsigned* coop_state = nullptr, int* coop_epoch = nullptr, unsigned* coop_aborts = nullptr, bool* coop_used_out = nullptr);
//   coop_xcd (0..7, or -1: off) + coop_state (32 device bytes of the calling lane, zero at first use) + coop_epoch (the lane's
//   HOST counter of such solves, advanced here): the 128-channel levels of --mode 16x as ONE launch on that XCD (solve.hip
//   ns_coop128_kernel); lanes that may solve at the same time are given different XCDs
hipError_t launch_assemble(int C, const double* eig_c, const double* eig_s, double alpha, double rel_thresh,
                           double* M, double* b, void* workspace, size_t workspace_bytes, hipStream_t s);

// ---- fold csF = M x + b into a decoder's first conv:  W' = W o M, b' = bias + W o b
//      w_oihw: device [cout][cin][3][3] fp32 (original weights), out: packed weights + bias for ConvDesc
//      maxbits_dev (optional): receives max |W'| as float bits for launch_split_pack(..., have_max = true)
// the same fold for the wide models' decoders (cin = 256 / 512) as an fp64 matrix-core GEMM (solve.hip fold_gemm_kernel): `rows` =
// the layer's weights as doubles [(o, tap)][c] followed by their tap sums [o][c] (fold_gemm_rows_doubles; launch_fold_rows, once)
bool fold_gemm_capable(int cout, int cin, int cout_pad);
size_t fold_gemm_rows_doubles(int cout, int cin);
hipError_t launch_fold_rows(const float* w_oihw, int cout, int cin, double* rows, hipStream_t s);
hipError_t launch_fold_gemm(const double* rows, const float* bias, int cout, int cin, int cout_pad, const double* M, const double* b,
                            float* wpk_out, float* bias_out, unsigned* maxbits_dev, hipStream_t s);
hipError_t launch_fold_affine(const float* w_oihw, const float* bias, int cout, int cin, int cout_pad,
                              const double* M, const double* b, float* wpk_out, float* bias_out,
                              unsigned* maxbits_dev, hipStream_t s);
// The same fold without C x C products on the content side's critical path (misc.hip): style-side part once per style
// (buf: fold_style_doubles(cout, cin) doubles), then Wc = cov_c^(-1/2), mu_c -> folded weights + bias + per-output-row max |W'|
// (rowmax_dev: cout_pad words, consumed by launch_split_pack(..., nmax = cout_pad)); cin <= 128
bool fold_fast_capable(int cin);
size_t fold_style_doubles(int cout, int cin);
hipError_t launch_fold_style(const float* w_oihw, int cout, int cin, const double* Ss, const double* mu_s, double* buf, hipStream_t s);
hipError_t launch_fold_fast(const float* w_oihw, const float* bias, int cout, int cin, int cout_pad, const double* style_buf, const double* Wc,
                            const double* mu_c, double alpha, float* wpk_out, float* bias_out, unsigned* rowmax_dev, hipStream_t s);
// pack [cout][cin][3][3] (+ bias) into the conv kernel's layout (device side, used for the apply conv)
hipError_t launch_pack_center_tap(const double* M, const double* b, int C, int cout_pad, float* wpk_out,
                                  float* bias_out, hipStream_t s);
