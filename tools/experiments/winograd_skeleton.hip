// VERDICT r4 task 1: "build the fewer-products convolution family and put a number on it" -- the go / no-go measurement.
//
// DESIGN 10.2 (round 4) validated Winograd's ARITHMETIC in f16x3 against the reference's pixels and showed on paper that the fused
// F(2x2, 3x3) kernel does not fit (16 accumulator sets, 4x the input volume in LDS, 3.5x the weight bytes).  The "cheaper first cut" the
// verdict names is F(2, 3) along x only: per output PAIR and input row 4 products instead of 6 (1.5x fewer MFMAs), 4 accumulator
// sets per pair instead of 2 (= 2x the registers per output), a transformed input V of 2x the halo volume, 12 weight slabs
// (4 positions x 3 rows) instead of 9 (1.33x).  This file measures what such a kernel can reach ON THE CHIP, as instruction-stream
// skeletons in the style of mfma_loop.hip (real DMA, real LDS layouts and operand reads, real MFMA counts, real transform arithmetic
// on random data, accumulator stores; no border logic, results not checked -- an UPPER bound for a real kernel of the same shape):
//
//   D    the direct kernel's skeleton (mfma_loop.hip V5): 32 x 16 px x 64 couts per workgroup, two DMA stages of 75 KB, 8 operand
//        reads per 12 MFMAs, accumulators stored every 8th chunk (cin = 128)
//   W    F(2, 3)-x as it FITS in 160 KB: 32 x 8 px (128 pairs) x 32 couts per workgroup -- halo 22 KB x 2 stages (DMA), weights
//        12 slabs = 24 KB x 2 stages (DMA), V 40 KB x 1 (a second V stage or 64 couts' weights x 2 do not fit: 170 / 178 KB), so per
//        chunk: barrier, TRANSFORM (halo -> fp32 -> 4 positions -> split hi / lo -> ds_write), barrier, 18 MFMAs per wave at 6
//        operand reads per 6 MFMAs; wave = (pair of N tiles, position): 32 accumulator registers
//   Wt   W's transform phase alone (what the split arithmetic + LDS round trip costs per chunk)
//   Wm   W without the transform: V arrives by DMA at 2x the bytes (as if the PRODUCING layer's epilogue had written the transformed
//        map -- 2x the activation bytes in HBM): the upper bound of that architecture
// All in "algorithmic TFLOP/s" = 2 * 9 * 16 * pixels * couts per chunk / time: directly comparable, the direct kernel's unit.
//   hipcc --offload-arch=gfx950 -O3 -o winograd_skeleton winograd_skeleton.hip && RANDOM_FILL=1 ./winograd_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// ------------------------------------------------------------------------------------------------ D: direct skeleton (= mfma_loop.hip kd<true>)
constexpr int NPP = 640, COW = 64, FHW = 34;
__global__ __launch_bounds__(512) void k_direct(float* out, int chunks, const u32x4* gact, size_t stride16, const u32x4* gwgt, u32x4* gout, int store_every) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * NPP + 36 * COW;
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  for (int e = tid; e < 2 * STAGE; e += 512) lds[e] = u32x4{0x3c003c00u + e, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
  __syncthreads();
  f32x16 acc[2][2];
  for (int c = 0; c < 2; ++c) for (int p = 0; p < 2; ++p) for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;
  auto issue = [&](int ch, int stage) {
    u32x4* act = lds + stage * STAGE;
    u32x4* wgt = act + 4 * NPP;
    const u32x4* ga = gact + ((size_t)blockIdx.x * chunks + ch) * stride16;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = wave + 8 * i;
      __builtin_amdgcn_global_load_lds(ga + idx * 64 + lane, (lds_ptr)(act + idx * 64), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int idx = wave + 8 * i; idx = idx < 36 ? idx : 35;
      __builtin_amdgcn_global_load_lds(gwgt + (size_t)((ch & 7) * 36 + idx) * 64 + lane, (lds_ptr)(wgt + idx * 64), 16, 0, 0);
    }
  };
  issue(0, 0);
  for (int ch = 0; ch < chunks; ++ch) {
    const int stage = ch & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ch + 1 < chunks) issue(ch + 1, stage ^ 1);
    const u32x4* act = lds + stage * STAGE;
    const u32x4* wgt = act + 4 * NPP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      f16x8 bh[2], bl[2], ah[2], al[2];
      for (int p = 0; p < 2; ++p) {
        const int pix = (wave * 2 + p + dy) * FHW + li + dx;
        bh[p] = __builtin_bit_cast(f16x8, act[(0 * 2 + kh) * NPP + pix]);
        bl[p] = __builtin_bit_cast(f16x8, act[(1 * 2 + kh) * NPP + pix]);
      }
      for (int c = 0; c < 2; ++c) {
        ah[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 0) * 2 + kh) * COW + c * 32 + li]);
        al[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 1) * 2 + kh) * COW + c * 32 + li]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bh[p], acc[c][p], 0, 0, 0);
          acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bl[p], acc[c][p], 0, 0, 0);
          acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c], bh[p], acc[c][p], 0, 0, 0);
        }
    }
    if ((ch % store_every) == store_every - 1) {
      u32x4* go = gout + ((size_t)blockIdx.x * (chunks / store_every) + ch / store_every) * 8192 + wave * 1024 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            go[((c * 2 + p) * 4 + q) * 64] = u32x4{__float_as_uint(acc[c][p][4 * q]), __float_as_uint(acc[c][p][4 * q + 1]), __float_as_uint(acc[c][p][4 * q + 2]), __float_as_uint(acc[c][p][4 * q + 3])};
            acc[c][p][4 * q] = 0.f;
          }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 2; ++c) for (int p = 0; p < 2; ++p) for (int r = 0; r < 16; ++r) s += acc[c][p][r];
  out[blockIdx.x * 512 + tid] = s;
}

// ------------------------------------------------------------------------------------------------ W: F(2, 3) along x, the variant that fits
// LDS (16-byte slots): halo stage = 352 px x 4 pieces (34 x 10 = 340 halo pixels, DMA'd as 22 wave-instructions), weight stage =
// [12 (position, dy)][hl][kh][32 couts] = 1536, V = [4 positions][10 rows][16 pairs] records x 4 pieces = 2560.
constexpr int WH_PX = 352, WH = WH_PX * 4, WU = 12 * 2 * 2 * 32, WV = 4 * 10 * 16 * 4;
constexpr int W_STAGE = WH + WU;                       // one DMA stage: 2944 slots = 46 wave-instructions
__device__ __forceinline__ int rec_slot(int rec, int q) { return rec * 4 + (q ^ ((rec >> 2) & 3)); }   // conv3x3_sp.hip sp_slot

struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float x0, float x1) {      // conv_f16_dev.h split2: 1.5 instructions per value
  const f16x2 h = {(_Float16)x0, (_Float16)x1};
  HiLo r;
  r.hi = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r.lo) : "v"(r.hi), "v"(x0), "v"(x1));
  return r;
}
__device__ __forceinline__ float clamp_pm(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

// MODE 0: W (transform + MFMA), 1: Wt (transform only), 2: Wm (V by DMA, MFMA only)
template <int MODE>
__global__ __launch_bounds__(512) void k_wino(float* out, int chunks, const u32x4* gact, size_t stride16, const u32x4* gwgt, u32x4* gout, int store_every) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  // MODE 2 has no halo: its DMA stage is [V 2560 | U 1536]; MODE 0 / 1: two stages [halo | U] + one V
  constexpr int STG = MODE == 2 ? WV + WU : W_STAGE;
  u32x4* vbuf1 = lds + 2 * STG;                          // MODE 0 / 1: the single V buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int total = 2 * STG + (MODE == 2 ? 0 : WV);
  for (int e = tid; e < total; e += 512) lds[e] = u32x4{0x3c003c00u + e, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
  __syncthreads();
  const int ntp = wave & 1, pos = wave >> 1;             // this wave: N tiles 2 ntp, 2 ntp + 1 (rows 4 ntp .. 4 ntp + 3), one position
  f32x16 acc[2];
  for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  constexpr int NDMA = STG / 64;                         // 46 (MODE 0 / 1) or 64 (MODE 2) wave-instructions per chunk
  auto issue = [&](int ch, int stage) {
    const u32x4* ga = gact + ((size_t)blockIdx.x * chunks + ch) * stride16;
    constexpr int NACT = (MODE == 2 ? WV : WH) / 64;
#pragma unroll
    for (int i = 0; i < (NDMA + 7) / 8; ++i) {
      int idx = wave + 8 * i; idx = idx < NDMA ? idx : NDMA - 1;
      const u32x4* g = idx < NACT ? ga + idx * 64 + lane : gwgt + (size_t)((ch & 7) * 24 + (idx - NACT)) * 64 + lane;
      __builtin_amdgcn_global_load_lds(g, (lds_ptr)(lds + stage * STG + idx * 64), 16, 0, 0);
    }
  };
  issue(0, 0);
  for (int ch = 0; ch < chunks; ++ch) {
    const int stage = ch & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ch + 1 < chunks) issue(ch + 1, stage ^ 1);
    const u32x4* st = lds + stage * STG;
    const u32x4* vsrc = MODE == 2 ? st : vbuf1;
    const u32x4* wgt = st + (MODE == 2 ? WV : WH);
    if (MODE != 2) {
      // ---- transform: item = (input row 0..9, pair 0..15, channel half kh) = 320 items on 512 lanes; an item reads the hi and lo piece of
      // four pixels (8 x ds_read_b128), rebuilds d = hi + lo (v_fma_mix_f32-class: one instruction per value), forms the four positions
      // (one add / sub each), splits (1.5 per value + the clamp) and writes 4 x (hi, lo) pieces
      const u32x4* halo = st;
      const int item = tid;
      if (item < 320) {
        const int ikh = item & 1, pr = (item >> 1) & 15, row = item >> 5;
        float d[4][8];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int pix = row * FHW + 2 * pr + x;
          const f16x8 h = __builtin_bit_cast(f16x8, halo[rec_slot(pix, 2 * ikh)]);
          const f16x8 l = __builtin_bit_cast(f16x8, halo[rec_slot(pix, 2 * ikh + 1)]);
#pragma unroll
          for (int c = 0; c < 8; ++c) d[x][c] = (float)h[c] + (float)l[c];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          u32x4 vh, vl;
#pragma unroll
          for (int c = 0; c < 8; c += 2) {
            float v0, v1;
            if (p == 0) { v0 = d[0][c] - d[2][c]; v1 = d[0][c + 1] - d[2][c + 1]; }
            else if (p == 1) { v0 = d[1][c] + d[2][c]; v1 = d[1][c + 1] + d[2][c + 1]; }
            else if (p == 2) { v0 = d[2][c] - d[1][c]; v1 = d[2][c + 1] - d[1][c + 1]; }
            else { v0 = d[1][c] - d[3][c]; v1 = d[1][c + 1] - d[3][c + 1]; }
            const HiLo t = split2(clamp_pm(v0), clamp_pm(v1));
            vh[c >> 1] = t.hi; vl[c >> 1] = t.lo;
          }
          const int rec = (p * 10 + row) * 16 + pr;
          vbuf1[rec_slot(rec, 2 * ikh)] = vh;
          vbuf1[rec_slot(rec, 2 * ikh + 1)] = vl;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (MODE != 1) {
      // ---- products: per dy, A = this position's slab (hi, lo), B = the two N tiles' records (hi, lo): 6 reads, 6 MFMAs
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int row = 4 * ntp + 2 * n + (li >> 4) + dy, rec = (pos * 10 + row) * 16 + (li & 15);
          bh[n] = __builtin_bit_cast(f16x8, vsrc[rec_slot(rec, 2 * kh)]);
          bl[n] = __builtin_bit_cast(f16x8, vsrc[rec_slot(rec, 2 * kh + 1)]);
        }
        const f16x8 ah = __builtin_bit_cast(f16x8, wgt[(((pos * 3 + dy) * 2 + 0) * 2 + kh) * 32 + li]);
        const f16x8 al = __builtin_bit_cast(f16x8, wgt[(((pos * 3 + dy) * 2 + 1) * 2 + kh) * 32 + li]);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[n], acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[n], acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[n], acc[n], 0, 0, 0);
        }
      }
    } else {
      acc[0][0] += __uint_as_float(vbuf1[tid][0]);      // keep the transform alive
    }
    if ((ch % store_every) == store_every - 1) {
      // (the output transform Y0 = M0 + M1 + M2, Y1 = M1 - M2 - M3 across the four position waves is left out: a few adds and one LDS
      // exchange per `store_every` chunks; the stores are what counts: 256 px x 32 couts x 4 B = 32 KB per workgroup)
      u32x4* go = gout + ((size_t)blockIdx.x * (chunks / store_every) + ch / store_every) * 2048 + (wave >> 1) * 512 + (wave & 1) * 256 + lane;
      if ((wave >> 1) < 2) {     // two of the four position waves' worth of registers = the 2 outputs per pair
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            go[(n * 2 + q) * 64] = u32x4{__float_as_uint(acc[n][8 * q]), __float_as_uint(acc[n][8 * q + 1]), __float_as_uint(acc[n][8 * q + 2]), __float_as_uint(acc[n][8 * q + 3])};
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[n][4 * q] = 0.f;
    }
  }
  float s = 0.f;
  for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ void fill_rand(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const unsigned lo = (x & 0x83ffu) | 0x3800u, hi = ((x >> 16) & 0x83ffu) | 0x3800u;   // two f16 in [-2, 2)
    p[i] = lo | (hi << 16);
  }
}

int main() {
  const int wgs = 256, ch = 400, se = 8;                // 8 chunks per output tile: cin = 128
  float* out; hipMalloc(&out, wgs * 512 * 4);
  u32x4 *ga, *gw, *go;
  const size_t per = 2560;                               // 16-byte units per (workgroup, chunk): covers 40 KB (direct) and V's 40 KB
  const size_t big = (size_t)wgs * ch * per;
  hipMalloc(&ga, big * 16); hipMalloc(&gw, 8 * 36 * 64 * 16); hipMalloc(&go, (size_t)wgs * (ch / se) * 8192 * 16);
  hipMemset(ga, 0x3c, big * 16); hipMemset(gw, 0x3c, 8 * 36 * 64 * 16);
  if (getenv("RANDOM_FILL")) {
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (unsigned*)ga, big * 4, 1u);
    hipLaunchKernelGGL(fill_rand, dim3(64), dim3(256), 0, 0, (unsigned*)gw, (size_t)8 * 36 * 64 * 4, 7u);
    hipDeviceSynchronize();
    printf("random operands in [-2,2):\n");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, size_t lds, double px, double co, size_t stride16) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, out, 16, ga, stride16, gw, go, se);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed (%s)\n", name, hipGetErrorString(hipGetLastError())); return; }
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, out, ch, ga, stride16, gw, go, se); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double flop = (double)wgs * ch * px * co * 16 * 9 * 2;
    printf("%-78s %7.3f ms  %6.1f algorithmic TF  (%.0f ns per chunk-job; LDS %.1f KB)\n", name, best, flop / best / 1e9, best * 1e6 / ch, lds / 1024.0);
  };
  run(k_direct, "D   direct, 32x16 px x 64 co, 2 stages, 8 reads / 12 MFMAs", (size_t)2 * (4 * NPP + 36 * COW) * 16, 512, 64, 2560);
  run(k_wino<0>, "W   F(2,3)-x as it fits: 32x8 px x 32 co, transform + MFMA, V single-buffered", (size_t)(2 * W_STAGE + WV) * 16, 256, 32, WH);
  run(k_wino<1>, "Wt  W's transform phase alone", (size_t)(2 * W_STAGE + WV) * 16, 256, 32, WH);
  run(k_wino<2>, "Wm  W with V delivered by DMA (producer-side transform, 2x activation bytes)", (size_t)2 * (WV + WU) * 16, 256, 32, WV);
  return 0;
}
