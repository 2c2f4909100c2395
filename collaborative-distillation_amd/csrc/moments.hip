// Raw fp64 moments of an NHWC fp32 feature map:  sum[c] = SUM_p x[p][c],  sumsq[a][b] = SUM_p x[p][a]*x[p][b].
//
// This is the data-dependent half of the reference's whitening/colouring transform
// (PytorchWCT/util_wct.py:68-70 content, :94-96 style): torch.mean(cF,1), cF - mean, mm(cF, cF.t())/(hw-1),
// all in fp64 on the host.  Here the features never leave HBM: one pass reads them once (coalesced 16-B
// loads staged through LDS), converts to fp64 in registers and accumulates x x^T on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), so products and sums are fp64 exactly like the reference's.  mean / covariance
// follow in solve.hip as  mu = sum/n,  cov = (sumsq - n mu mu^T)/(n-1)
// (raw sums, not centred ones, because they are what a content-sharded run all-reduces across GPUs).
//
// Decomposition: the C x C output is cut into 16x16 tiles, upper triangle only (NP pairs I <= J).
// grid = (pixel chunks, pair groups); a workgroup (4 waves) walks its pixel chunk in LDS tiles of 64 pixels;
// wave w owns up to 6 tile pairs (interleaved over the group's 24) for the WHOLE chunk, so accumulators stay
// in registers and there is no cross-wave reduction.  LDS row stride Cs == 16 (mod 32) dwords makes the
// operand reads (lane = channel l&15 of pixel l>>4) bank-conflict free.  Partial tiles go to a workspace and a
// second kernel adds them in a fixed order -> bitwise reproducible, no atomics.
//
// F32 variant (round 4; debug key "mom32", default on for maps of >= MOM32_MIN_PIXELS pixels): the products and the sums of
// MOM32_FLUSH x 4 = 64 consecutive pixels run on v_mfma_f32_16x16x4_f32 (twice the fp64 matrix-core rate, no v_cvt_f64_f32 in front of
// every operand -- the LDS-read -> convert -> fp64-MFMA chain is what bounds the fp64 form, tools/experiments/cvt_rate.hip), and each
// such block is then added to the fp64 accumulators.  Arithmetic: a block's 64 fp32 products x_a x_b are summed in fp32 (relative
// error <= 64 x 2^-24 worst case, ~1e-6 typical, zero-mean), the N / 64 blocks in fp64 -- the total's relative error is ~1e-6 /
// sqrt(N / 64): 1e-8 at the 518 400 pixels of relu3_1 at 4K, against 1e-16 for the fp64 form and ~1e-7 for a covariance that the
// reference's own fp32 FEATURES already carry (each x is an fp32 rounding of the exact activation).  Measured end to end: see DESIGN 9.
#include "wct_common.h"
#include "conv_f16_dev.h"
#include <algorithm>
#include <utility>
#include <cstdlib>

namespace {


struct MomArgs {
  const float* x;
  int C, T, NP, NPG, NPC, Cs, MP, pixsplit;  // MP: pixels per LDS tile
  int pw;                // tile pairs a wave owns at most (3 or 6); a pair group is 4 * pw pairs
  long npix, chunk;      // pixels per chunk (multiple of MP)
  int wfull, x0, wwin;   // window: pixel p -> (row p / wwin, col x0 + p % wwin) of a map of width wfull
  double* part_sq;       // [NPC][NP][256]
  double* part_sum;      // [NPC][T*16]
};

constexpr int MAXLD = 8;  // upper bound of float4 loads per thread per tile (MP * C / 4 / 256 <= 8)

// THREE tile pairs x steps [st0, st1) of the LDS tile, 4 steps at a time: all 24 LDS operand reads of a group are
// issued before the first conversion / MFMA and nothing in here is conditional, so the scheduler overlaps LDS
// latency with the matrix pipe instead of serialising read -> wait -> MFMA.  A wave's pair list is processed in
// triples (unused slots alias a valid LDS column; their accumulators are never stored).
constexpr int MOM32_FLUSH = 16;     // steps (of 4 pixels) whose products are summed in fp32 before they are added to the fp64 accumulators

// F32 = false: fp64 products on v_mfma_f64_16x16x4_f64, accumulators in ITS D layout (row = (lane >> 4) + 4 reg).
// F32 = true : fp32 products on v_mfma_f32_16x16x4_f32 per block of MOM32_FLUSH steps, then added in fp64; the fp64 accumulators are then
//              in the fp32 instruction's D layout (row = 4 (lane >> 4) + reg) -- mom_row() below is what the stores use.
template <bool F32>
__device__ __forceinline__ int mom_row(int pk, int r) { return F32 ? 4 * pk + r : pk + 4 * r; }

template <bool F32>
__device__ __forceinline__ void tile_steps3(const float* lds, int Cs, int st0, int st1, int pk, const int* offA, const int* offB,
                                            f64x4& c0, f64x4& c1, f64x4& c2, double& s0, double& s1, double& s2) {
  const int oa0 = offA[0], oa1 = offA[1], oa2 = offA[2], ob0 = offB[0], ob1 = offB[1], ob2 = offB[2];
  if constexpr (F32) {
    for (int sb = st0; sb < st1; sb += MOM32_FLUSH) {
      const int se = sb + MOM32_FLUSH < st1 ? sb + MOM32_FLUSH : st1;
      f32x4 f0 = f32x4{0.f, 0.f, 0.f, 0.f}, f1 = f0, f2 = f0;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int st = sb; st < se; st += 4) {
        float av[4][3], bv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* row = lds + ((st + u) * 4 + pk) * Cs;
          av[u][0] = row[oa0]; bv[u][0] = row[ob0];
          av[u][1] = row[oa1]; bv[u][1] = row[ob1];
          av[u][2] = row[oa2]; bv[u][2] = row[ob2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          t0 += av[u][0]; t1 += av[u][1]; t2 += av[u][2];
          f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bv[u][0], f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bv[u][1], f1, 0, 0, 0);
          f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bv[u][2], f2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0[r] += (double)f0[r]; c1[r] += (double)f1[r]; c2[r] += (double)f2[r]; }
      s0 += (double)t0; s1 += (double)t1; s2 += (double)t2;
    }
    return;
  }
  for (int st = st0; st < st1; st += 4) {
    float av[4][3], bv[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = lds + ((st + u) * 4 + pk) * Cs;
      av[u][0] = row[oa0]; bv[u][0] = row[ob0];
      av[u][1] = row[oa1]; bv[u][1] = row[ob1];
      av[u][2] = row[oa2]; bv[u][2] = row[ob2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double a0 = (double)av[u][0], a1 = (double)av[u][1], a2 = (double)av[u][2];
      s0 += a0; s1 += a1; s2 += a2;
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, (double)bv[u][0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, (double)bv[u][1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, (double)bv[u][2], c2, 0, 0, 0);
    }
  }
}

// two pairs (l1_moments_kernel's packed tiling of 24 channels).  oa / ob: [pair][parity of the step u] column offsets of the lane -- two
// sets because l1_moments_kernel's feature tile XOR-swizzles its columns with bits of the pixel index (see there); st0 is a multiple of 4.
template <bool F32>
__device__ __forceinline__ void tile_steps2(const float* lds, int Cs, int st0, int st1, int pk, const int (&oa)[2][2], const int (&ob)[2][2],
                                            f64x4& c0, f64x4& c1, double& s0, double& s1) {
  if constexpr (F32) {
    for (int sb = st0; sb < st1; sb += MOM32_FLUSH) {
      const int se = sb + MOM32_FLUSH < st1 ? sb + MOM32_FLUSH : st1;
      f32x4 f0 = f32x4{0.f, 0.f, 0.f, 0.f}, f1 = f0;
      float t0 = 0.f, t1 = 0.f;
      for (int st = sb; st < se; st += 4) {
        float av[4][2], bv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* row = lds + ((st + u) * 4 + pk) * Cs;
          av[u][0] = row[oa[0][u & 1]]; bv[u][0] = row[ob[0][u & 1]];
          av[u][1] = row[oa[1][u & 1]]; bv[u][1] = row[ob[1][u & 1]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          t0 += av[u][0]; t1 += av[u][1];
          f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bv[u][0], f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bv[u][1], f1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0[r] += (double)f0[r]; c1[r] += (double)f1[r]; }
      s0 += (double)t0; s1 += (double)t1;
    }
    return;
  }
  for (int st = st0; st < st1; st += 4) {
    float av[4][2], bv[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = lds + ((st + u) * 4 + pk) * Cs;
      av[u][0] = row[oa[0][u & 1]]; bv[u][0] = row[ob[0][u & 1]];
      av[u][1] = row[oa[1][u & 1]]; bv[u][1] = row[ob[1][u & 1]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double a0 = (double)av[u][0], a1 = (double)av[u][1];
      s0 += a0; s1 += a1;
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, (double)bv[u][0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, (double)bv[u][1], c1, 0, 0, 0);
    }
  }
}

// NLD: float4 load slots per thread per tile = ceil(MP * C / 4 / 256); PW: tile pairs a wave can own (3 / 6 / 9 -- sized to
// the problem so that small-C launches do not carry 9 accumulators); DEPTH: tiles in flight towards HBM (1 or 2)
template <int NLD, int PW, int DEPTH, bool F32>
__global__ __launch_bounds__(256, 2) void moments_kernel(MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);  // [MP][Cs]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches below
  const int c = lane & 15, pk = lane >> 4;
  const int pc = blockIdx.x, pg = blockIdx.y;
  const int MP = a.MP;
  // work split inside the workgroup:
  //   pixel split (NP <= 6, i.e. C <= 48): every wave owns ALL pairs for a quarter of each tile's pixels (3..6
  //     independent accumulator chains per wave, all four waves busy); the four partial sets are added in LDS;
  //   pair split: wave w owns pairs pg * 4 PW + w + 4j for the whole tile.
  const bool pixsplit = a.pixsplit != 0;
  int offA[PW], offB[PW], pidx[PW];
  bool diag[PW];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int idx = pixsplit ? j : pg * (4 * PW) + wave + 4 * j;
    offA[j] = offB[j] = 0; pidx[j] = 0; diag[j] = false;
    if (idx < a.NP) {
      int I = 0, rem = idx;
      while (rem >= a.T - I) { rem -= a.T - I; ++I; }
      offA[j] = I * 16 + c; offB[j] = (I + rem) * 16 + c; pidx[j] = idx; diag[j] = rem == 0;
      cnt = j + 1;
    }
  }
  cnt = __builtin_amdgcn_readfirstlane(cnt);
  const int steps = MP / 4;
  const int st0 = pixsplit ? wave * (steps / 4) : 0, st1 = pixsplit ? st0 + steps / 4 : steps;
  f64x4 acc[PW];
  double s[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) { acc[j] = f64x4{0., 0., 0., 0.}; s[j] = 0.; }
  // zero the whole tile once: columns >= C (C = 24 -> T*16 = 32) and the row padding are never loaded
  for (int e = tid; e < MP * a.Cs; e += 256) lds[e] = 0.f;

  const long p0 = (long)pc * a.chunk, p1 = min(a.npix, p0 + a.chunk);
  const int c4n = a.C >> 2;
  // this thread's float4 slots of a tile (the same for every tile): pixel-in-tile and channel quad
  int lpix[NLD], lc4[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int e = tid + 256 * k;
    lpix[k] = e / c4n; lc4[k] = e - lpix[k] * c4n;
    if (lpix[k] >= MP) lpix[k] = -1;
  }
  // Loads are UNCONDITIONAL (addresses clamped into the chunk; out-of-range slots are zeroed when they are written
  // to LDS): a load guarded by a per-lane condition makes hipcc branch around it and wait for it on the spot,
  // which serialises the tile fetch and defeats the prefetch (cdna_hip_programming.md, "three .s-level traps" (c)).
  auto fetch = [&](long pt, f32x4 (&v)[NLD]) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      long pp = pt + (lpix[k] < 0 ? 0 : lpix[k]);
      pp = pp < p1 ? pp : p1 - 1;
      long g = pp;
      if (a.wwin != a.wfull) { const long r = pp / a.wwin; g = r * a.wfull + a.x0 + (pp - r * a.wwin); }
      v[k] = *reinterpret_cast<const f32x4*>(a.x + g * a.C + (lpix[k] < 0 ? 0 : lc4[k]) * 4);
    }
  };
  // software pipeline, TWO tiles deep: while tile t is multiplied, the loads of tiles t+1 and t+2 are in flight (one LDS
  // buffer, the next two tiles wait in registers).  One tile ahead is not enough: the workgroups of a launch run in
  // lockstep, so with a single tile in flight the whole chip alternates between "everybody loads" (HBM queues drain in
  // ~4 us) and "everybody multiplies" (HBM idle) -- measured 1.5 TB/s at C = 32 with depth 1.
  f32x4 nxa[NLD], nxb[DEPTH == 2 ? NLD : 1];
  fetch(p0, nxa);
  if constexpr (DEPTH == 2) fetch(p0 + MP < p1 ? p0 + MP : p0, nxb);
  auto step = [&](long pt, f32x4 (&cur)[NLD]) {
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (lpix[k] >= 0) {
        const bool ok = pt + lpix[k] < p1;
        *reinterpret_cast<f32x4*>(lds + lpix[k] * a.Cs + lc4[k] * 4) = ok ? cur[k] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
    if (pt + DEPTH * MP < p1) fetch(pt + DEPTH * MP, cur);   // refill the register set that was just written out
    if (cnt > 0) tile_steps3<F32>(lds, a.Cs, st0, st1, pk, offA, offB, acc[0], acc[1], acc[2], s[0], s[1], s[2]);
    if constexpr (PW > 3) { if (cnt > 3) tile_steps3<F32>(lds, a.Cs, st0, st1, pk, offA + 3, offB + 3, acc[3], acc[4], acc[5], s[3], s[4], s[5]); }
    if constexpr (PW > 6) { if (cnt > 6) tile_steps3<F32>(lds, a.Cs, st0, st1, pk, offA + 6, offB + 6, acc[6], acc[7], acc[8], s[6], s[7], s[8]); }
  };
  if constexpr (DEPTH == 2) {
    for (long pt = p0; pt < p1; pt += 2 * MP) {
      step(pt, nxa);
      if (pt + MP < p1) step(pt + MP, nxb);
    }
  } else {
    for (long pt = p0; pt < p1; pt += MP) step(pt, nxa);
  }
  // D layout: col = lane & 15, row = mom_row<F32>(lane >> 4, reg)
  if (pixsplit) {
    // add the four waves' partial sets through LDS (reusing the tile buffer; sized for it on the host)
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);  // [4][NP][256] + [4][T*16]
    double* reds = red + (size_t)4 * a.NP * 256;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      if (j < cnt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((size_t)wave * a.NP + j) * 256 + mom_row<F32>(pk, r) * 16 + c] = acc[j][r];
        if (diag[j]) {
          double v = s[j];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          if (pk == 0) reds[wave * a.T * 16 + offA[j]] = v;
        }
      }
    }
    __syncthreads();
    const int nsq = a.NP * 256;
    for (int e = tid; e < nsq; e += 256)
      a.part_sq[(size_t)pc * nsq + e] = (red[e] + red[nsq + e]) + (red[2 * nsq + e] + red[3 * nsq + e]);
    const int ns = a.T * 16;
    for (int e = tid; e < ns; e += 256)
      a.part_sum[(size_t)pc * ns + e] = (reds[e] + reds[ns + e]) + (reds[2 * ns + e] + reds[3 * ns + e]);
    return;
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    if (j < cnt) {
      double* dst = a.part_sq + ((size_t)pc * a.NP + pidx[j]) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[mom_row<F32>(pk, r) * 16 + c] = acc[j][r];
      if (diag[j]) {  // the diagonal tile's owner also owns sum over that tile's channels
        double v = s[j];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (pk == 0) a.part_sum[(size_t)pc * a.T * 16 + offA[j]] = v;
      }
    }
  }
}

// ---- round 5: the big shallow maps (C = 32 / 64, fp32 block products) WITHOUT the LDS round trip.
// v_mfma_f32_16x16x4_f32 takes A[i][k] from lane (i = l & 15, k = l >> 4) and B[k][j] from lane (j = l & 15, k = l >> 4): with
// i = j = "channel slot" and k = "pixel of the step", the A and the B operand of a covariance block ARE the registers a coalesced load
// delivers -- lane (cl, k) loads the R = C / 16 consecutive channels R cl .. R cl + R - 1 of pixel 4 step + k (16 lanes x 4 R bytes = one
// pixel's whole C-vector, contiguous), and component r of that load, seen across the 16 lanes, is the channel set S_r = {R i + r}.
// mfma(X_r, X_s) is then the 16 x 16 block of products between S_r and S_s: a fixed PERMUTATION of the C x C matrix, R (R + 1) / 2 blocks
// (3 / 10 = the number of canonical tile pairs), un-permuted once per workgroup when the partials are written.  No ds_write, no
// ds_read, no barrier in the pixel loop (the kernel above spends its time in the LDS-read -> MFMA chain: 1.6 TB/s at 24 % matrix-core
// busy, profiles/r04a_*), 16 loads per wave and 64-pixel block in flight.
// Arithmetic = the F32 variant's, to the bit for C = 32: the same 64-pixel blocks (block b of a chunk belongs to wave b & 3), the same
// step order inside a block, the same (w0 + w1) + (w2 + w3) combination of the waves and the same chunking -- only WHICH lane holds a
// product differs, and x_a x_b = x_b x_a.  For C = 64 the kernel above cuts its 96-pixel tiles into blocks of 64 + 32 pixels; here
// every block is 64 (raw sums within ~1e-8, the F32 variant's own noise level).
template <int R>
__global__ __launch_bounds__(256, 2) void moments_reg_kernel(MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPR = R * (R + 1) / 2, C = 16 * R, T = R;    // T tiles of 16 channels; NP = T (T + 1) / 2 = NPR canonical pairs
  typedef float vecR __attribute__((ext_vector_type(R)));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = lane & 15, k = lane >> 4;
  const long p0 = (long)blockIdx.x * a.chunk, p1 = min(a.npix, p0 + a.chunk);
  f64x4 acc[NPR];
  double s[R];
#pragma unroll
  for (int j = 0; j < NPR; ++j) acc[j] = f64x4{0., 0., 0., 0.};
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = 0.;
  const bool windowed = a.wwin != a.wfull;
  for (long pb = p0 + 64 * wave; pb < p1; pb += 256) {
    // the block's 16 loads, all issued before the first product (unconditional: addresses clamped, out-of-range pixels zeroed below)
    vecR v[16];
    long q = pb + k;
    long row = 0; int col = 0;
    if (windowed) { row = q / a.wwin; col = (int)(q - row * a.wwin); }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      long qq = q + 4 * st;
      long g;
      if (windowed) {
        long rr = row; int cc = col + 4 * st;
        while (cc >= a.wwin) { cc -= a.wwin; ++rr; }
        if (qq >= p1) { const long ql = p1 - 1; rr = ql / a.wwin; cc = (int)(ql - rr * a.wwin); }
        g = rr * a.wfull + a.x0 + cc;
      } else {
        g = qq < p1 ? qq : p1 - 1;
      }
      v[st] = *reinterpret_cast<const vecR*>(a.x + g * C + R * cl);
    }
    f32x4 f[NPR];
    float t[R];
#pragma unroll
    for (int j = 0; j < NPR; ++j) f[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const bool ok = q + 4 * st < p1;
      float x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { x[r] = ok ? v[st][r] : 0.f; t[r] += x[r]; }
      int j = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c2 = r; c2 < R; ++c2, ++j) f[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r], x[c2], f[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NPR; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] += (double)f[j][e];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] += (double)t[r];
  }
  // ---- the four waves' partial sets -> canonical tile pairs, through TWO LDS sets: waves 0 / 2 write sets A / B, waves 1 / 3 add theirs,
  // the result is A + B = (w0 + w1) + (w2 + w3) -- the kernel above's order -- in 2 x (NP x 256 + C) doubles (12 / 41 KB) instead of 4 sets
  constexpr int nsq = NPR * 256, ns = C;
  double* red = reinterpret_cast<double*>(smem);          // [2][nsq]
  double* reds = red + (size_t)2 * nsq;                   // [2][ns]
  double* mine = red + (size_t)(wave >> 1) * nsq;
  double* mines = reds + (size_t)(wave >> 1) * ns;
  auto pair_of = [](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };   // canonical index of tile pair I <= J
#pragma unroll
  for (int phase = 0; phase < 2; ++phase) {
    if ((wave & 1) == phase) {
      const bool add = phase == 1;
      auto put = [&](int idx, double val) { mine[idx] = add ? mine[idx] + val : val; };   // every canonical entry: exactly one lane of a wave
      int j = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c2 = r; c2 < R; ++c2, ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // D layout of the fp32 instruction: row i = 4 (lane >> 4) + e, column jj = lane & 15
            const int ch_a = R * (4 * k + e) + r, ch_b = R * cl + c2;
            const int I = ch_a >> 4, J = ch_b >> 4, ia = ch_a & 15, ib = ch_b & 15;
            const double val = acc[j][e];
            if (I < J) put(pair_of(I, J) * 256 + ia * 16 + ib, val);
            else if (I > J) { if (r != c2) put(pair_of(J, I) * 256 + ib * 16 + ia, val); }   // (r == r blocks hold the mirrored entry themselves)
            else {
              put(pair_of(I, I) * 256 + ia * 16 + ib, val);
              if (r != c2) put(pair_of(I, I) * 256 + ib * 16 + ia, val);
            }
          }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double v2 = s[r];
        v2 += __shfl_xor(v2, 16);
        v2 += __shfl_xor(v2, 32);
        if (k == 0) mines[R * cl + r] = add ? mines[R * cl + r] + v2 : v2;
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < nsq; e += 256) a.part_sq[(size_t)blockIdx.x * nsq + e] = red[e] + red[nsq + e];
  for (int e = tid; e < ns; e += 256) a.part_sum[(size_t)blockIdx.x * ns + e] = reds[e] + reds[ns + e];
}

// ---- round 5: the WIDE maps (C a multiple of 128: 128 / 256 / 512, --mode original and the deep 16x levels), channel-BLOCKED.
// The LDS kernel above gives a wave up to six tile pairs and splits C (C / 16 + 1) / 2 tile pairs over pair groups that each re-read the
// whole C-vector of every pixel: 22 passes over the map at C = 512 (measured 452 us for 240 x 134 x 512: L2 traffic, 25 % of the fp64
// matrix-core peak).  Here a workgroup owns a PAIR OF 128-CHANNEL BLOCKS (BI <= BJ: 1 / 3 / 10 block pairs) for a pixel chunk and reads only
// those 2 x 128 channels; lane (cl, k) loads channels 8 cl .. 8 cl + 7 of each block for pixel 4 step + k (16 lanes x 32 B = the block,
// contiguous), component r of a block across the 16 lanes is the channel set {8 i + r}, and mfma(A_r, B_s) is one 16 x 16 block of the
// 128 x 128 block pair -- operands straight from the loads' registers as in moments_reg_kernel, 64 component pairs (36 on a diagonal block
// pair: r <= s) dealt to the four waves, 16 (9) accumulators each, every wave over ALL pixels of the chunk.  Passes over the map:
// (C / 128 + 1) / 2 = 2.5 at C = 512.  Every canonical entry is written by exactly one lane of one wave: no cross-wave reduction.
// F32 = false: fp64 products (v_cvt_f64_f32 in registers, v_mfma_f64_16x16x4_f64), every step straight into the fp64 accumulators.
// F32 = true : fp32 products, 64-pixel blocks summed in fp32, block totals in fp64 (the F32 variant's arithmetic; block boundaries at
//              multiples of 64 pixels from the chunk start).  Sums (not second moments) come from the diagonal block pairs.
// the j-th component pair (r, s) of wave `wave`: pairs enumerated r-major (diagonal block pair: s >= r only), dealt round robin to the four
// waves.  constexpr: the operand registers of every MFMA are then compile-time choices (a run-time index would put the operands in scratch)
constexpr int blk_pair(bool diag, int nw, int wave, int j, bool want_s) {
  int idx = 0, n = 0;
  for (int r = 0; r < 8; ++r)
    for (int c2 = diag ? r : 0; c2 < 8; ++c2, ++idx)
      if (idx % nw == wave) { if (n == j) return want_s ? c2 : r; ++n; }
  return -1;
}
constexpr int blk_pairs(bool diag, int nw, int wave) {      // component pairs of wave `wave` of `nw`
  int idx = 0, n = 0;
  for (int r = 0; r < 8; ++r)
    for (int c2 = diag ? r : 0; c2 < 8; ++c2, ++idx)
      if (idx % nw == wave) ++n;
  return n;
}
template <typename F, int... J>
__device__ __forceinline__ void blk_for_each(F&& fn, std::integer_sequence<int, J...>) { (fn(std::integral_constant<int, J>{}), ...); }
// every product of a step, the operand choices as constant expressions (fold over the pair index)
template <bool DIAG, int NW, int WAVE, int... J>
__device__ __forceinline__ void blk_products_f64(const double (&da)[8], const double (&db)[8], f64x4* acc, std::integer_sequence<int, J...>) {
  ((acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(da[blk_pair(DIAG, NW, WAVE, J, false)], db[blk_pair(DIAG, NW, WAVE, J, true)], acc[J], 0, 0, 0)), ...);
}
template <bool DIAG, int NW, int WAVE, int... J>
__device__ __forceinline__ void blk_products_f32(const float (&xa)[8], const float (&xb)[8], f32x4* f, std::integer_sequence<int, J...>) {
  ((f[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[blk_pair(DIAG, NW, WAVE, J, false)], xb[blk_pair(DIAG, NW, WAVE, J, true)], f[J], 0, 0, 0)), ...);
}

template <bool F32, bool DIAG, int NW, int WAVE>
__device__ __forceinline__ void moments_blk_body(const MomArgs& a, int BI, int BJ) {
  constexpr int R = 8, NPW = blk_pairs(DIAG, NW, WAVE);
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, k = lane >> 4;
  const int C = a.C, T = a.T;
  const long p0 = (long)blockIdx.x * a.chunk, p1 = min(a.npix, p0 + a.chunk);
  f64x4 acc[NPW];
  f32x4 f[F32 ? NPW : 1];
  double sd[R];
  float sf[R];
#pragma unroll
  for (int j = 0; j < NPW; ++j) acc[j] = f64x4{0., 0., 0., 0.};
#pragma unroll
  for (int r = 0; r < R; ++r) { sd[r] = 0.; sf[r] = 0.f; }
  if constexpr (F32) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) f[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool windowed = a.wwin != a.wfull;
  const float* xa = a.x + BI * 128 + R * cl;
  const float* xb = a.x + BJ * 128 + R * cl;
  auto gidx = [&](long q) -> long {
    const long qc = q < p1 ? q : p1 - 1;
    if (!windowed) return qc;
    const long row = qc / a.wwin;
    return row * a.wfull + a.x0 + (qc - row * a.wwin);
  };
  constexpr int DEP = 2;                                      // steps of loads in flight ahead of the products
  f32x4 va[DEP][2], vb[DEP][2];
  auto fetch = [&](long q, int slot) {
    const long g = gidx(q) * C;
    va[slot][0] = *reinterpret_cast<const f32x4*>(xa + g);
    va[slot][1] = *reinterpret_cast<const f32x4*>(xa + g + 4);
    if constexpr (!DIAG) {
      vb[slot][0] = *reinterpret_cast<const f32x4*>(xb + g);
      vb[slot][1] = *reinterpret_cast<const f32x4*>(xb + g + 4);
    }
  };
#pragma unroll
  for (int d = 0; d < DEP; ++d) fetch(p0 + k + 4 * d, d);
  long step = 0;
  for (long q = p0 + k; q - k < p1; q += 4 * DEP) {
#pragma unroll
    for (int d = 0; d < DEP; ++d) {
      const long qq = q + 4 * d;
      const bool ok = qq < p1;                               // (steps past the chunk's end multiply zeros)
      float xa_[R], xb_[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        xa_[r] = ok ? va[d][r >> 2][r & 3] : 0.f;
        if constexpr (DIAG) xb_[r] = xa_[r];
        else xb_[r] = ok ? vb[d][r >> 2][r & 3] : 0.f;
      }
      fetch(qq + 4 * DEP, d);                                 // refill the slot just read (addresses clamped; zeroed at use)
      if constexpr (F32) {
        if constexpr (DIAG) {
#pragma unroll
          for (int r = 0; r < R; ++r) sf[r] += xa_[r];
        }
        blk_products_f32<DIAG, NW, WAVE>(xa_, xb_, f, std::make_integer_sequence<int, NPW>{});
        ++step;
        if ((step & 15) == 0) {                               // a 64-pixel block is complete
#pragma unroll
          for (int j = 0; j < NPW; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] += (double)f[j][e];
            f[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if constexpr (DIAG) {
#pragma unroll
            for (int r = 0; r < R; ++r) { sd[r] += (double)sf[r]; sf[r] = 0.f; }
          }
        }
      } else {
        double da[R], db[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { da[r] = (double)xa_[r]; db[r] = DIAG ? da[r] : (double)xb_[r]; }
        if constexpr (DIAG) {
#pragma unroll
          for (int r = 0; r < R; ++r) sd[r] += da[r];
        }
        blk_products_f64<DIAG, NW, WAVE>(da, db, acc, std::make_integer_sequence<int, NPW>{});
      }
    }
  }
  if constexpr (F32) {                                        // the last, partial block
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] += (double)f[j][e];
#pragma unroll
    for (int r = 0; r < R; ++r) sd[r] += (double)sf[r];
  }
  // ---- partials in the canonical format [chunk][tile pair I <= J][16 x 16]; D layout: row = mom_row<F32>(lane >> 4, e), column = lane & 15
  auto pair_of = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };
  double* dst = a.part_sq + (size_t)blockIdx.x * a.NP * 256;
  auto emit = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int r = blk_pair(DIAG, NW, WAVE, j, false), c2 = blk_pair(DIAG, NW, WAVE, j, true);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch_a = BI * 128 + R * mom_row<F32>(k, e) + r, ch_b = BJ * 128 + R * cl + c2;
      const int I = ch_a >> 4, J = ch_b >> 4, ia = ch_a & 15, ib = ch_b & 15;
      const double val = acc[j][e];
      if (I < J) dst[(size_t)pair_of(I, J) * 256 + ia * 16 + ib] = val;
      else if (I > J) { if (r != c2) dst[(size_t)pair_of(J, I) * 256 + ib * 16 + ia] = val; }   // (only on a diagonal block pair; r == r blocks hold the mirror themselves)
      else {
        dst[(size_t)pair_of(I, I) * 256 + ia * 16 + ib] = val;
        if (r != c2) dst[(size_t)pair_of(I, I) * 256 + ib * 16 + ia] = val;
      }
    }
  };
  blk_for_each(emit, std::make_integer_sequence<int, NPW>{});
  if constexpr (DIAG && WAVE == 0) {                          // channel sums of block BI: every wave holds them, wave 0 writes
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double v2 = sd[r];
      v2 += __shfl_xor(v2, 16);
      v2 += __shfl_xor(v2, 32);
      if (k == 0) a.part_sum[(size_t)blockIdx.x * T * 16 + BI * 128 + R * cl + r] = v2;
    }
  }
}

// NW waves per workgroup: 4 for the fp64 form (16 / 9 accumulator sets per wave), 8 for the fp32-block form (8 / 5 sets of fp32 AND fp64 accumulators)
template <bool F32, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void moments_blk_kernel(MomArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nblk = a.C >> 7;
  int BI = 0, rem = (int)blockIdx.y;                          // blockIdx.y enumerates the block pairs BI <= BJ
  while (rem >= nblk - BI) { rem -= nblk - BI; ++BI; }
  const int BJ = BI + rem;
#define WCT_BLK_CASE(W) case W: if (BI == BJ) moments_blk_body<F32, true, NW, W>(a, BI, BJ); else moments_blk_body<F32, false, NW, W>(a, BI, BJ); break;
  switch (wave) {
    WCT_BLK_CASE(0) WCT_BLK_CASE(1) WCT_BLK_CASE(2)
    default:
      if constexpr (NW == 4) { if (BI == BJ) moments_blk_body<F32, true, NW, 3>(a, BI, BJ); else moments_blk_body<F32, false, NW, 3>(a, BI, BJ); }
      else {
        switch (wave) {
          WCT_BLK_CASE(3) WCT_BLK_CASE(4) WCT_BLK_CASE(5) WCT_BLK_CASE(6)
          default: if (BI == BJ) moments_blk_body<F32, true, NW, 7>(a, BI, BJ); else moments_blk_body<F32, false, NW, 7>(a, BI, BJ); break;
        }
      }
      break;
  }
#undef WCT_BLK_CASE
}

// ---- level 1 of the 16x cascade: moments of relu1_1 = relu(conv11(image)) WITHOUT the feature map in HBM (level1.hip).
// A persistent workgroup walks 32 x 8 image tiles: conv11 (f16x3, l1_conv_group) writes the tile's 256 x C features
// into the LDS tile the pair loop above consumes (pixel split: every wave owns all three tile pairs for a quarter of
// the pixels); pixels outside the window [x0, x1) x [0, H) are written as zeros.  Partials: one set per workgroup.
struct L1MomArgs {
  const float* img;
  L1Conv c;
  int H, W, x0, x1, tiles_x, tiles_y;
  int Cs;                // LDS row stride (floats): 44 (launch_l1_moments)
  double* part_sq;       // [grid][3][256]
  double* part_sum;      // [grid][32]
  unsigned* sat;         // sticky saturation counter of the context (image values beyond the f16 range), may be null
};

template <bool F32>
__global__ __launch_bounds__(256, 2) void l1_moments_kernel(L1MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = 3, T = 2, MP = 256;
  u32x2* imgH = reinterpret_cast<u32x2*>(smem);
  u32x2* imgL = imgH + IMG_E;
  float* feat = reinterpret_cast<float*>(imgL + IMG_E);   // [256][Cs]; reused for the cross-wave reduction at the end
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;     // conv roles; the pair loop uses the same split (c = li, pk = kq)
  const int ntiles = a.tiles_x * a.tiles_y;
  const unsigned txm = tile_div_magic(a.tiles_x);   // once per workgroup; tile_rc() then stays on the scalar unit
  if (tid < 4) { imgH[NPI2 + tid] = u32x2{0u, 0u}; imgL[NPI2 + tid] = u32x2{0u, 0u}; }
  for (int e = tid; e < MP * a.Cs; e += 256) feat[e] = 0.f;   // columns >= 32 are never written
  L1Weights w;
  l1_load_weights(a.c, li, kq, IMG_E, w);
  // 24 channels in TWO 16x16 products instead of the three tile pairs (0,0), (0,1), (1,1) of the padded 32:
  //   product 0: rows ch 0..15 x cols ch 0..15                                   = tile (0,0)
  //   product 1: rows ch 8..23 x cols ch (16..23, 0..7): its blocks are rows 8..15 of tile (0,1), the whole 8x8 of tile (1,1),
  //              rows 0..7 of tile (0,1) TRANSPOSED (x_a x_b = x_b x_a: the same products in the same pixel order, bit for bit)
  //              and a duplicate of part of tile (0,0) that is dropped
  // -- a third less work on the fp64 matrix cores, which bound this kernel.  The partials leave in the standard three-tile format.
  const int oa1 = 8 + li, ob1 = li < 8 ? 16 + li : li - 8;
  // LDS layout of the feature tile (round 5): row stride Cs = 48 floats and column c of pixel p stored at c ^ (((p >> 1) & 3) << 2).
  //   reads (ds_read_b32, groups of 32 lanes = rows pk in {0, 1} or {2, 3} x 16 columns): 48 p mod 32 = 16 (p & 1) puts the two rows of a
  //     group on opposite halves of the 32 banks, and the XOR (uniform within a group: (p >> 1) & 3 is the same for rows 4k + {0, 1} and for
  //     4k + {2, 3}) permutes columns inside their aligned 16-blocks -- conflict-free.  With the former stride 44 every read was a 2-way
  //     conflict (rows 44 apart: banks 0-15 against 12-27): profiles/r04b_sq_counters_fused_ends.txt, 26 % of this kernel's LDS cycles.
  //   writes (ds_write_b128, groups of 8 neighbouring pixels of a row): 16-byte slot index mod 8 = 4 ((px & 1) ^ ct) + (kq ^ ((px >> 1) & 3)):
  //     eight distinct slots of the 128-byte bank row -- conflict-free (a plain stride of 48 would be 4-way: why rounds 3-4 kept 44).
  // Row p = 4 (st + u) + pk of the pair loop (st a multiple of 4): (p >> 1) & 3 = pk >> 1 for even u, 2 + (pk >> 1) for odd u.
#ifdef WCT_L1MOM_CS44
  constexpr bool L1M_SWZ = false;
#else
  constexpr bool L1M_SWZ = true;
#endif
  const int sw0 = L1M_SWZ ? (kq >> 1) << 2 : 0, sw1 = L1M_SWZ ? (2 + (kq >> 1)) << 2 : 0;
  const int oa[2][2] = {{li ^ sw0, li ^ sw1}, {oa1 ^ sw0, oa1 ^ sw1}}, ob[2][2] = {{li ^ sw0, li ^ sw1}, {ob1 ^ sw0, ob1 ^ sw1}};
  f64x4 acc[2];
  double s[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { acc[j] = f64x4{0., 0., 0., 0.}; s[j] = 0.; }
  int soff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int e = tid + 256 * k;
    e = e < NPI2 ? e : NPI2 - 1;
    soff[k] = (e / I2W) * a.W + e % I2W;
  }
  const int xhi = a.x1 < a.W ? a.x1 : a.W;
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
    __syncthreads();   // image window in LDS; the previous tile's features consumed
    const int vn = v + gridDim.x;
    if (vn < ntiles) head_fetch(a.img, a.H, a.W, a.tiles_x, txm, pxr, soff, xcd_swizzle(vn, ntiles), tid);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int py = wave * 2 + r, px = h * 16 + li;
        const int base = (py + 1) * I2W + px + 1;
        const int gy = ty0 + py, gx = tx0 + px;
        const bool in = gy < a.H && gx >= a.x0 && gx < xhi;
        f32x4 xs[2];
        l1_conv_pair(imgH, base, w, xs[0], xs[1]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          *reinterpret_cast<f32x4*>(feat + (py * 32 + px) * a.Cs + ((ct * 16 + 4 * kq) ^ (L1M_SWZ ? ((px >> 1) & 3) << 2 : 0))) = in ? xs[ct] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
    const int st0 = wave * (MP / 16);
    tile_steps2<F32>(feat, a.Cs, st0, st0 + MP / 16, kq, oa, ob, acc[0], acc[1], s[0], s[1]);
    if (vn < ntiles) { head_pin(pxr); head_commit(pxr, imgH, imgL, tid, sat); }
  }
  sat.commit(a.sat);
  // add the four waves' partial sets through LDS, fixed order (as moments_kernel's pixel-split tail)
  __syncthreads();
  double* red = reinterpret_cast<double*>(feat);   // [4][NP][256] + [4][T*16]
  double* reds = red + (size_t)4 * NP * 256;
  {
    double* mine = red + (size_t)wave * NP * 256;
    // tiles (0,1) and (1,1): zero (padding channels 24..31), then product 1's blocks scattered to their places.  One wave's LDS
    // operations execute in order, and the regions are per wave.
    for (int e = lane; e < 2 * 256; e += 64) mine[256 + e] = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mom_row<F32>(kq, r);
      mine[row * 16 + li] = acc[0][r];                                   // tile (0,0)
      const int ra = 8 + row, cb = ob1;                                   // channels of product 1's entry
      if (ra < 16 && cb >= 16) mine[256 + ra * 16 + (cb - 16)] = acc[1][r];             // tile (0,1) rows 8..15
      else if (ra >= 16 && cb >= 16) mine[512 + (ra - 16) * 16 + (cb - 16)] = acc[1][r];   // tile (1,1)
      else if (ra >= 16) mine[256 + cb * 16 + (ra - 16)] = acc[1][r];                   // tile (0,1) rows 0..7, transposed
    }
    // channel sums: product 0's rows are channels 0..15, product 1's rows 8..15 are channels 16..23
    double t0 = s[0], t1 = s[1];
    t0 += __shfl_xor(t0, 16); t0 += __shfl_xor(t0, 32);
    t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
    if (kq == 0) { reds[wave * T * 16 + li] = t0; reds[wave * T * 16 + 16 + li] = 0.; }   // channels 24..31: padding
    if (kq == 0 && li >= 8) reds[wave * T * 16 + 8 + li] = t1;
  }
  __syncthreads();
  const int nsq = NP * 256, ns = T * 16;
  for (int e = tid; e < nsq; e += 256)
    a.part_sq[(size_t)blockIdx.x * nsq + e] = (red[e] + red[nsq + e]) + (red[2 * nsq + e] + red[3 * nsq + e]);
  for (int e = tid; e < ns; e += 256)
    a.part_sum[(size_t)blockIdx.x * ns + e] = (reds[e] + reds[ns + e]) + (reds[2 * ns + e] + reds[3 * ns + e]);
}

// (Round 6, measured and not kept: the big maps' products as FOUR exact split-f16 products per pair on v_mfma_f32_16x16x32_f16 with 32 pixels in the K
//  dimension -- conv11 transposed so that a lane holds four neighbouring pixels of one channel, [channel][pixel] f16 planes, the channel sums as products
//  against a matrix of ones: 12 MFMAs of 16 cycles per 32 pixels instead of 16 of 32 cycles, 6 ds_read_b128 instead of 32 ds_read_b32.  Same-box A/B at
//  4K + 2K style: 0.317-0.320 ms per step against 0.305-0.308 for the fp32 form above, and 1.3e-6 from the fp64 form where the fp32 form sits at ~1e-8
//  (x' = hi + lo is not x).  The kernel is bound by conv11's epilogue and the LDS stores, not by the pair loop's matrix time.  profiles/r06_l1_moments_f16_ab.txt)

// partial -> final, fixed summation order (bitwise reproducible): a 256-thread block owns 16 consecutive output
// elements x 16 slices of the chunk index; slices are combined through LDS in slice order.
__global__ __launch_bounds__(256) void moments_reduce_kernel(MomArgs a, double* sum, double* sumsq) {
  __shared__ double red[16][17];
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + el;
  const long nsq = (long)a.NP * 256, nsum = (long)a.T * 16;
  double v = 0.;
  if (e < nsq) {
    for (int pc = sl; pc < a.NPC; pc += 16) v += a.part_sq[(size_t)pc * nsq + e];
  } else if (e < nsq + nsum) {
    for (int pc = sl; pc < a.NPC; pc += 16) v += a.part_sum[(size_t)pc * nsum + (e - nsq)];
  }
  red[sl][el] = v;
  __syncthreads();
  if (sl != 0) return;
  v = 0.;
#pragma unroll
  for (int k = 0; k < 16; ++k) v += red[k][el];
  if (e < nsq) {
    const int pair = (int)(e >> 8), r = (int)((e >> 4) & 15), cc = (int)(e & 15);
    int I = 0, rem = pair;
    while (rem >= a.T - I) { rem -= a.T - I; ++I; }
    const int J = I + rem;
    const int ra = I * 16 + r, cb = J * 16 + cc;
    if (ra < a.C && cb < a.C) {
      sumsq[(size_t)ra * a.C + cb] = v;
      if (I != J) sumsq[(size_t)cb * a.C + ra] = v;  // a diagonal tile holds both of its halves itself
    }
  } else if (e < nsq + nsum) {
    const int ch = (int)(e - nsq);
    if (ch < a.C) sum[ch] = v;
  }
}

MomArgs plan(int C, long npix) {
  MomArgs a{};
  a.C = C; a.T = (C + 15) / 16; a.NP = a.T * (a.T + 1) / 2;
  a.pixsplit = a.NP <= 6;
  // pairs per wave: 3 or 6 -- never more, so that two tiles of prefetch fit the register file beside the accumulators;
  // larger C splits the pairs over NPG workgroup groups (each re-reads the features, mostly from L2)
  a.pw = a.pixsplit ? (a.NP <= 3 ? 3 : 6) : (a.NP <= 12 ? 3 : 6);
  a.NPG = a.pixsplit ? 1 : (a.NP + 4 * a.pw - 1) / (4 * a.pw);
  a.Cs = (a.T & 1) ? a.T * 16 : a.T * 16 + 16;  // == 16 (mod 32) dwords
  // MP * C / 4 <= maxld * 256 float4 slots; multiple of 16.  Six pairs per wave (C >= 64) leave room for 6 load slots only: with 8
  // the two-tile prefetch + 48 accumulator registers spilled 18 VGPRs into scratch inside the pixel loop
  static const int maxld6 = [] { const char* e = wct_debug_env("WCT_MOM_MAXLD6"); return e ? atoi(e) : 6; }();
  const int maxld = a.pw == 6 ? maxld6 : MAXLD;
  a.MP = std::max(16, std::min(256, (maxld * 256 * 4 / C) / 16 * 16));
  const int MP = a.MP;
  a.npix = npix;
  static long npc_target = [] { const char* e = wct_debug_env("WCT_MOM_NPC"); return e ? atol(e) : 512L; }();
  long npc = npc_target / a.NPG;           // ~8 workgroups per CU in total
  if (a.NP > 16 && npc > 512) npc = 512;   // bound the partial buffer (NPC * NP * 2 KB)
  const long maxc = (npix + MP - 1) / MP;
  if (npc > maxc) npc = maxc;
  if (npc < 1) npc = 1;
  long chunk = (npix + npc - 1) / npc;
  chunk = (chunk + MP - 1) / MP * MP;
  a.chunk = chunk;
  a.NPC = (int)((npix + chunk - 1) / chunk);
  if (a.NPC < 1) a.NPC = 1;
  return a;
}

// chunking of moments_blk_kernel: ~1024 workgroups over all block pairs, chunks multiples of 64 pixels
void blk_plan(int C, long npix, long& chunk, int& npc_out) {
  const int nb = C >> 7, nbp = nb * (nb + 1) / 2;
  long npc_target = 512;      // workgroups over all block pairs (two per CU); every chunk costs NP x 2 KB of partials: not more
  if (const char* e = wct_debug_env("WCT_MOM_BLK_WGS")) npc_target = atol(e);
  const long npc = std::max(1L, std::min(npc_target / nbp, (npix + 255) / 256));
  chunk = ((npix + npc - 1) / npc + 63) / 64 * 64;
  npc_out = (int)((npix + chunk - 1) / chunk);
}
}  // namespace

size_t moments_workspace_bytes(int C, long npix) {
  MomArgs a = plan(C, npix);
  int npc = a.NPC;
  if (C >= 256 && (C & 127) == 0) { long ch; int n2; blk_plan(C, npix, ch, n2); npc = std::max(npc, n2); }
  return ((size_t)npc * a.NP * 256 + (size_t)npc * a.T * 16) * sizeof(double);
}

hipError_t launch_moments(const float* feat, int C, int h, int wfull, int x0, int x1, double* sum, double* sumsq,
                          void* ws, size_t ws_bytes, hipStream_t s, bool f32_products) {
  if (x0 < 0 || x1 > wfull || x1 <= x0 || h < 1) return hipErrorInvalidValue;
  const long npix = (long)h * (x1 - x0);
  if (C < 4 || npix < 1 || (C & 3)) return hipErrorInvalidValue;
  MomArgs a = plan(C, npix);
  if (ws_bytes < moments_workspace_bytes(C, npix)) return hipErrorOutOfMemory;
  a.x = feat;
  a.wfull = wfull; a.x0 = x0; a.wwin = x1 - x0;
  a.part_sq = reinterpret_cast<double*>(ws);
  a.part_sum = a.part_sq + (size_t)a.NPC * a.NP * 256;
  // big shallow maps with fp32 block products: operands straight from the loads' registers (moments_reg_kernel); WCT_MOM_REG=0: the LDS kernel
  static const int reg_env = [] { const char* e = wct_debug_env("WCT_MOM_REG"); return e ? atoi(e) : 1; }();
  if (reg_env && f32_products && (C == 32 || C == 64)) {
    // same chunking as the LDS kernel (chunks are multiples of its tile MP; here they must be multiples of 256: four waves x 64 pixels)
    a.chunk = (a.chunk + 255) / 256 * 256;
    a.NPC = (int)((npix + a.chunk - 1) / a.chunk);
    if (ws_bytes < ((size_t)a.NPC * a.NP * 256 + (size_t)a.NPC * a.T * 16) * sizeof(double)) return hipErrorOutOfMemory;
    a.part_sum = a.part_sq + (size_t)a.NPC * a.NP * 256;
    const size_t ldsr = ((size_t)2 * a.NP * 256 + 2 * C) * sizeof(double);      // 12.5 / 42 KB
    auto kr = C == 32 ? moments_reg_kernel<2> : moments_reg_kernel<4>;
    if (ldsr > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kr, dim3((unsigned)a.NPC), dim3(256), ldsr, s, a);
    const long ne = (long)a.NP * 256 + a.T * 16;
    hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, a, sum, sumsq);
    return hipGetLastError();
  }
  // wide maps: channel-blocked, operands from registers (moments_blk_kernel); WCT_MOM_BLK=0: the LDS kernel
  static const int blk_env = [] { const char* e = wct_debug_env("WCT_MOM_BLK"); return e ? atoi(e) : 1; }();
  // measured (tools/experiments/mom_reg_ab.py blk), fp64 form: C = 512 at 240 x 134: 369 -> 237 us; C = 256: 1.05x; C = 128 and maps of a few hundred
  // pixels: SLOWER (every chunk writes NP x 2 KB of partials) -> C >= 512 and >= 4096 pixels.  fp32-block form (eight waves): C >= 256.
  const bool blk64 = !f32_products && C >= 512 && (C & 127) == 0 && npix >= 4096;
  // fp32-block form (eight waves): measured C = 128: 0.64-0.8x, C = 256: 1.08-1.15x, C = 512: 1.63x -- but its raw sums sit 1.4-4x further from fp64 than the
  // LDS kernel's (7e-9 against 2e-9 at C = 256) and config 3's level-isolated check at C = 256 moved from 4.8e-6 to 6.9e-5: experiments only (WCT_MOM_BLK=2).
  // (Why: at C >= 256 the LDS kernel's tile is 16 pixels, so ITS fp32 blocks are 16 pixels, not 64: a block's error grows as n^1.5, their number as 1/n.)
  const bool blk32 = f32_products && C >= 256 && (C & 127) == 0 && blk_env == 2;
  if (blk_env && (blk64 || blk32)) {
    const int nb = C >> 7, nbp = nb * (nb + 1) / 2;
    blk_plan(C, npix, a.chunk, a.NPC);
    if (ws_bytes < ((size_t)a.NPC * a.NP * 256 + (size_t)a.NPC * a.T * 16) * sizeof(double)) return hipErrorOutOfMemory;
    a.part_sum = a.part_sq + (size_t)a.NPC * a.NP * 256;
    if (f32_products) hipLaunchKernelGGL((moments_blk_kernel<true, 8>), dim3((unsigned)a.NPC, (unsigned)nbp), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((moments_blk_kernel<false, 4>), dim3((unsigned)a.NPC, (unsigned)nbp), dim3(256), 0, s, a);
    const long ne = (long)a.NP * 256 + a.T * 16;
    hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, a, sum, sumsq);
    return hipGetLastError();
  }
  size_t lds = (size_t)a.MP * a.Cs * sizeof(float);
  if (a.pixsplit) lds = std::max(lds, ((size_t)4 * a.NP * 256 + 4 * a.T * 16) * sizeof(double));
  const int nld = (a.MP * (C / 4) + 255) / 256;
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.NPC, (unsigned)a.NPG), dim3(256), lds, s, a);
    return hipSuccess;
  };
  hipError_t le;
#define WCT_MOM_CASE(N) \
  case N: le = f32_products ? (a.pw == 3 ? go(moments_kernel<N, 3, 2, true>) : go(moments_kernel<N, 6, 2, true>)) \
                            : (a.pw == 3 ? go(moments_kernel<N, 3, 2, false>) : go(moments_kernel<N, 6, 2, false>)); break;
  switch (nld) {
    WCT_MOM_CASE(1) WCT_MOM_CASE(2) WCT_MOM_CASE(3) WCT_MOM_CASE(4) WCT_MOM_CASE(5) WCT_MOM_CASE(6) WCT_MOM_CASE(7)
    default: le = f32_products ? (a.pw == 3 ? go(moments_kernel<8, 3, 2, true>) : go(moments_kernel<8, 6, 2, true>))
                               : (a.pw == 3 ? go(moments_kernel<8, 3, 2, false>) : go(moments_kernel<8, 6, 2, false>)); break;
  }
#undef WCT_MOM_CASE
  if (le != hipSuccess) return le;
  const long ne = (long)a.NP * 256 + a.T * 16;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, a, sum, sumsq);
  return hipGetLastError();
}

size_t l1_moments_workspace_bytes() { return (size_t)2 * num_cus() * (3 * 256 + 32) * sizeof(double); }

hipError_t launch_l1_moments(const ConvDesc& e, const float* img, int H, int W, int x0, int x1, double* sum, double* sumsq,
                             void* ws, size_t ws_bytes, hipStream_t s, bool f32_products) {
  if (!l1_capable(e) || H < 2 || W < 2 || x0 < 0 || x1 > W || x1 <= x0) return hipErrorInvalidValue;
  if (ws_bytes < l1_moments_workspace_bytes()) return hipErrorOutOfMemory;
  L1MomArgs a;
  a.img = img;
  a.c.w = reinterpret_cast<const u32x4*>(e.l1w16); a.c.b = e.l1bias; a.c.inv = e.l1inv;
  a.H = H; a.W = W; a.x0 = x0; a.x1 = x1; a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + 7) / 8;
  // LDS row stride of the feature tile (floats).  44: the 16-byte conv11 stores of eight neighbouring pixels fall on distinct banks (48
  // made them 4-way conflicts: 64 LDS cycles per store pair instead of 16) at the price of 2-way conflicts on the pair loop's 4-byte
  // reads (model: 320 vs 416 LDS cycles per wave and tile; measured -6 %)
#ifdef WCT_L1MOM_CS44
  a.Cs = 44;
#else
  a.Cs = 48;      // with the kernel's column swizzle: reads and writes free of bank conflicts (l1_moments_kernel)
#endif
  a.sat = e.sat;
  const int ntiles = a.tiles_x * a.tiles_y, grid = ntiles < 2 * num_cus() ? ntiles : 2 * num_cus();
  a.part_sq = reinterpret_cast<double*>(ws);
  a.part_sum = a.part_sq + (size_t)grid * 3 * 256;
  const size_t lds = (size_t)2 * IMG_E * 8 + (size_t)256 * a.Cs * sizeof(float);   // 52 KB (the reduction needs 25 KB of it)
  auto kern = f32_products ? l1_moments_kernel<true> : l1_moments_kernel<false>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  MomArgs m{};
  m.C = e.cout; m.T = 2; m.NP = 3; m.NPC = grid; m.part_sq = a.part_sq; m.part_sum = a.part_sum;
  const long ne = (long)m.NP * 256 + m.T * 16;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, m, sum, sumsq);
  return hipGetLastError();
}
