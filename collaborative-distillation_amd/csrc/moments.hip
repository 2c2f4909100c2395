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
// wave w owns up to 9 tile pairs (interleaved over the group's 36) for the WHOLE chunk, so accumulators stay
// in registers and there is no cross-wave reduction.  LDS row stride Cs == 16 (mod 32) dwords makes the
// operand reads (lane = channel l&15 of pixel l>>4) bank-conflict free.  Partial tiles go to a workspace and a
// second kernel adds them in a fixed order -> bitwise reproducible, no atomics.
#include "wct_common.h"

namespace {

constexpr int MP = 64;   // pixels per LDS tile
constexpr int PPW = 9;   // tile pairs per wave
constexpr int PPG = 4 * PPW;

struct MomArgs {
  const float* x;
  int C, T, NP, NPG, NPC, Cs;
  long npix, chunk;      // pixels per chunk (multiple of MP)
  int wfull, x0, wwin;   // window: pixel p -> (row p / wwin, col x0 + p % wwin) of a map of width wfull
  double* part_sq;       // [NPC][NP][256]
  double* part_sum;      // [NPC][T*16]
};

__global__ __launch_bounds__(256) void moments_kernel(MomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);  // [MP][Cs]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, pk = lane >> 4;
  const int pc = blockIdx.x, pg = blockIdx.y;
  // pairs owned by this wave: idx = pg*PPG + wave + 4*j
  int offA[PPW], offB[PPW], pidx[PPW];
  bool diag[PPW];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int idx = pg * PPG + wave + 4 * j;
    offA[j] = offB[j] = 0; pidx[j] = 0; diag[j] = false;
    if (idx < a.NP) {
      int I = 0, rem = idx;
      while (rem >= a.T - I) { rem -= a.T - I; ++I; }
      offA[j] = I * 16 + c; offB[j] = (I + rem) * 16 + c; pidx[j] = idx; diag[j] = rem == 0;
      cnt = j + 1;
    }
  }
  f64x4 acc[PPW];
  double s[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) { acc[j] = f64x4{0., 0., 0., 0.}; s[j] = 0.; }
  // zero the whole tile once: columns >= C (C = 24 -> T*16 = 32) and the row padding are never loaded
  for (int e = tid; e < MP * a.Cs; e += 256) lds[e] = 0.f;

  const long p0 = (long)pc * a.chunk, p1 = min(a.npix, p0 + a.chunk);
  const int c4n = a.C >> 2;
  for (long pt = p0; pt < p1; pt += MP) {
    __syncthreads();
    for (int e = tid; e < MP * c4n; e += 256) {
      const int pix = e / c4n, c4 = e - pix * c4n;
      const long pp = pt + pix;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pp < p1) {
        long g = pp;
        if (a.wwin != a.wfull) { const long r = pp / a.wwin; g = r * a.wfull + a.x0 + (pp - r * a.wwin); }
        v = *reinterpret_cast<const f32x4*>(a.x + g * a.C + c4 * 4);
      }
      *reinterpret_cast<f32x4*>(lds + pix * a.Cs + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int st = 0; st < MP / 4; ++st) {
      const float* row = lds + (st * 4 + pk) * a.Cs;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        if (j < cnt) {
          const double av = (double)row[offA[j]], bv = (double)row[offB[j]];
          s[j] += av;
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // D layout (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    if (j < cnt && pg * PPG + wave + 4 * j < a.NP) {
      double* dst = a.part_sq + ((size_t)pc * a.NP + pidx[j]) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(pk + 4 * r) * 16 + c] = acc[j][r];
      if (diag[j]) {  // the diagonal tile's owner also owns sum over that tile's channels
        double v = s[j];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (pk == 0) a.part_sum[(size_t)pc * a.T * 16 + offA[j]] = v;
      }
    }
  }
}

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
  a.NPG = (a.NP + PPG - 1) / PPG;
  a.Cs = (a.T & 1) ? a.T * 16 : a.T * 16 + 16;  // == 16 (mod 32) dwords
  a.npix = npix;
  long npc = 2048 / a.NPG;                 // ~8 workgroups per CU in total
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

}  // namespace

size_t moments_workspace_bytes(int C, long npix) {
  MomArgs a = plan(C, npix);
  return ((size_t)a.NPC * a.NP * 256 + (size_t)a.NPC * a.T * 16) * sizeof(double);
}

hipError_t launch_moments(const float* feat, int C, int h, int wfull, int x0, int x1, double* sum, double* sumsq,
                          void* ws, size_t ws_bytes, hipStream_t s) {
  if (x0 < 0 || x1 > wfull || x1 <= x0 || h < 1) return hipErrorInvalidValue;
  const long npix = (long)h * (x1 - x0);
  if (C < 4 || npix < 1 || (C & 3)) return hipErrorInvalidValue;
  MomArgs a = plan(C, npix);
  if (ws_bytes < moments_workspace_bytes(C, npix)) return hipErrorOutOfMemory;
  a.x = feat;
  a.wfull = wfull; a.x0 = x0; a.wwin = x1 - x0;
  a.part_sq = reinterpret_cast<double*>(ws);
  a.part_sum = a.part_sq + (size_t)a.NPC * a.NP * 256;
  const size_t lds = (size_t)MP * a.Cs * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(moments_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(moments_kernel, dim3((unsigned)a.NPC, (unsigned)a.NPG), dim3(256), lds, s, a);
  const long ne = (long)a.NP * 256 + a.T * 16;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 15) / 16)), dim3(256), 0, s, a, sum, sumsq);
  return hipGetLastError();
}
