// Raw fp64 moments of an NHWC fp32 feature map:  sum[c] = SUM_p x[p][c],  sumsq[a][b] = SUM_p x[p][a]*x[p][b].
//
// This is the data-dependent half of the reference's whitening/colouring transform
// (PytorchWCT/util_wct.py:68-70 content, :94-96 style): torch.mean(cF,1), cF - mean, mm(cF, cF.t())/(hw-1),
// all in fp64 on the host.  Here the features never leave HBM: one pass reads them once, converts to fp64
// in registers and accumulates x x^T on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), so the products and
// sums are fp64 exactly like the reference's.  mean / covariance follow in solve.hip as
//   mu = sum/n,  cov = (sumsq - n mu mu^T)/(n-1)
// (raw sums, not centred ones, because they are what a content-sharded run all-reduces across GPUs).
//
// Decomposition: the C x C output is cut into 16x16 tiles, upper triangle only.  A work item is
// (pixel chunk, tile row I, up to 8 tile columns J >= I); one wave per item, no LDS, no inter-wave sync.
// Per 4 pixels the wave loads 1 + cnt dwords per lane (64-B channel runs) and issues cnt MFMAs.  Partial
// tiles go to a workspace and a second kernel adds them in a fixed order -> bitwise reproducible.
#include "wct_common.h"

namespace {

constexpr int JW = 8;  // tile columns per work item

__host__ __device__ inline int items_per_chunk(int T) {
  int n = 0;
  for (int I = 0; I < T; ++I) n += (T - I + JW - 1) / JW;
  return n;
}

struct MomArgs {
  const float* x;
  int C, T, NP, NITEMS, NPC;
  long npix, chunk;  // pixels per chunk (multiple of 4)
  int wfull, x0, wwin;  // window: pixel p -> (row p / wwin, col x0 + p % wwin) of a map of width wfull
  double* part_sq;   // [NPC][NP][256]
  double* part_sum;  // [NPC][T*16]
};

__device__ __forceinline__ int pair_index(int I, int J, int T) { return I * T - (I * (I - 1)) / 2 + (J - I); }

__global__ __launch_bounds__(256) void moments_kernel(MomArgs a) {
  const int lane = threadIdx.x & 63;
  const long wg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= (long)a.NPC * a.NITEMS) return;
  const int item = (int)(wg % a.NITEMS);
  const int pc = (int)(wg / a.NITEMS);
  // decode item -> (I, J0, cnt)
  int I = 0, J0 = 0, cnt = 0;
  {
    int it = item;
    for (I = 0; I < a.T; ++I) {
      const int g = (a.T - I + JW - 1) / JW;
      if (it < g) { J0 = I + it * JW; cnt = min(JW, a.T - J0); break; }
      it -= g;
    }
  }
  const int c = lane & 15, pk = lane >> 4;
  const long p0 = (long)pc * a.chunk, p1 = min(a.npix, p0 + a.chunk);
  f64x4 acc[JW];
#pragma unroll
  for (int j = 0; j < JW; ++j) acc[j] = f64x4{0., 0., 0., 0.};
  double s = 0.;
  const int ca = I * 16 + c;
  const bool va = ca < a.C;
  for (long p = p0; p < p1; p += 4) {
    const long pp = p + pk;
    const bool vp = pp < p1;
    long pix = pp;
    if (a.wwin != a.wfull) { const long r = pp / a.wwin; pix = r * a.wfull + a.x0 + (pp - r * a.wwin); }
    const float* row = a.x + pix * a.C;
    const double av = (vp && va) ? (double)row[ca] : 0.;
    s += av;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      if (j < cnt) {
        const int cb = (J0 + j) * 16 + c;
        const double bv = (vp && cb < a.C) ? (double)row[cb] : 0.;
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[j], 0, 0, 0);
      }
    }
  }
  // D layout (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int j = 0; j < JW; ++j) {
    if (j < cnt) {
      double* dst = a.part_sq + ((size_t)pc * a.NP + pair_index(I, J0 + j, a.T)) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(pk + 4 * r) * 16 + c] = acc[j][r];
    }
  }
  if (J0 == I) {  // this item owns the diagonal tile -> it also owns sum over tile I's channels
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (pk == 0) a.part_sum[(size_t)pc * a.T * 16 + ca] = s;
  }
}

__global__ void moments_reduce_kernel(MomArgs a, double* sum, double* sumsq) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nsq = (long)a.NP * 256;
  if (e < nsq) {
    const int pair = (int)(e >> 8), r = (int)((e >> 4) & 15), cc = (int)(e & 15);
    double v = 0.;
    for (int pc = 0; pc < a.NPC; ++pc) v += a.part_sq[(size_t)pc * nsq + e];
    // pair -> (I, J)
    int I = 0, rem = pair;
    while (rem >= a.T - I) { rem -= a.T - I; ++I; }
    const int J = I + rem;
    const int ra = I * 16 + r, cb = J * 16 + cc;
    if (ra < a.C && cb < a.C) {
      sumsq[(size_t)ra * a.C + cb] = v;
      if (I != J) sumsq[(size_t)cb * a.C + ra] = v;
      else if (ra > cb) { /* lower half of a diagonal tile: written by its mirror element */ }
    }
  } else if (e < nsq + a.T * 16) {
    const int ch = (int)(e - nsq);
    if (ch < a.C) {
      double v = 0.;
      for (int pc = 0; pc < a.NPC; ++pc) v += a.part_sum[(size_t)pc * a.T * 16 + ch];
      sum[ch] = v;
    }
  }
}

MomArgs plan(int C, long npix) {
  MomArgs a{};
  a.C = C; a.T = (C + 15) / 16; a.NP = a.T * (a.T + 1) / 2; a.NITEMS = items_per_chunk(a.T);
  a.npix = npix;
  long npc = 2048 / a.NITEMS;
  if (npc < 1) npc = 1;
  const long maxc = (npix + 63) / 64;  // at least 64 pixels per chunk
  if (npc > maxc) npc = maxc;
  if (npc < 1) npc = 1;
  long chunk = (npix + npc - 1) / npc;
  chunk = (chunk + 3) / 4 * 4;
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
  if (C < 1 || npix < 1 || (C & 3)) return hipErrorInvalidValue;
  MomArgs a = plan(C, npix);
  if (ws_bytes < moments_workspace_bytes(C, npix)) return hipErrorOutOfMemory;
  a.x = feat;
  a.wfull = wfull; a.x0 = x0; a.wwin = x1 - x0;
  a.part_sq = reinterpret_cast<double*>(ws);
  a.part_sum = a.part_sq + (size_t)a.NPC * a.NP * 256;
  const long waves = (long)a.NPC * a.NITEMS;
  hipLaunchKernelGGL(moments_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  const long ne = (long)a.NP * 256 + a.T * 16;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, a, sum, sumsq);
  return hipGetLastError();
}
