// From raw moments to the affine map of the whitening/colouring transform, entirely on the device.
//
// Restates the C x C part of PytorchWCT/util_wct.py:62-131 + :219 (which the reference runs on the host
// in fp64 with LAPACK):
//   mu = sum/n ; cov = (sumsq - n mu mu^T)/(n-1)                       (:68-70, :94-96, unbiased)
//   cov = V diag(lambda) V^T                                            (:74, :100  torch.svd)
//   Wc = Vc diag(lambda_c^-1/2) Vc^T ; Ss = Vs diag(lambda_s^1/2) Vs^T   (:117-119, :124-125)
//   T = Ss Wc ; csF = alpha (T (cF - mu_c) + mu_s) + (1-alpha) cF       (:120, :125-126, :219)
//        = M cF + b,  M = alpha T + (1-alpha) I,  b = alpha (mu_s - T mu_c)
// The reference multiplies the C x hw feature map by Wc and then by Ss (two skinny fp64 GEMMs, its
// dominant cost); here T is formed once (C^3) and applied inside the decoder's first convolution.
//
// Two steps so that the style side can run ahead on its own stream (it does not depend on the content):
//   launch_eig       moments of ONE feature map -> EigResult {G = V diag(lambda), lambda, mu, floor}
//   launch_assemble  two EigResults -> M, b
//
// Eigen-decomposition: one-sided (Hestenes) Jacobi in fp64 on G = cov.  Rotating column pairs of G from
// the right until all columns are mutually orthogonal leaves G = V diag(lambda): the column norms are the
// eigenvalues and the normalised columns the eigenvectors, so no separate V is accumulated and a C<=128
// problem (128 x 130 doubles) lives entirely in one CU's 160 KB LDS.  A round of the round-robin
// tournament rotates n/2 disjoint pairs concurrently (16 lanes per pair); n-1 rounds make a sweep.
// Rank policy: the reference keeps every singular value >= 1e-100 (util_wct.py:25,82-86), i.e. all of
// them -- null directions get lambda ~ 1e-15 from LAPACK and are multiplied into exactly-zero centred
// features.  Jacobi returns the same directions with tiny norms; directions with
// lambda <= max(REL_THRESH * lambda_max, ABS_FLOOR * max E[x^2]) are dropped (REL_THRESH 1e-12: round-off puts null
// directions at <= 1e-16 lambda_max, while GENUINE directions 1e-10..1e-11 below the top one exist -- --mode original on
// generated weights has them -- and the reference whitens those like any other), which reproduces the reference to
// <= 2e-6 in every regime of tests/golden/g3.  The absolute floor is the round-off level of the covariance itself (formed from raw fp64
// sums of magnitude E[x^2]); it makes a constant feature map (cov = 0 up to round-off) whiten to exactly 0 like
// the reference's exact-zero centred features do (k_c = 0 -> target = s_mean, util_wct.py:82-86,117-126).
#include "wct_common.h"
#include <cstdlib>

namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int MAX_SWEEPS = 40;
constexpr double ROT_TOL = 1e-12;  // relative off-diagonal; quadratic convergence overshoots this by far
constexpr double ABS_FLOOR = 1e-13;
static const double REL_THRESH = [] { const char* e = wct_debug_env("WCT_REL_THRESH"); return e ? atof(e) : 1e-12; }();

__device__ __forceinline__ void tournament_pair(int n, int round, int k, int& p, int& q) {
  // circle method: player n-1 stays, the others rotate
  const int m = n - 1;
  if (k == 0) { p = m; q = round % m; }
  else { p = (round + k) % m; q = (round - k + m) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

// 1/sqrt(x) to fp64 round-off: hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// Jacobi rotation that zeroes the (p,q) entry of G^T G, division-free:
//   d = be - al, h = hypot(d, 2 ga), c^2 = (1 + |d|/h)/2, s = sign(d) ga / (h c), t = s / c
// (equivalent to Rutishauser's t = sign(zeta)/(|zeta| + sqrt(1 + zeta^2)), zeta = d/(2 ga): the small-angle root)
__device__ __forceinline__ double rotation(double al, double be, double ga, double& c, double& s) {
  const double d = be - al, g2 = 2.0 * ga;
  const double rh = rsqrt_nr(d * d + g2 * g2);       // 1/h
  const double c2 = 0.5 + 0.5 * fabs(d) * rh;        // in [0.5, 1]
  const double rc = rsqrt_nr(c2);                    // 1/c
  c = c2 * rc;
  const double sg = d >= 0 ? ga : -ga;
  s = sg * rh * rc;
  return s * rc;                                     // t
}

// ---- EigResult layout (doubles): G[C*C] | lam[C] | mu[C] | floor | pad(3) | F[C*C]
//      F = cov^(-1/2) (content) or cov^(+1/2) (style) on the live subspace -- what launch_assemble consumes
__host__ __device__ inline size_t eig_doubles(int C) { return 2 * (size_t)C * C + 2 * (size_t)C + 4; }
__host__ __device__ inline size_t eig_F_offset(int C) { return (size_t)C * C + 2 * (size_t)C + 4; }

// element e of the covariance (+ mean, + floor by the first wave): shared by cov_kernel (grid-wide) and the single-workgroup
// kernels that start from the raw moments (ns_lds_kernel, ns_prep_kernel) -- one arithmetic, bit-identical results
__device__ __forceinline__ void cov_floor(int C, double n, const double* sumsq, double* res, int lane) {
  double ex2 = 0.;                              // eigenvalue floor from max E[x^2]: one wave, strided + shuffle max
  for (int j = lane; j < C; j += 64) ex2 = fmax(ex2, sumsq[(size_t)j * C + j]);   // (a single thread walking the diagonal cost 19 us of dependent loads)
  for (int o = 32; o > 0; o >>= 1) ex2 = fmax(ex2, __shfl_xor(ex2, o));
  if (lane == 0) res[(size_t)C * C + 2 * C] = ABS_FLOOR * ex2 / n;
}
__device__ __forceinline__ double cov_value(int r, int c, int C, double n, const double* sum, const double* sumsq, double diag_add) {
  const double mr = sum[r] / n, mc = sum[c] / n;
  // symmetric by construction: use the (min,max) entry for both halves
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return (sumsq[(size_t)lo * C + hi] - n * mr * mc) / (n - 1.0) + (r == c ? diag_add : 0.0);   // diag_add: `--numpy` (+ I)
}
__device__ __forceinline__ void cov_element(long e, int C, double n, const double* sum, const double* sumsq, double* res, double diag_add) {
  const int r = (int)(e / C), c = (int)(e % C);
  res[e] = cov_value(r, c, C, n, sum, sumsq, diag_add);
  if (c == 0) res[(size_t)C * C + C + r] = sum[r] / n;  // mu
}

__global__ void cov_kernel(int C, double n, const double* sum, const double* sumsq, double* res, double diag_add) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 64) cov_floor(C, n, sumsq, res, threadIdx.x);
  if (e >= (long)C * C) return;
  cov_element(e, C, n, sum, sumsq, res, diag_add);
}

// the covariance by ONE workgroup (the caller synchronises afterwards): the single-workgroup solvers start from the raw moments
__device__ __forceinline__ void cov_by_block(int C, double n, const double* sum, const double* sumsq, double* res, double diag_add, int tid, int nt) {
  if (tid < 64) cov_floor(C, n, sumsq, res, tid);
  for (long e = tid; e < (long)C * C; e += nt) cov_element(e, C, n, sum, sumsq, res, diag_add);
}

// =====================================================================================================
// Primary path: coupled Newton-Schulz iteration for cov^(1/2) and cov^(-1/2) -- all GEMMs, so it runs on the
// whole chip's fp64 matrix cores instead of one CU (a Jacobi sweep is a chain of n-1 dependent rounds):
//     Y0 = A/s, Z0 = I;   T = (3I - Z Y)/2;  Y <- Y T;  Z <- T Z;     Y -> (A/s)^(1/2),  Z -> (A/s)^(-1/2)
// with s = ||A||_F (eigenvalues of A/s in (0,1]).  Convergence is quadratic once ||I - Z Y|| < 1; covariances
// of this path (cond 1e1..1e5 on the live block) take 10..18 iterations to 1e-14 (numpy prototype matched eigh to
// 1e-13).  Dead channels (variance at round-off level) are replaced by an identity block and zeroed in the
// result, which is the pseudo-inverse square root the Jacobi path forms by dropping them.
// The launch schedule is fixed (no host round trip): stage kernels of iteration k return at once when the
// residual of iteration k-1 is already below NS_TOL.  If the budget runs out (singular or very ill-conditioned
// matrix, e.g. fewer pixels than channels) ok stays 0 and the Jacobi path below takes over (C <= 128: one gated
// launch).  C > 128 -- and the 128-channel level of a model that has wider ones -- runs the DEFLATED, optimally scaled
// form of the same iteration instead (below: singular matrices included, 19 iterations); only its outcome is read back.
constexpr int NS_MAXIT = 26, NS_MAXIT_REG = 96;   // default budget; size of the residual array (WCT_NS_MAXIT may raise the budget)
constexpr double NS_TOL = 1e-7;   // on max|ZY - I| BEFORE an update; the update squares it (quadratic convergence)
// Condition gate of the inverse square root: Z -> (A/s)^(-1/2), so ||Z||_F >= (lambda_min/s)^(-1/2) (and <= sqrt(C) times it).
// Above 1e6 (lambda_min <~ 1e-12 s; the eps-regularised null space of a rank-deficient covariance lands at 3e7) the
// converged iterate is the inverse of round-off, not the pseudo-inverse the reference's exactly-zero centred components
// make of it (util_wct.py:117-120): ok stays 0 and the Jacobi path drops those directions (REL_THRESH, above).  Below it
// every direction is genuine and kept, as the reference does; the iteration's error there is ~10 cond eps (8e-5 of max|M|
// at lambda_min = 1e-11 lambda_max, tools/experiments/solve_cond.py).
static const double NS_ZMAX = [] { const char* e = wct_debug_env("WCT_NS_ZMAX"); return e ? atof(e) : 1e6; }();

struct NsWs {           // carved from the eig workspace
  double* Y[2]; double* Z[2]; double* T;
  int* dead;            // [C]
  double* scal;         // [0] = s (Frobenius norm)
  unsigned long long* resid;  // [maxit + 1] max |ZY - I| per iteration, as double bits (non-negative -> integer order)
  double* zfro;         // [maxit + 1] ||Z_k||_F^2 of the iterate entering iteration k (live block), for the condition gate
  int* iters;           // iterations actually executed
  int* ok;              // 1: F holds the Newton-Schulz result
  unsigned* coop;       // single-launch iteration (ns_coop128_kernel): [0] barrier arrivals, [1] abort flag, [2] XCC id + 1 of participant 0.
                        // A lane owns TWO such sets and alternates: a solve counts in one and zeroes the other for its successor, so a
                        // solve that went wrong half-way leaves nothing behind that the next one could trip over
};

__device__ __forceinline__ void ns_init_body(const double* res, int C, double eps_rel, const NsWs& w, int maxit, double* red, int* sdead) {
  const int tid = threadIdx.x;
  const double floor_ = res[(size_t)C * C + 2 * C];
  for (int j = tid; j < C; j += 1024) { const int d = !(res[(size_t)j * C + j] > floor_); w.dead[j] = d; sdead[j] = d; }
  for (int k = tid; k <= maxit; k += 1024) { w.resid[k] = 0ull; w.zfro[k] = 0.; }  // atomic targets; a skipped iteration leaves resid 0 = "converged"
  if (tid == 0) { *w.iters = 0; *w.ok = 0; }
  __syncthreads();
  // ||A_live||_F by ONE workgroup (a fixed summation order: the scale must not depend on scheduling); rows in turn, the
  // row's columns across the threads, 4 independent rows in flight (the first version -- one element per thread per trip,
  // flags from global memory -- took 214 us at C = 512)
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  {
    const int nrg = 1024 / C > 0 ? 1024 / C : 1;   // row groups in parallel (C = 512: 2)
    const int grp = tid / C, c = tid - grp * C;
    const bool live_c = grp < nrg && !sdead[c];
#pragma unroll 2
    for (int r = 4 * grp; r < C; r += 4 * nrg) {
      double v0 = 0., v1 = 0., v2 = 0., v3 = 0.;
      if (live_c) {
        v0 = res[(size_t)r * C + c];
        if (r + 1 < C) v1 = res[(size_t)(r + 1) * C + c];
        if (r + 2 < C) v2 = res[(size_t)(r + 2) * C + c];
        if (r + 3 < C) v3 = res[(size_t)(r + 3) * C + c];
      }
      if (!sdead[r]) s0 += v0 * v0;
      if (r + 1 < C && !sdead[r + 1]) s1 += v1 * v1;
      if (r + 2 < C && !sdead[r + 2]) s2 += v2 * v2;
      if (r + 3 < C && !sdead[r + 3]) s3 += v3 * v3;
    }
  }
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    double f = sqrt(red[0]);
    if (!(f > 0.)) f = 1.;            // all channels dead: Y0 = I, result zeroed anyway
    w.scal[0] = f * (1.0 + eps_rel);   // keeps the spectrum of (A + eps f I)/s inside (0, 1]
    w.scal[1] = eps_rel * f;
  }
}

__global__ __launch_bounds__(1024) void ns_init_kernel(const double* res, int C, double eps_rel, NsWs w, int maxit) {
  __shared__ double red[1024];
  __shared__ int sdead[512];
  ns_init_body(res, C, eps_rel, w, maxit, red, sdead);
}

__device__ __forceinline__ void ns_fill_element(long e, const double* res, int C, int Cp, const NsWs& w, const int* dead, double s0, double s1) {
  const int r = (int)(e / Cp), c = (int)(e % Cp);
  double y = r == c ? 1.0 : 0.0;     // padding rows/cols and dead channels: identity block
  if (r < C && c < C && !dead[r] && !dead[c]) y = (res[(size_t)r * C + c] + (r == c ? s1 : 0.0)) / s0;
  w.Y[0][e] = y;
  w.Z[0][e] = r == c ? 1.0 : 0.0;
}

__global__ void ns_fill_kernel(const double* res, int C, int Cp, NsWs w) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Cp * Cp) return;
  ns_fill_element(e, res, C, Cp, w, w.dead, w.scal[0], w.scal[1]);
}

// Cp = 128 of the 16x levels: covariance + dead flags + Frobenius scale + Y0 / Z0 by ONE workgroup in ONE launch -- the three
// kernels above each take 5-6 us of which most is launch latency, and they sit on the content side's critical path twice per
// frame.  Same element arithmetic and the same summation order as cov_kernel / ns_init_kernel / ns_fill_kernel: bit-identical.
__global__ __launch_bounds__(1024) void ns_prep_kernel(int C, int Cp, double n, const double* sum, const double* sumsq, double* res, double diag_add,
                                                         double eps_rel, NsWs w, int maxit) {
  __shared__ double red[1024];
  __shared__ int sdead[512];
  cov_by_block(C, n, sum, sumsq, res, diag_add, threadIdx.x, 1024);
  __syncthreads();                   // the covariance (global memory, written by this workgroup) is visible to all its threads
  ns_init_body(res, C, eps_rel, w, maxit, red, sdead);
  __syncthreads();
  const double s0 = w.scal[0], s1 = w.scal[1];
  for (long e = threadIdx.x; e < (long)Cp * Cp; e += 1024) ns_fill_element(e, res, C, Cp, w, sdead, s0, s1);
}

// one 16x16 output tile per wave, 2x2 tiles per workgroup: D = P Q with the TRUE row-major operands.  (Reading P
// transposed because "every iterate is symmetric" is tempting -- both operands would be coalesced -- but it makes the
// iteration unstable: the antisymmetric part of the round-off is amplified and it diverges for cond >~ 3e3.)
__device__ __forceinline__ f64x4 tile_gemm(const double* P, const double* Q, int Cp, int i0, int j0, int lane) {
  const int li = lane & 15, kk = lane >> 4;
  f64x4 acc = f64x4{0., 0., 0., 0.};
  const double* pp = P + (size_t)(i0 + li) * Cp + kk;
  const double* qq = Q + (size_t)kk * Cp + j0 + li;
  for (int k0 = 0; k0 < Cp; k0 += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = pp[k0 + 4 * u]; b[u] = qq[(size_t)(k0 + 4 * u) * Cp]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;  // row = kk + 4 * reg, col = li
}

__device__ __forceinline__ bool ns_converged(const NsWs& w, int it) {
  return it > 0 && __longlong_as_double((long long)w.resid[it - 1]) < NS_TOL;
}

// stage 1 of iteration `it`: T = ca I - cb Z Y (plain iteration: 1.5, 0.5; scaled: 1.5 mu, 0.5 mu^3) ; resid[it] = max |Z Y - I|
__global__ __launch_bounds__(256) void ns_stage1_kernel(NsWs w, int Cp, int it, double ca, double cb) {
  if (ns_converged(w, it)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
  const int cur = it & 1;
  const f64x4 acc = tile_gemm(w.Z[cur], w.Y[cur], Cp, i0, j0, lane);
  const int li = lane & 15, kk = lane >> 4;
  double m = 0.;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = i0 + kk + 4 * r, col = j0 + li;
    const double zy = acc[r], d = zy - (row == col ? 1.0 : 0.0);
    m = (d == d) ? fmax(m, fabs(d)) : __longlong_as_double(0x7ff0000000000000ll);  // fmax would swallow a NaN
    w.T[(size_t)row * Cp + col] = (row == col ? ca : 0.0) - cb * zy;
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (lane == 0) atomicMax(&w.resid[it], (unsigned long long)__double_as_longlong(m));
}

// stage 2: Y' = Y T (blockIdx.z = 0), Z' = T Z (blockIdx.z = 1), into the other ping-pong buffer
__global__ __launch_bounds__(256) void ns_stage2_kernel(NsWs w, int Cp, int it) {
  if (ns_converged(w, it)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
  const int cur = it & 1, nxt = cur ^ 1;
  const bool zside = blockIdx.z == 1;
  const f64x4 acc = zside ? tile_gemm(w.T, w.Z[cur], Cp, i0, j0, lane) : tile_gemm(w.Y[cur], w.T, Cp, i0, j0, lane);
  double* out = zside ? w.Z[nxt] : w.Y[nxt];
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(size_t)(i0 + kk + 4 * r) * Cp + j0 + li] = acc[r];
  if (zside) {   // ||Z'||_F^2 (the identity block of dead / padding channels adds at most Cp: irrelevant against the gate)
    double q = acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) atomicAdd(&w.zfro[it + 1], q);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *w.iters = it + 1;
}

// ---- C > 128 (Cp = 256, 512): one 32x32 output tile per WORKGROUP, the k range split over its 4 waves and summed through LDS
//      in a fixed order.  A whole-k wave per 16x16 tile is a chain of Cp/16 dependent load batches with one 8-byte operand
//      load per lane per MFMA (~30 us per launch at Cp = 512, against 3.4 us of fp64 MFMA work); here a wave holds 2x2 MFMA
//      tiles (half the operand bytes per MFMA) over a quarter of k, and lane group kk owns k = k0 + 4 kk + u of a 16-block, so
//      its four P values are 32 contiguous bytes (two 16-byte loads; the wave reads whole 128-byte lines of 16 rows instead
//      of 32 bytes out of each).  The k order differs from tile_gemm's -- both are "true row-major operands".
struct Acc32 { f64x4 t[4]; };   // MFMA tile (ri, cj) at t[2 ri + cj]: rows i0 + 16 ri + kk + 4 reg, cols j0 + 16 cj + li

// One 16-block of k for the 2x2 MFMA tiles of a wave: 4 16-byte loads of P (two row groups), 8 8-byte loads of Q.
struct KBlock { f64x2 a0l, a0h, a1l, a1h; double b0[4], b1[4]; };
__device__ __forceinline__ void kblock_load(KBlock& b, const double* p0, const double* p1, const double* q0, int Cp, int k0) {
  b.a0l = *reinterpret_cast<const f64x2*>(p0 + k0); b.a0h = *reinterpret_cast<const f64x2*>(p0 + k0 + 2);
  b.a1l = *reinterpret_cast<const f64x2*>(p1 + k0); b.a1h = *reinterpret_cast<const f64x2*>(p1 + k0 + 2);
#pragma unroll
  for (int u = 0; u < 4; ++u) { b.b0[u] = q0[(size_t)(k0 + u) * Cp]; b.b1[u] = q0[(size_t)(k0 + u) * Cp + 16]; }
}
__device__ __forceinline__ void kblock_mfma(const KBlock& b, Acc32& acc) {
  const double a0[4] = {b.a0l[0], b.a0l[1], b.a0h[0], b.a0h[1]}, a1[4] = {b.a1l[0], b.a1l[1], b.a1h[0], b.a1h[1]};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc.t[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b.b0[u], acc.t[0], 0, 0, 0);
    acc.t[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b.b1[u], acc.t[1], 0, 0, 0);
    acc.t[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b.b0[u], acc.t[2], 0, 0, 0);
    acc.t[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b.b1[u], acc.t[3], 0, 0, 0);
  }
}

// NS = k slices = waves per workgroup: 4 (Cp = 128) or 8 (Cp >= 256: at one 4-wave workgroup per CU these launches are bound by
// each wave's serial chain of k-blocks -- 8 at Cp = 512 -- not by the matrix cores; twice the waves halve the chain)
template <int NS>
__device__ __forceinline__ void gemm32_splitk(const double* P, const double* Q, int Cp, int i0, int j0, int lane, int wave,
                                               double (*red)[16 * 64], Acc32& acc) {
  const int li = lane & 15, kk = lane >> 4;
  const int kq = Cp / NS, kbeg = wave * kq, kend = kbeg + kq;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc.t[t] = f64x4{0., 0., 0., 0.};
  const double* p0 = P + (size_t)(i0 + li) * Cp + 4 * kk;
  const double* p1 = p0 + (size_t)16 * Cp;
  const double* q0 = Q + (size_t)(4 * kk) * Cp + j0 + li;
  // (An explicit software pipeline over pairs of k-blocks -- the next pair's loads in flight during this pair's 32 MFMAs -- was
  // measured in round 3: the C = 512 matrix functions got 6 % SLOWER, 3.08 -> 3.28 ms per cfg3 frame and side; the compiler's own
  // schedule of the 2-unrolled loop stays.)
  KBlock x;
#pragma unroll 2
  for (int k0 = kbeg; k0 < kend; k0 += 16) { kblock_load(x, p0, p1, q0, Cp, k0); kblock_mfma(x, acc); }
  if (wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][(t * 4 + r) * 64 + lane] = acc.t[t][r];
  }
  __syncthreads();
  if (!wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int at = (t * 4 + r) * 64 + lane;
        double v = acc.t[t][r];
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) v += red[q][at];      // slice order: fixed, reproducible
        acc.t[t][r] = v;
      }
  }
}

template <int NS>
__global__ __launch_bounds__(64 * NS) void ns_stage1_wide_kernel(NsWs w, int Cp, int it, double ca, double cb) {
  if (ns_converged(w, it)) return;
  __shared__ double red[NS - 1][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, cur = it & 1;
  Acc32 acc;
  gemm32_splitk<NS>(w.Z[cur], w.Y[cur], Cp, i0, j0, lane, wave, red, acc);
  if (wave) return;
  const int li = lane & 15, kk = lane >> 4;
  double m = 0.;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * (t >> 1) + kk + 4 * r, col = j0 + 16 * (t & 1) + li;
      const double zy = acc.t[t][r], d = zy - (row == col ? 1.0 : 0.0);
      m = (d == d) ? fmax(m, fabs(d)) : __longlong_as_double(0x7ff0000000000000ll);
      w.T[(size_t)row * Cp + col] = (row == col ? ca : 0.0) - cb * zy;
    }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (lane == 0) atomicMax(&w.resid[it], (unsigned long long)__double_as_longlong(m));
}

template <int NS>
__global__ __launch_bounds__(64 * NS) void ns_stage2_wide_kernel(NsWs w, int Cp, int it) {
  if (ns_converged(w, it)) return;
  __shared__ double red[NS - 1][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, cur = it & 1, nxt = cur ^ 1;
  const bool zside = blockIdx.z == 1;
  Acc32 acc;
  gemm32_splitk<NS>(zside ? w.T : w.Y[cur], zside ? w.Z[cur] : w.T, Cp, i0, j0, lane, wave, red, acc);
  if (wave) return;
  double* out = zside ? w.Z[nxt] : w.Y[nxt];
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(i0 + 16 * (t >> 1) + kk + 4 * r) * Cp + j0 + 16 * (t & 1) + li] = acc.t[t][r];
  if (zside) {   // ||Z'||_F^2 for the condition gate of the plain (C <= 128) iteration
    double q = 0.;
#pragma unroll
    for (int t = 0; t < 4; ++t) q += acc.t[t][0] * acc.t[t][0] + acc.t[t][1] * acc.t[t][1] + acc.t[t][2] * acc.t[t][2] + acc.t[t][3] * acc.t[t][3];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) atomicAdd(&w.zfro[it + 1], q);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *w.iters = it + 1;
}

// ---- the decoder-side fold of the wide models (--mode original, cin = 256 / 512): W'[(o, tap)][i] = SUM_c W[(o, tap)][c] M[c][i]
// as a [cout * 9] x [cin] x [cin] fp64 GEMM on the matrix cores, with this file's 32 x 32 split-k tiles (the rows are the
// decoder's first-conv weights as doubles, [(o, tap)][c], prepared once at load: fold_rows_kernel).  It replaces misc.hip's
// fold_block_kernel on those layers -- fp64 VALU FMAs fed by wave-uniform LDS reads: 271 us per launch at cin = cout = 512, 9 TF --
// which sits on the content lane between the inverse square root and the decoder.  Epilogue: the conv kernels' packed fp32
// layout + max |w'| for the f16 split's scale; the workgroups past the GEMM rows form b'[o] = bias[o] + SUM_c Wsum[o][c] b[c].
template <int NS>
__global__ __launch_bounds__(64 * NS) void fold_gemm_kernel(const double* Wd, const double* Wsum, const float* bias, int cout, int cin, int cout_pad,
                                                             const double* M, const double* b, float* wpk, float* bias_out, unsigned* maxbits) {
  __shared__ double red[NS - 1][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = cout * 9;
  if ((int)blockIdx.y * 32 >= rows) {       // bias rows: one wave per output channel
    if (blockIdx.x != 0) return;
    const int o = ((int)blockIdx.y - rows / 32) * NS + wave;
    if (o >= cout_pad) return;
    double part = 0.;
    if (o < cout)
      for (int c = lane; c < cin; c += 64) part += Wsum[(size_t)o * cin + c] * b[c];
    for (int k = 32; k > 0; k >>= 1) part += __shfl_xor(part, k);
    if (lane == 0) bias_out[o] = o < cout ? (float)((double)bias[o] + part) : 0.f;
    return;
  }
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  Acc32 acc;
  gemm32_splitk<NS>(Wd, M, cin, i0, j0, lane, wave, red, acc);
  if (wave) return;
  const int li = lane & 15, kk = lane >> 4;
  float mx = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * (t >> 1) + kk + 4 * r, i = j0 + 16 * (t & 1) + li;
      const int o = row / 9, tap = row - 9 * o;
      const float v = (float)acc.t[t][r];
      mx = fmaxf(mx, fabsf(v));
      wpk[((((size_t)(i >> 4) * 9 + tap) * 4 + ((i >> 2) & 3)) * cout_pad + o) * 4 + (i & 3)] = v;
    }
  for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
  if (maxbits && lane == 0) atomicMax(maxbits, __float_as_uint(mx));
}

// OIHW fp32 [o][c][9] -> rows[(o * 9 + tap)][c] and wsum[o][c] = SUM_tap w (tap order 0..8), as doubles
__global__ void fold_rows_kernel(const float* w, int cout, int cin, double* rows, double* wsum) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)cout * cin) return;
  const int o = (int)(e / cin), c = (int)(e % cin);
  double ws = 0.;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const double v = (double)w[e * 9 + t];
    rows[((size_t)o * 9 + t) * cin + c] = v;
    ws += v;
  }
  wsum[e] = ws;
}

// ---- Cp = 128 (the 128-channel levels of --mode 16x): covariance, scaling, the whole iteration and the result in ONE launch.
// Alone on the GPU the 1 + 2 x 16 + 1 launches above take ~0.13 ms.  In a stylise call they never are alone: the other lane runs
// persistent convolution kernels whose workgroups own a CU each (all of its LDS and registers) for the kernel's lifetime, so
// every one of the dependent launches queues for a CU again -- measured with HIP events in an overlapped 4K step
// (tools/experiments/lane_timeline.py): 0.59 / 0.32 ms for the two content-lane solves, 0.76 / 0.71 ms for the style-lane ones,
// and what the content lane waits there is step time.  One launch queues once.
// The 32 workgroups that the dispatcher places on ONE XCD (workgroup b -> XCD b mod 8; every participant compares its XCC_ID with
// participant 0's) work inside the kernel:
//   front end   every participant derives dead flags, Frobenius scale (ns_init_body's summation order, on its own) from the raw
//               moments, and writes ITS four rows of the covariance, of Y0 and of Z0
//   iteration   16 participants form T (stage 1), all 32 form Y' and Z' (stage 2); a software barrier on an agent-scope counter
//               after each stage; the iterates travel with sc1 (agent-scope, L1-bypassing) buffer loads / stores
//   back end    ns_final_kernel's scaling of ITS four rows of the result
// Element arithmetic, tile products, k split over the four waves, every summation order: the multi-launch path's -- the result is
// that path's bit for bit (tools/experiments/ns_coop_probe.hip; tests/test_hip_parity.py).
// It cannot hang: a participant that waits 5 ms at a barrier (or finds itself on another XCD) raises the abort flag and everyone
// leaves; the gated Jacobi launch behind every solve looks at that flag as well as at `ok` and does the solve.
constexpr int COOP_NW = 32, COOP_MAXIT = 32;
struct NsSched { double ca[COOP_MAXIT], cb[COOP_MAXIT]; };
struct CoopArgs {
  int C; double n; const double* sum; const double* sumsq; double* res; double diag_add, eps_rel;
  NsWs w; int maxit, inverse; double zmax; int* info; int xcd;
  unsigned* coop_next;
  unsigned* aborts;      // the lane's count of aborted single-launch solves (may be null): raised once per aborted solve
};
constexpr int BUF_SC1 = 16;   // buffer cache policy: agent scope (never served from this CU's L1)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coop_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 128 * 128 * 8, 0x00020000);
}
__device__ __forceinline__ f64x2 coop_ld2(__amdgpu_buffer_rsrc_t r, int elem) {
  return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, elem * 8, 0, BUF_SC1));
}
__device__ __forceinline__ double coop_ld1(__amdgpu_buffer_rsrc_t r, int elem) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, elem * 8, 0, BUF_SC1));
}
__device__ __forceinline__ void coop_st1(__amdgpu_buffer_rsrc_t r, int elem, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, elem * 8, 0, BUF_SC1);
}
// gemm32_splitk<4> at Cp = 128 (two k-blocks per wave, both in flight), operands through sc1 buffer loads
__device__ __forceinline__ void gemm32_coop(const double* P, const double* Q, int i0, int j0, int lane, int wave, double (*red)[16 * 64], Acc32& acc) {
  constexpr int CP = 128;
  const int li = lane & 15, kk = lane >> 4, kbeg = wave * (CP / 4);
  const __amdgpu_buffer_rsrc_t rp = coop_rsrc(P), rq = coop_rsrc(Q);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc.t[t] = f64x4{0., 0., 0., 0.};
  const int p0 = (i0 + li) * CP + 4 * kk, p1 = p0 + 16 * CP, q0 = (4 * kk) * CP + j0 + li;
  KBlock x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int k0 = kbeg + 16 * b;
    x[b].a0l = coop_ld2(rp, p0 + k0); x[b].a0h = coop_ld2(rp, p0 + k0 + 2);
    x[b].a1l = coop_ld2(rp, p1 + k0); x[b].a1h = coop_ld2(rp, p1 + k0 + 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) { x[b].b0[u] = coop_ld1(rq, q0 + (k0 + u) * CP); x[b].b1[u] = coop_ld1(rq, q0 + (k0 + u) * CP + 16); }
  }
  kblock_mfma(x[0], acc);
  kblock_mfma(x[1], acc);
  if (wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][(t * 4 + r) * 64 + lane] = acc.t[t][r];
  }
  __syncthreads();
  if (!wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int at = (t * 4 + r) * 64 + lane;
        acc.t[t][r] = ((acc.t[t][r] + red[0][at]) + red[1][at]) + red[2][at];
      }
  }
  __syncthreads();     // red is reused by the next product
}
// all of this workgroup's stores have reached L2, then: arrive, wait for the other participants (or for the abort flag)
// The watchdog: 5 ms of the 100 MHz wall clock (a healthy solve takes ~0.1 ms in all; a participant that cannot be placed because the
// other lane's persistent workgroups own the XCD waits for one of their jobs, < 1 ms).  Whoever raises the abort flag FIRST counts the
// solve in the lane's abort counter: the host mirrors it and stops using the single launch on a lane where it keeps failing
// (wct_api.hip coop_usable) instead of paying the timeout plus the ~2 ms Jacobi net on every solve, silently.
constexpr long long COOP_TIMEOUT_TICKS = 500000ll;
__device__ __forceinline__ void coop_abort(const NsWs& w, unsigned bit, unsigned* aborts) {
  const unsigned old = __hip_atomic_fetch_or(&w.coop[1], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == 0u && aborts) __hip_atomic_fetch_add(aborts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool coop_barrier(const NsWs& w, unsigned& target, int* ok_s, unsigned* aborts) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  target += COOP_NW;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&w.coop[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();     // 100 MHz
    bool ok = true;
    while (__hip_atomic_load(&w.coop[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS || __hip_atomic_load(&w.coop[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        coop_abort(w, 1u, aborts);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *ok_s = ok;
  }
  __syncthreads();
  return *ok_s != 0;
}

__global__ __launch_bounds__(256) void ns_coop128_kernel(CoopArgs a, NsSched sc) {
  const bool inject = (a.xcd & 16) != 0;             // debug key "nscoop" = 2: behave as if a participant sat on another XCD
  if ((int)(blockIdx.x & 7) != (a.xcd & 7)) return;  // the workgroups the dispatcher places on one XCD
  constexpr int CP = 128;
  __shared__ double red[3][16 * 64];                  // the front end's 1024 partial sums, then the k-slice sums of the products
  __shared__ int sdead[CP];
  __shared__ int ok_s;
  const NsWs& w = a.w;
  const int C = a.C, tid = threadIdx.x, me = blockIdx.x >> 3;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;
  const double n = a.n;
  // ---- front end (ns_prep_kernel's arithmetic; nothing here reads what another participant writes)
  if (me == 0) {
    if (tid < 4) a.coop_next[tid] = 0u;               // the state the NEXT single-launch solve of this lane will use (nobody touches it now)
    if (tid < 64) cov_floor(C, n, a.sumsq, a.res, tid);
    for (int k = tid; k <= a.maxit; k += 256) { w.resid[k] = 0ull; w.zfro[k] = 0.; }   // atomic targets of the iteration (first touched after the barriers below)
    if (tid == 0) { *w.iters = 0; *w.ok = 0; __hip_atomic_store(&w.coop[2], xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  {
    double ex2 = 0.;                                  // cov_floor's value, by every participant for itself
    for (int j = lane; j < C; j += 64) ex2 = fmax(ex2, a.sumsq[(size_t)j * C + j]);
    for (int o = 32; o > 0; o >>= 1) ex2 = fmax(ex2, __shfl_xor(ex2, o));
    const double floor_ = ABS_FLOOR * ex2 / n;
    for (int j = tid; j < CP; j += 256) {
      const int d = j < C ? !(cov_value(j, j, C, n, a.sum, a.sumsq, a.diag_add) > floor_) : 1;
      sdead[j] = d;
      if (me == 0 && j < C) w.dead[j] = d;
    }
  }
  __syncthreads();
  double* part = &red[0][0];
  {
    // ||A_live||_F: ns_init_body's partition over 1024 threads and its tree, four of those threads per thread here
    const int nrg = 1024 / C > 0 ? 1024 / C : 1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int vt = tid + 256 * v, grp = vt / C, c = vt - grp * C;
      const bool live_c = grp < nrg && !sdead[c];
      double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
      for (int r = 4 * grp; r < C; r += 4 * nrg) {
        double v0 = 0., v1 = 0., v2 = 0., v3 = 0.;
        if (live_c) {
          v0 = cov_value(r, c, C, n, a.sum, a.sumsq, a.diag_add);
          if (r + 1 < C) v1 = cov_value(r + 1, c, C, n, a.sum, a.sumsq, a.diag_add);
          if (r + 2 < C) v2 = cov_value(r + 2, c, C, n, a.sum, a.sumsq, a.diag_add);
          if (r + 3 < C) v3 = cov_value(r + 3, c, C, n, a.sum, a.sumsq, a.diag_add);
        }
        if (!sdead[r]) s0 += v0 * v0;
        if (r + 1 < C && !sdead[r + 1]) s1 += v1 * v1;
        if (r + 2 < C && !sdead[r + 2]) s2 += v2 * v2;
        if (r + 3 < C && !sdead[r + 3]) s3 += v3 * v3;
      }
      part[vt] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      for (int i = tid; i < o; i += 256) part[i] += part[i + o];
      __syncthreads();
    }
  }
  double fro = sqrt(part[0]);
  if (!(fro > 0.)) fro = 1.;
  const double s0 = fro * (1.0 + a.eps_rel), s1 = a.eps_rel * fro;
  if (me == 0 && tid == 0) { w.scal[0] = s0; w.scal[1] = s1; }
  __syncthreads();       // part[] (red) is free again
  {
    // rows 4 me .. 4 me + 3: covariance + mean for whoever reads `res` later; Y0 = (A + eps f I) / s, Z0 = I (ns_fill_element)
    const __amdgpu_buffer_rsrc_t ry = coop_rsrc(w.Y[0]), rz = coop_rsrc(w.Z[0]);
    for (int e = tid; e < 4 * CP; e += 256) {
      const int r = 4 * me + e / CP, c = e % CP;
      double y = r == c ? 1.0 : 0.0;
      if (r < C && c < C) {
        const double cv = cov_value(r, c, C, n, a.sum, a.sumsq, a.diag_add);
        a.res[(size_t)r * C + c] = cv;
        if (c == 0) a.res[(size_t)C * C + C + r] = a.sum[r] / n;
        if (!sdead[r] && !sdead[c]) y = (cv + (r == c ? s1 : 0.0)) / s0;
      }
      coop_st1(ry, r * CP + c, y);
      coop_st1(rz, r * CP + c, r == c ? 1.0 : 0.0);
    }
  }
  unsigned target = 0;
  if (!coop_barrier(w, target, &ok_s, a.aborts)) return;
  // all participants on participant 0's XCD: the exchange is coherent across XCDs too (ns_coop_probe -DSPREAD), but 15 % slower
  if (tid == 0 && (__hip_atomic_load(&w.coop[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc_id() + 1u || (inject && me == 7)))
    coop_abort(w, 2u, a.aborts);
  if (!coop_barrier(w, target, &ok_s, a.aborts)) return;
  if (__hip_atomic_load(&w.coop[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  // ---- iteration
  const int tile = me & 15, zside = me >> 4;         // stage 1: tiles by participants 0..15; stage 2: Y' by 0..15, Z' by 16..31
  const int i0 = (tile >> 2) * 32, j0 = (tile & 3) * 32;
  int nit = 0;
  for (int it = 0; it < a.maxit; ++it) {
    if (it > 0 && __longlong_as_double((long long)__hip_atomic_load(&w.resid[it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < NS_TOL) break;
    const int cur = it & 1, nxt = cur ^ 1;
    if (!zside) {       // ns_stage1_wide_kernel
      Acc32 acc;
      gemm32_coop(w.Z[cur], w.Y[cur], i0, j0, lane, wave, red, acc);
      if (!wave) {
        const __amdgpu_buffer_rsrc_t rt = coop_rsrc(w.T);
        const double ca = sc.ca[it], cb = sc.cb[it];
        double m = 0.;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i0 + 16 * (t >> 1) + kk + 4 * r, col = j0 + 16 * (t & 1) + li;
            const double zy = acc.t[t][r], d = zy - (row == col ? 1.0 : 0.0);
            m = (d == d) ? fmax(m, fabs(d)) : __longlong_as_double(0x7ff0000000000000ll);
            coop_st1(rt, row * CP + col, (row == col ? ca : 0.0) - cb * zy);
          }
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(&w.resid[it], (unsigned long long)__double_as_longlong(m));
      }
    }
    if (!coop_barrier(w, target, &ok_s, a.aborts)) return;
    {                   // ns_stage2_wide_kernel
      Acc32 acc;
      gemm32_coop(zside ? w.T : w.Y[cur], zside ? w.Z[cur] : w.T, i0, j0, lane, wave, red, acc);
      if (!wave) {
        const __amdgpu_buffer_rsrc_t ro = coop_rsrc(zside ? w.Z[nxt] : w.Y[nxt]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) coop_st1(ro, (i0 + 16 * (t >> 1) + kk + 4 * r) * CP + j0 + 16 * (t & 1) + li, acc.t[t][r]);
        if (zside) {   // ||Z'||_F^2 for the condition gate
          double q = 0.;
#pragma unroll
          for (int t = 0; t < 4; ++t) q += acc.t[t][0] * acc.t[t][0] + acc.t[t][1] * acc.t[t][1] + acc.t[t][2] * acc.t[t][2] + acc.t[t][3] * acc.t[t][3];
          for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
          if (lane == 0) atomicAdd(&w.zfro[it + 1], q);
        }
      }
    }
    if (!coop_barrier(w, target, &ok_s, a.aborts)) return;
    nit = it + 1;
  }
  // ---- back end (ns_final_kernel): every participant reaches the same verdict from the same counters
  bool ok = nit >= 1 && __longlong_as_double((long long)__hip_atomic_load(&w.resid[nit - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < NS_TOL;
  if (ok && a.inverse) {
    const unsigned long long zb = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&w.zfro[nit]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = __longlong_as_double((long long)zb) <= a.zmax * a.zmax;   // condition gate (NS_ZMAX)
  }
  if (me == 0 && tid == 0) { *w.iters = nit; *w.ok = ok ? 1 : 0; if (a.info && ok) *a.info = nit; }
  if (!ok) return;
  {
    const __amdgpu_buffer_rsrc_t rf = coop_rsrc(a.inverse ? w.Z[nit & 1] : w.Y[nit & 1]);
    const double sc_out = a.inverse ? rsqrt(s0) : sqrt(s0);
    for (int e = tid; e < 4 * CP; e += 256) {
      const int r = 4 * me + e / CP, c = e % CP;
      if (r >= C || c >= C) continue;
      double v = 0.;
      if (!sdead[r] && !sdead[c]) v = coop_ld1(rf, r * CP + c) * sc_out;
      a.res[eig_F_offset(C) + (size_t)r * C + c] = v;
    }
  }
}

// F = Z / sqrt(s) (inverse) or Y * sqrt(s), dead rows/cols zeroed; ok = converged
__global__ void ns_final_kernel(double* res, int C, int Cp, int inverse, NsWs w, int maxit, double zmax, int* info, const double* deflated,
                                int* ok_defer) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // n = executed iterations; the iterate lives in buffer n & 1.  resid[n-1] was measured on the iterate BEFORE the
  // last executed update, which squares it.
  const int n = *w.iters;
  bool ok = n >= 1 && n <= maxit && __longlong_as_double((long long)w.resid[n - 1]) < NS_TOL;
  if (ok && inverse && !deflated) ok = w.zfro[n] <= zmax * zmax;   // condition gate (NS_ZMAX)
  if (e == 0) { *w.ok = ok ? 1 : 0; if (ok_defer) *ok_defer = ok ? 1 : 0; if (info && ok) *info = n; }
  if (e >= (long)C * C || !ok) return;
  const int r = (int)(e / C), c = (int)(e % C);
  const double s = w.scal[0];
  double v = 0.;
  if (!w.dead[r] && !w.dead[c]) {
    const size_t at = (size_t)r * Cp + c;
    v = (deflated ? deflated[at] : inverse ? w.Z[n & 1][at] : w.Y[n & 1][at]) * (inverse ? rsqrt(s) : sqrt(s));
  }
  res[eig_F_offset(C) + e] = v;
}

// ---- C > 128: deflated iteration.  The wide layers of --mode original meet singular covariances (fewer pixels than
//      channels) and spectra graded over 10+ decades, and a Jacobi sweep there is C - 1 dependent launches (27..68 ms per
//      matrix).  So the iteration runs on B = A + delta I with delta = NS_DEFLATE ||A||_F (cond(B) <= 1e12: <= 41 iterations),
//      and the pseudo-inverse square root of A itself is recovered from Z = B^(-1/2), Y = B^(1/2) with a few more GEMMs:
//          E = delta Z^2 = delta (A + delta)^-1                eigenvalue delta/(lambda+delta): ~0 live, ~1 null
//          P = h^5(I - E),  h(x) = 3x^2 - 2x^3                  0 and 1 are super-attracting: a hard projector onto lambda > delta
//          A^(-1/2) = P Z (I - E)^(-1/2),  A^(1/2) = P Y (I - E)^(1/2)     (Taylor series in E, 5 terms: |E| <= 1e-4 on directions
//                                                                with lambda >= 1e-8 ||A||, exact to round-off there)
//      numpy prototype against eigh + threshold: <= 2e-8 for rank-deficient covariances (n = 16 .. C-1 pixels) and for graded
//      spectra down to 1e-11; directions within a decade of delta get a soft weight instead of the hard cut (no
//      implementation agrees with another there: the reference amplifies its own fp32 conv round-off by 1e6 in them).
constexpr double NS_DEFLATE = 1e-12;

// D = alpha P Q + beta R + gamma I ; optionally D2 = I - D
template <int NS>
__global__ __launch_bounds__(64 * NS) void ns_gemm_kernel(const double* P, const double* Q, double* D, double* D2, int Cp, double alpha,
                                                           const double* R, double beta, double gamma) {
  __shared__ double red[NS - 1][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  Acc32 acc;
  gemm32_splitk<NS>(P, Q, Cp, i0, j0, lane, wave, red, acc);
  if (wave) return;
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * (t >> 1) + kk + 4 * r, col = j0 + 16 * (t & 1) + li;
      const size_t at = (size_t)row * Cp + col;
      double v = alpha * acc.t[t][r] + (row == col ? gamma : 0.0);
      if (R) v += beta * R[at];
      D[at] = v;
      if (D2) D2[at] = (row == col ? 1.0 : 0.0) - v;
    }
}

// the converged iterate into buffer 0 (the post-processing addresses fixed buffers); S1 = a E + g I
__global__ void ns_settle_kernel(NsWs w, int Cp) {
  if (!(*w.iters & 1)) return;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Cp * Cp) return;
  w.Y[0][e] = w.Y[1][e];
  w.Z[0][e] = w.Z[1][e];
}
__global__ void ns_axpi_kernel(const double* E, double* D, int Cp, double a, double g) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Cp * Cp) return;
  D[e] = a * E[e] + ((e / Cp) == (e % Cp) ? g : 0.0);
}

// ---- C <= 64: the whole coupled iteration in ONE workgroup, Y / Z / T resident in LDS (3 x 34 KB at Cp = 64).
//      One launch replaces init + fill + 2 * NS_MAXIT stage launches + final: at these sizes a stage is a few hundred
//      MFMAs, so the multi-launch schedule above is pure launch latency (~5 us per stage, ~250 us per solve), while
//      one CU's fp64 matrix cores need 0.75 us (Cp = 32) / 6 us (Cp = 64) per iteration.
//      One wave per 16x16 output tile; the arithmetic (operands, k order, residual test) is that of the stage kernels.
template <int CP>
__device__ __forceinline__ f64x4 tile_gemm_lds(const double* P, const double* Q, int i0, int j0, int lane) {
  constexpr int LD = CP + 2;   // row stride == 2 (mod 32) doubles: the A-operand reads (16 rows x 4 columns) are bank-conflict free
  const int li = lane & 15, kk = lane >> 4;
  f64x4 acc = f64x4{0., 0., 0., 0.};
  const double* pp = P + (i0 + li) * LD + kk;
  const double* qq = Q + kk * LD + j0 + li;
#pragma unroll 2
  for (int k0 = 0; k0 < CP; k0 += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = pp[k0 + 4 * u]; b[u] = qq[(k0 + 4 * u) * LD]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;  // row = kk + 4 * reg, col = li
}

template <int CP>
__global__ __launch_bounds__((CP / 16) * (CP / 16) * 64) void ns_lds_kernel(double* res, int C, int inverse, double eps_rel,
                                                                               int maxit, double zmax, int* ok_out, int* info,
                                                                               double npix, const double* sum, const double* sumsq, double diag_add,
                                                                               NsSched sched, int nsched) {
  constexpr int LD = CP + 2, TPR = CP / 16, NW = TPR * TPR, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem_ns[];
  double* Y = reinterpret_cast<double*>(smem_ns);
  double* Z = Y + CP * LD;
  double* T = Z + CP * LD;
  double* red = T + CP * LD;                       // [NW] + [1]
  int* dead = reinterpret_cast<int*>(red + NW + 2);  // [CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const int i0 = (wave / TPR) * 16, j0 = (wave % TPR) * 16;
  // covariance, means and eigenvalue floor from the raw moments first (cov_kernel's arithmetic; one launch less per solve)
  cov_by_block(C, npix, sum, sumsq, res, diag_add, tid, NT);
  __syncthreads();
  const double floor_ = res[(size_t)C * C + 2 * C];
  for (int j = tid; j < CP; j += NT) dead[j] = (j >= C) || !(res[(size_t)j * C + j] > floor_);   // padding = identity block too
  __syncthreads();
  // Frobenius norm of the live block
  double sq = 0.;
  for (int e = tid; e < C * C; e += NT) {
    const int r = e / C, c = e - r * C;
    if (!dead[r] && !dead[c]) sq += res[e] * res[e];
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0) red[wave] = sq;
  __syncthreads();
  double f = 0.;
  for (int w = 0; w < NW; ++w) f += red[w];
  f = sqrt(f);
  if (!(f > 0.)) f = 1.;                 // all channels dead: Y0 = I, result zeroed anyway
  const double s = f * (1.0 + eps_rel), shift = eps_rel * f;   // spectrum of (A + eps f I)/s inside (0, 1]
  for (int e = tid; e < CP * CP; e += NT) {
    const int r = e / CP, c = e - r * CP;
    double y = r == c ? 1.0 : 0.0;
    if (!dead[r] && !dead[c]) y = (res[(size_t)r * C + c] + (r == c ? shift : 0.0)) / s;
    Y[r * LD + c] = y;
    Z[r * LD + c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  int n = 0;
  double prev = 1e300;   // residual measured before the last executed update
  for (int it = 0; it < maxit; ++it) {
    if (it > 0 && prev < NS_TOL) break;
    // stage 1: T = ca I - cb Z Y (ca = 1.5, cb = 0.5 unscaled; the scaled schedule of launch_eig for the first `nsched` steps), residual = max |Z Y - I|
    const double ca = it < nsched ? sched.ca[it] : 1.5, cb = it < nsched ? sched.cb[it] : 0.5;
    const f64x4 zy = tile_gemm_lds<CP>(Z, Y, i0, j0, lane);
    double m = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + kk + 4 * r, col = j0 + li;
      const double d = zy[r] - (row == col ? 1.0 : 0.0);
      m = (d == d) ? fmax(m, fabs(d)) : __longlong_as_double(0x7ff0000000000000ll);  // fmax would swallow a NaN
      T[row * LD + col] = (row == col ? ca : 0.0) - cb * zy[r];
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = 0.;
    for (int w = 0; w < NW; ++w) m = fmax(m, red[w]);   // +inf (from a NaN) survives fmax
    prev = m;
    // stage 2: Y <- Y T, Z <- T Z (tiles held in registers until every wave has read the old iterates)
    const f64x4 yn = tile_gemm_lds<CP>(Y, T, i0, j0, lane);
    const f64x4 zn = tile_gemm_lds<CP>(T, Z, i0, j0, lane);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Y[(i0 + kk + 4 * r) * LD + j0 + li] = yn[r];
      Z[(i0 + kk + 4 * r) * LD + j0 + li] = zn[r];
    }
    __syncthreads();
    n = it + 1;
  }
  bool ok = n >= 1 && prev < NS_TOL;
  if (ok && inverse) {   // condition gate (NS_ZMAX)
    double q = 0.;
    for (int e = tid; e < CP * CP; e += NT) { const double z = Z[(e / CP) * LD + (e % CP)]; q += z * z; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    __syncthreads();
    if (lane == 0) red[wave] = q;
    __syncthreads();
    q = 0.;
    for (int w = 0; w < NW; ++w) q += red[w];
    ok = q <= zmax * zmax;
  }
  if (tid == 0) { *ok_out = ok ? 1 : 0; if (info && ok) *info = n; }
  if (!ok) return;
  const double sc = inverse ? rsqrt(s) : sqrt(s);
  const double* src = inverse ? Z : Y;
  for (int e = tid; e < C * C; e += NT) {
    const int r = e / C, c = e - r * C;
    res[eig_F_offset(C) + e] = (!dead[r] && !dead[c]) ? src[r * LD + c] * sc : 0.;
  }
}

// ---- all-reduce of an fp64 value over the LPP (4, 8 or 16) lanes of a pair with DPP moves (no LDS round trips):
//      quad_perm[1,0,3,2], quad_perm[2,3,0,1] -> quad sums; row_half_mirror -> 8-lane sums; row_mirror -> 16.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int LPP>
__device__ __forceinline__ double reduce_pair(double v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  if constexpr (LPP >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if constexpr (LPP >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// ---- C <= 128: whole problem in LDS, one workgroup per matrix.
//  * channels whose variance is at round-off level (exactly-dead ReLU channels: 29..77 of 128 at relu5_1/relu4_1,
//    SURVEY 7) are compacted away first -- their rows/columns of cov are zero, so they are eigenvectors with
//    lambda = 0 and the Jacobi problem shrinks to the live block;
//  * column norms are cached in LDS and updated by the rotation (alpha' = alpha - t*gamma, beta' = beta + t*gamma);
//    round 0 of every sweep recomputes them exactly, so only ONE dot product per pair is reduced per round;
//  * the kernel is bound by fp64 VALU issue on ONE CU: LPP lanes share a pair (each lane owns 2-row groups read as
//    ds_read_b128), so the per-pair rotation arithmetic is amortised over 64/LPP pairs per wave instruction.
template <int LPP>
__global__ __launch_bounds__(64 * LPP) void jacobi_lds_kernel(double* res, int n_full, int* info, const int* ns_ok, double expo, double rel_thresh, unsigned* coop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the Newton-Schulz path converged: nothing to do (uniform branch, before any barrier).  `coop` (the single-launch iteration's
  // state, or null): an aborted iteration never wrote `ok`, so its abort flag counts as "not converged"
  if (*ns_ok != 0 && (!coop || coop[1] == 0u)) return;
  __builtin_amdgcn_s_setprio(3);  // latency-critical single-CU kernel: win issue arbitration against co-resident conv waves
  const int tid = threadIdx.x;
  double* Gg = res;
  double* lamg = res + (size_t)n_full * n_full;
  const double floor_ = res[(size_t)n_full * n_full + 2 * n_full];
  // carve: G [n_full][LD] | norm2 [n_full] | live [n_full] ints | flags.  LD: rows padded to a multiple of 32 (whole
  // 2*LPP-row groups are read; rows >= n hold zeros and stay zero under rotations) + 2 (bank spread)
  const int LD = (n_full + 31) / 32 * 32 + 2;
  double* G = reinterpret_cast<double*>(smem);
  double* norm2 = G + (size_t)n_full * LD;
  int* live = reinterpret_cast<int*>(norm2 + n_full);
  volatile int* rotated = live + n_full;      // [0] rotation flag, [1] n (live count, even)
  if (tid == 0) {
    int nl = 0;
    for (int j = 0; j < n_full; ++j)
      if (Gg[(size_t)j * n_full + j] > floor_) live[nl++] = j;
    if (nl & 1) {  // the tournament needs an even count: add one dead channel back (a zero row/column)
      for (int j = 0; j < n_full; ++j)
        if (!(Gg[(size_t)j * n_full + j] > floor_)) { live[nl++] = j; break; }
    }
    rotated[0] = 0;
    rotated[1] = nl;
  }
  __syncthreads();
  const int n = rotated[1];
  for (int e = tid; e < n_full * LD; e += blockDim.x) G[e] = 0.;   // rows >= n of a column read as 0 below
  __syncthreads();
  for (int e = tid; e < n * n; e += blockDim.x) {
    const int cj = e / n, r = e - cj * n;
    G[cj * LD + r] = Gg[(size_t)live[cj] * n_full + live[r]];
  }
  __syncthreads();
  const int npairs = n >> 1, m_ = n - 1;
  const int pair = tid / LPP, sub = tid % LPP;
  const bool active = pair < npairs;
  constexpr int MAXG = 128 / (2 * LPP);        // 2-row groups per lane
  const int ngroups = (n + 2 * LPP - 1) / (2 * LPP);
  int sweep = 0;
  for (; sweep < MAX_SWEEPS && n >= 2; ++sweep) {
    // circle method: player n-1 stays, the others rotate; positions advance by one each round (no modulo)
    int pa = pair % (m_ > 0 ? m_ : 1), pb = (m_ - pair) % (m_ > 0 ? m_ : 1);
    for (int round = 0; round < n - 1; ++round) {
      if (active) {
        int p = pair == 0 ? m_ : pa, q = pair == 0 ? pa : pb;  // pair 0: (n-1, round mod m)
        if (p > q) { const int t_ = p; p = q; q = t_; }
        const double* colp = G + p * LD + 2 * sub;
        const double* colq = G + q * LD + 2 * sub;
        f64x2 gp[MAXG], gq[MAXG];
        double ga0 = 0., ga1 = 0.;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
          if (g < ngroups) {
            gp[g] = *reinterpret_cast<const f64x2*>(colp + 2 * LPP * g);
            gq[g] = *reinterpret_cast<const f64x2*>(colq + 2 * LPP * g);
            ga0 += gp[g][0] * gq[g][0]; ga1 += gp[g][1] * gq[g][1];
          }
        }
        double al, be;
        if (round == 0) {
          double a0 = 0., a1 = 0., b0 = 0., b1 = 0.;
#pragma unroll
          for (int g = 0; g < MAXG; ++g) {
            if (g < ngroups) {
              a0 += gp[g][0] * gp[g][0]; a1 += gp[g][1] * gp[g][1];
              b0 += gq[g][0] * gq[g][0]; b1 += gq[g][1] * gq[g][1];
            }
          }
          al = reduce_pair<LPP>(a0 + a1); be = reduce_pair<LPP>(b0 + b1);
        } else {
          al = norm2[p]; be = norm2[q];
        }
        const double ga = reduce_pair<LPP>(ga0 + ga1);
        bool rot = false;
        double t = 0.;
        if (ga * ga > (ROT_TOL * ROT_TOL) * al * be && al * be > 0.) {
          double c, s;
          t = rotation(al, be, ga, c, s);
          double* wp = G + p * LD + 2 * sub;
          double* wq = G + q * LD + 2 * sub;
#pragma unroll
          for (int g = 0; g < MAXG; ++g) {
            if (g < ngroups) {
              f64x2 np_, nq_;
              np_[0] = c * gp[g][0] - s * gq[g][0]; np_[1] = c * gp[g][1] - s * gq[g][1];
              nq_[0] = s * gp[g][0] + c * gq[g][0]; nq_[1] = s * gp[g][1] + c * gq[g][1];
              *reinterpret_cast<f64x2*>(wp + 2 * LPP * g) = np_;
              *reinterpret_cast<f64x2*>(wq + 2 * LPP * g) = nq_;
            }
          }
          rot = true;
        }
        if (sub == 0) {
          if (rot) rotated[0] = 1;
          if (rot || round == 0) { norm2[p] = al - t * ga; norm2[q] = be + t * ga; }
        }
      }
      pa = pa + 1 == m_ ? 0 : pa + 1;
      pb = pb + 1 == m_ ? 0 : pb + 1;
      __syncthreads();
    }
    const int any = rotated[0];
    __syncthreads();
    if (tid == 0) rotated[0] = 0;
    __syncthreads();
    if (!any) break;
  }
  // eigenvalues = column norms (recomputed exactly); G = V diag(lambda) written back in the FULL index space
  for (int e = tid; e < n_full * n_full; e += blockDim.x) Gg[e] = 0.;
  for (int j = tid; j < n_full; j += blockDim.x) lamg[j] = 0.;
  __syncthreads();
  for (int j = tid; j < n; j += blockDim.x) {
    double s = 0.;
    for (int r = 0; r < n; ++r) s += G[j * LD + r] * G[j * LD + r];
    lamg[j] = sqrt(s);
  }
  for (int e = tid; e < n * n; e += blockDim.x) {
    const int cj = e / n, r = e - cj * n;
    Gg[(size_t)cj * n_full + live[r]] = G[cj * LD + r];  // column-major: column cj, full row index live[r]
  }
  if (tid == 0 && info) *info = 100 + sweep;   // 100 + sweeps: the Jacobi fallback ran
  // F = SUM_{j live} lambda_j^(expo-2) G[:,j] G[:,j]^T -- sym_power_kernel's arithmetic and summation order, by this one workgroup
  // (the fallback is the rare path: its second gated launch cost every solve 4 us + a launch gap for nothing)
  __syncthreads();                    // G (global, column-major) and lambda as written above are visible to the workgroup
  double* wj = G;                      // the LDS copy is no longer needed
  double lmax = 0.;
  for (int j = 0; j < n_full; ++j) lmax = fmax(lmax, lamg[j]);
  const double thr = fmax(rel_thresh * lmax, floor_);
  for (int j = tid; j < n_full; j += blockDim.x) {
    const double l = lamg[j];
    wj[j] = (l > thr && l > 0.) ? pow(l, expo - 2.0) : 0.;
  }
  __syncthreads();
  double* out = res + eig_F_offset(n_full);
  for (int e = tid; e < n_full * n_full; e += blockDim.x) {
    const int a = e / n_full, b = e - a * n_full;
    double sacc = 0.;
    for (int j = 0; j < n_full; ++j) sacc += wj[j] * Gg[(size_t)j * n_full + a] * Gg[(size_t)j * n_full + b];
    out[e] = sacc;
  }
}

// ---- C > 128: columns stay in global memory (L2 resident); one launch per tournament round, one wave per pair.
//      flags[sweep] is raised when any rotation happened in that sweep; rounds of sweep s+1 return at once when
//      flags[s] == 0, so a fixed launch schedule needs no host round trip.
__global__ __launch_bounds__(256) void jacobi_round_global_kernel(double* G, int n, int round, int sweep, int* fl) {
  if (sweep > 0 && fl[sweep - 1] == 0) return;
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (n >> 1)) return;
  int p, q;
  tournament_pair(n, round, pair, p, q);
  constexpr int MAXR = 512 / 64;
  double gp[MAXR], gq[MAXR];
  double al = 0., be = 0., ga = 0.;
#pragma unroll
  for (int m = 0; m < MAXR; ++m) {
    const int r = lane + 64 * m;
    gp[m] = r < n ? G[(size_t)p * n + r] : 0.;
    gq[m] = r < n ? G[(size_t)q * n + r] : 0.;
    al += gp[m] * gp[m]; be += gq[m] * gq[m]; ga += gp[m] * gq[m];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { al += __shfl_xor(al, o); be += __shfl_xor(be, o); ga += __shfl_xor(ga, o); }
  if (ga * ga > (ROT_TOL * ROT_TOL) * al * be && al * be > 0.) {
    double c, s;
    rotation(al, be, ga, c, s);
#pragma unroll
    for (int m = 0; m < MAXR; ++m) {
      const int r = lane + 64 * m;
      if (r < n) {
        G[(size_t)p * n + r] = c * gp[m] - s * gq[m];
        G[(size_t)q * n + r] = s * gp[m] + c * gq[m];
      }
    }
    if (lane == 0) fl[sweep] = 1;
  }
}

__global__ void colnorm_kernel(double* res, int n, const int* fl, int* info) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    const double* g = res + (size_t)j * n;
    double s = 0.;
    for (int r = 0; r < n; ++r) s += g[r] * g[r];
    res[(size_t)n * n + j] = sqrt(s);
  }
  if (j == 0 && info) {
    int sw = 0;
    while (sw < MAX_SWEEPS && fl[sw]) ++sw;
    *info = 100 + sw;   // 100 + sweeps, like the LDS kernel
  }
}

// out[a][b] = sum_{j live} lambda_j^(expo-2) G[a,j] G[b,j]   (G = V diag(lambda), column-major)
__global__ __launch_bounds__(256) void sym_power_kernel(double* res, int n, double expo, double rel_thresh, const int* ns_ok) {
  if (ns_ok && *ns_ok) return;
  double* out = res + eig_F_offset(n);
  __shared__ double wj[512];
  const double* G = res;
  const double* lam = res + (size_t)n * n;
  const double floor_ = res[(size_t)n * n + 2 * n];
  double lmax = 0.;
  for (int j = 0; j < n; ++j) lmax = fmax(lmax, lam[j]);
  const double thr = fmax(rel_thresh * lmax, floor_);
  for (int j = threadIdx.x; j < n; j += 256) {
    const double l = lam[j];
    wj[j] = (l > thr && l > 0.) ? pow(l, expo - 2.0) : 0.;
  }
  __syncthreads();
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)n * n) return;
  const int a = (int)(e / n), b = (int)(e % n);
  double s = 0.;
  for (int j = 0; j < n; ++j) s += wj[j] * G[(size_t)j * n + a] * G[(size_t)j * n + b];
  out[e] = s;
}

}  // namespace

size_t eig_result_bytes(int C) { return eig_doubles(C) * sizeof(double); }
size_t eig_result_F_offset(size_t C) { return eig_F_offset((int)C); }
static inline int ns_pad(int C) { return C > 128 ? (C + 63) / 64 * 64 : (C + 31) / 32 * 32; }   // split-k: k quarters in steps of 16
size_t eig_workspace_bytes(int C) {
  const size_t cp2 = (size_t)ns_pad(C) * ns_pad(C);
  return 5 * cp2 * sizeof(double) + (size_t)C * sizeof(int) + 4 * sizeof(double) + 2 * (NS_MAXIT_REG + 2) * sizeof(unsigned long long) + 64;
}
size_t assemble_workspace_bytes(int C) { return (size_t)C * C * sizeof(double); }

bool eig_is_big(int C, bool wide_model) { return C > 128 || (wide_model && C > 64 && ns_pad(C) % 64 == 0); }

bool fold_gemm_capable(int cout, int cin, int cout_pad) { return cin >= 256 && cin % 128 == 0 && cout % 32 == 0 && cout_pad == cout; }
size_t fold_gemm_rows_doubles(int cout, int cin) { return (size_t)cout * 10 * cin; }

hipError_t launch_fold_rows(const float* w_oihw, int cout, int cin, double* rows, hipStream_t s) {
  const long n = (long)cout * cin;
  hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_oihw, cout, cin, rows, rows + (size_t)cout * 9 * cin);
  return hipGetLastError();
}

hipError_t launch_fold_gemm(const double* rows, const float* bias, int cout, int cin, int cout_pad, const double* M, const double* b,
                            float* wpk_out, float* bias_out, unsigned* maxbits_dev, hipStream_t s) {
  if (!fold_gemm_capable(cout, cin, cout_pad)) return hipErrorInvalidValue;
  if (maxbits_dev) {
    hipError_t e = hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
  }
  constexpr int NS = 8;
  const dim3 grid((unsigned)(cin / 32), (unsigned)(cout * 9 / 32 + (cout_pad + NS - 1) / NS));
  hipLaunchKernelGGL(fold_gemm_kernel<NS>, grid, dim3(64 * NS), 0, s, rows, rows + (size_t)cout * 9 * cin, bias, cout, cin, cout_pad, M, b, wpk_out,
                     bias_out, maxbits_dev);
  return hipGetLastError();
}

hipError_t launch_eig(int C, double n, const double* sum, const double* sumsq, int inverse, double* res, int* info_dev,
                      void* ws, size_t ws_bytes, hipStream_t s, double diag_add, bool wide_model, int* ok_defer, int coop_xcd, unsigned* coop_state, int* coop_epoch,
                      unsigned* coop_aborts, bool* coop_used_out) {
  if (C < 2 || (C & 1) || C > 512 || n < 2) return hipErrorInvalidValue;  // unbiased covariance needs n >= 2
  if (ws_bytes < eig_workspace_bytes(C)) return hipErrorOutOfMemory;
  const size_t cc = (size_t)C * C;
  const int Cp = ns_pad(C);
  const size_t cp2 = (size_t)Cp * Cp;
  NsWs w;
  double* p = reinterpret_cast<double*>(ws);
  w.Y[0] = p; w.Y[1] = p + cp2; w.Z[0] = p + 2 * cp2; w.Z[1] = p + 3 * cp2; w.T = p + 4 * cp2;
  w.scal = p + 5 * cp2;
  w.resid = reinterpret_cast<unsigned long long*>(w.scal + 4);
  w.zfro = reinterpret_cast<double*>(w.resid + NS_MAXIT_REG + 2);
  w.iters = reinterpret_cast<int*>(w.zfro + NS_MAXIT_REG + 2);
  w.ok = w.iters + 1;
  w.coop = nullptr;
  w.dead = w.iters + 2;
  const bool big = eig_is_big(C, wide_model);   // deflated, scaled iteration + host check of the outcome (or the caller's: ok_defer)
  // the covariance: its own grid-wide launch only where one workgroup would be too slow (C > 128); the single-workgroup front
  // ends below (ns_lds_kernel, ns_prep_kernel) start from the raw moments themselves
  if (big) hipLaunchKernelGGL(cov_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, C, n, sum, sumsq, res, diag_add);
  static const int maxit_env = [] { const char* e = wct_debug_env("WCT_NS_MAXIT"); const int v = e ? atoi(e) : 0; return v < 1 ? 0 : (v > NS_MAXIT_REG ? NS_MAXIT_REG : v); }();
  // 64 < Cp <= 128 without deflation (the 128-channel levels of --mode 16x): the iteration is scaled with an ASSUMED lower
  // spectral bound 1e-5 (see the schedule below) -- a wrong guess costs iterations, never correctness (only the upper bound 1
  // matters for safety, eigenvalues below the guess still grow 2.6x per scaled step) -- which reaches cond ~1e7 in 16
  // iterations like the plain iteration in 26: 20 fewer always-enqueued stage launches per solve, 13/10 -> 11/11 executed
  // iterations on the 4K bench frame, 1.4 % of a cached-style frame (A/B on one box, tools/experiments/ns_guess.py).
  static const double guess_env = [] { const char* e = wct_debug_env("WCT_NS_GUESS"); return e ? atof(e) : -1.0; }();
  const double guess = big || Cp <= 64 ? 0.0 : (guess_env >= 0. ? guess_env : 1e-5);
  // Cp <= 64 (round 4): the same scaled start inside the LDS kernel -- 16 / 18 / 15 plain iterations at levels 3 / 2 / 1 of the 4K bench
  // frame.  The schedule covers its first COOP_MAXIT steps; beyond them (a spectrum far below the guess) the plain iteration goes on
  // up to the unchanged budget NS_MAXIT, so a wrong guess still costs iterations only.
  static const double guess64_env = [] { const char* e = wct_debug_env("WCT_NS_GUESS64"); return e ? atof(e) : -1.0; }();
  const double guess64 = guess64_env >= 0. ? guess64_env : 1e-5;
  // C > 128 (original mode): the deflated, optimally scaled iteration takes 19-20 iterations whatever the matrix
  const int maxit = maxit_env ? maxit_env : (C > 128 ? 24 : (guess > 0. ? 16 : NS_MAXIT));
  bool coop_used = false;
  if (Cp <= 64) {
    // one workgroup, iterates in LDS (see ns_lds_kernel)
    auto go = [&](auto kern, int cp) -> hipError_t {
      const int nw = (cp / 16) * (cp / 16);
      const size_t lds = (size_t)3 * cp * (cp + 2) * sizeof(double) + (size_t)(nw + 2) * sizeof(double) + (size_t)cp * sizeof(int);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      NsSched sc;
      int nsched = 0;
      if (guess64 > 0.) {
        double xl = sqrt(guess64);
        for (; nsched < COOP_MAXIT && nsched < maxit; ++nsched) {
          const double mu = xl <= 0.9 ? sqrt(3.0 / (1.0 + xl + xl * xl)) : 1.0;
          xl = mu * xl * (3.0 - mu * mu * xl * xl) / 2.0;
          sc.ca[nsched] = 1.5 * mu; sc.cb[nsched] = 0.5 * mu * mu * mu;
        }
      }
      hipLaunchKernelGGL(kern, dim3(1), dim3((unsigned)nw * 64), lds, s, res, C, inverse, 1e-15, maxit, NS_ZMAX, w.ok, info_dev, n, sum, sumsq, diag_add, sc, nsched);
      return hipSuccess;
    };
    hipError_t e = Cp == 32 ? go(ns_lds_kernel<32>, 32) : go(ns_lds_kernel<64>, 64);
    if (e != hipSuccess) return e;
  } else {
    static const bool sk_env = [] { const char* e = wct_debug_env("WCT_NS_SPLITK"); return e ? atoi(e) != 0 : true; }();
    const bool splitk128 = sk_env && Cp % 64 == 0;
    const bool coop = !big && Cp == 128 && splitk128 && coop_xcd >= 0 && coop_state && coop_epoch && maxit <= COOP_MAXIT;
    coop_used = coop;
    if (coop_used_out) *coop_used_out = coop;
    if (coop) { w.coop = coop_state + 4 * (*coop_epoch & 1); ++*coop_epoch; }
    if (big) {
      hipLaunchKernelGGL(ns_init_kernel, dim3(1), dim3(1024), 0, s, res, C, NS_DEFLATE, w, maxit);
      hipLaunchKernelGGL(ns_fill_kernel, dim3((unsigned)((cp2 + 255) / 256)), dim3(256), 0, s, res, C, Cp, w);
    } else if (!coop) {
      hipLaunchKernelGGL(ns_prep_kernel, dim3(1), dim3(1024), 0, s, C, Cp, n, sum, sumsq, res, diag_add, 1e-15, w, maxit);
    }
    const dim3 g1(Cp / 32, Cp / 32, 1), g2(Cp / 32, Cp / 32, 2);
    // Deflated problem: every eigenvalue of Z Y = B/s starts in [l0, 1] with l0 = delta/s KNOWN, so the iteration can be
    // scaled optimally (T = mu (3I - mu^2 Z Y)/2 with mu^2 = 3/(1 + x + x^2), x = sqrt of the current lower bound: the cubic
    // then maps both ends of [x, 1] to the same value) -- small eigenvalues grow 6.75x per step instead of 2.25x and the
    // whole spectrum arrives together: 19 iterations for ANY matrix instead of 15 (cond 1e3) .. 39 (singular).  The schedule
    // depends on l0 alone, so it is computed here; numpy prototype: same accuracy as the plain iteration.
    double xlow = big ? sqrt(NS_DEFLATE / (1.0 + NS_DEFLATE)) : (guess > 0. ? sqrt(guess) : 1.0);
    if (coop) {   // front end, the whole iteration and the scaled result in one launch (ns_coop128_kernel); same schedule, same arithmetic
      NsSched sc;
      for (int it = 0; it < maxit; ++it) {
        const double mu = xlow <= 0.9 ? sqrt(3.0 / (1.0 + xlow + xlow * xlow)) : 1.0;
        xlow = mu * xlow * (3.0 - mu * mu * xlow * xlow) / 2.0;
        sc.ca[it] = 1.5 * mu; sc.cb[it] = 0.5 * mu * mu * mu;
      }
      CoopArgs ca;
      ca.C = C; ca.n = n; ca.sum = sum; ca.sumsq = sumsq; ca.res = res; ca.diag_add = diag_add; ca.eps_rel = 1e-15;
      ca.w = w; ca.maxit = maxit; ca.inverse = inverse; ca.zmax = NS_ZMAX; ca.info = info_dev; ca.xcd = coop_xcd & 23;
      ca.coop_next = coop_state + 4 * (*coop_epoch & 1);     // (the epoch was advanced above: this is the other set)
      ca.aborts = coop_aborts;
      hipLaunchKernelGGL(ns_coop128_kernel, dim3(8 * COOP_NW), dim3(256), 0, s, ca, sc);
    }
    for (int it = 0; it < (coop ? 0 : maxit); ++it) {
      if (big) {
        const double mu = xlow <= 0.9 ? sqrt(3.0 / (1.0 + xlow + xlow * xlow)) : 1.0;
        xlow = mu * xlow * (3.0 - mu * mu * xlow * xlow) / 2.0;
        if (Cp >= 256 && Cp % 128 == 0) {
          hipLaunchKernelGGL(ns_stage1_wide_kernel<8>, g1, dim3(512), 0, s, w, Cp, it, 1.5 * mu, 0.5 * mu * mu * mu);
          hipLaunchKernelGGL(ns_stage2_wide_kernel<8>, g2, dim3(512), 0, s, w, Cp, it);
        } else {
          hipLaunchKernelGGL(ns_stage1_wide_kernel<4>, g1, dim3(256), 0, s, w, Cp, it, 1.5 * mu, 0.5 * mu * mu * mu);
          hipLaunchKernelGGL(ns_stage2_wide_kernel<4>, g2, dim3(256), 0, s, w, Cp, it);
        }
      } else {
        const double mu = xlow <= 0.9 ? sqrt(3.0 / (1.0 + xlow + xlow * xlow)) : 1.0;
        xlow = mu * xlow * (3.0 - mu * mu * xlow * xlow) / 2.0;
        const double ca = 1.5 * mu, cb = 0.5 * mu * mu * mu;
        if (splitk128) {   // Cp = 128, adaptive iteration on the split-k tiles (2 k-blocks per wave instead of 8)
          hipLaunchKernelGGL(ns_stage1_wide_kernel<4>, g1, dim3(256), 0, s, w, Cp, it, ca, cb);
          hipLaunchKernelGGL(ns_stage2_wide_kernel<4>, g2, dim3(256), 0, s, w, Cp, it);
        } else {
          hipLaunchKernelGGL(ns_stage1_kernel, g1, dim3(256), 0, s, w, Cp, it, ca, cb);
          hipLaunchKernelGGL(ns_stage2_kernel, g2, dim3(256), 0, s, w, Cp, it);
        }
      }
    }
    const double* deflated = nullptr;
    if (big) {
      const unsigned eb = (unsigned)((cp2 + 255) / 256);
      const double de = NS_DEFLATE / (1.0 + NS_DEFLATE);   // delta / s in the iteration's units
      auto gemm = [&](const double* P, const double* Q, double* D, double* D2, double alpha, const double* R, double beta, double gamma) {
        if (Cp >= 256 && Cp % 128 == 0) hipLaunchKernelGGL(ns_gemm_kernel<8>, g1, dim3(512), 0, s, P, Q, D, D2, Cp, alpha, R, beta, gamma);
        else hipLaunchKernelGGL(ns_gemm_kernel<4>, g1, dim3(256), 0, s, P, Q, D, D2, Cp, alpha, R, beta, gamma);
      };
      hipLaunchKernelGGL(ns_settle_kernel, dim3(eb), dim3(256), 0, s, w, Cp);
      const double* Rm = inverse ? w.Z[0] : w.Y[0];
      double* E = w.Y[1];
      double* wb[3] = {inverse ? w.Y[0] : w.Z[1], inverse ? w.Z[1] : w.T, inverse ? w.T : w.Z[0]};   // Z[0] is free once E exists
      gemm(w.Z[0], w.Z[0], E, wb[0], de, nullptr, 0., 0.);            // E = delta Z^2, P = I - E
      int ip = 0;
      for (int h = 0; h < 5; ++h) {                                    // P <- 3 P^2 - 2 P^3
        const int i2 = (ip + 1) % 3, in = (ip + 2) % 3;
        gemm(wb[ip], wb[ip], wb[i2], nullptr, 1., nullptr, 0., 0.);
        gemm(wb[i2], wb[ip], wb[in], nullptr, -2., wb[i2], 3., 0.);
        ip = in;
      }
      double* Sa = wb[(ip + 1) % 3];
      double* Sb = wb[(ip + 2) % 3];
      // (1 - x)^(-1/2) = 1 + x/2 + 3x^2/8 + 5x^3/16 + 35x^4/128 + 63x^5/256 ; (1 - x)^(1/2) = 1 - x/2 - x^2/8 - x^3/16 - 5x^4/128 - 7x^5/256
      static const double ci[6] = {1., 0.5, 3. / 8, 5. / 16, 35. / 128, 63. / 256}, cs[6] = {1., -0.5, -1. / 8, -1. / 16, -5. / 128, -7. / 256};
      const double* cf = inverse ? ci : cs;
      hipLaunchKernelGGL(ns_axpi_kernel, dim3(eb), dim3(256), 0, s, (const double*)E, Sa, Cp, cf[5], cf[4]);
      for (int k = 3; k >= 0; --k) {                                   // Horner: S <- S E + c_k I
        gemm(Sa, E, Sb, nullptr, 1., nullptr, 0., cf[k]);
        double* t = Sa; Sa = Sb; Sb = t;
      }
      gemm(Rm, Sa, Sb, nullptr, 1., nullptr, 0., 0.);                 // R S
      gemm(wb[ip], Sb, Sa, nullptr, 1., nullptr, 0., 0.);             // P R S
      deflated = Sa;
    }
    if (!coop)
      hipLaunchKernelGGL(ns_final_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, res, C, Cp, inverse, w, maxit, NS_ZMAX, info_dev, deflated,
                         big ? ok_defer : nullptr);
  }
  if (big && ok_defer) return hipGetLastError();   // the caller reads *ok_defer after ITS synchronisation point and re-runs without deferral if it is 0
  if (big) {
    // C > 128 has no single-CU Jacobi: the deflated iteration above covers singular and ill-conditioned covariances, and the
    // slow global-memory Jacobi (one launch per tournament round) is the net under it -- the outcome of the iteration is read
    // back (one 4-byte copy + stream sync per solve, original mode only) and Jacobi runs only if it did not converge.
    int ok_host = 0;
    hipError_t e = hipMemcpyAsync(&ok_host, w.ok, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (!ok_host) {
      int* flags = reinterpret_cast<int*>(w.Y[0]);  // the iteration's buffers are free now
      e = hipMemsetAsync(flags, 0, (MAX_SWEEPS + 1) * sizeof(int), s);
      if (e != hipSuccess) return e;
      // a sweep is C - 1 launches (~1.7 ms at C = 512), so the host looks at the sweep's flag before queueing the next one:
      // well-separated spectra stop after 7-10 sweeps, graded ones spanning > 12 decades need 30+ (16 fixed sweeps used to
      // leave those unconverged).
      const dim3 grid((unsigned)((C / 2 + 3) / 4));
      for (int sw = 0; sw < MAX_SWEEPS; ++sw) {
        for (int r = 0; r < C - 1; ++r)
          hipLaunchKernelGGL(jacobi_round_global_kernel, grid, dim3(256), 0, s, res, C, r, sw, flags);
        int rotated = 0;
        e = hipMemcpyAsync(&rotated, flags + sw, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        if (!rotated) break;
      }
      hipLaunchKernelGGL(colnorm_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, res, C, flags, info_dev);
      hipLaunchKernelGGL(sym_power_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, res, C, inverse ? -0.5 : 0.5, REL_THRESH,
                         (const int*)nullptr);
    }
  } else {
    const size_t lds = ((size_t)C * ((C + 31) / 32 * 32 + 2) + C) * sizeof(double) + (size_t)(C + 4) * sizeof(int);
    static int lpp = [] { const char* e = wct_debug_env("WCT_JACOBI_LPP"); return e ? atoi(e) : 16; }();
    auto go = [&](auto kern, int LPPv) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      const unsigned threads = (unsigned)(((C / 2) * LPPv + 63) / 64 * 64);
      hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, s, res, C, info_dev, (const int*)w.ok, inverse ? -0.5 : 0.5, REL_THRESH,
                         coop_used ? w.coop : (unsigned*)nullptr);
      return hipSuccess;
    };
    // ONE gated launch: Jacobi and the symmetric power of its result (it returns at once when the iteration converged)
    hipError_t e = lpp == 16 ? go(jacobi_lds_kernel<16>, 16) : lpp == 8 ? go(jacobi_lds_kernel<8>, 8) : go(jacobi_lds_kernel<4>, 4);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

namespace {
// one workgroup per row a:  T[a][:] = Ss[a][:] Wc,  M[a][:] = alpha T[a][:] + (1 - alpha) I[a][:],
// b[a] = alpha (mu_s[a] - T[a][:] mu_c)   -- the whole assembly in ONE launch (was: element-wise matmul + bias kernel)
__global__ __launch_bounds__(256) void assemble_row_kernel(int C, double alpha, const double* Ss, const double* Wc, const double* mu_c,
                                                             const double* mu_s, double* T, double* M, double* bvec) {
  __shared__ double srow[512];
  __shared__ double red[256];
  const int a = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < C; k += 256) srow[k] = Ss[(size_t)a * C + k];
  __syncthreads();
  double dot = 0.;
  for (int b0 = tid; b0 < C; b0 += 256) {
    double s = 0.;
    for (int k = 0; k < C; ++k) s += srow[k] * Wc[(size_t)k * C + b0];
    T[(size_t)a * C + b0] = s;
    M[(size_t)a * C + b0] = alpha * s + (a == b0 ? 1.0 - alpha : 0.0);
    dot += s * mu_c[b0];
  }
  red[tid] = dot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) bvec[a] = alpha * (mu_s[a] - red[0]);
}
}  // namespace

hipError_t launch_assemble(int C, const double* eig_c, const double* eig_s, double alpha, double rel_thresh, double* M,
                           double* b, void* ws, size_t ws_bytes, hipStream_t s) {
  (void)rel_thresh;
  if (ws_bytes < assemble_workspace_bytes(C)) return hipErrorOutOfMemory;
  const size_t cc = (size_t)C * C;
  double* T = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(assemble_row_kernel, dim3((unsigned)C), dim3(256), 0, s, C, alpha, eig_s + eig_F_offset(C), eig_c + eig_F_offset(C),
                     eig_c + cc + C, eig_s + cc + C, T, M, b);
  return hipGetLastError();
}
