// Probe for a SINGLE-LAUNCH coupled Newton-Schulz iteration at Cp = 128 (solve.hip's multi-launch schedule is 2 launches per
// iteration at ~5.5 us each, almost all of it kernel-boundary latency: launch + cold L2 after the boundary's cache maintenance).
// Here the 32 workgroups the dispatcher places on ONE XCD (workgroup b -> XCD b mod 8, xcd_probe.hip) iterate inside one
// kernel: software barrier on an agent-scope counter in that XCD's L2, iterates exchanged through global memory with sc1
// (agent-scope: L1-bypassing) buffer loads / stores -- proper vector memory instructions the compiler can pipeline, unlike
// __hip_atomic_load, which it serialises with a wait after every load (round 1's attempt: 12 k cycles per tile product).
//   hipcc --offload-arch=gfx950 -O3 -o ns_coop_probe ns_coop_probe.hip && ./ns_coop_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CP = 128, MAXIT = 16;
constexpr double NS_TOL = 1e-7;
struct Ws { double *Y[2], *Z[2], *T; unsigned long long* resid; int* iters; unsigned* bar; unsigned* abort_; unsigned* xcc; };
struct Sched { double ca[MAXIT], cb[MAXIT]; };
struct Acc32 { f64x4 t[4]; };

// ------------------------------------------------------------------------------------------------ multi-launch baseline
__device__ __forceinline__ void gemm32(const double* P, const double* Q, int i0, int j0, int lane, int wave, double (*red)[16 * 64], Acc32& acc) {
  const int li = lane & 15, kk = lane >> 4;
  const int kq = CP >> 2, kbeg = wave * kq, kend = kbeg + kq;
  for (int t = 0; t < 4; ++t) acc.t[t] = f64x4{0., 0., 0., 0.};
  const double* p0 = P + (size_t)(i0 + li) * CP + 4 * kk;
  const double* p1 = p0 + (size_t)16 * CP;
  const double* q0 = Q + (size_t)(4 * kk) * CP + j0 + li;
#pragma unroll 2
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    const f64x2 a0l = *reinterpret_cast<const f64x2*>(p0 + k0), a0h = *reinterpret_cast<const f64x2*>(p0 + k0 + 2);
    const f64x2 a1l = *reinterpret_cast<const f64x2*>(p1 + k0), a1h = *reinterpret_cast<const f64x2*>(p1 + k0 + 2);
    double b0[4], b1[4];
    for (int u = 0; u < 4; ++u) { b0[u] = q0[(size_t)(k0 + u) * CP]; b1[u] = q0[(size_t)(k0 + u) * CP + 16]; }
    const double a0[4] = {a0l[0], a0l[1], a0h[0], a0h[1]}, a1[4] = {a1l[0], a1l[1], a1h[0], a1h[1]};
    for (int u = 0; u < 4; ++u) {
      acc.t[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc.t[0], 0, 0, 0);
      acc.t[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc.t[1], 0, 0, 0);
      acc.t[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc.t[2], 0, 0, 0);
      acc.t[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc.t[3], 0, 0, 0);
    }
  }
  if (wave) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) red[wave - 1][(t * 4 + r) * 64 + lane] = acc.t[t][r];
  __syncthreads();
  if (!wave) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) {
    const int at = (t * 4 + r) * 64 + lane;
    acc.t[t][r] = ((acc.t[t][r] + red[0][at]) + red[1][at]) + red[2][at];
  }
}
__device__ __forceinline__ bool conv_(const Ws& w, int it) { return it > 0 && __longlong_as_double((long long)w.resid[it - 1]) < NS_TOL; }
__global__ __launch_bounds__(256) void stage1(Ws w, int it, double ca, double cb) {
  if (conv_(w, it)) return;
  __shared__ double red[3][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, cur = it & 1;
  Acc32 acc;
  gemm32(w.Z[cur], w.Y[cur], i0, j0, lane, wave, red, acc);
  if (wave) return;
  const int li = lane & 15, kk = lane >> 4;
  double m = 0.;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) {
    const int row = i0 + 16 * (t >> 1) + kk + 4 * r, col = j0 + 16 * (t & 1) + li;
    const double zy = acc.t[t][r], d = zy - (row == col ? 1.0 : 0.0);
    m = fmax(m, fabs(d));
    w.T[(size_t)row * CP + col] = (row == col ? ca : 0.0) - cb * zy;
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (lane == 0) atomicMax(&w.resid[it], (unsigned long long)__double_as_longlong(m));
}
__global__ __launch_bounds__(256) void stage2(Ws w, int it) {
  if (conv_(w, it)) return;
  __shared__ double red[3][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, cur = it & 1, nxt = cur ^ 1;
  const bool zside = blockIdx.z == 1;
  Acc32 acc;
  gemm32(zside ? w.T : w.Y[cur], zside ? w.Z[cur] : w.T, i0, j0, lane, wave, red, acc);
  if (wave) return;
  double* out = zside ? w.Z[nxt] : w.Y[nxt];
  const int li = lane & 15, kk = lane >> 4;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) out[(size_t)(i0 + 16 * (t >> 1) + kk + 4 * r) * CP + j0 + 16 * (t & 1) + li] = acc.t[t][r];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *w.iters = it + 1;
}

// ------------------------------------------------------------------------------------------------ single launch, one XCD
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
constexpr int SC1 = 16;   // buffer cache policy: agent scope (never served from this CU's L1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, CP * CP * 8, 0x00020000); }
__device__ __forceinline__ f64x2 ld2(__amdgpu_buffer_rsrc_t r, int elem) {
  return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, elem * 8, 0, SC1));
}
__device__ __forceinline__ double ld1(__amdgpu_buffer_rsrc_t r, int elem) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, elem * 8, 0, SC1));
}
__device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t r, int elem, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, elem * 8, 0, SC1);
}
// the same tile product as gemm32 (same k split, same order: bit-identical), operands through sc1 buffer loads
__device__ __forceinline__ void gemm32c(const double* P, const double* Q, int i0, int j0, int lane, int wave, double (*red)[16 * 64], Acc32& acc) {
  const int li = lane & 15, kk = lane >> 4;
  const int kq = CP >> 2, kbeg = wave * kq;
  const __amdgpu_buffer_rsrc_t rp = rsrc(P), rq = rsrc(Q);
  for (int t = 0; t < 4; ++t) acc.t[t] = f64x4{0., 0., 0., 0.};
  const int p0 = (i0 + li) * CP + 4 * kk, p1 = p0 + 16 * CP, q0 = (4 * kk) * CP + j0 + li;
  f64x2 a0l[2], a0h[2], a1l[2], a1h[2];
  double b0[2][4], b1[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {     // both k-blocks of this wave's quarter in flight
    const int k0 = kbeg + 16 * s;
    a0l[s] = ld2(rp, p0 + k0); a0h[s] = ld2(rp, p0 + k0 + 2); a1l[s] = ld2(rp, p1 + k0); a1h[s] = ld2(rp, p1 + k0 + 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) { b0[s][u] = ld1(rq, q0 + (k0 + u) * CP); b1[s][u] = ld1(rq, q0 + (k0 + u) * CP + 16); }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const double a0[4] = {a0l[s][0], a0l[s][1], a0h[s][0], a0h[s][1]}, a1[4] = {a1l[s][0], a1l[s][1], a1h[s][0], a1h[s][1]};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc.t[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[s][u], acc.t[0], 0, 0, 0);
      acc.t[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[s][u], acc.t[1], 0, 0, 0);
      acc.t[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[s][u], acc.t[2], 0, 0, 0);
      acc.t[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[s][u], acc.t[3], 0, 0, 0);
    }
  }
  if (wave) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) red[wave - 1][(t * 4 + r) * 64 + lane] = acc.t[t][r];
  __syncthreads();
  if (!wave) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) {
    const int at = (t * 4 + r) * 64 + lane;
    acc.t[t][r] = ((acc.t[t][r] + red[0][at]) + red[1][at]) + red[2][at];
  }
  __syncthreads();     // red is reused by the next product
}

// all of this workgroup's stores have reached L2, then: arrive, wait for the other NW - 1 (or for the abort flag)
__device__ __forceinline__ bool xcd_barrier(const Ws& w, unsigned& target, unsigned nwg) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  target += nwg;
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(w.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    bool ok = true;
    while (__hip_atomic_load(w.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > 4000000 || __hip_atomic_load(w.abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // ~0.1 s: never hangs
        __hip_atomic_store(w.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

constexpr int NW = 32;
__global__ __launch_bounds__(256) void ns_coop(Ws w, Sched sc, int maxit) {
#ifdef SPREAD   // -DSPREAD: the 32 participants are workgroups 0..31 = four on each XCD: do sc1 loads / stores and agent-scope atomics
                // keep the iterates coherent ACROSS the XCDs' L2s?  (bitwise comparison with the multi-launch schedule below)
  if (blockIdx.x >= NW) return;
  __shared__ double red[3][16 * 64];
  const int me = blockIdx.x;
#else
  if ((blockIdx.x & 7) != 0) return;                 // the workgroups the dispatcher places on one XCD
  __shared__ double red[3][16 * 64];
  const int me = blockIdx.x >> 3;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
  // all participants must really share an XCD (an L2): compare with workgroup 0's
  if (threadIdx.x == 0) {
    if (me == 0) __hip_atomic_store(w.xcc, xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned target = 0;
  if (!xcd_barrier(w, target, NW)) return;
#ifndef SPREAD
  if (threadIdx.x == 0 && __hip_atomic_load(w.xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc_id() + 1u)
    __hip_atomic_store(w.abort_, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  const int tile = me & 15, half = me >> 4;          // stage 1: tiles by workgroups 0..15; stage 2: Y' by half 0, Z' by half 1
  const int i0 = (tile >> 2) * 32, j0 = (tile & 3) * 32;
  int n = 0;
  for (int it = 0; it < maxit; ++it) {
    if (it > 0) {
      const double prev = __longlong_as_double((long long)__hip_atomic_load(&w.resid[it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (prev < NS_TOL) break;
    }
    const int cur = it & 1, nxt = cur ^ 1;
    if (half == 0) {
      Acc32 acc;
      gemm32c(w.Z[cur], w.Y[cur], i0, j0, lane, wave, red, acc);
      if (!wave) {
        const __amdgpu_buffer_rsrc_t rt = rsrc(w.T);
        double m = 0.;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) {
          const int row = i0 + 16 * (t >> 1) + kk + 4 * r, col = j0 + 16 * (t & 1) + li;
          const double zy = acc.t[t][r], d = zy - (row == col ? 1.0 : 0.0);
          m = fmax(m, fabs(d));
          st1(rt, row * CP + col, (row == col ? sc.ca[it] : 0.0) - sc.cb[it] * zy);
        }
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        if (lane == 0) __hip_atomic_fetch_max(&w.resid[it], (unsigned long long)__double_as_longlong(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!xcd_barrier(w, target, NW)) return;
    {
      Acc32 acc;
      gemm32c(half ? w.T : w.Y[cur], half ? w.Z[cur] : w.T, i0, j0, lane, wave, red, acc);
      if (!wave) {
        const __amdgpu_buffer_rsrc_t ro = rsrc(half ? w.Z[nxt] : w.Y[nxt]);
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r)
          st1(ro, (i0 + 16 * (t >> 1) + kk + 4 * r) * CP + j0 + 16 * (t & 1) + li, acc.t[t][r]);
      }
    }
    if (!xcd_barrier(w, target, NW)) return;
    n = it + 1;
  }
  if (me == 0 && threadIdx.x == 0) *w.iters = n;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  // SPD matrix with a graded spectrum (cond 3e4), scaled like solve.hip does (Frobenius norm), dead channels as identity
  std::vector<double> A(CP * CP, 0.), Q(CP * CP);
  srand(7);
  for (auto& q : Q) q = rand() / (double)RAND_MAX - 0.5;
  for (int j = 0; j < CP; ++j) {       // Gram-Schmidt
    for (int k = 0; k < j; ++k) { double d = 0; for (int i = 0; i < CP; ++i) d += Q[i * CP + j] * Q[i * CP + k]; for (int i = 0; i < CP; ++i) Q[i * CP + j] -= d * Q[i * CP + k]; }
    double nn = 0; for (int i = 0; i < CP; ++i) nn += Q[i * CP + j] * Q[i * CP + j]; nn = sqrt(nn); for (int i = 0; i < CP; ++i) Q[i * CP + j] /= nn;
  }
  for (int i = 0; i < CP; ++i) for (int j = 0; j < CP; ++j) { double s = 0; for (int k = 0; k < CP; ++k) s += Q[i * CP + k] * pow(3e4, -k / (double)(CP - 1)) * Q[j * CP + k]; A[i * CP + j] = s; }
  double fro = 0; for (double a : A) fro += a * a; fro = sqrt(fro);
  std::vector<double> Y0(CP * CP), Z0(CP * CP, 0.);
  for (int i = 0; i < CP * CP; ++i) Y0[i] = A[i] / fro;
  for (int i = 0; i < CP; ++i) Z0[i * CP + i] = 1.;
  Sched sc;
  double xlow = sqrt(1e-5);
  for (int it = 0; it < MAXIT; ++it) {
    const double mu = xlow <= 0.9 ? sqrt(3.0 / (1.0 + xlow + xlow * xlow)) : 1.0;
    xlow = mu * xlow * (3.0 - mu * mu * xlow * xlow) / 2.0;
    sc.ca[it] = 1.5 * mu; sc.cb[it] = 0.5 * mu * mu * mu;
  }
  Ws w;
  double* buf; CK(hipMalloc(&buf, 5 * CP * CP * 8));
  w.Y[0] = buf; w.Y[1] = buf + CP * CP; w.Z[0] = buf + 2 * CP * CP; w.Z[1] = buf + 3 * CP * CP; w.T = buf + 4 * CP * CP;
  CK(hipMalloc(&w.resid, (MAXIT + 2) * 8)); CK(hipMalloc(&w.iters, 64)); CK(hipMalloc(&w.bar, 64));
  w.abort_ = w.bar + 4; w.xcc = w.bar + 8;
  auto reset = [&]() {
    hipMemcpy(w.Y[0], Y0.data(), CP * CP * 8, hipMemcpyHostToDevice); hipMemcpy(w.Z[0], Z0.data(), CP * CP * 8, hipMemcpyHostToDevice);
    hipMemset(w.resid, 0, (MAXIT + 2) * 8); hipMemset(w.iters, 0, 64); hipMemset(w.bar, 0, 64);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> Zm(CP * CP), Ym(CP * CP), Zc(CP * CP), Yc(CP * CP);
  int itm = 0, itc = 0;
  float ms_multi = 0, ms_coop = 0;
  const int reps = 50;
  for (int rep = 0; rep < reps + 3; ++rep) {
    reset(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < MAXIT; ++it) {
      hipLaunchKernelGGL(stage1, dim3(4, 4, 1), dim3(256), 0, 0, w, it, sc.ca[it], sc.cb[it]);
      hipLaunchKernelGGL(stage2, dim3(4, 4, 2), dim3(256), 0, 0, w, it);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 3) ms_multi += ms;
  }
  CK(hipMemcpy(&itm, w.iters, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(Zm.data(), w.Z[itm & 1], CP * CP * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Ym.data(), w.Y[itm & 1], CP * CP * 8, hipMemcpyDeviceToHost));
  int mismatches = 0, aborts = 0;
  for (int rep = 0; rep < reps + 3; ++rep) {
    reset(); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(ns_coop, dim3(NW * 8), dim3(256), 0, 0, w, sc, MAXIT);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 3) ms_coop += ms;
    unsigned ab = 0; hipMemcpy(&ab, w.abort_, 4, hipMemcpyDeviceToHost); aborts += ab != 0;
    CK(hipMemcpy(&itc, w.iters, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Zc.data(), w.Z[itc & 1], CP * CP * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Yc.data(), w.Y[itc & 1], CP * CP * 8, hipMemcpyDeviceToHost));
    mismatches += itc != itm || memcmp(Zc.data(), Zm.data(), CP * CP * 8) != 0 || memcmp(Yc.data(), Ym.data(), CP * CP * 8) != 0;
  }
  // ||Z A Z - I|| with Z = Zc / sqrt(fro)
  double worst = 0;
  std::vector<double> ZA(CP * CP);
  for (int i = 0; i < CP; ++i) for (int j = 0; j < CP; ++j) { double s = 0; for (int k = 0; k < CP; ++k) s += Zc[i * CP + k] * A[k * CP + j]; ZA[i * CP + j] = s / fro; }
  for (int i = 0; i < CP; ++i) for (int j = 0; j < CP; ++j) { double s = 0; for (int k = 0; k < CP; ++k) s += ZA[i * CP + k] * Zc[k * CP + j]; worst = fmax(worst, fabs(s - (i == j))); }
  printf("multi-launch: %d iterations, %.1f us per solve (32 launches)\n", itm, ms_multi / reps * 1e3);
  printf("single launch: %d iterations, %.1f us per solve; bitwise mismatches vs multi-launch %d / %d, aborts %d, max|Z A Z - I| = %.2e\n", itc,
         ms_coop / reps * 1e3, mismatches, reps + 3, aborts, worst);
  return 0;
}
