/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the dense fp32 operators the reference obtains from
 * torch.nn on its hot path (they live in ATen, not in /root/reference):
 *   ReflectionPad2d(1) + Conv2d(3x3, stride 1) + bias (+ ReLU)   model/model_cd.py:726-742 ...
 *   Conv2d 1x1 3->3 ("conv0")                                    model/model_cd.py:725
 *   MaxPool2d(2,2), floor mode                                   model/model_cd.py:709,728
 *   UpsamplingNearest2d(scale_factor=2)                          model/model_cd.py:261,278
 * Layout: NCHW fp32, N=1, weights OIHW -- exactly the reference's tensors.
 * Accumulation is fp32 in (c, ky, kx) order; ATen's order is unspecified, so parity
 * against the goldens is checked to ~1e-5 relative, not bitwise.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The .so is built in the dev container and travels to the GPU box, whose host CPU may
 * differ: dispatch the one hot loop at load time instead of compiling with -march=native. */
__attribute__((target_clones("default", "avx2", "avx512f")))
void oracle_row_fma3(float* out, const float* in, float w0, float w1, float w2, int W) {
  for (int xx = 0; xx < W; ++xx) {
    float a = out[xx];
    a += w0 * in[xx];
    a += w1 * in[xx + 1];
    a += w2 * in[xx + 2];
    out[xx] = a;
  }
}

static inline int reflect(int i, int n) {
  /* ReflectionPad2d(1): index -1 -> 1, n -> n-2 */
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* y[K,H,W] = relu?( bias + sum_{c,ky,kx} w[k,c,ky,kx] * xpad[c, y+ky, x+kx] ) */
int oracle_conv3x3_reflect(const float* x, int C, int H, int W, const float* w, const float* b,
                           int K, int relu, float* y) {
  if (H < 2 || W < 2) return -1; /* reflect pad of 1 needs >=2 samples (torch raises too) */
  const int Wp = W + 2, Hp = H + 2;
  float* xp = (float*)malloc((size_t)C * Hp * Wp * sizeof(float));
  if (!xp) return -2;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int yy = 0; yy < Hp; ++yy) {
      const float* src = x + ((size_t)c * H + reflect(yy - 1, H)) * W;
      float* dst = xp + ((size_t)c * Hp + yy) * Wp;
      dst[0] = src[1];
      memcpy(dst + 1, src, (size_t)W * sizeof(float));
      dst[W + 1] = src[W - 2];
    }
  const int RB = 4; /* rows per block: keeps the C x (RB+2) x Wp input slab cache-resident */
  const int nblk = (H + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
  for (int blk = 0; blk < nblk; ++blk) {
    const int y0 = blk * RB, y1 = (y0 + RB < H) ? y0 + RB : H;
    for (int k = 0; k < K; ++k) {
      for (int yy = y0; yy < y1; ++yy) {
        float* out = y + ((size_t)k * H + yy) * W;
        const float bias = b ? b[k] : 0.f;
        for (int xx = 0; xx < W; ++xx) out[xx] = bias;
        for (int c = 0; c < C; ++c) {
          const float* wk = w + ((size_t)k * C + c) * 9;
          for (int ky = 0; ky < 3; ++ky) {
            const float* in = xp + ((size_t)c * Hp + yy + ky) * Wp;
            const float w0 = wk[ky * 3 + 0], w1 = wk[ky * 3 + 1], w2 = wk[ky * 3 + 2];
            oracle_row_fma3(out, in, w0, w1, w2, W);
          }
        }
        if (relu)
          for (int xx = 0; xx < W; ++xx) out[xx] = out[xx] > 0.f ? out[xx] : 0.f;
      }
    }
  }
  free(xp);
  return 0;
}

/* ---- fp64 arm ("truth"): the same operators with double tensors and double accumulation.  Not a restatement of
 * anything the reference runs (its convolutions are fp32, model_cd.py) -- it is the yardstick against which the
 * tests measure how far the fp32 reference arithmetic itself, and the HIP path, sit from the exact result
 * (tests/test_hip_parity.py::test_error_budget_vs_fp64_truth). */
__attribute__((target_clones("default", "avx2", "avx512f")))
void oracle_row_fma3_f64(double* out, const double* in, double w0, double w1, double w2, int W) {
  for (int xx = 0; xx < W; ++xx) {
    double a = out[xx];
    a += w0 * in[xx];
    a += w1 * in[xx + 1];
    a += w2 * in[xx + 2];
    out[xx] = a;
  }
}

int oracle_conv3x3_reflect_f64(const double* x, int C, int H, int W, const float* w, const float* b,
                               int K, int relu, double* y) {
  if (H < 2 || W < 2) return -1;
  const int Wp = W + 2, Hp = H + 2;
  double* xp = (double*)malloc((size_t)C * Hp * Wp * sizeof(double));
  if (!xp) return -2;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int yy = 0; yy < Hp; ++yy) {
      const double* src = x + ((size_t)c * H + reflect(yy - 1, H)) * W;
      double* dst = xp + ((size_t)c * Hp + yy) * Wp;
      dst[0] = src[1];
      memcpy(dst + 1, src, (size_t)W * sizeof(double));
      dst[W + 1] = src[W - 2];
    }
  const int RB = 4;
  const int nblk = (H + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
  for (int blk = 0; blk < nblk; ++blk) {
    const int y0 = blk * RB, y1 = (y0 + RB < H) ? y0 + RB : H;
    for (int k = 0; k < K; ++k) {
      for (int yy = y0; yy < y1; ++yy) {
        double* out = y + ((size_t)k * H + yy) * W;
        const double bias = b ? (double)b[k] : 0.0;
        for (int xx = 0; xx < W; ++xx) out[xx] = bias;
        for (int c = 0; c < C; ++c) {
          const float* wk = w + ((size_t)k * C + c) * 9;
          for (int ky = 0; ky < 3; ++ky) {
            const double* in = xp + ((size_t)c * Hp + yy + ky) * Wp;
            oracle_row_fma3_f64(out, in, (double)wk[ky * 3 + 0], (double)wk[ky * 3 + 1], (double)wk[ky * 3 + 2], W);
          }
        }
        if (relu)
          for (int xx = 0; xx < W; ++xx) out[xx] = out[xx] > 0.0 ? out[xx] : 0.0;
      }
    }
  }
  free(xp);
  return 0;
}

/* y[K,H,W] = bias + sum_c w[k,c] x[c]  (1x1 conv, no activation) */
int oracle_conv1x1(const float* x, int C, int H, int W, const float* w, const float* b, int K, float* y) {
  const size_t n = (size_t)H * W;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    float* out = y + (size_t)k * n;
    const float bias = b ? b[k] : 0.f;
    for (size_t i = 0; i < n; ++i) out[i] = bias;
    for (int c = 0; c < C; ++c) {
      const float wk = w[(size_t)k * C + c];
      const float* in = x + (size_t)c * n;
      for (size_t i = 0; i < n; ++i) out[i] += wk * in[i];
    }
  }
  return 0;
}

/* floor-mode 2x2/2 max pool: drops an odd last row / column */
int oracle_maxpool2(const float* x, int C, int H, int W, float* y) {
  const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int yy = 0; yy < Ho; ++yy) {
      const float* r0 = x + ((size_t)c * H + 2 * yy) * W;
      const float* r1 = r0 + W;
      float* out = y + ((size_t)c * Ho + yy) * Wo;
      for (int xx = 0; xx < Wo; ++xx) {
        float a = r0[2 * xx], b2 = r0[2 * xx + 1], c2 = r1[2 * xx], d = r1[2 * xx + 1];
        float m0 = a > b2 ? a : b2, m1 = c2 > d ? c2 : d;
        out[xx] = m0 > m1 ? m0 : m1;
      }
    }
  return 0;
}

/* nearest x2: out[i,j] = in[i/2, j/2] */
int oracle_upsample2(const float* x, int C, int H, int W, float* y) {
  const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int yy = 0; yy < Ho; ++yy) {
      const float* in = x + ((size_t)c * H + yy / 2) * W;
      float* out = y + ((size_t)c * Ho + yy) * Wo;
      for (int xx = 0; xx < Wo; ++xx) out[xx] = in[xx / 2];
    }
  return 0;
}
