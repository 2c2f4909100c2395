/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE, and NOT a restatement of anything the reference runs.
 *
 * CPU emulation of CANDIDATE device arithmetic for the reference's 3x3 convolutions (model/model_cd.py:726-742,
 * model/model_original.py:492-511: ReflectionPad2d(1) + Conv2d(3x3) + bias + ReLU), used to DECIDE on the CPU whether an
 * arithmetic is worth building as a kernel (VERDICT r3 task 3; tools/experiments/winograd_emul.py):
 *
 *   algo 0  direct nine-tap convolution
 *   algo 1  Winograd F(2x2, 3x3): U = G g G^T (fp64, once per filter), V = B^T d B (fp32 adds), M = sum_c U .* V, Y = A^T M A (fp32)
 *   split 0 fp32 operands
 *   split 1 "f16x3": every operand x = hi + lo, hi = f16(x), lo = f16(x - hi) (round-to-nearest-even, f16 subnormals, +-65504
 *           clamp), product = hi*hi + hi*lo + lo*hi accumulated in fp32 -- the product's split-f16 MFMA arithmetic
 *           (collaborative-distillation_amd/csrc/conv_f16_dev.h).  Weights are scaled by a power of two per layer before the split
 *           (like the packed device weights); under Winograd the TRANSFORMED tiles U, V are what is split.
 *   The input activations are first rounded to hi + lo as well (the SP16 activation format holds exactly that).
 *
 * Layout NCHW fp32, N = 1, weights OIHW -- the oracle's tensors (conv_ref.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float f16_round(float x) {
  /* the value of (float)(_Float16)x with the product's +-65504 clamp in front */
  float ax = fabsf(x);
  if (!(ax < 65504.f)) return x < 0 ? -65504.f : 65504.f; /* also NaN -> finite, like v_med3_f32 */
  if (ax < 6.103515625e-05f) return rintf(x * 16777216.f) * (1.0f / 16777216.f); /* subnormal: spacing 2^-24 */
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0xFFFu + ((u >> 13) & 1u);
  u &= ~0x1FFFu;
  memcpy(&x, &u, 4);
  return x;
}

static inline void split16(float x, float* hi, float* lo) {
  const float h = f16_round(x);
  *hi = h;
  *lo = f16_round(x - h);
}

static inline int reflect1(int i, int n) {
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

/* padded copy (reflect 1) with one extra zero row / column so that odd sizes can run whole 4x4 Winograd tiles */
static float* pad_input(const float* x, int C, int H, int W, int split, int* Hp_, int* Wp_) {
  const int Hp = H + 3, Wp = W + 3;
  float* xp = (float*)calloc((size_t)C * Hp * Wp, sizeof(float));
  if (!xp) return NULL;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int yy = 0; yy < H + 2; ++yy) {
      const float* src = x + ((size_t)c * H + reflect1(yy - 1, H)) * W;
      float* dst = xp + ((size_t)c * Hp + yy) * Wp;
      for (int xx = 0; xx < W + 2; ++xx) {
        float v = src[reflect1(xx - 1, W)];
        if (split) { float h, l; split16(v, &h, &l); v = h + l; } /* what an SP16 activation holds */
        dst[xx] = v;
      }
    }
  *Hp_ = Hp;
  *Wp_ = Wp;
  return xp;
}

static float pow2_scale(const double* a, size_t n) {
  /* power of two that brings max|a| just under 2^14: hi/lo of the scaled weights stay clear of f16 subnormals as far as possible */
  double m = 0;
  for (size_t i = 0; i < n; ++i) m = fmax(m, fabs(a[i]));
  if (m == 0) return 1.f;
  int e;
  frexp(m, &e); /* m = f * 2^e, f in [0.5, 1) */
  return (float)ldexp(1.0, 14 - e);
}

/* ---- direct, fp32 or f16x3 */
static int conv_direct(const float* x, int C, int H, int W, const float* w, const float* b, int K, int relu, float* y, int split) {
  int Hp, Wp;
  float* xp = pad_input(x, C, H, W, split, &Hp, &Wp);
  if (!xp) return -2;
  const size_t nw = (size_t)K * C * 9;
  double* wd = (double*)malloc(nw * sizeof(double));
  float *wh = (float*)malloc(nw * sizeof(float)), *wl = (float*)malloc(nw * sizeof(float));
  for (size_t i = 0; i < nw; ++i) wd[i] = w[i];
  const float sc = split ? pow2_scale(wd, nw) : 1.f;
  for (size_t i = 0; i < nw; ++i) {
    if (split) split16(w[i] * sc, &wh[i], &wl[i]); else { wh[i] = w[i]; wl[i] = 0.f; }
  }
  free(wd);
  float *xh = NULL, *xl = NULL;
  if (split) {
    const size_t n = (size_t)C * Hp * Wp;
    xh = (float*)malloc(n * sizeof(float));
    xl = (float*)malloc(n * sizeof(float));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) split16(xp[i], &xh[i], &xl[i]);
  }
  const float inv = 1.f / sc;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int k = 0; k < K; ++k)
    for (int yy = 0; yy < H; ++yy) {
      float* out = y + ((size_t)k * H + yy) * W;
      for (int xx = 0; xx < W; ++xx) out[xx] = 0.f;
      for (int c = 0; c < C; ++c)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const size_t wi = ((size_t)k * C + c) * 9 + ky * 3 + kx;
            const float h = wh[wi], l = wl[wi];
            const size_t off = ((size_t)c * Hp + yy + ky) * Wp + kx;
            if (split) {
              const float *ih = xh + off, *il = xl + off;
              for (int xx = 0; xx < W; ++xx) {
                float a = out[xx];
                a += h * ih[xx];
                a += h * il[xx];
                a += l * ih[xx];
                out[xx] = a;
              }
            } else {
              const float* in = xp + off;
              for (int xx = 0; xx < W; ++xx) out[xx] += h * in[xx];
            }
          }
      const float bias = b ? b[k] : 0.f;
      for (int xx = 0; xx < W; ++xx) {
        float v = out[xx] * inv + bias;
        out[xx] = (relu && !(v > 0.f)) ? 0.f : v;
      }
    }
  free(xp); free(wh); free(wl); free(xh); free(xl);
  return 0;
}

/* ---- Winograd F(2x2, 3x3), fp32 or f16x3 on the transformed tiles */
static int conv_wino(const float* x, int C, int H, int W, const float* w, const float* b, int K, int relu, float* y, int split) {
  int Hp, Wp;
  float* xp = pad_input(x, C, H, W, split, &Hp, &Wp);
  if (!xp) return -2;
  /* U = G g G^T in fp64; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] */
  const size_t nu = (size_t)K * C * 16;
  double* ud = (double*)malloc(nu * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int kc = 0; kc < K * C; ++kc) {
    const float* g = w + (size_t)kc * 9;
    double t[4][3];
    for (int j = 0; j < 3; ++j) {
      const double g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
      t[0][j] = g0; t[1][j] = 0.5 * (g0 + g1 + g2); t[2][j] = 0.5 * (g0 - g1 + g2); t[3][j] = g2;
    }
    double* u = ud + (size_t)kc * 16;
    for (int i = 0; i < 4; ++i) {
      u[i * 4 + 0] = t[i][0];
      u[i * 4 + 1] = 0.5 * (t[i][0] + t[i][1] + t[i][2]);
      u[i * 4 + 2] = 0.5 * (t[i][0] - t[i][1] + t[i][2]);
      u[i * 4 + 3] = t[i][2];
    }
  }
  const float sc = split ? pow2_scale(ud, nu) : 1.f;
  /* [k][p][c] so that the c loop walks contiguously */
  float *uh = (float*)malloc(nu * sizeof(float)), *ul = (float*)malloc(nu * sizeof(float));
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < 16; ++p) {
        const double v = ud[((size_t)k * C + c) * 16 + p] * sc;
        const size_t o = ((size_t)k * 16 + p) * C + c;
        if (split) split16((float)v, &uh[o], &ul[o]); else { uh[o] = (float)v; ul[o] = 0.f; }
      }
  free(ud);
  const int TY = (H + 1) / 2, TX = (W + 1) / 2;
  const float inv = 1.f / sc;
  int rc = 0;
#pragma omp parallel
  {
    float* vh = (float*)malloc((size_t)16 * C * TX * sizeof(float));
    float* vl = (float*)malloc((size_t)16 * C * TX * sizeof(float));
    float* acc = (float*)malloc((size_t)16 * TX * sizeof(float));
    if (!vh || !vl || !acc) rc = -2;
#pragma omp for schedule(dynamic, 1)
    for (int ty = 0; ty < TY; ++ty) {
      if (rc) continue;
      /* V = B^T d B per tile and channel, fp32; B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] */
      for (int c = 0; c < C; ++c) {
        const float* r0 = xp + ((size_t)c * Hp + 2 * ty) * Wp;
        for (int tx = 0; tx < TX; ++tx) {
          float d[4][4], t[4][4];
          for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) d[i][j] = r0[(size_t)i * Wp + 2 * tx + j];
          for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
          }
          for (int i = 0; i < 4; ++i) {
            const float v0 = t[i][0] - t[i][2], v1 = t[i][1] + t[i][2], v2 = t[i][2] - t[i][1], v3 = t[i][1] - t[i][3];
            const float v[4] = {v0, v1, v2, v3};
            for (int j = 0; j < 4; ++j) {
              const size_t o = ((size_t)(i * 4 + j) * C + c) * TX + tx;
              if (split) split16(v[j], &vh[o], &vl[o]); else { vh[o] = v[j]; vl[o] = 0.f; }
            }
          }
        }
      }
      for (int k = 0; k < K; ++k) {
        for (int p = 0; p < 16; ++p) {
          float* a = acc + (size_t)p * TX;
          for (int tx = 0; tx < TX; ++tx) a[tx] = 0.f;
          const float *uhp = uh + ((size_t)k * 16 + p) * C, *ulp = ul + ((size_t)k * 16 + p) * C;
          for (int c = 0; c < C; ++c) {
            const float h = uhp[c], l = ulp[c];
            const float *ih = vh + ((size_t)p * C + c) * TX, *il = vl + ((size_t)p * C + c) * TX;
            if (split) {
              for (int tx = 0; tx < TX; ++tx) {
                float s = a[tx];
                s += h * ih[tx];
                s += h * il[tx];
                s += l * ih[tx];
                a[tx] = s;
              }
            } else {
              for (int tx = 0; tx < TX; ++tx) a[tx] += h * ih[tx];
            }
          }
        }
        /* Y = A^T M A; A^T = [[1,1,1,0],[0,1,-1,-1]] */
        const float bias = b ? b[k] : 0.f;
        for (int tx = 0; tx < TX; ++tx) {
          float m[4][4], s[2][4];
          for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = acc[(size_t)p * TX + tx];
          for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
          }
          for (int i = 0; i < 2; ++i) {
            const int yy = 2 * ty + i;
            if (yy >= H) break;
            const float o0 = s[i][0] + s[i][1] + s[i][2], o1 = s[i][1] - s[i][2] - s[i][3];
            float v0 = o0 * inv + bias, v1 = o1 * inv + bias;
            if (relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
            float* out = y + ((size_t)k * H + yy) * W;
            out[2 * tx] = v0;
            if (2 * tx + 1 < W) out[2 * tx + 1] = v1;
          }
        }
      }
    }
    free(vh); free(vl); free(acc);
  }
  free(xp); free(uh); free(ul);
  return rc;
}

int oracle_conv3x3_emul(const float* x, int C, int H, int W, const float* w, const float* b, int K, int relu, float* y,
                        int algo, int split) {
  if (H < 2 || W < 2) return -1;
  return algo ? conv_wino(x, C, H, W, w, b, K, relu, y, split) : conv_direct(x, C, H, W, w, b, K, relu, y, split);
}
