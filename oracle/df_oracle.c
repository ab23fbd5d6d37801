/* Plain-C restatement of the stencil / loss / conv arithmetic on the north-star path -- TEST INFRASTRUCTURE ONLY.
 *
 * Second CPU implementation next to oracle/df_oracle.py (NumPy): scalar loops in index form, written from
 * SURVEY.md Appendix A.  Pinned by the same golden vectors (tests/golden/stencils.npz, captured by executing the
 * reference's own ops.py) through tests/test_oracle_c.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (deep_fluids_amd) never does.
 *
 * Layout: channels-last fp32, [B,Y,X,C] / [B,Z,Y,X,C]  ("x: bzyxd", reference ops.py:228).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Thread count of the OpenMP loops below, set / read through the runtime itself (an OMP_NUM_THREADS exported after the host process
 * has already initialised an OpenMP runtime -- PyTorch's -- is ignored): bench.py's cpu_baseline reports what dfo_max_threads() says. */
void dfo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int dfo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* D_a f[i] = f[i+1]-f[i] (i <= n-2), D_a f[n-1] = D_a f[n-2]  -- reference ops.py:214-217, 243-253, 269-270 */
static inline float fdiff(const float* f, int64_t i, int64_t n, int64_t stride) {
  if (i == n - 1) return f[0] - f[-stride];
  return f[stride] - f[0];
}

/* curl, reference ops.py:264-274: psi [B,Y,X,1] -> u [B,Y,X,2] = (D_y psi, -D_x psi) */
void dfo_curl2d(const float* psi, float* u, int64_t B, int64_t Y, int64_t X) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t y = 0; y < Y; ++y)
      for (int64_t x = 0; x < X; ++x) {
        const int64_t v = (b * Y + y) * X + x;
        u[v * 2 + 0] = fdiff(psi + v, y, Y, X);
        u[v * 2 + 1] = -fdiff(psi + v, x, X, 1);
      }
}

/* jacobian, reference ops.py:205-225: x [B,Y,X,2] -> j [..,4] = (dudx,dudy,dvdx,dvdy), w [..,1] = dvdx-dudy */
void dfo_jacobian2d(const float* x, float* j, float* w, int64_t B, int64_t Y, int64_t X) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t y = 0; y < Y; ++y)
      for (int64_t xx = 0; xx < X; ++xx) {
        const int64_t v = (b * Y + y) * X + xx;
        const float dudx = fdiff(x + v * 2 + 0, xx, X, 2), dudy = fdiff(x + v * 2 + 0, y, Y, 2 * X);
        const float dvdx = fdiff(x + v * 2 + 1, xx, X, 2), dvdy = fdiff(x + v * 2 + 1, y, Y, 2 * X);
        j[v * 4 + 0] = dudx; j[v * 4 + 1] = dudy; j[v * 4 + 2] = dvdx; j[v * 4 + 3] = dvdy;
        w[v] = dvdx - dudy;
      }
}

/* jacobian3, reference ops.py:227-262: x [B,Z,Y,X,3] -> j [..,9], c [..,3] */
void dfo_jacobian3d(const float* x, float* j, float* c, int64_t B, int64_t Z, int64_t Y, int64_t X) {
  const int64_t sx = 3, sy = 3 * X, sz = 3 * X * Y;
#pragma omp parallel for collapse(2)
  for (int64_t b = 0; b < B; ++b)
    for (int64_t z = 0; z < Z; ++z)
      for (int64_t y = 0; y < Y; ++y)
        for (int64_t xx = 0; xx < X; ++xx) {
          const int64_t v = ((b * Z + z) * Y + y) * X + xx;
          float d[3][3];
          for (int comp = 0; comp < 3; ++comp) {
            const float* p = x + v * 3 + comp;
            d[comp][0] = fdiff(p, xx, X, sx);
            d[comp][1] = fdiff(p, y, Y, sy);
            d[comp][2] = fdiff(p, z, Z, sz);
          }
          if (j)
            for (int comp = 0; comp < 3; ++comp)
              for (int ax = 0; ax < 3; ++ax) j[v * 9 + comp * 3 + ax] = d[comp][ax];
          if (c) {
            c[v * 3 + 0] = d[2][1] - d[1][2];   /* dwdy - dvdz */
            c[v * 3 + 1] = d[0][2] - d[2][0];   /* dudz - dwdx */
            c[v * 3 + 2] = d[1][0] - d[0][1];   /* dvdx - dudy */
          }
        }
}

/* mean |a-b|, reference trainer.py:170-171 (fp64 accumulation) */
double dfo_l1_mean(const float* a, const float* b, int64_t n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += fabs((double)a[i] - (double)b[i]);
  return s / (double)n;
}

/* The reference graph's tail for one batch (trainer.py:30, trainer3.py:18,24,49-51), op by op like the TF graph:
 *   jx = jacobian3(x)[0];  u = jacobian3(psi)[1];  ju = jacobian3(u)[0];  l1 = mean|u - x|;  jl1 = mean|ju - jx|.
 * `ju`, `jx` are caller-provided scratch [B,Z,Y,X,9].  Used by bench.py's cpu_baseline leg (host-side stencil baseline). */
void dfo_velocity_tail3d(const float* psi, const float* x, float* u, float* ju, float* jx, double* l1, double* jl1, int64_t B,
                         int64_t Z, int64_t Y, int64_t X) {
  const int64_t n = B * Z * Y * X;
  dfo_jacobian3d(x, jx, (float*)0, B, Z, Y, X);
  dfo_jacobian3d(psi, (float*)0, u, B, Z, Y, X);
  dfo_jacobian3d(u, ju, (float*)0, B, Z, Y, X);
  *l1 = dfo_l1_mean(u, x, n * 3);
  *jl1 = dfo_l1_mean(ju, jx, n * 9);
}

/* slim.conv3d / conv2d, k = 3, stride 1, SAME, channels-last, TF weights [kz,ky,kx,Cin,Cout] (kz = 1: 2-D);
 * fp64 accumulation; optional bias and lrelu (reference ops.py:9-16). */
void dfo_conv_same(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t D, int64_t H,
                   int64_t W, int64_t Cin, int64_t Cout, int kz, int lrelu, float leak) {
  const int pz = kz / 2;
#pragma omp parallel for collapse(2)
  for (int64_t b = 0; b < B; ++b)
    for (int64_t z = 0; z < D; ++z) {
      double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
      for (int64_t yy = 0; yy < H; ++yy)
        for (int64_t xx = 0; xx < W; ++xx) {
          for (int64_t co = 0; co < Cout; ++co) acc[co] = bias ? (double)bias[co] : 0.0;
          for (int dz = 0; dz < kz; ++dz)
            for (int dy = 0; dy < 3; ++dy)
              for (int dx = 0; dx < 3; ++dx) {
                const int64_t zs = z + dz - pz, ys = yy + dy - 1, xs = xx + dx - 1;
                if (zs < 0 || zs >= D || ys < 0 || ys >= H || xs < 0 || xs >= W) continue;
                const float* xp = x + (((b * D + zs) * H + ys) * W + xs) * Cin;
                const float* wp = w + ((int64_t)((dz * 3 + dy) * 3 + dx)) * Cin * Cout;
                for (int64_t ci = 0; ci < Cin; ++ci) {
                  const double xv = xp[ci];
                  for (int64_t co = 0; co < Cout; ++co) acc[co] += xv * (double)wp[ci * Cout + co];
                }
              }
          float* yp = y + (((b * D + z) * H + yy) * W + xx) * Cout;
          for (int64_t co = 0; co < Cout; ++co) {
            double v = acc[co];
            if (lrelu) v = v > leak * v ? v : leak * v;
            yp[co] = (float)v;
          }
        }
      free(acc);
    }
}
