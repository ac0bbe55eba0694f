/* Sanitizer self-test of the C restatement (built with -fsanitize=address,undefined by tests/test_native_sanitizers.py):
 * V V^H = 1, gradient == central differences, threaded batch == serial batch, on small random cases. */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int aqc_ref_apply(int n, int ent, const int32_t* blocks, int L, int trotter, int second, const double* thetas, long ncols, int inverse, double* data);
int aqc_ref_grad(int n, int ent, const int32_t* blocks, int L, int trotter, int second, const double* thetas, long ncols, double* w, double* z,
                 int from, int to, int front, double* grad);
int aqc_ref_eval_batch(int n, int ent, const int32_t* blocks, int L, int trotter, int second, int B, const double* thetas, const double* y,
                       long x_index, int threads, double* hs, double* grads);

static double urand(void) { return (double)rand() / RAND_MAX; }

int main(void) {
    srand(7);
    int failures = 0;
    for (int ent = 0; ent < 3; ++ent)
        for (long ncols = 1; ncols <= 3; ncols += 2) {
            const int n = 4, L = 7, tpb = ent == 2 ? 5 : 4, T = 3 * n + tpb * L;
            const long N = (1L << n) * ncols;
            int32_t blocks[2 * 7];
            for (int i = 0; i < L; ++i) { blocks[i] = rand() % n; do { blocks[L + i] = rand() % n; } while (blocks[L + i] == blocks[i]); }
            double* th = malloc(sizeof(double) * T);
            for (int i = 0; i < T; ++i) th[i] = 6.28 * urand() - 3.14;
            double _Complex *x = malloc(sizeof(double _Complex) * N), *y = malloc(sizeof(double _Complex) * N), *v = malloc(sizeof(double _Complex) * N);
            double _Complex *w = malloc(sizeof(double _Complex) * N), *z = malloc(sizeof(double _Complex) * N), *g = malloc(sizeof(double _Complex) * T);
            for (long i = 0; i < N; ++i) { x[i] = urand() + I * urand(); y[i] = urand() - I * urand(); v[i] = y[i]; }
            aqc_ref_apply(n, ent, blocks, L, 0, 0, th, ncols, 1, (double*)v);      /* v = V^H y */
            for (long i = 0; i < N; ++i) { w[i] = x[i]; z[i] = v[i]; }
            aqc_ref_grad(n, ent, blocks, L, 0, 0, th, ncols, (double*)w, (double*)z, 0, L, 1, (double*)g);
            aqc_ref_apply(n, ent, blocks, L, 0, 0, th, ncols, 0, (double*)v);      /* V V^H y = y */
            double err = 0;
            for (long i = 0; i < N; ++i) err = fmax(err, cabs(v[i] - y[i]));
            for (int t = 0; t < T; t += 5) {                                       /* d/dtheta <V x|y> */
                double _Complex f[2];
                for (int s = 0; s < 2; ++s) {
                    th[t] += s ? -2e-6 : 1e-6;
                    for (long i = 0; i < N; ++i) w[i] = x[i];
                    aqc_ref_apply(n, ent, blocks, L, 0, 0, th, ncols, 0, (double*)w);
                    f[s] = 0;
                    for (long i = 0; i < N; ++i) f[s] += conj(w[i]) * y[i];
                }
                th[t] += 1e-6;
                err = fmax(err, cabs((f[0] - f[1]) / 2e-6 - g[t]) * 1e-3);
            }
            if (err > 1e-9) { printf("ent %d ncols %ld: error %g\n", ent, ncols, err); ++failures; }
            free(th); free(x); free(y); free(v); free(w); free(z); free(g);
        }
    {   /* Trotter ansatz, batch on 1 and 3 threads */
        const int n = 4, L = 9, T = 3 * n + 4 * L, B = 3;
        const int32_t blocks[18] = {1, 0, 1, 3, 2, 3, 2, 1, 2, /* targets */ 0, 1, 0, 2, 3, 2, 1, 2, 1};
        double th[3 * 48], hs1[6], hs3[6];
        double _Complex y[16];
        double* g1 = malloc(sizeof(double) * 2 * B * T);
        double* g3 = malloc(sizeof(double) * 2 * B * T);
        for (int i = 0; i < B * T; ++i) th[i] = urand();
        for (int i = 0; i < 16; ++i) y[i] = urand() + I * urand();
        if (aqc_ref_eval_batch(n, 0, blocks, L, 1, 1, B, th, (double*)y, 5, 1, hs1, g1) || aqc_ref_eval_batch(n, 0, blocks, L, 1, 1, B, th, (double*)y, 5, 3, hs3, g3)) ++failures;
        for (int i = 0; i < 2 * B * T; ++i) if (g1[i] != g3[i]) { ++failures; break; }
        for (int i = 0; i < 2 * B; ++i) if (hs1[i] != hs3[i]) { ++failures; break; }
        free(g1); free(g3);
    }
    printf(failures ? "FAILED\n" : "ok\n");
    return failures != 0;
}
