/*
 * C restatement of the aqc-research fidelity/gradient hot path (CPU).
 *
 * TEST INFRASTRUCTURE ONLY.  Checker and CPU baseline for the HIP path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this
 * file.  The product package never does.
 *
 * Parity status: PINNED -- tests/test_oracle_c.py checks every entry point against the .npz
 * fixtures under tests/golden (outputs of the reference itself) and against the NumPy restatement.
 *
 * The algorithm is the reference's own: one full pass over the array per elementary gate, one
 * full pass per inner product (core_operations.py:606-1019, core_op_matrix.py:480-762).  Nothing
 * is fused or tiled here on purpose: this is what the reference's CPU path does, compiled.
 *
 * Layout: a (2^n x ncols) row-major complex128 array; a state vector is ncols = 1.  Qubit q is
 * bit q of the row index (core_operations.py:34-43, core_op_matrix.py:56).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex c128;

typedef struct {
    int n, ent, L, trotter, second, tpb, tail;
    const int32_t* blocks; /* [2][L] */
    long ncols, total;
} ref_t;

static int ref_init(ref_t* a, int n, int ent, const int32_t* blocks, int L, int trotter, int second, long ncols) {
    if (n < 1 || n > 30 || ent < 0 || ent > 2 || L < 0 || ncols < 1) return 1;
    a->n = n; a->ent = ent; a->L = L; a->trotter = trotter; a->second = second;
    a->tpb = ent == 2 ? 5 : 4;
    a->tail = (trotter && second) ? 3 * (n / 2) : 0; /* parametric_circuit.py:328-333 */
    a->blocks = blocks; a->ncols = ncols; a->total = ((long)1 << n) * ncols;
    for (int i = 0; i < L; ++i) {
        const int c = blocks[i], t = blocks[L + i];
        if (c < 0 || c >= n || t < 0 || t >= n || c == t) return 1;
    }
    return 0;
}

/* ---- one-qubit gates: pairs (i, i + h), h = ncols << q  (core_operations.py:164-264) ---- */
#define FOR_PAIRS(a, q)                                   \
    const long h = (a)->ncols << (q);                     \
    for (long b = 0; b < (a)->total; b += 2 * h)          \
        for (long i = b; i < b + h; ++i)

static void rz(const ref_t* a, c128* d, int q, double ang) {
    const c128 e0 = cexp(-0.5 * I * ang), e1 = cexp(0.5 * I * ang);
    FOR_PAIRS(a, q) { d[i] *= e0; d[i + h] *= e1; }
}
static void ry(const ref_t* a, c128* d, int q, double ang) {
    const double c = cos(0.5 * ang), s = sin(0.5 * ang);
    FOR_PAIRS(a, q) { const c128 x = d[i], y = d[i + h]; d[i] = c * x - s * y; d[i + h] = s * x + c * y; }
}
static void rx(const ref_t* a, c128* d, int q, double ang) {
    const double c = cos(0.5 * ang);
    const c128 s = -I * sin(0.5 * ang);
    FOR_PAIRS(a, q) { const c128 x = d[i], y = d[i + h]; d[i] = c * x + s * y; d[i + h] = s * x + c * y; }
}

/* ---- entanglers (core_operations.py:422-558) ---- */
static void entangle(const ref_t* a, c128* d, int qc, int qt, double ang) {
    const long rows = (long)1 << a->n, k = a->ncols;
    const c128 ph = cexp(I * ang);
    for (long r = 0; r < rows; ++r) {
        if (!((r >> qc) & 1)) continue;
        if (a->ent == 0) { /* CX: swap |c=1,t=0> <-> |c=1,t=1> */
            if ((r >> qt) & 1) continue;
            c128 *p = d + r * k, *p1 = d + (r | ((long)1 << qt)) * k;
            for (long j = 0; j < k; ++j) { const c128 t = p[j]; p[j] = p1[j]; p1[j] = t; }
        } else if ((r >> qt) & 1) {
            c128* p = d + r * k;
            if (a->ent == 1) for (long j = 0; j < k; ++j) p[j] = -p[j];          /* CZ */
            else             for (long j = 0; j < k; ++j) p[j] *= ph;             /* CP */
        }
    }
}

/* ---- inner products 0.5j <P w|z>  (core_operations.py:267-351) ---- */
static c128 dot_x(const ref_t* a, const c128* w, const c128* z, int q) {
    c128 s = 0;
    FOR_PAIRS(a, q) s += conj(w[i + h]) * z[i] + conj(w[i]) * z[i + h];
    return 0.5 * I * s;
}
static c128 dot_y(const ref_t* a, const c128* w, const c128* z, int q) {
    c128 s = 0;
    FOR_PAIRS(a, q) s += conj(w[i]) * z[i + h] - conj(w[i + h]) * z[i];
    return 0.5 * s;
}
static c128 dot_z(const ref_t* a, const c128* w, const c128* z, int q) {
    c128 s = 0;
    FOR_PAIRS(a, q) s += conj(w[i]) * z[i] - conj(w[i + h]) * z[i + h];
    return 0.5 * I * s;
}
/* -1j <P11 w|z> before the CP gate (core_op_matrix.py:430-477, core_operations.py:972-975) */
static c128 dot_cp11(const ref_t* a, const c128* w, const c128* z, int qc, int qt) {
    const long rows = (long)1 << a->n, k = a->ncols;
    c128 s = 0;
    for (long r = 0; r < rows; ++r)
        if (((r >> qc) & 1) && ((r >> qt) & 1))
            for (long j = 0; j < k; ++j) s += conj(w[r * k + j]) * z[r * k + j];
    return -I * s;
}

static void rs(const ref_t* a, c128* d, int q, double ang) { if (a->ent == 0) rx(a, d, q, ang); else rz(a, d, q, ang); }

/* V (core_operations.py:606-710) */
static void apply_v(const ref_t* a, const double* th, c128* d) {
    const double* t2 = th + 3 * a->n;
    for (int q = 0; q < a->n; ++q) {
        rz(a, d, q, th[3 * q + 2]); ry(a, d, q, th[3 * q + 1]); rz(a, d, q, th[3 * q + 0]);
    }
    for (int i = 0; i < a->L + a->tail; ++i) {
        const int j = i % a->L, c = a->blocks[j], t = a->blocks[a->L + j];
        const double* b = t2 + (long)a->tpb * j;
        if (a->trotter && i % 3 == 0) rz(a, d, c, -M_PI / 2);
        entangle(a, d, c, t, a->tpb == 5 ? b[4] : 0.0);
        ry(a, d, c, b[0]); rz(a, d, c, b[1]); ry(a, d, t, b[2]); rs(a, d, t, b[3]);
        if (a->trotter && i % 3 == 2) rz(a, d, t, M_PI / 2);
    }
}

/* V^H (core_operations.py:713-820) */
static void apply_vh(const ref_t* a, const double* th, c128* d) {
    const double* t2 = th + 3 * a->n;
    for (int i = a->L + a->tail - 1; i >= 0; --i) {
        const int j = i % a->L, c = a->blocks[j], t = a->blocks[a->L + j];
        const double* b = t2 + (long)a->tpb * j;
        if (a->trotter && i % 3 == 2) rz(a, d, t, -M_PI / 2);
        rs(a, d, t, -b[3]); ry(a, d, t, -b[2]); rz(a, d, c, -b[1]); ry(a, d, c, -b[0]);
        entangle(a, d, c, t, a->tpb == 5 ? -b[4] : 0.0);
        if (a->trotter && i % 3 == 0) rz(a, d, c, M_PI / 2);
    }
    for (int q = 0; q < a->n; ++q) {
        rz(a, d, q, -th[3 * q + 0]); ry(a, d, q, -th[3 * q + 1]); rz(a, d, q, -th[3 * q + 2]);
    }
}

/* forward w/z sweep (core_operations.py:918-1019, core_op_matrix.py:713-762); w, z overwritten */
static void sweep(const ref_t* a, const double* th, c128* w, c128* z, int from, int to, int front, c128* grad) {
    const int T = 3 * a->n + a->tpb * a->L;
    const double* t2 = th + 3 * a->n;
    c128* g2 = grad + 3 * a->n;
    for (int i = 0; i < T; ++i) grad[i] = 0;
    for (int q = 0; q < a->n; ++q) {
        rz(a, w, q, th[3 * q + 2]); rz(a, z, q, th[3 * q + 2]); if (front) grad[3 * q + 2] = dot_z(a, w, z, q);
        ry(a, w, q, th[3 * q + 1]); ry(a, z, q, th[3 * q + 1]); if (front) grad[3 * q + 1] = dot_y(a, w, z, q);
        rz(a, w, q, th[3 * q + 0]); rz(a, z, q, th[3 * q + 0]); if (front) grad[3 * q + 0] = dot_z(a, w, z, q);
    }
    for (int i = 0; i < a->L + a->tail; ++i) {
        const int j = i % a->L, c = a->blocks[j], t = a->blocks[a->L + j];
        const double* b = t2 + (long)a->tpb * j;
        c128* g = g2 + (long)a->tpb * j;
        const int live = from <= j && j < to;
        if (a->trotter && i % 3 == 0) { rz(a, w, c, -M_PI / 2); rz(a, z, c, -M_PI / 2); }
        const double ang = a->tpb == 5 ? b[4] : 0.0;
        if (live && a->tpb == 5) g[4] += dot_cp11(a, w, z, c, t);
        entangle(a, z, c, t, ang); entangle(a, w, c, t, ang);
        ry(a, w, c, b[0]); ry(a, z, c, b[0]); if (live) g[0] += dot_y(a, w, z, c);
        rz(a, w, c, b[1]); rz(a, z, c, b[1]); if (live) g[1] += dot_z(a, w, z, c);
        ry(a, w, t, b[2]); ry(a, z, t, b[2]); if (live) g[2] += dot_y(a, w, z, t);
        rs(a, w, t, b[3]); rs(a, z, t, b[3]); if (live) g[3] += a->ent == 0 ? dot_x(a, w, z, t) : dot_z(a, w, z, t);
        if (a->trotter && i % 3 == 2) { rz(a, w, t, M_PI / 2); rz(a, z, t, M_PI / 2); }
    }
}

/* ---- exported entry points (ctypes: oracle/aqc_ref.py) ------------------------------------------ */

/* data <- V data (inverse = 0) or V^H data (inverse = 1); data is (2^n x ncols) row-major complex128 */
int aqc_ref_apply(int n, int ent, const int32_t* blocks, int L, int trotter, int second, const double* thetas,
                  long ncols, int inverse, double* data) {
    ref_t a;
    if (ref_init(&a, n, ent, blocks, L, trotter, second, ncols)) return 1;
    if (inverse) apply_vh(&a, thetas, (c128*)data); else apply_v(&a, thetas, (c128*)data);
    return 0;
}

/* grad[T] (complex) of <V x|y> given w = x and z = V^H y; both are overwritten */
int aqc_ref_grad(int n, int ent, const int32_t* blocks, int L, int trotter, int second, const double* thetas,
                 long ncols, double* w, double* z, int from, int to, int front, double* grad) {
    ref_t a;
    if (ref_init(&a, n, ent, blocks, L, trotter, second, ncols)) return 1;
    sweep(&a, thetas, (c128*)w, (c128*)z, from, to, front, (c128*)grad);
    return 0;
}

/* B independent objective+gradient evaluations of the state-vector path (the bench's unit of work):
 * per lane b: z = V(theta_b)^H y; hs[b] = z[x_index]; grad_b = sweep(w = |x_index>, z).
 * Lanes are spread over `threads` OpenMP threads -- one evaluation per core at a time, which is how
 * the reference uses cores (job_executor.py:141). */
int aqc_ref_eval_batch(int n, int ent, const int32_t* blocks, int L, int trotter, int second, int B,
                       const double* thetas /* [B][T] */, const double* y, long x_index, int threads,
                       double* hs /* [B] c128 */, double* grads /* [B][T] c128 */) {
    ref_t a;
    if (ref_init(&a, n, ent, blocks, L, trotter, second, 1)) return 1;
    const int T = 3 * n + a.tpb * L;
    const long N = a.total;
    if (x_index < 0 || x_index >= N) return 1;
    int rc = 0;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        c128* w = (c128*)malloc(sizeof(c128) * N);
        c128* z = (c128*)malloc(sizeof(c128) * N);
        if (!w || !z) { rc = 2; free(w); free(z); continue; }
        memcpy(z, y, sizeof(c128) * N);
        apply_vh(&a, thetas + (long)b * T, z);
        ((c128*)hs)[b] = z[x_index];
        memset(w, 0, sizeof(c128) * N);
        w[x_index] = 1.0;
        sweep(&a, thetas + (long)b * T, w, z, 0, L, 1, (c128*)grads + (long)b * T);
        free(w); free(z);
    }
    return rc;
}

/* ---- coordinate descent (core_op_matrix.py:765-917) --------------------------------------------------------------- */
static double cd_delta(c128 prod, c128 grad, double d) {   /* _delta_theta, :833-850 */
    const double tol = 1.4901161193847656e-08, lr = M_PI / 16, maxdt = M_PI / 4;
    double d1 = (-2.0 * creal(conj(prod) * grad)) / (d * d);
    const double d2 = (-2.0 * (creal(grad) * creal(grad) + cimag(grad) * cimag(grad)) +
                       0.5 * (creal(prod) * creal(prod) + cimag(prod) * cimag(prod))) / (d * d);
    double dt;
    if (d2 < tol) { d1 /= fmax(fabs(d1), 1.0); dt = -lr * d1; } else dt = -d1 / d2;
    const double r = fabs(dt / maxdt);
    return r <= 1.0 ? dt : dt / r;
}
static c128 vdot_all(const ref_t* a, const c128* w, const c128* z) {
    c128 s = 0;
    for (long i = 0; i < a->total; ++i) s += conj(w[i]) * z[i];
    return s;
}
/* one parameter: grad and prod with the current operands, z rotated by the OLD angle, w by the NEW one (:855-912) */
static void cd_step(const ref_t* a, c128* w, c128* z, int q, int kind /* 0 y, 1 z, 2 x */, double* theta, double d) {
    const c128 grad = kind == 0 ? dot_y(a, w, z, q) : kind == 1 ? dot_z(a, w, z, q) : dot_x(a, w, z, q);
    const c128 prod = vdot_all(a, w, z);
    void (*rot)(const ref_t*, c128*, int, double) = kind == 0 ? ry : kind == 1 ? rz : rx;
    rot(a, z, q, *theta);
    *theta += cd_delta(prod, grad, d);
    rot(a, w, q, *theta);
}
static double cd_sweep(const ref_t* a, double* th, const c128* target, c128* w, c128* z) {
    const long d = (long)1 << a->n;
    memset(w, 0, sizeof(c128) * a->total);
    for (long i = 0; i < d; ++i) w[i * d + i] = 1.0;
    memcpy(z, target, sizeof(c128) * a->total);
    apply_vh(a, th, z);
    double* t2 = th + 3 * a->n;
    for (int q = 0; q < a->n; ++q) {
        cd_step(a, w, z, q, 1, th + 3 * q + 2, (double)d);
        cd_step(a, w, z, q, 0, th + 3 * q + 1, (double)d);
        cd_step(a, w, z, q, 1, th + 3 * q + 0, (double)d);
    }
    for (int j = 0; j < a->L; ++j) {
        const int c = a->blocks[j], t = a->blocks[a->L + j];
        double* b = t2 + 4L * j;
        entangle(a, z, c, t, 0.0); entangle(a, w, c, t, 0.0);
        cd_step(a, w, z, c, 0, b + 0, (double)d);
        cd_step(a, w, z, c, 1, b + 1, (double)d);
        cd_step(a, w, z, t, 0, b + 2, (double)d);
        cd_step(a, w, z, t, a->ent == 0 ? 2 : 1, b + 3, (double)d);
    }
    const c128 p = vdot_all(a, w, z) / (double)d;
    return 1.0 - (creal(p) * creal(p) + cimag(p) * cimag(p));
}
/* `nsweeps` consecutive sweeps for each of B lanes (thetas [B][T] updated in place, targets [B][d][d] or one shared [d][d]),
 * lanes spread over `threads` OpenMP threads; fobj [B][nsweeps].  cx / cz only, no Trotter decorations (matrix path). */
int aqc_ref_cd_sweeps(int n, int ent, const int32_t* blocks, int L, int B, double* thetas, const double* targets, int shared_target,
                      int nsweeps, int threads, double* fobj) {
    ref_t a;
    if (ent == 2 || ref_init(&a, n, ent, blocks, L, 0, 0, (long)1 << n)) return 1;
    const int T = 3 * n + 4 * L;
    int rc = 0;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        c128* w = (c128*)malloc(sizeof(c128) * a.total);
        c128* z = (c128*)malloc(sizeof(c128) * a.total);
        if (!w || !z) { rc = 2; free(w); free(z); continue; }
        const c128* u = (const c128*)targets + (shared_target ? 0 : (long)b * a.total);
        for (int s = 0; s < nsweeps; ++s) fobj[(long)b * nsweeps + s] = cd_sweep(&a, thetas + (long)b * T, u, w, z);
        free(w); free(z);
    }
    return rc;
}
