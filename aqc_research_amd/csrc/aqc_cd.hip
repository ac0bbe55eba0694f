// Coordinate-descent sweep (core_op_matrix.py:765-917): per parameter one reduction pass
// (grad = 0.5j<P w|z>, prod = <w|z>) and one update pass that derives the Newton / gradient
// step on the device (every workgroup recomputes the same scalar from the same partials in the
// same order) and applies the rotation with the OLD angle to z and the NEW angle to w.
// Parameters are strictly sequential (Gauss-Seidel), so the sweep is a chain of small launches
// with no host round trip in between.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"

namespace aqc {

typedef double2 cplx;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ size_t pair_index(size_t g, int hbit) {
    const size_t lo = g & (((size_t)1 << hbit) - 1);
    return ((g >> hbit) << (hbit + 1)) | lo;
}

// kind: 0 = Y (ry), 1 = Z (rz), 2 = X (rx).  part[blk] = {sum for grad, sum for prod}
__global__ __launch_bounds__(256) void cd_dot_kernel(const cplx* __restrict__ w, const cplx* __restrict__ z, size_t npairs,
                                                     int hbit, int kind, cplx* part) {
    __shared__ double sm[4][4];
    const size_t h = (size_t)1 << hbit;
    double gr = 0, gi = 0, pr = 0, pi = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(g, hbit);
        const cplx w0 = w[i0], w1 = w[i0 + h], z0 = z[i0], z1 = z[i0 + h];
        // conj(a)*b = (ax bx + ay by) + i (ax by - ay bx)
        const double c00r = w0.x * z0.x + w0.y * z0.y, c00i = w0.x * z0.y - w0.y * z0.x;
        const double c11r = w1.x * z1.x + w1.y * z1.y, c11i = w1.x * z1.y - w1.y * z1.x;
        pr += c00r + c11r;
        pi += c00i + c11i;
        if (kind == 1) {
            gr += c00r - c11r;
            gi += c00i - c11i;
        } else {
            const double c01r = w0.x * z1.x + w0.y * z1.y, c01i = w0.x * z1.y - w0.y * z1.x;
            const double c10r = w1.x * z0.x + w1.y * z0.y, c10i = w1.x * z0.y - w1.y * z0.x;
            if (kind == 0) { gr += c01r - c10r; gi += c01i - c10i; } else { gr += c01r + c10r; gi += c01i + c10i; }
        }
    }
    gr = wsum(gr); gi = wsum(gi); pr = wsum(pr); pi = wsum(pi);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wave][0] = gr; sm[wave][1] = gi; sm[wave][2] = pr; sm[wave][3] = pi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sm[i][0]; b += sm[i][1]; c += sm[i][2]; d += sm[i][3]; }
        part[2 * blockIdx.x] = make_double2(a, b);
        part[2 * blockIdx.x + 1] = make_double2(c, d);
    }
}

__device__ __forceinline__ void rot_pair(int kind, cplx& a0, cplx& a1, double c, double s) {
    if (kind == 0) {  // Ry
        const cplx t0 = make_double2(c * a0.x - s * a1.x, c * a0.y - s * a1.y);
        const cplx t1 = make_double2(s * a0.x + c * a1.x, s * a0.y + c * a1.y);
        a0 = t0; a1 = t1;
    } else if (kind == 1) {  // Rz
        a0 = make_double2(c * a0.x + s * a0.y, c * a0.y - s * a0.x);
        a1 = make_double2(c * a1.x - s * a1.y, c * a1.y + s * a1.x);
    } else {  // Rx
        const cplx t0 = make_double2(c * a0.x + s * a1.y, c * a0.y - s * a1.x);
        const cplx t1 = make_double2(s * a0.y + c * a1.x, c * a1.y - s * a0.x);
        a0 = t0; a1 = t1;
    }
}

__global__ __launch_bounds__(256) void cd_update_kernel(cplx* __restrict__ w, cplx* __restrict__ z, size_t npairs, int hbit,
                                                        int kind, const cplx* part, int nparts, const double* theta_in,
                                                        double* theta_out, int tindex, double dim) {
    __shared__ double sh[4];
    if (threadIdx.x == 0) {  // identical in every workgroup: same partials, same order
        double gr = 0, gi = 0, pr = 0, pi = 0;
        for (int i = 0; i < nparts; ++i) { gr += part[2 * i].x; gi += part[2 * i].y; pr += part[2 * i + 1].x; pi += part[2 * i + 1].y; }
        // grad = f * S with f = 0.5 (Y) or 0.5j (Z, X)   (core_op_matrix.py:284-389)
        double g_re, g_im;
        if (kind == 0) { g_re = 0.5 * gr; g_im = 0.5 * gi; } else { g_re = -0.5 * gi; g_im = 0.5 * gr; }
        // _delta_theta (core_op_matrix.py:833-850)
        const double d2n = dim * dim;
        double d1 = (-2.0 * (pr * g_re + pi * g_im)) / d2n;
        const double d2 = (-2.0 * (g_re * g_re + g_im * g_im) + 0.5 * (pr * pr + pi * pi)) / d2n;
        const double tol = 1.4901161193847656e-08, lr = 0.19634954084936207, maxdt = 0.78539816339744831;
        double dt;
        if (d2 < tol) { d1 /= fmax(fabs(d1), 1.0); dt = -lr * d1; } else { dt = -d1 / d2; }
        const double r = fabs(dt / maxdt);
        if (!(r <= 1.0)) dt = dt / r;
        const double t_old = theta_in[tindex], t_new = t_old + dt;
        double s, c;
        sincos(0.5 * t_old, &s, &c); sh[0] = c; sh[1] = s;
        sincos(0.5 * t_new, &s, &c); sh[2] = c; sh[3] = s;
        if (blockIdx.x == 0) theta_out[tindex] = t_new;
    }
    __syncthreads();
    const double co = sh[0], so = sh[1], cn = sh[2], sn = sh[3];
    const size_t h = (size_t)1 << hbit;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(g, hbit);
        cplx z0 = z[i0], z1 = z[i0 + h], w0 = w[i0], w1 = w[i0 + h];
        rot_pair(kind, z0, z1, co, so);  // z <- R(theta_old) z
        rot_pair(kind, w0, w1, cn, sn);  // w <- R(theta_new) w
        z[i0] = z0; z[i0 + h] = z1; w[i0] = w0; w[i0 + h] = w1;
    }
}

// CX (ent 0) or CZ (ent 1) on both operands
__global__ __launch_bounds__(256) void cd_entangle_kernel(cplx* __restrict__ w, cplx* __restrict__ z, size_t ngroups, int cbit,
                                                          int tbit, int ent) {
    const int lo = min(cbit, tbit), hi = max(cbit, tbit);
    const size_t ic = (size_t)1 << cbit, it = (size_t)1 << tbit;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(pair_index(g, lo), hi);
        if (ent == 0) {
            cplx t = w[i0 + ic]; w[i0 + ic] = w[i0 + ic + it]; w[i0 + ic + it] = t;
            t = z[i0 + ic]; z[i0 + ic] = z[i0 + ic + it]; z[i0 + ic + it] = t;
        } else {
            cplx t = w[i0 + ic + it]; w[i0 + ic + it] = make_double2(-t.x, -t.y);
            t = z[i0 + ic + it]; z[i0 + ic + it] = make_double2(-t.x, -t.y);
        }
    }
}

// ---- the whole sweep -- or many sweeps -- as ONE persistent launch ------------------------------------------------------
// Up to 6 qubits the two d x d operands of the walk fit one workgroup's LDS (2 x 16 KiB at d = 32, 2 x 64 KiB at d = 64), so a
// workgroup per lane holds w and z for the whole walk: no launch per parameter, no HBM round trip between parameters.  Lanes
// are independent problems (random restarts of one ansatz, different targets).  Per sweep and lane:
//   theta -> (cos, sin) of every half angle (all threads in parallel);
//   z <- target (HBM -> LDS), z <- V(theta)^H z gate by gate in LDS (v_dagger_mul_mat, core_op_matrix.py:562-642), w <- I;
//   the Gauss-Seidel walk of core_op_matrix.py:852-912: per parameter the two inner products in one pass over the pairs
//   (fixed-order reduction: lanes by butterfly, waves in order), the Newton / gradient step worked out by EVERY thread from
//   the same four wave partials (identical arithmetic, no broadcast), z rotated by the old angle, w by the new one;
//   fobj = 1 - |<w|z>|^2 / d^2 (:917).
// Consecutive parameters on the same qubit touch the same pairs of the same threads: the barrier between them is skipped.
struct CdStep {
    int32_t kind;     // 0 Ry, 1 Rz, 2 Rx: one parameter; 3 CX, 4 CZ: the block's entangler on both operands
    int32_t hbit;     // address bit of the rotated qubit / of the control
    int32_t hbit2;    // address bit of the target (entanglers)
    int32_t tindex;   // index of the parameter
};

__device__ __forceinline__ void cd_delta(int kind, double gr, double gi, double pr, double pi, double dim, double t_old, double& t_new) {
    // grad = f * S with f = 0.5 (Y) or 0.5j (Z, X)   (core_op_matrix.py:284-389)
    double g_re, g_im;
    if (kind == 0) { g_re = 0.5 * gr; g_im = 0.5 * gi; } else { g_re = -0.5 * gi; g_im = 0.5 * gr; }
    // _delta_theta (core_op_matrix.py:833-850)
    const double d2n = dim * dim;
    double d1 = (-2.0 * (pr * g_re + pi * g_im)) / d2n;
    const double d2 = (-2.0 * (g_re * g_re + g_im * g_im) + 0.5 * (pr * pr + pi * pi)) / d2n;
    const double tol = 1.4901161193847656e-08, lr = 0.19634954084936207, maxdt = 0.78539816339744831;
    double dt;
    if (d2 < tol) { d1 /= fmax(fabs(d1), 1.0); dt = -lr * d1; } else { dt = -d1 / d2; }
    const double r = fabs(dt / maxdt);
    if (!(r <= 1.0)) dt = dt / r;
    t_new = t_old + dt;
}

__device__ __forceinline__ void cd_entangle_lds(cplx* a, int npairs_half, int cbit, int tbit, int cz, int tid, int nthreads) {
    const int lo = min(cbit, tbit), hi = max(cbit, tbit);
    const int ic = 1 << cbit, it = 1 << tbit;
    for (int g = tid; g < npairs_half; g += nthreads) {
        const int i0 = (int)pair_index(pair_index((size_t)g, lo), hi);
        if (!cz) { const cplx t = a[i0 + ic]; a[i0 + ic] = a[i0 + ic + it]; a[i0 + ic + it] = t; }
        else { const cplx t = a[i0 + ic + it]; a[i0 + ic + it] = make_double2(-t.x, -t.y); }
    }
}

__global__ __launch_bounds__(256) void cd_persistent_kernel(const CdStep* __restrict__ prog, int nsteps, int nbits, int col_bits,
                                                            const cplx* __restrict__ target, size_t lane_stride, double* thetas, int T,
                                                            double* fobj, int nsweeps, int max_steps) {
    extern __shared__ double cd_lds[];
    const int N = 1 << nbits, npairs = N >> 1;
    cplx* w = reinterpret_cast<cplx*>(cd_lds);
    cplx* z = w + N;
    double* th = reinterpret_cast<double*>(z + N);
    double2* cs = reinterpret_cast<double2*>(th + ((T + 1) & ~1));
    double* red = reinterpret_cast<double*>(cs + T);            // [4 waves][4]
    const int tid = threadIdx.x, lane = blockIdx.x, wave = tid >> 6;
    const double dim = (double)(1 << (nbits - col_bits));
    const int cmask = (1 << col_bits) - 1;
    double* my_thetas = thetas + (size_t)lane * T;
    const cplx* y = target + (size_t)lane * lane_stride;
    for (int t = tid; t < T; t += 256) th[t] = my_thetas[t];
    __syncthreads();
    for (int sweep = 0; sweep < nsweeps; ++sweep) {
        for (int t = tid; t < T; t += 256) { double s, c; sincos(0.5 * th[t], &s, &c); cs[t] = make_double2(c, s); }
        for (int e = tid; e < N; e += 256) {
            z[e] = y[e];
            w[e] = make_double2(((e >> col_bits) == (e & cmask)) ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        // z <- V^H z: the gates of the walk in reverse order, every rotation inverted
        for (int i = nsteps - 1; i >= 0; --i) {
            const CdStep st = prog[i];
            if (st.kind >= 3) {
                cd_entangle_lds(z, N >> 2, st.hbit, st.hbit2, st.kind == 4, tid, 256);
            } else {
                const double2 c = cs[st.tindex];
                const int h = 1 << st.hbit;
                for (int g = tid; g < npairs; g += 256) {
                    const int i0 = (int)pair_index((size_t)g, st.hbit);
                    cplx a0 = z[i0], a1 = z[i0 + h];
                    rot_pair(st.kind, a0, a1, c.x, -c.y);
                    z[i0] = a0; z[i0 + h] = a1;
                }
            }
            // the next (earlier) gate pairs the same elements in the same threads when it acts on the same qubit
            if (!(i > 0 && st.kind < 3 && prog[i - 1].kind < 3 && prog[i - 1].hbit == st.hbit)) __syncthreads();
        }
        // the walk
        int done = 0;
        for (int i = 0; i < nsteps; ++i) {
            const CdStep st = prog[i];
            if (st.kind >= 3) {
                cd_entangle_lds(z, N >> 2, st.hbit, st.hbit2, st.kind == 4, tid, 256);
                cd_entangle_lds(w, N >> 2, st.hbit, st.hbit2, st.kind == 4, tid, 256);
                __syncthreads();
                continue;
            }
            if (max_steps >= 0 && done >= max_steps) break;    // (tests: stop after a given number of parameter steps)
            ++done;
            const int h = 1 << st.hbit, kind = st.kind;
            double gr = 0, gi = 0, pr = 0, pi = 0;
            for (int g = tid; g < npairs; g += 256) {
                const int i0 = (int)pair_index((size_t)g, st.hbit);
                const cplx w0 = w[i0], w1 = w[i0 + h], z0 = z[i0], z1 = z[i0 + h];
                const double c00r = w0.x * z0.x + w0.y * z0.y, c00i = w0.x * z0.y - w0.y * z0.x;
                const double c11r = w1.x * z1.x + w1.y * z1.y, c11i = w1.x * z1.y - w1.y * z1.x;
                pr += c00r + c11r;
                pi += c00i + c11i;
                if (kind == 1) {
                    gr += c00r - c11r;
                    gi += c00i - c11i;
                } else {
                    const double c01r = w0.x * z1.x + w0.y * z1.y, c01i = w0.x * z1.y - w0.y * z1.x;
                    const double c10r = w1.x * z0.x + w1.y * z0.y, c10i = w1.x * z0.y - w1.y * z0.x;
                    if (kind == 0) { gr += c01r - c10r; gi += c01i - c10i; } else { gr += c01r + c10r; gi += c01i + c10i; }
                }
            }
            gr = wsum(gr); gi = wsum(gi); pr = wsum(pr); pi = wsum(pi);
            if ((tid & 63) == 0) { red[4 * wave] = gr; red[4 * wave + 1] = gi; red[4 * wave + 2] = pr; red[4 * wave + 3] = pi; }
            __syncthreads();
            gr = (red[0] + red[4]) + (red[8] + red[12]);
            gi = (red[1] + red[5]) + (red[9] + red[13]);
            pr = (red[2] + red[6]) + (red[10] + red[14]);
            pi = (red[3] + red[7]) + (red[11] + red[15]);
            const double t_old = th[st.tindex];
            double t_new;
            cd_delta(kind, gr, gi, pr, pi, dim, t_old, t_new);
            const double2 co = cs[st.tindex];
            double sn, cn;
            sincos(0.5 * t_new, &sn, &cn);
            for (int g = tid; g < npairs; g += 256) {
                const int i0 = (int)pair_index((size_t)g, st.hbit);
                cplx z0 = z[i0], z1 = z[i0 + h], w0 = w[i0], w1 = w[i0 + h];
                rot_pair(kind, z0, z1, co.x, co.y);     // z <- R(theta_old) z
                rot_pair(kind, w0, w1, cn, sn);         // w <- R(theta_new) w
                z[i0] = z0; z[i0 + h] = z1; w[i0] = w0; w[i0 + h] = w1;
            }
            // red / th are read by everybody before this point of the NEXT step's first barrier, so one barrier here orders
            // both the scratch and -- when the next step pairs other elements -- the operands
            __syncthreads();
            if (tid == 0) th[st.tindex] = t_new;   // (this step's entry is not read again before the next sweep's barriers)
        }
        // fobj = 1 - |<w|z> / d|^2
        double pr = 0, pi = 0;
        for (int e = tid; e < N; e += 256) {
            const cplx a = w[e], b = z[e];
            pr += a.x * b.x + a.y * b.y;
            pi += a.x * b.y - a.y * b.x;
        }
        pr = wsum(pr); pi = wsum(pi);
        if ((tid & 63) == 0) { red[4 * wave] = pr; red[4 * wave + 1] = pi; }
        __syncthreads();
        if (tid == 0) {
            const double a = (red[0] + red[4]) + (red[8] + red[12]), b = (red[1] + red[5]) + (red[9] + red[13]);
            fobj[(size_t)lane * nsweeps + sweep] = 1.0 - (a * a + b * b) / (dim * dim);
        }
        __syncthreads();
    }
    for (int t = tid; t < T; t += 256) my_thetas[t] = th[t];
}

size_t cd_persistent_lds_bytes(int nbits, int T) {
    return ((size_t)2 << nbits) * sizeof(cplx) + (size_t)((T + 1) & ~1) * sizeof(double) + (size_t)T * sizeof(double2) + 16 * sizeof(double);
}

hipError_t launch_cd_persistent(const void* prog, int nsteps, int nbits, int col_bits, const void* target, size_t lane_stride, double* thetas,
                                int T, double* fobj, int nsweeps, int max_steps, int batch, hipStream_t s) {
    const size_t lds = cd_persistent_lds_bytes(nbits, T);
    static size_t granted = 0;
    if (lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cd_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted = lds;
    }
    cd_persistent_kernel<<<batch, 256, lds, s>>>(static_cast<const CdStep*>(prog), nsteps, nbits, col_bits, static_cast<const cplx*>(target),
                                                 lane_stride, thetas, T, fobj, nsweeps, max_steps);
    return hipGetLastError();
}

static unsigned cd_blocks(size_t items) { return (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (items + 255) / 256)); }

int cd_num_parts(size_t npairs) { return (int)cd_blocks(npairs); }

hipError_t launch_cd_dot(const void* w, const void* z, size_t npairs, int hbit, int kind, void* part, hipStream_t s) {
    cd_dot_kernel<<<cd_blocks(npairs), 256, 0, s>>>(static_cast<const cplx*>(w), static_cast<const cplx*>(z), npairs, hbit, kind,
                                                     static_cast<cplx*>(part));
    return hipGetLastError();
}
hipError_t launch_cd_update(void* w, void* z, size_t npairs, int hbit, int kind, const void* part, int nparts,
                            const double* theta_in, double* theta_out, int tindex, double dim, hipStream_t s) {
    cd_update_kernel<<<cd_blocks(npairs), 256, 0, s>>>(static_cast<cplx*>(w), static_cast<cplx*>(z), npairs, hbit, kind,
                                                        static_cast<const cplx*>(part), nparts, theta_in, theta_out, tindex, dim);
    return hipGetLastError();
}
hipError_t launch_cd_entangle(void* w, void* z, size_t ngroups, int cbit, int tbit, int ent, hipStream_t s) {
    cd_entangle_kernel<<<cd_blocks(ngroups), 256, 0, s>>>(static_cast<cplx*>(w), static_cast<cplx*>(z), ngroups, cbit, tbit, ent);
    return hipGetLastError();
}

}  // namespace aqc
