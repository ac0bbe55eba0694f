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
