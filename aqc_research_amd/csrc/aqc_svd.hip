// Complex SVD by one-sided (Hestenes) Jacobi on the device, for the truncated 2-qubit gates of the MPS engine
// (the step qiskit-aer performs inside AerSimulator(method="matrix_product_state"), reached from
// mps_dot_objective.py:380-468 / mps_operations.py:252-257).  The work matrix W is column-major (rows x cols,
// every column contiguous); plane rotations make its columns mutually orthogonal and are accumulated in the
// column-major unitary V (cols x cols): on exit W = A V, sigma_j = |W_j|, so A = sum_j W_j V_j^H.
// One launch = one round of the round-robin tournament: cols/2 disjoint column pairs, one workgroup per pair.
// Everything is deterministic: fixed pair order, fixed-order block reductions, no float atomics.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"
#include "aqc_math.h"

namespace aqc {

constexpr int kSvdThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = kSvdThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// One wave per column pair (4 pairs per workgroup): no barriers, the four sums go through one shuffle butterfly.
__global__ __launch_bounds__(kSvdThreads) void jacobi_round_kernel(cplx* __restrict__ W, int rows, cplx* __restrict__ V, int cols,
                                                                   const int2* __restrict__ pairs, int npairs, double tol,
                                                                   int* __restrict__ rotations) {
    const int lane = threadIdx.x & 63, pair = blockIdx.x * (kSvdThreads / 64) + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int2 pq = pairs[pair];
    if (pq.x < 0 || pq.y < 0 || pq.x >= cols || pq.y >= cols) return;   // bye of an odd tournament
    cplx* wp = W + (size_t)pq.x * rows;
    cplx* wq = W + (size_t)pq.y * rows;
    double a = 0.0, b = 0.0, gr = 0.0, gi = 0.0;
    for (int i = lane; i < rows; i += 64) {
        const cplx x = wp[i], y = wq[i];
        a += x.x * x.x + x.y * x.y;
        b += y.x * y.x + y.y * y.y;
        gr += x.x * y.x + x.y * y.y;      // conj(x) * y
        gi += x.x * y.y - x.y * y.x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {   // xor butterfly: every lane ends with the same totals, fixed order
        a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64);
        gr += __shfl_xor(gr, off, 64); gi += __shfl_xor(gi, off, 64);
    }
    const double g2 = gr * gr + gi * gi;
    if (g2 <= tol * tol * a * b || g2 == 0.0) return;                    // already orthogonal (or a zero column)
    if (lane == 0) atomicAdd(rotations, 1);
    const double g = sqrt(g2);
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    const double er = gr / g, ei = -gi / g;                              // e^{-i phi}, phi = arg(gamma)
    // (x, y) <- (c x - s e^{-i phi} y,  s x + c e^{-i phi} y): makes the pair's Gram matrix diagonal
    cplx* vp = V + (size_t)pq.x * cols;
    cplx* vq = V + (size_t)pq.y * cols;
    for (int i = lane; i < rows + cols; i += 64) {
        cplx* xp = i < rows ? wp + i : vp + (i - rows);
        cplx* yp = i < rows ? wq + i : vq + (i - rows);
        const cplx x = *xp, y0 = *yp;
        const cplx y = make_double2(y0.x * er - y0.y * ei, y0.x * ei + y0.y * er);
        *xp = make_double2(c * x.x - s * y.x, c * x.y - s * y.y);
        *yp = make_double2(s * x.x + c * y.x, s * x.y + c * y.y);
    }
}

__global__ void svd_identity_kernel(cplx* V, int cols) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)cols * cols) V[i] = make_double2((i / cols) == (i % cols) ? 1.0 : 0.0, 0.0);
}

__global__ __launch_bounds__(kSvdThreads) void svd_norms_kernel(const cplx* __restrict__ W, int rows, double* __restrict__ sigma) {
    __shared__ double red[kSvdThreads];
    const cplx* w = W + (size_t)blockIdx.x * rows;
    double a = 0.0;
    for (int i = threadIdx.x; i < rows; i += kSvdThreads) a += w[i].x * w[i].x + w[i].y * w[i].y;
    a = block_sum(a, red);
    if (threadIdx.x == 0) sigma[blockIdx.x] = sqrt(a);
}

// Whole SVD in ONE launch when the work matrix and V fit into the LDS of one workgroup (rows, cols <= 64: bond
// dimensions up to 32): 32 half-waves, one column pair each per round, __syncthreads between rounds, the sweep
// loop and the convergence test inside the kernel.  Replaces ~500 launches of jacobi_round_kernel.
constexpr int kSmallMax = 64;
__global__ __launch_bounds__(1024) void jacobi_small_kernel(cplx* __restrict__ W, int rows, cplx* __restrict__ V, int cols,
                                                            const int2* __restrict__ pairs, int rounds, int per_round, double tol,
                                                            int max_sweeps, int* __restrict__ sweeps_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* sw = reinterpret_cast<cplx*>(smem);            // [cols][rows]
    cplx* sv = sw + (size_t)cols * rows;                  // [cols][cols]
    __shared__ int rotated;
    const int tid = threadIdx.x, grp = tid >> 5, lane = tid & 31;
    for (int i = tid; i < rows * cols; i += blockDim.x) sw[i] = W[i];
    for (int i = tid; i < cols * cols; i += blockDim.x) sv[i] = make_double2((i / cols) == (i % cols) ? 1.0 : 0.0, 0.0);
    __syncthreads();
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            const int2 pq = grp < per_round ? pairs[r * per_round + grp] : make_int2(-1, -1);
            if (pq.x >= 0 && pq.y >= 0 && pq.x < cols && pq.y < cols) {
                cplx* wp = sw + (size_t)pq.x * rows;
                cplx* wq = sw + (size_t)pq.y * rows;
                double a = 0.0, b = 0.0, gr = 0.0, gi = 0.0;
                for (int i = lane; i < rows; i += 32) {
                    const cplx x = wp[i], y = wq[i];
                    a += x.x * x.x + x.y * x.y;
                    b += y.x * y.x + y.y * y.y;
                    gr += x.x * y.x + x.y * y.y;
                    gi += x.x * y.y - x.y * y.x;
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {   // butterfly inside the half-wave: every lane gets the totals
                    a += __shfl_xor(a, off, 32); b += __shfl_xor(b, off, 32);
                    gr += __shfl_xor(gr, off, 32); gi += __shfl_xor(gi, off, 32);
                }
                const double g2 = gr * gr + gi * gi;
                if (g2 > tol * tol * a * b && g2 != 0.0) {
                    if (lane == 0) atomicAdd(&rotated, 1);
                    const double g = sqrt(g2);
                    const double zeta = (b - a) / (2.0 * g);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    const double er = gr / g, ei = -gi / g;
                    cplx* vp = sv + (size_t)pq.x * cols;
                    cplx* vq = sv + (size_t)pq.y * cols;
                    for (int i = lane; i < rows + cols; i += 32) {
                        cplx* xp = i < rows ? wp + i : vp + (i - rows);
                        cplx* yp = i < rows ? wq + i : vq + (i - rows);
                        const cplx x = *xp, y0 = *yp;
                        const cplx y = make_double2(y0.x * er - y0.y * ei, y0.x * ei + y0.y * er);
                        *xp = make_double2(c * x.x - s * y.x, c * x.y - s * y.y);
                        *yp = make_double2(s * x.x + c * y.x, s * x.y + c * y.y);
                    }
                }
            }
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (any == 0) { ++sweep; break; }
    }
    for (int i = tid; i < rows * cols; i += blockDim.x) W[i] = sw[i];
    for (int i = tid; i < cols * cols; i += blockDim.x) V[i] = sv[i];
    if (tid == 0) *sweeps_out = sweep;
}

// Generic entry (aqc_svd): A row-major (m x n) -> Jacobi work matrix, and (W, V, order, sigma) -> U (m x k), Vh (k x n).
// mode 0 (n <= m): work = column-major A;  mode 1: work = column-major A^H (the conjugate of row-major A).
__global__ void svd_load_kernel(const cplx* __restrict__ a, int m, int n, int mode, cplx* __restrict__ work) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)m * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    const cplx v = a[idx];
    if (mode == 0) work[(size_t)j * m + i] = v;
    else work[idx] = make_double2(v.x, -v.y);
}
__global__ void svd_assemble_kernel(const cplx* __restrict__ W, const cplx* __restrict__ V, const int* __restrict__ ord,
                                    const double* __restrict__ sigma, int m, int n, int k, int mode, cplx* __restrict__ u,
                                    cplx* __restrict__ vh, double* __restrict__ s_sorted) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nu = (size_t)m * k, nv = (size_t)k * n;
    if (idx < nu) {
        const int i = (int)(idx / k), j = (int)(idx - (size_t)i * k), c = ord[j];
        const double inv = sigma[c] > 0.0 ? 1.0 / sigma[c] : 0.0;
        cplx v = mode == 0 ? W[(size_t)c * m + i] : V[(size_t)c * m + i];
        if (mode == 0) { v.x *= inv; v.y *= inv; }
        u[idx] = v;
    } else if (idx < nu + nv) {
        const size_t e = idx - nu;
        const int j = (int)(e / n), i = (int)(e - (size_t)j * n), c = ord[j];
        const double inv = sigma[c] > 0.0 ? 1.0 / sigma[c] : 0.0;
        cplx v = mode == 0 ? V[(size_t)c * n + i] : W[(size_t)c * n + i];
        if (mode == 1) { v.x *= inv; v.y *= inv; }
        vh[e] = make_double2(v.x, -v.y);
    } else if (idx < nu + nv + (size_t)k) {
        const int j = (int)(idx - nu - nv);
        s_sorted[j] = sigma[ord[j]];
    }
}
hipError_t launch_svd_load(const void* a, int m, int n, int mode, void* work, hipStream_t s) {
    const size_t total = (size_t)m * n;
    svd_load_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(a), m, n, mode, static_cast<cplx*>(work));
    return hipGetLastError();
}
hipError_t launch_svd_assemble(const void* W, const void* V, const int* ord, const double* sigma, int m, int n, int k, int mode, void* u, void* vh,
                               double* s_sorted, hipStream_t s) {
    const size_t total = (size_t)m * k + (size_t)k * n + k;
    svd_assemble_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(W), static_cast<const cplx*>(V), ord, sigma, m, n, k, mode,
                                                                        static_cast<cplx*>(u), static_cast<cplx*>(vh), s_sorted);
    return hipGetLastError();
}

bool svd_fits_small(int rows, int cols) { return rows <= kSmallMax && cols <= kSmallMax && cols >= 2; }
hipError_t launch_jacobi_small(void* W, int rows, void* V, int cols, const void* pairs, int rounds, int per_round, double tol, int max_sweeps,
                               int* sweeps_out, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * kSmallMax * kSmallMax * (int)sizeof(cplx));   // + the static flag < 160 KiB
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t lds = sizeof(cplx) * ((size_t)cols * rows + (size_t)cols * cols);
    jacobi_small_kernel<<<1, 1024, lds, s>>>(static_cast<cplx*>(W), rows, static_cast<cplx*>(V), cols, static_cast<const int2*>(pairs), rounds,
                                             per_round, tol, max_sweeps, sweeps_out);
    return hipGetLastError();
}

hipError_t launch_svd_identity(void* V, int cols, hipStream_t s) {
    const size_t total = (size_t)cols * cols;
    svd_identity_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(V), cols);
    return hipGetLastError();
}
hipError_t launch_jacobi_round(void* W, int rows, void* V, int cols, const void* pairs, int npairs, double tol, int* rotations, hipStream_t s) {
    jacobi_round_kernel<<<(npairs + kSvdThreads / 64 - 1) / (kSvdThreads / 64), kSvdThreads, 0, s>>>(
        static_cast<cplx*>(W), rows, static_cast<cplx*>(V), cols, static_cast<const int2*>(pairs), npairs, tol, rotations);
    return hipGetLastError();
}
hipError_t launch_svd_norms(const void* W, int rows, int cols, double* sigma, hipStream_t s) {
    svd_norms_kernel<<<cols, kSvdThreads, 0, s>>>(static_cast<const cplx*>(W), rows, sigma);
    return hipGetLastError();
}

}  // namespace aqc

// ---- pieces of a 2-qubit gate on two adjacent MPS sites ------------------------------------------------
namespace aqc {

struct Gate16 { cplx m[16]; };

// theta0: row-major (2 chi_l) x (2 chi_r), rows (a, l), columns (b, r) = T_q . [T_{q+1}[0] | T_{q+1}[1]].
// theta'[(a',l),(b',r)] = lam_left[l] * sum_{ab} G[2a'+b'][2a+b] theta0[(a,l),(b,r)], written as the Jacobi work
// matrix: mode 0 (columns <= rows) column-major theta'; mode 1 column-major theta'^H (= conj of row-major theta').
__global__ void mps_theta_kernel(const cplx* __restrict__ theta0, const double* __restrict__ lam_left, int chil, int chir, Gate16 g,
                                 int mode, cplx* __restrict__ work) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= chil * chir) return;
    const int l = idx / chir, r = idx - l * chir;
    const int m = 2 * chil, n = 2 * chir;
    const double sc = lam_left ? lam_left[l] : 1.0;
    cplx in[4], out[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const cplx v = theta0[(size_t)(a * chil + l) * n + b * chir + r];
            in[2 * a + b] = make_double2(sc * v.x, sc * v.y);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            re += g.m[4 * i + j].x * in[j].x - g.m[4 * i + j].y * in[j].y;
            im += g.m[4 * i + j].x * in[j].y + g.m[4 * i + j].y * in[j].x;
        }
        out[i] = make_double2(re, im);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const size_t row = a * chil + l, col = b * chir + r;
            const cplx v = out[2 * a + b];
            if (mode == 0) work[col * m + row] = v;
            else work[row * n + col] = make_double2(v.x, -v.y);
        }
}

// New site tensors from the converged Jacobi pair (W, V), keeping columns ord[0..k):
// mode 0: W = theta' Vj  =>  U S = W, V^H = Vj^H;   mode 1: W = theta'^H Vj  =>  U = Vj, S V^H = W^H.
// T_q'[a][l][j] = (U S)[(a,l), j] / lam_left[l];  T_{q+1}'[b][j][r] = V^H[j, (b,r)]  (lambda_{q+1} is already inside).
__global__ void mps_split_kernel(const cplx* __restrict__ W, const cplx* __restrict__ V, const int* __restrict__ ord,
                                 const double* __restrict__ sigma, const double* __restrict__ lam_left, int chil, int chir, int k,
                                 int mode, double rescale, cplx* __restrict__ tq, cplx* __restrict__ tq1) {
    const int m = 2 * chil, n = 2 * chir;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_left = (size_t)m * k, n_right = (size_t)k * n;
    if (idx < n_left) {
        const int row = (int)(idx / k), j = (int)(idx - (size_t)row * k);
        const int l = row % chil, c = ord[j];
        const double inv = 1.0 / (lam_left ? lam_left[l] : 1.0);
        cplx v;
        if (mode == 0) { v = W[(size_t)c * m + row]; v.x *= rescale * inv; v.y *= rescale * inv; }
        else { v = V[(size_t)c * m + row]; const double f = sigma[c] * rescale * inv; v.x *= f; v.y *= f; }
        tq[idx] = v;
    } else if (idx < n_left + n_right) {
        const size_t e = idx - n_left;                       // e = (b * k + j) * chir + r
        const int r = (int)(e % chir), bj = (int)(e / chir), j = bj % k, b = bj / k;
        const int c = ord[j], col = b * chir + r;
        cplx v;
        if (mode == 0) { v = V[(size_t)c * n + col]; v.y = -v.y; }
        else { v = W[(size_t)c * n + col]; const double f = 1.0 / sigma[c]; v.x *= f; v.y *= -f; }
        tq1[e] = v;
    }
}

// t[row][col] *= (mul ? lam[col] : 1 / lam[col])   -- import (Gamma -> Gamma lambda) and export of MPS tensors
__global__ void mps_colscale_kernel(cplx* t, const double* lam, size_t rows, int cols, int mul) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const double s = mul ? lam[i % cols] : 1.0 / lam[i % cols];
    t[i].x *= s;
    t[i].y *= s;
}

hipError_t launch_mps_theta(const void* theta0, const double* lam_left, int chil, int chir, const double* g16, int mode, void* work, hipStream_t s) {
    Gate16 g;
    for (int i = 0; i < 16; ++i) g.m[i] = make_double2(g16[2 * i], g16[2 * i + 1]);
    const int total = chil * chir;
    mps_theta_kernel<<<(total + 255) / 256, 256, 0, s>>>(static_cast<const cplx*>(theta0), lam_left, chil, chir, g, mode, static_cast<cplx*>(work));
    return hipGetLastError();
}
hipError_t launch_mps_split(const void* W, const void* V, const int* ord, const double* sigma, const double* lam_left, int chil, int chir,
                            int k, int mode, double rescale, void* tq, void* tq1, hipStream_t s) {
    const size_t total = (size_t)2 * chil * k + (size_t)k * 2 * chir;
    mps_split_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(W), static_cast<const cplx*>(V), ord, sigma, lam_left,
                                                                     chil, chir, k, mode, rescale, static_cast<cplx*>(tq), static_cast<cplx*>(tq1));
    return hipGetLastError();
}
hipError_t launch_mps_colscale(void* t, const double* lam, size_t rows, int cols, int mul, hipStream_t s) {
    const size_t total = rows * cols;
    mps_colscale_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(t), lam, rows, cols, mul);
    return hipGetLastError();
}

}  // namespace aqc
