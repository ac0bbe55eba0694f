// Gate-level building blocks of the reference exposed one call at a time:
// one elementary gate, or one inner product, over a (2^n x ncols) row-major complex128 array; a state vector
// is ncols = 1, qubit q is bit q of the row index (core_operations.py:34-43, core_op_matrix.py:56).  One pass
// over the data per call, like the reference's functions: these are the pieces the fused stage kernels are
// made of, kept for callers (and tests) that use them directly -- not the hot path.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"
#include "aqc_math.h"

namespace aqc {

__device__ __forceinline__ cplx gmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx gadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }

struct Gate4 { cplx m[16]; };   // 1-qubit gates use m[0..3] (row-major 2x2), 2-qubit ones the 4x4 on index 2*bit_c + bit_t

// dst <- (I (x) g (x) I) src  (gate2x2_mul_vec, core_operations.py:46-119; gate2x2_mul_mat, core_op_matrix.py:392-427)
__global__ void gate1q_kernel(const cplx* src, cplx* dst,   /* dst may be src */ size_t npairs, size_t ncols, int q, Gate4 g) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const size_t rp = p / ncols, c = p - rp * ncols;
    const size_t lo = rp & (((size_t)1 << q) - 1), r0 = ((rp >> q) << (q + 1)) | lo;
    const size_t i0 = r0 * ncols + c, i1 = i0 + (ncols << q);
    const cplx a0 = src[i0], a1 = src[i1];
    dst[i0] = gadd(gmul(g.m[0], a0), gmul(g.m[1], a1));
    dst[i1] = gadd(gmul(g.m[2], a0), gmul(g.m[3], a1));
}

// dst <- (4x4 on qubits (qc, qt)) src.  The entanglers are special cases the host encodes in the matrix:
// CX / CZ / CP (core_operations.py:422-558), the CP derivative i e^{i phi} |11><11| (:561-603), and a whole
// unit-block c00 (x) t + c11 (x) t.g (block_mul_vec, :354-419).
__global__ void gate2q_kernel(const cplx* src, cplx* dst,   /* dst may be src */ size_t nquads, size_t ncols, int qc, int qt, Gate4 g) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nquads) return;
    const size_t rq = p / ncols, c = p - rq * ncols;
    const int qlo = qc < qt ? qc : qt, qhi = qc < qt ? qt : qc;
    size_t r = ((rq >> qlo) << (qlo + 1)) | (rq & (((size_t)1 << qlo) - 1));
    r = ((r >> qhi) << (qhi + 1)) | (r & (((size_t)1 << qhi) - 1));
    const size_t sc = ncols << qc, st = ncols << qt, i00 = r * ncols + c;
    const cplx a[4] = {src[i00], src[i00 + st], src[i00 + sc], src[i00 + sc + st]};
    cplx o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cplx s = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s = gadd(s, gmul(g.m[4 * i + j], a[j]));
        o[i] = s;
    }
    dst[i00] = o[0]; dst[i00 + st] = o[1]; dst[i00 + sc] = o[2]; dst[i00 + sc + st] = o[3];
}

// Inner products: kind 0/1/2 = <X w|z>, <Y w|z>/(-i) form, <Z w|z> pairs on qubit q0 (dot_x/y/z, core_operations.py:
// 267-351; x/y/z_dot_mat, core_op_matrix.py:284-389); kind 3 = <P11(q0,q1) w|z> (derv_cphase, :430-477).  Per-block
// partial sums in a fixed order; gate_dot_final adds the blocks in order and applies 0.5j / 0.5 / 0.5j / -1j.
constexpr int kDotThreads = 256;
__global__ __launch_bounds__(kDotThreads) void gate_dot_kernel(const cplx* __restrict__ w, const cplx* __restrict__ z, size_t nitems,
                                                               size_t ncols, int kind, int q0, int q1, cplx* __restrict__ partial) {
    __shared__ cplx red[kDotThreads];
    cplx acc = make_double2(0.0, 0.0);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < nitems; p += (size_t)gridDim.x * blockDim.x) {
        const size_t rp = p / ncols, c = p - rp * ncols;
        if (kind < 3) {
            const size_t r0 = ((rp >> q0) << (q0 + 1)) | (rp & (((size_t)1 << q0) - 1));
            const size_t i0 = r0 * ncols + c, i1 = i0 + (ncols << q0);
            const cplx w0 = w[i0], w1 = w[i1], z0 = z[i0], z1 = z[i1];
            if (kind == 0) { cmacc(acc, w1, z0); cmacc(acc, w0, z1); }
            else if (kind == 1) { cmacc(acc, w0, z1); cmsub(acc, w1, z0); }
            else { cmacc(acc, w0, z0); cmsub(acc, w1, z1); }
        } else {
            const int qlo = q0 < q1 ? q0 : q1, qhi = q0 < q1 ? q1 : q0;
            size_t r = ((rp >> qlo) << (qlo + 1)) | (rp & (((size_t)1 << qlo) - 1));
            r = ((r >> qhi) << (qhi + 1)) | (r & (((size_t)1 << qhi) - 1));
            const size_t i11 = (r | ((size_t)1 << q0) | ((size_t)1 << q1)) * ncols + c;
            cmacc(acc, w[i11], z[i11]);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kDotThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = gadd(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void gate_dot_final(const cplx* __restrict__ partial, int nparts, int kind, cplx* __restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    cplx s = make_double2(0.0, 0.0);
    for (int i = 0; i < nparts; ++i) s = gadd(s, partial[i]);
    if (kind == 1) *out = make_double2(0.5 * s.x, 0.5 * s.y);          // 0.5 <..>   (dot_y: the i of Y is folded in)
    else if (kind == 3) *out = make_double2(s.y, -s.x);                 // -1j <P11 w|z>
    else *out = make_double2(-0.5 * s.y, 0.5 * s.x);                    // 0.5j <..>
}

hipError_t launch_gate1q(const void* src, void* dst, int n, size_t ncols, int q, const double* g /* 4 c128 */, hipStream_t s) {
    Gate4 m = {};
    for (int i = 0; i < 4; ++i) m.m[i] = make_double2(g[2 * i], g[2 * i + 1]);
    const size_t npairs = ((size_t)1 << (n - 1)) * ncols;
    gate1q_kernel<<<(unsigned)((npairs + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(src), static_cast<cplx*>(dst), npairs, ncols, q, m);
    return hipGetLastError();
}

hipError_t launch_gate2q(const void* src, void* dst, int n, size_t ncols, int qc, int qt, const double* g /* 16 c128 */, hipStream_t s) {
    Gate4 m;
    for (int i = 0; i < 16; ++i) m.m[i] = make_double2(g[2 * i], g[2 * i + 1]);
    const size_t nquads = ((size_t)1 << (n - 2)) * ncols;
    gate2q_kernel<<<(unsigned)((nquads + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(src), static_cast<cplx*>(dst), nquads, ncols, qc, qt, m);
    return hipGetLastError();
}

int gate_dot_parts(int n, size_t ncols, int kind) {
    const size_t nitems = ((size_t)1 << (n - (kind == 3 ? 2 : 1))) * ncols;
    const size_t blocks = (nitems + kDotThreads - 1) / kDotThreads;
    return (int)(blocks < 1024 ? (blocks ? blocks : 1) : 1024);
}

hipError_t launch_gate_dot(const void* w, const void* z, int n, size_t ncols, int kind, int q0, int q1, void* partial, void* out, hipStream_t s) {
    const size_t nitems = ((size_t)1 << (n - (kind == 3 ? 2 : 1))) * ncols;
    const int nparts = gate_dot_parts(n, ncols, kind);
    gate_dot_kernel<<<nparts, kDotThreads, 0, s>>>(static_cast<const cplx*>(w), static_cast<const cplx*>(z), nitems, ncols, kind, q0, q1, static_cast<cplx*>(partial));
    gate_dot_final<<<1, 64, 0, s>>>(static_cast<const cplx*>(partial), nparts, kind, static_cast<cplx*>(out));
    return hipGetLastError();
}

}  // namespace aqc
