// gfx950 (MI355X / CDNA4) kernels of the fidelity/gradient path.
//
// Stage kernels: one workgroup = one tile of 2^k complex128 amplitudes (those that differ
// only in the stage's k local address bits) of one batch lane.  The tile is gathered from
// HBM into LDS in runs of 2^low_bits contiguous elements (16 B per lane, coalesced), every
// gate group of the stage is applied in LDS, the per-parameter inner products
// 0.5j<P w|z> are reduced in flight (wave-64 butterfly -> per-wave LDS slot -> fixed-order
// sum, no float atomics => run-to-run identical), and the tile is scattered back.
//
// All arithmetic is complex fp64 on the vector ALU (the path is ~1 flop/byte per gate group
// before fusion and LDS/VALU bound after it; fp64 MFMA has no higher peak on gfx950).
#include <hip/hip_runtime.h>

#include "aqc_device.h"
#include "aqc_launch.h"
#include "aqc_math.h"

namespace aqc {

template <int ENT>
__device__ __forceinline__ void rs2(cplx& a0, cplx& a1, double c, double s) {
    if (ENT == 0) rx2(a0, a1, c, s); else rz2(a0, a1, c, s);
}
// amplitudes of a 2-qubit group are indexed a[2*cbit + tbit]
template <int ENT>
__device__ __forceinline__ void entangle(cplx* a, double c4, double s4) {
    if (ENT == 0) { const cplx t = a[2]; a[2] = a[3]; a[3] = t; }          // CX
    else if (ENT == 1) { a[3].x = -a[3].x; a[3].y = -a[3].y; }             // CZ
    else { a[3] = cmul(a[3], c4, s4); }                                     // CP(theta4)
}

// Unit block (C (x) T) CG, forward order  (core_operations.py:686-708)
template <int ENT>
__device__ __forceinline__ void block_fwd(cplx* a, const double* cf, int flags) {
    if (flags & 1) { rz2(a[0], a[2], kR, -kR); rz2(a[1], a[3], kR, -kR); }
    entangle<ENT>(a, cf[8], cf[9]);
    ry2(a[0], a[2], cf[0], cf[1]); ry2(a[1], a[3], cf[0], cf[1]);
    rz2(a[0], a[2], cf[2], cf[3]); rz2(a[1], a[3], cf[2], cf[3]);
    ry2(a[0], a[1], cf[4], cf[5]); ry2(a[2], a[3], cf[4], cf[5]);
    rs2<ENT>(a[0], a[1], cf[6], cf[7]); rs2<ENT>(a[2], a[3], cf[6], cf[7]);
    if (flags & 2) { rz2(a[0], a[1], kR, kR); rz2(a[2], a[3], kR, kR); }
}
// Conjugate-transposed unit block  (core_operations.py:787-809)
template <int ENT>
__device__ __forceinline__ void block_inv(cplx* a, const double* cf, int flags) {
    if (flags & 2) { rz2(a[0], a[1], kR, -kR); rz2(a[2], a[3], kR, -kR); }
    rs2<ENT>(a[0], a[1], cf[6], -cf[7]); rs2<ENT>(a[2], a[3], cf[6], -cf[7]);
    ry2(a[0], a[1], cf[4], -cf[5]); ry2(a[2], a[3], cf[4], -cf[5]);
    rz2(a[0], a[2], cf[2], -cf[3]); rz2(a[1], a[3], cf[2], -cf[3]);
    ry2(a[0], a[2], cf[0], -cf[1]); ry2(a[1], a[3], cf[0], -cf[1]);
    entangle<ENT>(a, cf[8], -cf[9]);
    if (flags & 1) { rz2(a[0], a[2], kR, kR); rz2(a[1], a[3], kR, kR); }
}

struct TileCtx {
    size_t base;      // element offset of the tile inside the lane
    unsigned* dlo;    // LDS copies of the deposit tables
    unsigned* dhi;
};

__device__ __forceinline__ TileCtx tile_setup(const DevStage* st, unsigned* tables) {
    TileCtx t;
    t.dlo = tables;
    t.dhi = tables + 64;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) t.dlo[i] = st->dlo[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) t.dhi[i] = st->dhi[i];
    size_t base = 0;
    const unsigned tile = blockIdx.x;
    const int nub = st->nub;
    for (int i = 0; i < nub; ++i) base |= (size_t)((tile >> i) & 1u) << st->ubits[i];
    t.base = base;
    return t;
}

// Gate-group descriptors and their coefficients are staged in LDS in chunks: fetching them per group costs two
// dependent global loads (descriptor -> coefficient address), ~1.5 us that a single-wave workgroup cannot hide.
constexpr int kOpChunk = 32;
__device__ __forceinline__ void stage_ops(const StageArgs& a, const DevStage* st, const double* coef, int first,
                                          DevOp* s_ops, double* s_cf) {
    const int cnt = min(kOpChunk, st->nops - first);
    const DevOp* ops = a.ops + st->op_begin + first;
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) s_ops[t] = ops[t];
    for (int t = threadIdx.x; t < cnt * 10; t += blockDim.x) {
        const int o = t / 10;
        s_cf[t] = coef[(size_t)ops[o].coef * kCoefStride + (t - 10 * o)];
    }
}
__device__ __forceinline__ DevOp uniform_op(const DevOp& v) {   // same value in every lane -> scalar registers
    DevOp o;
    o.type = __builtin_amdgcn_readfirstlane(v.type); o.p0 = __builtin_amdgcn_readfirstlane(v.p0);
    o.p1 = __builtin_amdgcn_readfirstlane(v.p1);     o.flags = __builtin_amdgcn_readfirstlane(v.flags);
    o.coef = __builtin_amdgcn_readfirstlane(v.coef); o.slot = __builtin_amdgcn_readfirstlane(v.slot);
    o.jblock = __builtin_amdgcn_readfirstlane(v.jblock); o.pad = 0;
    return o;
}

// ------------------------------------------------------------------------------------------
// V / V^H on one vector per lane
// ------------------------------------------------------------------------------------------
template <int ENT, bool INV>
__global__ __launch_bounds__(512) void apply_stage_kernel(StageArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage* st = a.stage;
    const int k = st->k;
    const unsigned tsize = 1u << k;
    cplx* tile = reinterpret_cast<cplx*>(smem);
    unsigned* tables = reinterpret_cast<unsigned*>(smem + (size_t)tsize * sizeof(cplx));
    const TileCtx tc = tile_setup(st, tables);
    __syncthreads();

    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    const cplx* src = a.in0 + lane_off;
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x)
        tile[l] = src[tc.dlo[l & 63u] + tc.dhi[l >> 6]];

    const double* coef = a.coef + (size_t)blockIdx.y * a.ncoef * kCoefStride;
    const int nops = st->nops;
    DevOp* s_ops = reinterpret_cast<DevOp*>(tables + 320);
    double* s_cf = reinterpret_cast<double*>(s_ops + kOpChunk);
    for (int i = 0; i < nops; ++i) {
        const int ci = i & (kOpChunk - 1);
        if (ci == 0) {
            if (i) __syncthreads();   // every wave is done with the previous chunk
            stage_ops(a, st, coef, i, s_ops, s_cf);
        }
        __syncthreads();
        const DevOp op = uniform_op(s_ops[ci]);
        double cf[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) cf[j] = s_cf[ci * 10 + j];
        if (op.type == 1) {
            const int plo = min(op.p0, op.p1), phi = max(op.p0, op.p1);
            const unsigned ic = 1u << op.p0, it = 1u << op.p1;
            for (unsigned g = threadIdx.x; g < (tsize >> 2); g += blockDim.x) {
                const unsigned i0 = insert_zero(insert_zero(g, plo), phi);
                cplx v[4] = {tile[i0], tile[i0 + it], tile[i0 + ic], tile[i0 + ic + it]};
                if (INV) block_inv<ENT>(v, cf, op.flags); else block_fwd<ENT>(v, cf, op.flags);
                tile[i0] = v[0]; tile[i0 + it] = v[1]; tile[i0 + ic] = v[2]; tile[i0 + ic + it] = v[3];
            }
        } else {
            const unsigned h = 1u << op.p0;
            for (unsigned g = threadIdx.x; g < (tsize >> 1); g += blockDim.x) {
                const unsigned i0 = insert_zero(g, op.p0);
                cplx a0 = tile[i0], a1 = tile[i0 + h];
                if (INV) {  // (Rz Ry Rz)^H  (core_operations.py:812-818)
                    rz2(a0, a1, cf[0], -cf[1]); ry2(a0, a1, cf[2], -cf[3]); rz2(a0, a1, cf[4], -cf[5]);
                } else {    // Rz(t0) Ry(t1) Rz(t2), rightmost first  (core_operations.py:671-677)
                    rz2(a0, a1, cf[4], cf[5]); ry2(a0, a1, cf[2], cf[3]); rz2(a0, a1, cf[0], cf[1]);
                }
                tile[i0] = a0; tile[i0 + h] = a1;
            }
        }
    }
    __syncthreads();
    cplx* dst = a.out0 + lane_off;
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x)
        dst[tc.dlo[l & 63u] + tc.dhi[l >> 6]] = tile[l];
}

// ------------------------------------------------------------------------------------------
// forward w/z sweep with in-flight inner products  (core_operations.py:918-1019)
// ------------------------------------------------------------------------------------------
template <int ENT>
__global__ __launch_bounds__(512) void sweep_stage_kernel(StageArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage* st = a.stage;
    const int k = st->k;
    const unsigned tsize = 1u << k;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    cplx* tz = tw + tsize;
    unsigned* tables = reinterpret_cast<unsigned*>(tz + tsize);
    double* scratch = reinterpret_cast<double*>(tables + 320);  // [2][nwaves * 4 rows][10]: row partials of <=5 inner products
    const TileCtx tc = tile_setup(st, tables);
    __syncthreads();

    const int nwaves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    {
        const cplx* sw = a.in0 + lane_off;
        const cplx* sz = a.in1 + lane_off;
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
            const unsigned off = tc.dlo[l & 63u] + tc.dhi[l >> 6];
            tw[l] = sw[off];
            tz[l] = sz[off];
        }
    }
    const double* coef = a.coef + (size_t)blockIdx.y * a.ncoef * kCoefStride;
    cplx* partial = a.partial + (size_t)blockIdx.y * a.nslots * a.ntiles_max;
    constexpr int ND_BLOCK = (ENT == 2) ? 5 : 4;

    int pend_slot = -1, pend_nd = 0, pend_par = 0;
    auto flush = [&]() {  // fixed-order sum over the row partials of the previous group's inner products
        if (pend_slot >= 0 && (int)threadIdx.x < pend_nd) {
            const double* s = scratch + (size_t)pend_par * nwaves * 40 + 2 * threadIdx.x;
            double re = 0.0, im = 0.0;
            for (int r = 0; r < nwaves * 4; ++r) { re += s[r * 10]; im += s[r * 10 + 1]; }
            partial[(size_t)(pend_slot + threadIdx.x) * a.ntiles_max + blockIdx.x] = make_double2(re, im);
        }
    };

    const int nops = st->nops;
    DevOp* s_ops = reinterpret_cast<DevOp*>(scratch + (size_t)2 * nwaves * 40);
    double* s_cf = reinterpret_cast<double*>(s_ops + kOpChunk);
    for (int i = 0; i < nops; ++i) {
        const int ci = i & (kOpChunk - 1);
        if (ci == 0) {
            if (i) __syncthreads();   // every wave is done with the previous chunk
            stage_ops(a, st, coef, i, s_ops, s_cf);
        }
        __syncthreads();
        const DevOp op = uniform_op(s_ops[ci]);
        double cf[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) cf[j] = s_cf[ci * 10 + j];
        flush();
        cplx d[kSlotsPerGroup];
#pragma unroll
        for (int j = 0; j < kSlotsPerGroup; ++j) d[j] = make_double2(0.0, 0.0);
        bool dots;
        int nd;
        if (op.type == 1) {
            dots = (op.jblock >= a.from) && (op.jblock < a.to);
            nd = ND_BLOCK;
            const int plo = min(op.p0, op.p1), phi = max(op.p0, op.p1);
            const unsigned ic = 1u << op.p0, it = 1u << op.p1;
            const int flags = op.flags;
            for (unsigned g = threadIdx.x; g < (tsize >> 2); g += blockDim.x) {
                const unsigned i0 = insert_zero(insert_zero(g, plo), phi);
                cplx w[4] = {tw[i0], tw[i0 + it], tw[i0 + ic], tw[i0 + ic + it]};
                cplx z[4] = {tz[i0], tz[i0 + it], tz[i0 + ic], tz[i0 + ic + it]};
                if (flags & 1) {
                    rz2(w[0], w[2], kR, -kR); rz2(w[1], w[3], kR, -kR);
                    rz2(z[0], z[2], kR, -kR); rz2(z[1], z[3], kR, -kR);
                }
                if (ENT == 2) cmacc(d[4], w[3], z[3]);  // -i<P11 w|z>, pre-gate (core_op_matrix.py:430-477)
                entangle<ENT>(z, cf[8], cf[9]);
                entangle<ENT>(w, cf[8], cf[9]);
                // control qubit: Ry(t0), Rz(t1)
                ry2(w[0], w[2], cf[0], cf[1]); ry2(w[1], w[3], cf[0], cf[1]);
                ry2(z[0], z[2], cf[0], cf[1]); ry2(z[1], z[3], cf[0], cf[1]);
                cmacc(d[0], w[0], z[2]); cmsub(d[0], w[2], z[0]);
                cmacc(d[0], w[1], z[3]); cmsub(d[0], w[3], z[1]);
                rz2(w[0], w[2], cf[2], cf[3]); rz2(w[1], w[3], cf[2], cf[3]);
                rz2(z[0], z[2], cf[2], cf[3]); rz2(z[1], z[3], cf[2], cf[3]);
                cmacc(d[1], w[0], z[0]); cmsub(d[1], w[2], z[2]);
                cmacc(d[1], w[1], z[1]); cmsub(d[1], w[3], z[3]);
                // target qubit: Ry(t2), Rs(t3)
                ry2(w[0], w[1], cf[4], cf[5]); ry2(w[2], w[3], cf[4], cf[5]);
                ry2(z[0], z[1], cf[4], cf[5]); ry2(z[2], z[3], cf[4], cf[5]);
                cmacc(d[2], w[0], z[1]); cmsub(d[2], w[1], z[0]);
                cmacc(d[2], w[2], z[3]); cmsub(d[2], w[3], z[2]);
                rs2<ENT>(w[0], w[1], cf[6], cf[7]); rs2<ENT>(w[2], w[3], cf[6], cf[7]);
                rs2<ENT>(z[0], z[1], cf[6], cf[7]); rs2<ENT>(z[2], z[3], cf[6], cf[7]);
                if (ENT == 0) {  // <X w|z>
                    cmacc(d[3], w[1], z[0]); cmacc(d[3], w[0], z[1]);
                    cmacc(d[3], w[3], z[2]); cmacc(d[3], w[2], z[3]);
                } else {         // <Z w|z>
                    cmacc(d[3], w[0], z[0]); cmsub(d[3], w[1], z[1]);
                    cmacc(d[3], w[2], z[2]); cmsub(d[3], w[3], z[3]);
                }
                if (flags & 2) {
                    rz2(w[0], w[1], kR, kR); rz2(w[2], w[3], kR, kR);
                    rz2(z[0], z[1], kR, kR); rz2(z[2], z[3], kR, kR);
                }
                tw[i0] = w[0]; tw[i0 + it] = w[1]; tw[i0 + ic] = w[2]; tw[i0 + ic + it] = w[3];
                tz[i0] = z[0]; tz[i0 + it] = z[1]; tz[i0 + ic] = z[2]; tz[i0 + ic + it] = z[3];
            }
            // factors: dot_y -> 0.5, dot_z / dot_x -> 0.5j, cphase -> -1j
            d[0] = make_double2(0.5 * d[0].x, 0.5 * d[0].y);
            d[1] = make_double2(-0.5 * d[1].y, 0.5 * d[1].x);
            d[2] = make_double2(0.5 * d[2].x, 0.5 * d[2].y);
            d[3] = make_double2(-0.5 * d[3].y, 0.5 * d[3].x);
            d[4] = make_double2(d[4].y, -d[4].x);
        } else {
            dots = a.front != 0;
            nd = 3;
            const unsigned h = 1u << op.p0;
            for (unsigned g = threadIdx.x; g < (tsize >> 1); g += blockDim.x) {
                const unsigned i0 = insert_zero(g, op.p0);
                cplx w0 = tw[i0], w1 = tw[i0 + h], z0 = tz[i0], z1 = tz[i0 + h];
                rz2(w0, w1, cf[4], cf[5]); rz2(z0, z1, cf[4], cf[5]);   // Rz(t2)
                cmacc(d[0], w0, z0); cmsub(d[0], w1, z1);
                ry2(w0, w1, cf[2], cf[3]); ry2(z0, z1, cf[2], cf[3]);   // Ry(t1)
                cmacc(d[1], w0, z1); cmsub(d[1], w1, z0);
                rz2(w0, w1, cf[0], cf[1]); rz2(z0, z1, cf[0], cf[1]);   // Rz(t0)
                cmacc(d[2], w0, z0); cmsub(d[2], w1, z1);
                tw[i0] = w0; tw[i0 + h] = w1; tz[i0] = z0; tz[i0 + h] = z1;
            }
            d[0] = make_double2(-0.5 * d[0].y, 0.5 * d[0].x);
            d[1] = make_double2(0.5 * d[1].x, 0.5 * d[1].y);
            d[2] = make_double2(-0.5 * d[2].y, 0.5 * d[2].x);
        }
        if (dots) {
            // transposing DPP butterfly over each row of 16 lanes; lanes 0..7 of a row hold its totals
            double* s = scratch + ((size_t)(i & 1) * nwaves * 4 + wave * 4 + (lane >> 4)) * 10;
            const double v[8] = {d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
            const double tot = reduce8_row(v, lane);
            if ((lane & 15) < 8) s[lane & 7] = tot;
            if (ENT == 2 && nd == 5) {   // CP: fifth inner product
                const double w8[8] = {d[4].x, d[4].y, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                const double t2 = reduce8_row(w8, lane);
                if ((lane & 15) < 2) s[8 + (lane & 7)] = t2;
            }
            pend_slot = op.slot; pend_nd = nd; pend_par = i & 1;
        } else {
            pend_slot = -1;
        }
    }
    __syncthreads();
    flush();
    {
        cplx* dw = a.out0 + lane_off;
        cplx* dz = a.out1 + lane_off;
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
            const unsigned off = tc.dlo[l & 63u] + tc.dhi[l >> 6];
            dw[off] = tw[l];
            dz[off] = tz[l];
        }
    }
}

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
// coefficient records from thetas: half-angle (cos, sin) pairs (+ full-angle pair for CP) and their
// lifting form (see aqc_device.h)
__device__ __forceinline__ double put_pair(double* cf, int j, double angle) {
    double s, c;
    sincos(angle, &s, &c);
    cf[2 * j] = c;
    cf[2 * j + 1] = s;
    double sg = 1.0;
    if (c < 0.0) { c = -c; s = -s; sg = -1.0; }
    cf[kLiftOffset + 2 * j] = -s / (1.0 + c);
    cf[kLiftOffset + 2 * j + 1] = s;
    return sg;
}
__global__ void coef_kernel(const double* thetas, double* coef, int n, int nblocks, int tpb, int batch) {
    const int ncoef = n + nblocks + 1;  // + one constant record: Rz(+-pi/2) of the Trotter decoration
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * ncoef) return;
    const int b = idx / ncoef, ci = idx % ncoef;
    const int T = 3 * n + tpb * nblocks;
    const double* th = thetas + (size_t)b * T;
    double* cf = coef + (size_t)idx * kCoefStride;
    for (int j = 0; j < kCoefStride; ++j) cf[j] = 0.0;
    double sg = 1.0;
    if (ci == n + nblocks) {
        put_pair(cf, 0, 0.78539816339744831);  // pi/4
        cf[2] = 1.0;
    } else if (ci < n) {
        for (int j = 0; j < 3; ++j) sg *= put_pair(cf, j, 0.5 * th[3 * ci + j]);
    } else {
        const double* t = th + 3 * n + (size_t)tpb * (ci - n);
        for (int j = 0; j < 4; ++j) sg *= put_pair(cf, j, 0.5 * t[j]);
        if (tpb == 5) put_pair(cf, 4, t[4]); else { cf[8] = 1.0; cf[9] = 0.0; }
    }
    cf[kSignOffset] = sg;
}
// lane sign = product of the record signs over the gate groups actually executed (the virtual trailing
// half-layer of a 2nd-order Trotter ansatz re-uses the first `tail` block records: their signs square away)
__global__ void sign_kernel(double* coef, int n, int nblocks, int tail) {
    const int ncoef = n + nblocks + 1, lane = threadIdx.x;
    double* base = coef + (size_t)blockIdx.x * ncoef * kCoefStride;
    int neg = 0;
    for (int ci = lane; ci < n + nblocks; ci += 64)
        if (!(ci >= n && ci < n + tail) && base[(size_t)ci * kCoefStride + kSignOffset] < 0.0) ++neg;
    for (int o = 32; o > 0; o >>= 1) neg += __shfl_xor(neg, o, 64);
    if (lane == 0) base[(size_t)(n + nblocks) * kCoefStride + 2] = (neg & 1) ? -1.0 : 1.0;
}

// grads[b][t] = sum over the (<=2) slots feeding theta t, over tiles, in a fixed order.
// `mirror` (may be null): a second copy of the result written straight into pinned host memory -- the one-call
// evaluation path uses it instead of a device-to-host copy node for small batches.
__global__ void finalize_kernel(const cplx* partial, const int* theta_slots, const int* slot_ntiles, cplx* grads,
                                int T, int nslots, int ntiles_max, int n, int tpb, int from, int to, int front, cplx* mirror) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const bool on = t < 3 * n ? (front != 0) : ((t - 3 * n) / tpb >= from && (t - 3 * n) / tpb < to);
    double re = 0.0, im = 0.0;
    if (on) {
        for (int s = 0; s < 2; ++s) {
            const int slot = theta_slots[2 * t + s];
            if (slot < 0) continue;
            const cplx* p = partial + ((size_t)b * nslots + slot) * ntiles_max;
            const int nt = slot_ntiles[slot];
            for (int i = lane; i < nt; i += 64) { re += p[i].x; im += p[i].y; }
        }
    }
    re = wave_sum(re);
    im = wave_sum(im);
    if (lane == 0) {
        grads[(size_t)b * T + t] = make_double2(re, im);
        if (mirror) mirror[(size_t)b * T + t] = make_double2(re, im);
    }
}

__global__ void scatter_one_kernel(cplx* buf, size_t lane_stride, int batch, const long long* elem) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) buf[(size_t)b * lane_stride + (size_t)elem[b]] = make_double2(1.0, 0.0);
}

// lane b: clear the two positions written last time, then buf[elem[2b]] = coef[2b], buf[elem[2b+1]] = coef[2b+1] (elem < 0: none)
__global__ void scatter_two_kernel(cplx* buf, size_t lane_stride, int batch, const long long* elem, const cplx* coef, long long* prev) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    cplx* x = buf + (size_t)b * lane_stride;
    for (int k = 0; k < 2; ++k)
        if (prev[2 * b + k] >= 0) x[(size_t)prev[2 * b + k]] = make_double2(0.0, 0.0);
    for (int k = 0; k < 2; ++k) {
        const long long e = elem[2 * b + k];
        if (e >= 0) x[(size_t)e] = coef[2 * b + k];
        prev[2 * b + k] = e;
    }
}

__global__ void set_identity_kernel(cplx* buf, size_t lane_stride, int dim, int pitch, int batch) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)batch * dim) {
        const size_t b = idx / dim, r = idx % dim;
        buf[b * lane_stride + r * pitch + r] = make_double2(1.0, 0.0);
    }
}

__global__ void gather_kernel(const cplx* buf, size_t lane_stride, const long long* elem, int count, int batch, cplx* out,
                              cplx* mirror) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < batch * count) {
        const int b = idx / count, i = idx % count;
        const cplx v = buf[(size_t)b * lane_stride + (size_t)elem[i]];
        out[idx] = v;
        if (mirror) mirror[idx] = v;   // pinned host copy (see finalize_kernel)
    }
}

// <a|b> per lane, stage 1: per-workgroup partials; stage 2 sums them with one wave.
__global__ __launch_bounds__(256) void vdot_partial_kernel(const cplx* a, const cplx* b, size_t lane_stride, size_t count,
                                                           cplx* part) {
    __shared__ cplx sm[4];
    const cplx* pa = a + (size_t)blockIdx.y * lane_stride;
    const cplx* pb = b + (size_t)blockIdx.y * lane_stride;
    cplx acc = make_double2(0.0, 0.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        cmacc(acc, pa[i], pb[i]);
    const double re = wave_sum(acc.x), im = wave_sum(acc.y);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = make_double2(re, im);
    __syncthreads();
    if (threadIdx.x == 0) {
        cplx s = sm[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { s.x += sm[w].x; s.y += sm[w].y; }
        part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
}
__global__ void vdot_final_kernel(const cplx* part, int nparts, cplx* out) {
    const cplx* p = part + (size_t)blockIdx.x * nparts;
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) { re += p[i].x; im += p[i].y; }
    re = wave_sum(re);
    im = wave_sum(im);
    if (threadIdx.x == 0) out[blockIdx.x] = make_double2(re, im);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
constexpr size_t kOpStageBytes = kOpChunk * (sizeof(DevOp) + 10 * sizeof(double));
size_t apply_lds_bytes(int k) { return ((size_t)16 << k) + 320 * sizeof(unsigned) + kOpStageBytes; }
size_t sweep_lds_bytes(int k, int threads) {
    return ((size_t)32 << k) + 320 * sizeof(unsigned) + (size_t)2 * (threads / 64) * 4 * 10 * sizeof(double) + kOpStageBytes;
}

template <typename K>
static hipError_t allow_big_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t init_kernels() {
    hipError_t e;
#define AQC_TRY(x) if ((e = (x)) != hipSuccess) return e
    AQC_TRY(allow_big_lds(apply_stage_kernel<0, false>)); AQC_TRY(allow_big_lds(apply_stage_kernel<0, true>));
    AQC_TRY(allow_big_lds(apply_stage_kernel<1, false>)); AQC_TRY(allow_big_lds(apply_stage_kernel<1, true>));
    AQC_TRY(allow_big_lds(apply_stage_kernel<2, false>)); AQC_TRY(allow_big_lds(apply_stage_kernel<2, true>));
    AQC_TRY(allow_big_lds(sweep_stage_kernel<0>)); AQC_TRY(allow_big_lds(sweep_stage_kernel<1>));
    AQC_TRY(allow_big_lds(sweep_stage_kernel<2>));
#undef AQC_TRY
    return hipSuccess;
}

hipError_t launch_apply(int ent, bool inverse, int ntiles, int batch, int threads, int k, hipStream_t s, const StageArgs& a) {
    const dim3 grid(ntiles, batch), block(threads);
    const size_t lds = apply_lds_bytes(k);
    switch (ent * 2 + (inverse ? 1 : 0)) {
        case 0: apply_stage_kernel<0, false><<<grid, block, lds, s>>>(a); break;
        case 1: apply_stage_kernel<0, true><<<grid, block, lds, s>>>(a); break;
        case 2: apply_stage_kernel<1, false><<<grid, block, lds, s>>>(a); break;
        case 3: apply_stage_kernel<1, true><<<grid, block, lds, s>>>(a); break;
        case 4: apply_stage_kernel<2, false><<<grid, block, lds, s>>>(a); break;
        default: apply_stage_kernel<2, true><<<grid, block, lds, s>>>(a); break;
    }
    return hipGetLastError();
}

hipError_t launch_sweep(int ent, int ntiles, int batch, int threads, int k, hipStream_t s, const StageArgs& a) {
    const dim3 grid(ntiles, batch), block(threads);
    const size_t lds = sweep_lds_bytes(k, threads);
    switch (ent) {
        case 0: sweep_stage_kernel<0><<<grid, block, lds, s>>>(a); break;
        case 1: sweep_stage_kernel<1><<<grid, block, lds, s>>>(a); break;
        default: sweep_stage_kernel<2><<<grid, block, lds, s>>>(a); break;
    }
    return hipGetLastError();
}

hipError_t launch_coef(const double* thetas, double* coef, int n, int nblocks, int tpb, int tail, int batch, hipStream_t s) {
    const int total = batch * (n + nblocks + 1);
    coef_kernel<<<(total + 127) / 128, 128, 0, s>>>(thetas, coef, n, nblocks, tpb, batch);
    sign_kernel<<<batch, 64, 0, s>>>(coef, n, nblocks, tail);
    return hipGetLastError();
}

hipError_t launch_finalize(const void* partial, const int* theta_slots, const int* slot_ntiles, void* grads, int T,
                           int nslots, int ntiles_max, int n, int tpb, int from, int to, int front, int batch,
                           hipStream_t s, void* mirror) {
    finalize_kernel<<<dim3(T, batch), 64, 0, s>>>(static_cast<const cplx*>(partial), theta_slots, slot_ntiles,
                                                   static_cast<cplx*>(grads), T, nslots, ntiles_max, n, tpb, from, to, front,
                                                   static_cast<cplx*>(mirror));
    return hipGetLastError();
}

hipError_t launch_scatter_one(void* buf, size_t lane_stride, int batch, const long long* elem, hipStream_t s) {
    scatter_one_kernel<<<(batch + 63) / 64, 64, 0, s>>>(static_cast<cplx*>(buf), lane_stride, batch, elem);
    return hipGetLastError();
}

hipError_t launch_scatter_two(void* buf, size_t lane_stride, int batch, const long long* elem, const void* coef, long long* prev, hipStream_t s) {
    scatter_two_kernel<<<(batch + 63) / 64, 64, 0, s>>>(static_cast<cplx*>(buf), lane_stride, batch, elem, static_cast<const cplx*>(coef), prev);
    return hipGetLastError();
}

hipError_t launch_set_identity(void* buf, size_t lane_stride, int dim, int pitch, int batch, hipStream_t s) {
    const size_t total = (size_t)batch * dim;
    set_identity_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(buf), lane_stride, dim, pitch, batch);
    return hipGetLastError();
}

hipError_t launch_gather(const void* buf, size_t lane_stride, const long long* elem, int count, int batch, void* out,
                         hipStream_t s, void* mirror) {
    const int total = batch * count;
    gather_kernel<<<(total + 127) / 128, 128, 0, s>>>(static_cast<const cplx*>(buf), lane_stride, elem, count, batch,
                                                      static_cast<cplx*>(out), static_cast<cplx*>(mirror));
    return hipGetLastError();
}

hipError_t launch_vdot(const void* a, const void* b, size_t lane_stride, size_t count, int batch, void* part, int nparts,
                       void* out, hipStream_t s) {
    vdot_partial_kernel<<<dim3(nparts, batch), 256, 0, s>>>(static_cast<const cplx*>(a), static_cast<const cplx*>(b),
                                                            lane_stride, count, static_cast<cplx*>(part));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    vdot_final_kernel<<<batch, 64, 0, s>>>(static_cast<const cplx*>(part), nparts, static_cast<cplx*>(out));
    return hipGetLastError();
}

}  // namespace aqc
