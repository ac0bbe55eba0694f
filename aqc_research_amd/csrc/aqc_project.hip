// Passes over a full-size state of the projected route (host side and the mathematics: aqc_ws_project.cpp).  One kernel, two uses:
//     out[keep, c] = sum_k conj(S[k, c]) Y[k, keep]                  per item = (lane, first-stage tile that holds the lhs state)
//   * Y = z (the checkpoint of V^H) or the target y, k = the first stage's local bits outside T, keep = the bits T, S = psi (w after
//     the first stage):  the projection of z onto the subspace the lhs state spans when it enters the later stages;
//   * Y = the target y, k = the bits T, keep = the first stage's local bits outside T, S = the virtual basis pattern after the later
//     stages' gates:  the tile of (later stages)^H y that the lhs state and the gathered amplitudes live on.
// A (2^cb x 2^nk) by (2^nk x 2^nkeep) complex product per item on the fp64 matrix cores; Y is read once, which is what a launch costs.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"

namespace aqc {

using cplx = double2;
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
// Y is streamed (read once): no reason to keep it in the caches the small operand lives in
__device__ __forceinline__ cplx pj_stream(const cplx* p) { const double2_t v = __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(p)); return make_double2(v.x, v.y); }

__device__ __forceinline__ double4_t pj_mfma(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned pj_off(const ProjMap& m, unsigned i) { return m.tab ? m.tab[i] : (i << m.shift); }
__device__ __forceinline__ size_t pj_tile_bits(const ProjArgs& a, int tile) {   // element offset of a first-stage tile: the values of its non-local bits
    size_t e = 0;
    for (int i = 0; i < a.nub0; ++i) e |= (size_t)((tile >> i) & 1) << a.ubits0[i];
    return e;
}

// Wave = RB blocks of 16 values of `keep` (the B operand's columns; MFMA rows = 16 values of c), K = the summed index: per block of
// 16 values of k a lane loads the 4 elements k = 16 kb + 4 (lane / 16) + jj of its row(s) of Y and of its row of S.
template <int RB, int NB>
__global__ __launch_bounds__(256) void project_kernel(const ProjArgs a) {
    const int item = blockIdx.y;
    if (item >= *a.nitems) return;
    const TileItem it = a.items[item];
    const size_t ebits = pj_tile_bits(a, it.tile);
    const int nkeep = 1 << a.keep_bits, ncb = 1 << a.cb, nkb = 1 << (a.k_bits - 4);
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r16 = l & 15, kg = l >> 4;
    const int rb0 = (blockIdx.x * 4 + wave) * RB;
    if (rb0 * 16 >= nkeep) return;
    const size_t vbase = ((size_t)it.lane * 2 + it.slot) << a.nvp;
    const size_t real_base = (size_t)it.lane * a.lane_stride;
    const cplx* ybase = a.y + real_base + (ebits & a.ff_mask);
    const cplx* sbase = a.s + (a.s_virtual ? vbase : real_base + ebits);
    cplx* obase = a.out + (a.out_virtual ? vbase : real_base + ebits);
    const cplx* yrow[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int keep = (rb0 + q) * 16 + r16;
        yrow[q] = ybase + (keep < nkeep ? pj_off(a.y_keep, keep) : 0u);
    }
    for (int nb0 = 0; nb0 * 16 < ncb; nb0 += NB) {
        const cplx* srow[NB];
        bool valid[NB];
        double4_t re[RB][NB], im[RB][NB];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int c = (nb0 + p) * 16 + r16;
            valid[p] = c < ncb;
            srow[p] = sbase + (valid[p] ? pj_off(a.s_c, c) : 0u);
#pragma unroll
            for (int q = 0; q < RB; ++q) { re[q][p] = double4_t{0.0, 0.0, 0.0, 0.0}; im[q][p] = re[q][p]; }
        }
        unsigned ylow[4], slow[4];   // offset of k = offset of its block of 16 + offset of its low four bits (disjoint address bits, or a shift)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { ylow[jj] = pj_off(a.y_k, (unsigned)(4 * kg + jj)); slow[jj] = pj_off(a.s_k, (unsigned)(4 * kg + jj)); }
        for (int kb = 0; kb < nkb; ++kb) {
            const unsigned yb = pj_off(a.y_k, (unsigned)kb * 16u), sb = pj_off(a.s_k, (unsigned)kb * 16u);
            unsigned yo[4], so[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { yo[jj] = yb + ylow[jj]; so[jj] = sb + slow[jj]; }
            cplx zv[RB][4], pv[NB][4];
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) zv[q][jj] = pj_stream(&yrow[q][yo[jj]]);
#pragma unroll
            for (int p = 0; p < NB; ++p)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pv[p][jj] = valid[p] ? srow[p][so[jj]] : make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int p = 0; p < NB; ++p)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {   // conj(s) y = (sr yr + si yi) + i (sr yi - si yr)
                        re[q][p] = pj_mfma(pv[p][jj].x, zv[q][jj].x, re[q][p]);
                        re[q][p] = pj_mfma(pv[p][jj].y, zv[q][jj].y, re[q][p]);
                        im[q][p] = pj_mfma(pv[p][jj].x, zv[q][jj].y, im[q][p]);
                        im[q][p] = pj_mfma(-pv[p][jj].y, zv[q][jj].x, im[q][p]);
                    }
        }
        // D[row = lane / 16 + 4 r][col = lane % 16]: row = c within its block, col = keep within its block
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int keep = (rb0 + q) * 16 + r16;
            if (keep >= nkeep) continue;
            cplx* orow = obase + pj_off(a.o_keep, keep);
#pragma unroll
            for (int p = 0; p < NB; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = (nb0 + p) * 16 + kg + 4 * r;
                    if (c < ncb) orow[pj_off(a.o_c, c)] = make_double2(re[q][p][r], im[q][p][r]);
                }
        }
    }
}

// The same product when the summed index k runs along contiguous memory of BOTH operands (the projection: k = the low address bits) and
// `keep` / c do not: a lane of the matrix-core layout would then fetch 16 bytes of a row of its own -- 64 separate 64-byte sectors per
// wave instruction, which is what the L1 can look up, not what the memory delivers.  Here a wave fetches its 16 x 16 block of Y and of S
// as 256-byte runs (lane = 4 rows x 16 consecutive k), hands them through LDS tiles (rows padded to 17 elements) and reads them
// back in the operand layout; the next block's loads are in flight while the 16 MFMAs of this one run.
constexpr int kPjRow = 17;   // elements per LDS row
// (S -- the same 16 x 16 block for the four waves of a workgroup -- is fetched once per workgroup, a quarter by each wave, into a
// shared tile that is double-buffered over the blocks of k: one workgroup barrier per block.)
__global__ __launch_bounds__(256) void project_staged_kernel(const ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char pj_smem[];
    const int item = blockIdx.y;
    if (item >= *a.nitems) return;   // (uniform over the workgroup: before any barrier)
    const TileItem it = a.items[item];
    const size_t ebits = pj_tile_bits(a, it.tile);
    const int nkeep = 1 << a.keep_bits, ncb = 1 << a.cb, nkb = 1 << (a.k_bits - 4);
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r16 = l & 15, kg = l >> 4;
    const int rb = blockIdx.x * 4 + wave;
    const bool active = rb * 16 < nkeep;   // (a wave without rows still fetches its quarter of S and meets the barriers)
    cplx* zt = reinterpret_cast<cplx*>(pj_smem) + (size_t)wave * (16 * kPjRow);
    cplx* pt = reinterpret_cast<cplx*>(pj_smem) + 4 * (16 * kPjRow);   // [2][16][kPjRow]
    const size_t vbase = ((size_t)it.lane * 2 + it.slot) << a.nvp;
    const size_t real_base = (size_t)it.lane * a.lane_stride;
    const cplx* ybase = a.y + real_base + (ebits & a.ff_mask) + r16;
    const cplx* sbase = a.s + (a.s_virtual ? vbase : real_base + ebits) + r16;
    cplx* obase = a.out + (a.out_virtual ? vbase : real_base + ebits);
    unsigned zoff[4];   // rows 4 j + kg of the wave's block of Y
#pragma unroll
    for (int j = 0; j < 4; ++j) zoff[j] = active ? pj_off(a.y_keep, (unsigned)(rb * 16 + 4 * j + kg)) : 0u;
    for (int nb = 0; nb * 16 < ncb; ++nb) {
        const int c_mine = nb * 16 + 4 * wave + kg;   // the row of S this lane fetches
        const bool pvalid = c_mine < ncb;
        const unsigned poff = pvalid ? pj_off(a.s_c, (unsigned)c_mine) : 0u;
        double4_t re = {0.0, 0.0, 0.0, 0.0}, im = re;
        cplx zc[4], zn[4], pc, pn;
        {
            const unsigned yb = pj_off(a.y_k, 0u), sb = pj_off(a.s_k, 0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) zc[j] = active ? pj_stream(ybase + zoff[j] + yb) : make_double2(0.0, 0.0);
            pc = pvalid ? sbase[poff + sb] : make_double2(0.0, 0.0);
        }
        for (int kb = 0; kb < nkb; ++kb) {
            if (kb + 1 < nkb) {
                const unsigned yb = pj_off(a.y_k, (unsigned)(kb + 1) * 16u), sb = pj_off(a.s_k, (unsigned)(kb + 1) * 16u);
#pragma unroll
                for (int j = 0; j < 4; ++j) zn[j] = active ? pj_stream(ybase + zoff[j] + yb) : make_double2(0.0, 0.0);
                pn = pvalid ? sbase[poff + sb] : make_double2(0.0, 0.0);
            }
            cplx* ptb = pt + (kb & 1) * (16 * kPjRow);
#pragma unroll
            for (int j = 0; j < 4; ++j) zt[(4 * j + kg) * kPjRow + r16] = zc[j];
            ptb[(4 * wave + kg) * kPjRow + r16] = pc;
            __syncthreads();   // S of this block is complete (its buffer is written again two blocks on, behind the next barrier)
            cplx zv[4], pv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { zv[jj] = zt[r16 * kPjRow + 4 * kg + jj]; pv[jj] = ptb[r16 * kPjRow + 4 * kg + jj]; }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {   // conj(s) y = (sr yr + si yi) + i (sr yi - si yr)
                re = pj_mfma(pv[jj].x, zv[jj].x, re);
                re = pj_mfma(pv[jj].y, zv[jj].y, re);
                im = pj_mfma(pv[jj].x, zv[jj].y, im);
                im = pj_mfma(-pv[jj].y, zv[jj].x, im);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) zc[j] = zn[j];
            pc = pn;
        }
        __syncthreads();   // (the next block of columns starts with buffer 0 again)
        if (active) {
            cplx* orow = obase + pj_off(a.o_keep, (unsigned)(rb * 16 + r16));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = nb * 16 + kg + 4 * r;
                if (c < ncb) orow[pj_off(a.o_c, (unsigned)c)] = make_double2(re[r], im[r]);
            }
        }
    }
}

hipError_t launch_project(const ProjArgs& a, hipStream_t s) {
    if (a.staged) {   // (the host has checked: the low four bits of k are address bits 0..3 of both operands)
        if (a.keep_bits < 4 || a.k_bits < 4 || a.cb < 0 || a.cb > 12 || a.batch < 1 || !a.y || !a.s || !a.out) return hipErrorInvalidValue;
        const int blocks = 1 << (a.keep_bits - 4);
        project_staged_kernel<<<dim3((unsigned)((blocks + 3) / 4), (unsigned)(2 * a.batch)), 256, 6 * 16 * kPjRow * sizeof(cplx), s>>>(a);
        return hipGetLastError();
    }
    if (a.keep_bits < 4 || a.k_bits < 4 || a.cb < 0 || a.cb > 12 || a.batch < 1 || !a.y || !a.s || !a.out) return hipErrorInvalidValue;
    const int blocks = 1 << (a.keep_bits - 4);
    const bool wide = blocks >= 16;   // four blocks of `keep` per wave: S is fetched once for four columns of Y
    const dim3 grid((unsigned)((blocks / (wide ? 4 : 1) + 3) / 4), (unsigned)(2 * a.batch));
    if (wide) {
        if (a.cb <= 4) project_kernel<4, 1><<<grid, 256, 0, s>>>(a);
        else project_kernel<4, 2><<<grid, 256, 0, s>>>(a);
    } else {
        if (a.cb <= 4) project_kernel<1, 1><<<grid, 256, 0, s>>>(a);
        else if (a.cb == 5) project_kernel<1, 2><<<grid, 256, 0, s>>>(a);
        else project_kernel<1, 4><<<grid, 256, 0, s>>>(a);
    }
    return hipGetLastError();
}

// The lhs state of the virtual register, M_0[i_T, c] = [i_T on the shared bits = c][i_T on the other touched bits = the tile's], and
// the item list of the virtual stage launches (one workgroup per item; workgroup 0 also the counts).
__global__ __launch_bounds__(256) void project_init_kernel(const ProjArgs a) {
    const int item = blockIdx.x;
    const int nitems = *a.nitems;
    if (item == 0) {
        if (threadIdx.x == 0) *a.vcount = nitems * a.ntiles_v;
        for (int b = threadIdx.x; b < a.batch; b += 256) a.vlane_parts[b] = a.lane_parts[b] * a.ntiles_v;
    }
    if (item >= nitems) return;
    const TileItem it = a.items[item];
    const unsigned ebits = (unsigned)pj_tile_bits(a, it.tile);
    for (int tt = threadIdx.x; tt < a.ntiles_v; tt += 256)
        a.vitems[(size_t)item * a.ntiles_v + tt] = TileItem{it.lane, it.slot * a.ntiles_v + tt, it.slot * a.ntiles_v + tt, 0};
    cplx* vm = a.vm + (((size_t)it.lane * 2 + it.slot) << a.nvp);
    const int nv = a.t + a.cb;
    for (unsigned v = threadIdx.x; v < (1u << nv); v += 256) {
        const unsigned i_t = v & ((1u << a.t) - 1), c = v >> a.t;
        const unsigned t_bits = a.off_t[i_t];
        const bool one = (t_bits & a.cb_mask) == a.off_cb[c] && (t_bits & a.tf_mask) == (ebits & a.tf_mask);
        vm[v] = make_double2(one ? 1.0 : 0.0, 0.0);
    }
}
hipError_t launch_project_init(const ProjArgs& a, hipStream_t s) {
    if (a.batch < 1 || !a.vm || !a.vitems || !a.vcount || !a.vlane_parts) return hipErrorInvalidValue;
    project_init_kernel<<<dim3((unsigned)(2 * a.batch)), 256, 0, s>>>(a);
    return hipGetLastError();
}

// Amplitudes <g|V^H y> of gather indices g that lie OUTSIDE the lane's first-stage tile but share its index on the first stage's local
// bits (flip states on the other bits): with the virtual z, Y_0 = (later stages)^H proj(y), of an lhs state that is ONE basis state with
// coefficient 1,  <g|V^H y> = sum_c Y_0[(c on the shared bits, g on the touched bits outside the first stage), c].
__global__ __launch_bounds__(64) void project_amps_kernel(const ProjArgs a, const long long* gather, int ngather, const long long* supp, double2* small,
                                                          const double2* vy) {
    const int b = blockIdx.x;
    const long long e = supp[2 * (size_t)b];
    if (e < 0) return;
    unsigned fmask = 0;
    for (int i = 0; i < a.nub0; ++i) fmask |= 1u << a.ubits0[i];
    const cplx* y0 = vy + (((size_t)b * 2) << a.nvp);
    const int ncb = 1 << a.cb;
    for (int i = threadIdx.x; i < ngather; i += 64) {
        const unsigned g = (unsigned)gather[i];
        if ((((unsigned)e ^ g) & fmask) == 0) continue;   // inside the lane's tile: the gather has read it from Z
        unsigned it_g = 0;   // the entry's index on the touched bits outside the first stage: g's
        for (int j = 0; j < a.t; ++j)
            if (g & a.tf_mask & a.off_t[1u << j]) it_g |= 1u << j;
        double re = 0.0, im = 0.0;
        for (int c = 0; c < ncb; ++c) {   // ... and on the shared ones: c (it_of_c: the T index with those bits = c, the others 0)
            const cplx v = y0[(a.it_of_c[c] | it_g) + ((size_t)c << a.t)];
            re += v.x; im += v.y;
        }
        small[(size_t)b * ngather + i] = make_double2(re, im);
    }
}
hipError_t launch_project_amps(const ProjArgs& a, const long long* gather, int ngather, const long long* supp, void* small, const void* vy, hipStream_t s) {
    if (!gather || ngather < 1 || !supp || !small || !vy) return hipErrorInvalidValue;
    project_amps_kernel<<<dim3((unsigned)a.batch), 64, 0, s>>>(a, gather, ngather, supp, static_cast<double2*>(small), static_cast<const double2*>(vy));
    return hipGetLastError();
}

}  // namespace aqc
