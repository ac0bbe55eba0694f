// Passes over a full-size state of the projected route (host side and the mathematics: aqc_ws_project.cpp).
//   project_kernel<RB, NB>   the product below with both operands fetched in the matrix-core layout (good when `keep` runs along memory)
//   project_staged_kernel    the same through LDS tiles (when the SUMMED index runs along memory: the projection)
//   project_fused_kernel<QB> both uses below from ONE fetch of the target (objective by projection)
//   project_init_kernel, project_amps_kernel   the virtual lhs pattern / amplitudes read off the virtual z
// The product, two uses:
//     out[keep, c] = sum_k conj(S[k, c]) Y[k, keep]                  per item = (lane, first-stage tile that holds the lhs state)
//   * Y = z (the checkpoint of V^H) or the target y, k = the first stage's local bits outside T, keep = the bits T, S = psi (w after
//     the first stage):  the projection of z onto the subspace the lhs state spans when it enters the later stages;
//   * Y = the target y, k = the bits T, keep = the first stage's local bits outside T, S = the virtual basis pattern after the later
//     stages' gates:  the tile of (later stages)^H y that the lhs state and the gathered amplitudes live on.
// A (2^cb x 2^nk) by (2^nk x 2^nkeep) complex product per item on the fp64 matrix cores; Y is read once, which is what a launch costs.
#include <hip/hip_runtime.h>

#include <cstdlib>

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

// ---- both products of the objective by projection from ONE fetch of the target --------------------------------------------------
//     C[u, c]   = sum_{i_T} conj(M_end[i_T, c]) y[i_T, u]      (the lhs tile of (later stages)^H y; keeps u, sums over i_T)
//     Y[i_T, c] = sum_u     conj(psi[u, c])     y[i_T, u]      (the projection; keeps i_T, sums over u)
// A workgroup = one item, wave w = the 64 values u in [64 w, 64 w + 64) (four blocks of 16), walking over the blocks of 16 values of
// i_T.  The block of y a lane fetches (lane % 16 = u, 256-byte runs) IS the operand layout of the first product, whose sums complete
// inside the wave (C accumulates in registers over the walk); for the second product the block is transposed through a wave-private
// LDS tile, multiplied with the wave's 64 x 16 slice of psi (registers, fetched once), and the four waves' partial sums over u are added
// in a fixed order through a double-buffered LDS tile -- one workgroup barrier per block of i_T.  Needs 2^us <= 256 and cb <= 4.
// QB blocks of 16 values of u per wave, 16 / QB waves per workgroup (QB = 2: eight waves, two per SIMD -- the registers of a wave
// with four blocks leave room for one; sixteen waves with one block each measured 0.36 ms against 0.35 and were dropped).  Complex products in the three-multiplication form: with a = conj(s),
//     re = sr yr + si yi = A1 + A2,   im = sr yi - si yr = A3 + A1 - A2,   A3 = sum (sr + si)(yi - yr)
// -- three MFMAs per K-step instead of four, the two extra sums are a handful of vector adds per block.
template <int QB>
__global__ __launch_bounds__(64 * (16 / QB), 1) void project_fused_kernel(const ProjArgs a, const double2* __restrict__ mend, double2* __restrict__ ctile,
                                                                          double2* __restrict__ yout) {
    constexpr int NW = 16 / QB;
    constexpr int PB = QB >= 2 ? 2 : 1;          // blocks transposed at a time (tiles per wave)
    constexpr bool kRedDouble = QB >= 2;         // (sixteen waves: one reduction tile and a second barrier -- two would not fit the LDS)
    extern __shared__ __attribute__((aligned(16))) char pj_smem[];
    const int item = blockIdx.x;
    if (item >= *a.nitems) return;   // (uniform over the workgroup: before any barrier)
    const TileItem it = a.items[item];
    const size_t ebits = pj_tile_bits(a, it.tile);
    const int nu = 1 << a.us_bits, ncb = 1 << a.cb, nkb_all = 1 << (a.t - 4);
    // blockIdx.z: which share of the blocks of i_T (few items: the walk is split so that the launch fills the chip; the tile product's
    // sums over i_T are then partial -- one compact copy per share in cpart, added by project_csum_kernel)
    const int kb0 = (int)blockIdx.z * (nkb_all / (int)gridDim.z), nkb = kb0 + nkb_all / (int)gridDim.z;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r16 = l & 15, kg = l >> 4;
    cplx* tile = reinterpret_cast<cplx*>(pj_smem) + (size_t)wave * (PB * 16 * kPjRow);   // transposition tiles of the wave
    cplx* red = reinterpret_cast<cplx*>(pj_smem) + NW * (PB * 16 * kPjRow);              // [2 (or 1)][NW waves][256]
    const size_t vbase = ((size_t)it.lane * 2 + it.slot) << a.nvp;
    const size_t real_base = (size_t)it.lane * a.lane_stride;
    const cplx* ybase = a.y + real_base + (ebits & a.ff_mask);
    const cplx* wbase = a.s + real_base + ebits;          // psi: w after the first stage, on the item's tile
    const cplx* mbase = mend + vbase;                     // M_end, virtual layout: i_T + (c << t)
    yout += (size_t)blockIdx.y * a.part_stride;           // (the projection's partial sum over this workgroup's values of u)
    const bool cvalid = r16 < ncb;
    // this lane's slice of psi: A operand of the second product, rows c = r16, k = u = 16 (QB wave + q) + 4 kg + jj
    cplx pa[QB][4];
    double ps[QB][4];
    unsigned yoff[QB];   // element offset of u = 16 (QB wave + q) + r16: the lane's column of y in block q
    bool uvalid[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int ub = 16 * (int)blockIdx.y + QB * wave + q;   // (blockIdx.y: which 256 values of u -- more than 256: partial projections)
        uvalid[q] = ub * 16 < nu;
        yoff[q] = uvalid[q] ? a.off_us[ub * 16 + r16] : 0u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            pa[q][jj] = (uvalid[q] && cvalid) ? wbase[a.off_us[ub * 16 + 4 * kg + jj] + a.off_cb[r16]] : make_double2(0.0, 0.0);
            ps[q][jj] = pa[q][jj].x + pa[q][jj].y;
        }
    }
    double4_t c1[QB], c2[QB], c3[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) { c1[q] = double4_t{0.0, 0.0, 0.0, 0.0}; c2[q] = c1[q]; c3[q] = c1[q]; }
    unsigned tlow[4];   // rows i_T = 16 kb + 4 kg + jj of y
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) tlow[jj] = a.off_t[4 * kg + jj];
    cplx yv[QB][4];
    {
        const unsigned tb = a.off_t[kb0 * 16];
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) yv[q][jj] = uvalid[q] ? pj_stream(ybase + yoff[q] + tb + tlow[jj]) : make_double2(0.0, 0.0);
    }
    // M_end rows i_T = 16 kb + 4 kg + jj, column c = r16: A operand of the first product, fetched one block ahead (scattered 16-byte
    // loads from L2: their latency would otherwise open every block)
    cplx mn[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) mn[jj] = cvalid ? mbase[(size_t)(kb0 * 16 + 4 * kg + jj) + ((size_t)r16 << a.t)] : make_double2(0.0, 0.0);
    for (int kb = kb0; kb < nkb; ++kb) {
        cplx ma[4];
        double ms[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            ma[jj] = mn[jj];
            ms[jj] = ma[jj].x + ma[jj].y;
        }
        if (kb + 1 < nkb) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mn[jj] = cvalid ? mbase[(size_t)((kb + 1) * 16 + 4 * kg + jj) + ((size_t)r16 << a.t)] : make_double2(0.0, 0.0);
        }
        double4_t y1 = {0.0, 0.0, 0.0, 0.0}, y2 = y1, y3 = y1;
#pragma unroll
        for (int h0 = 0; h0 < QB; h0 += PB) {
#pragma unroll
            for (int qq = 0; qq < PB; ++qq) {   // first product on the fetched layout; the block goes to its transposition tile
                const int q = h0 + qq;
                cplx* tq = tile + qq * (16 * kPjRow);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) tq[(4 * kg + jj) * kPjRow + r16] = yv[q][jj];   // [row = i_T in the block][col = u in the block]
                double yd[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) yd[jj] = yv[q][jj].y - yv[q][jj].x;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    c1[q] = pj_mfma(ma[jj].x, yv[q][jj].x, c1[q]);
                    c2[q] = pj_mfma(ma[jj].y, yv[q][jj].y, c2[q]);
                    c3[q] = pj_mfma(ms[jj], yd[jj], c3[q]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (h0 + PB >= QB && kb + 1 < nkb) {   // the fetched block is consumed: the next one is requested under the second product
                const unsigned tb = a.off_t[(kb + 1) * 16];
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) yv[q][jj] = uvalid[q] ? pj_stream(ybase + yoff[q] + tb + tlow[jj]) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int qq = 0; qq < PB; ++qq) {   // second product: B[k = u][col = i_T] = the tile read across
                const int q = h0 + qq;
                const cplx* tq = tile + qq * (16 * kPjRow);
                cplx tr[4];
                double td[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { tr[jj] = tq[r16 * kPjRow + 4 * kg + jj]; td[jj] = tr[jj].y - tr[jj].x; }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    y1 = pj_mfma(pa[q][jj].x, tr[jj].x, y1);
                    y2 = pj_mfma(pa[q][jj].y, tr[jj].y, y2);
                    y3 = pj_mfma(ps[q][jj], td[jj], y3);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the tiles are written again by the next pair of blocks)
        }
        // the waves' sums over their values of u, added in wave order: D[row = c = kg + 4 r][col = i_T in the block = r16]
        cplx* rb = red + (kRedDouble ? (size_t)(kb & 1) * (NW * 256) : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) rb[wave * 256 + (kg + 4 * r) * 16 + r16] = make_double2(y1[r] + y2[r], y3[r] + y1[r] - y2[r]);
        __syncthreads();   // (this buffer is written again two blocks on, behind the next barrier)
        if (threadIdx.x < 256) {
            const int c = threadIdx.x >> 4, i_l = threadIdx.x & 15;
            double re = 0.0, im = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const cplx v = rb[w * 256 + threadIdx.x]; re += v.x; im += v.y; }
            if (c < ncb) yout[vbase + (size_t)(kb * 16 + i_l) + ((size_t)c << a.t)] = make_double2(re, im);
        }
        if (!kRedDouble) __syncthreads();
    }
    // C: D[row = c = kg + 4 r][col = u in the block = r16]
    if (gridDim.z > 1) {   // a share of the sums over i_T: compact copy [share][item][u][16]
        cplx* pbase = a.cpart + (((size_t)blockIdx.z * gridDim.x + item) << (a.us_bits + 4));
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if (!uvalid[q]) continue;
            const int u = (16 * (int)blockIdx.y + QB * wave + q) * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) pbase[(size_t)u * 16 + kg + 4 * r] = make_double2(c1[q][r] + c2[q][r], c3[q][r] + c1[q][r] - c2[q][r]);
        }
        return;
    }
    cplx* obase = ctile + real_base + ebits;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        if (!uvalid[q]) continue;
        cplx* orow = obase + yoff[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = kg + 4 * r;
            if (c < ncb) orow[a.off_cb[c]] = make_double2(c1[q][r] + c2[q][r], c3[q][r] + c1[q][r] - c2[q][r]);
        }
    }
}
// more than 256 values of u: the workgroups of an item leave partial projections in consecutive copies of the virtual register
// (part_stride apart); this adds copies 1 .. nparts - 1 into copy 0, in that order
__global__ __launch_bounds__(256) void project_sum_kernel(const ProjArgs a, double2* __restrict__ yout, int nparts) {
    const int item = blockIdx.y;
    if (item >= *a.nitems) return;
    const TileItem it = a.items[item];
    const size_t vbase = ((size_t)it.lane * 2 + it.slot) << a.nvp;
    const unsigned v = blockIdx.x * 256 + threadIdx.x;
    if (v >= (1u << (a.t + a.cb))) return;
    cplx acc = yout[vbase + v];
    for (int p = 1; p < nparts; ++p) { const cplx x = yout[(size_t)p * a.part_stride + vbase + v]; acc.x += x.x; acc.y += x.y; }
    yout[vbase + v] = acc;
}
// ... and the shares of the tile product over i_T (blockIdx.z of the fused pass): added in share order into the tile of ZW
__global__ __launch_bounds__(256) void project_csum_kernel(const ProjArgs a, double2* __restrict__ ctile, int nshares, int nitems_max) {
    const int item = blockIdx.y;
    if (item >= *a.nitems) return;
    const TileItem it = a.items[item];
    const size_t ebits = pj_tile_bits(a, it.tile);
    const unsigned e = blockIdx.x * 256 + threadIdx.x;   // u * 16 + c
    if (e >= (16u << a.us_bits)) return;
    const unsigned u = e >> 4, c = e & 15;
    if (c >= (1u << a.cb)) return;
    cplx acc = make_double2(0.0, 0.0);
    for (int z = 0; z < nshares; ++z) {
        const cplx x = a.cpart[(((size_t)z * nitems_max + item) << (a.us_bits + 4)) + e];
        acc.x += x.x; acc.y += x.y;
    }
    ctile[(size_t)it.lane * a.lane_stride + ebits + a.off_us[u] + a.off_cb[c]] = acc;
}
hipError_t launch_project_fused(const ProjArgs& a, const void* mend, void* ctile, void* yout, hipStream_t s) {
    if (a.t < 4 || a.us_bits < 4 || a.us_bits > 10 || (a.us_bits > 8 && a.part_stride == 0) || a.cb < 0 || a.cb > 4 || a.batch < 1 || !a.y || !a.s || !mend || !ctile || !yout || !a.off_us)
        return hipErrorInvalidValue;
    static const int qb = []() { const char* e = getenv("AQC_PROJECTED_FUSED_QB"); return e && atoi(e) == 4 ? 4 : 2; }();
    static bool attr_set[64] = {};   // (per device: the eight-wave form needs more than the default 64 KiB of LDS)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const size_t lds2 = (8 * 2 * 16 * kPjRow + 2 * 8 * 256) * sizeof(cplx), lds4 = (4 * 2 * 16 * kPjRow + 2 * 4 * 256) * sizeof(cplx);
    if (!attr_set[dev] || dev == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(project_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(project_fused_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int nparts = a.us_bits > 8 ? 1 << (a.us_bits - 8) : 1;
    int nshares = 1;   // few items: split the walk over i_T until the launch has ~512 workgroups (one per CU and a successor)
    if (a.cpart)
        nshares = a.cpart_shares;   // (sized by proj_alloc for this batch)
    const dim3 grid((unsigned)(2 * a.batch), (unsigned)nparts, (unsigned)nshares);
    const double2* m = static_cast<const double2*>(mend);
    double2* c = static_cast<double2*>(ctile);
    double2* y = static_cast<double2*>(yout);
    if (qb == 2) project_fused_kernel<2><<<grid, 512, lds2, s>>>(a, m, c, y);
    else project_fused_kernel<4><<<grid, 256, lds4, s>>>(a, m, c, y);
    if (nparts > 1) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        project_sum_kernel<<<dim3((unsigned)(((1u << (a.t + a.cb)) + 255) / 256), (unsigned)(2 * a.batch)), 256, 0, s>>>(a, y, nparts);
    }
    if (nshares > 1) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        project_csum_kernel<<<dim3((unsigned)(((16u << a.us_bits) + 255) / 256), (unsigned)(2 * a.batch)), 256, 0, s>>>(a, c, nshares, 2 * a.batch);
    }
    return hipGetLastError();
}

// Amplitudes <g|V^H y> of gather indices g that lie OUTSIDE the lane's first-stage tile but share its index on the first stage's local
// bits (flip states on the other bits): with the virtual z, Y_0 = (later stages)^H proj(y), of an lhs state that is ONE basis state with
// coefficient 1,  <g|V^H y> = sum_c Y_0[(c on the shared bits, g on the touched bits outside the first stage), c].
__global__ __launch_bounds__(64) void project_amps_kernel(const ProjArgs a, const long long* gather, int ngather, const long long* supp, double2* small,
                                                          const double2* vy, const double2* z) {
    // (the whole registered gather of such an evaluation: indices inside the lane's tile are read from Z, where V^H's last stage has just
    // put them; four lanes share the sum over c of an index outside it)
    const int b = blockIdx.x;
    const long long e = supp[2 * (size_t)b];
    if (e < 0) return;
    unsigned fmask = 0;
    for (int i = 0; i < a.nub0; ++i) fmask |= 1u << a.ubits0[i];
    const cplx* y0 = vy + (((size_t)b * 2) << a.nvp);
    const int ncb = 1 << a.cb, part = threadIdx.x & 3;
    for (int base = 0; base < ngather; base += 16) {
        const int i = base + (int)(threadIdx.x >> 2);
        double re = 0.0, im = 0.0;
        bool inside = false;
        unsigned g = 0;
        if (i < ngather) {
            g = (unsigned)gather[i];
            inside = (((unsigned)e ^ g) & fmask) == 0;
            if (!inside) {
                unsigned it_g = 0;   // the entry's index on the touched bits outside the first stage: g's
                for (int j = 0; j < a.t; ++j)
                    if (g & a.tf_mask & a.off_t[1u << j]) it_g |= 1u << j;
                for (int c = part; c < ncb; c += 4) {   // ... and on the shared ones: c (it_of_c: the T index with those bits = c, the others 0)
                    const cplx v = y0[(a.it_of_c[c] | it_g) + ((size_t)c << a.t)];
                    re += v.x; im += v.y;
                }
            }
        }
        re += __shfl_xor(re, 1, 64); im += __shfl_xor(im, 1, 64);
        re += __shfl_xor(re, 2, 64); im += __shfl_xor(im, 2, 64);
        if (i < ngather && part == 0) small[(size_t)b * ngather + i] = inside ? z[(size_t)b * a.lane_stride + g] : make_double2(re, im);
    }
}
hipError_t launch_project_amps(const ProjArgs& a, const long long* gather, int ngather, const long long* supp, void* small, const void* vy, const void* z,
                               hipStream_t s) {
    if (!gather || ngather < 1 || !supp || !small || !vy || !z) return hipErrorInvalidValue;
    project_amps_kernel<<<dim3((unsigned)a.batch), 64, 0, s>>>(a, gather, ngather, supp, static_cast<double2*>(small), static_cast<const double2*>(vy),
                                                               static_cast<const double2*>(z));
    return hipGetLastError();
}

}  // namespace aqc
