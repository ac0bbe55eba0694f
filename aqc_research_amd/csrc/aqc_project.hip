// Projection of z onto the subspace the lhs state of the sweep occupies when it enters the stages after the first
// (host side and the mathematics: aqc_ws_project.cpp).  Per item (lane, tile of the first stage that holds the lhs state):
//     y0[i_T, c] = sum_u conj(psi[u, c]) z[u, i_T]        m0[i_T, c] = [i_T restricted to the bits of c equals c][its other bits equal the tile's]
// a (2^cb x 2^us) by (2^us x 2^t) complex product on the fp64 matrix cores -- z is read once, which is what the launch costs.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"

namespace aqc {

using cplx = double2;
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double4_t pj_mfma(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// Workgroup = 4 waves, wave = 16 values of i_T (the B operand's columns; MFMA rows = 16 values of c), K = the summed bits u: per block of
// 16 values of u a lane loads 4 consecutive elements of its row of z (k-group = lane / 16: u = 16 kb + 4 (lane / 16) + jj) and of
// its row of psi -- 64 contiguous bytes each, 256 contiguous bytes per row and block.
template <int NB>
__global__ __launch_bounds__(256) void project_kernel(const ProjArgs a) {
    const int item = blockIdx.y;
    const int nitems = *a.nitems;
    if (blockIdx.x == 0 && item == 0) {   // bookkeeping of the virtual stage launches
        if (threadIdx.x == 0) *a.vcount = nitems * a.ntiles_v;
        for (int b = threadIdx.x; b < a.batch; b += 256) a.vlane_parts[b] = a.lane_parts[b] * a.ntiles_v;
    }
    if (item >= nitems) return;
    const TileItem it = a.items[item];
    size_t ebits = 0;   // element offset of the tile: the values of the first stage's non-local bits
    for (int i = 0; i < a.nub0; ++i) ebits |= (size_t)((it.tile >> i) & 1) << a.ubits0[i];
    if (blockIdx.x == 0)
        for (int tt = threadIdx.x; tt < a.ntiles_v; tt += 256)
            a.vitems[(size_t)item * a.ntiles_v + tt] = TileItem{it.lane, it.slot * a.ntiles_v + tt, it.slot * a.ntiles_v + tt, 0};
    const int rows = 1 << a.t, ncb = 1 << a.cb, nkb = 1 << (a.us - 4);
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r16 = l & 15, kg = l >> 4;
    const int rb = blockIdx.x * 4 + wave;
    if (rb * 16 >= rows) return;
    const size_t vbase = ((size_t)it.lane * 2 + it.slot) << a.nvp;
    const unsigned my_t = a.off_t[rb * 16 + r16];
    const cplx* zrow = a.zin + (size_t)it.lane * a.lane_stride + my_t + (ebits & a.ff_mask) + 4 * kg;
    for (int nb0 = 0; nb0 * 16 < ncb; nb0 += NB) {
        const cplx* prow[NB];
        bool valid[NB];
        double4_t re[NB], im[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int c = (nb0 + q) * 16 + r16;
            valid[q] = c < ncb;
            prow[q] = a.w + (size_t)it.lane * a.lane_stride + ebits + (valid[q] ? a.off_cb[c] : 0u) + 4 * kg;
            re[q] = double4_t{0.0, 0.0, 0.0, 0.0};
            im[q] = re[q];
        }
        for (int kb = 0; kb < nkb; ++kb) {
            const unsigned uo = a.off_usblk[kb];
            cplx zv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) zv[jj] = zrow[uo + jj];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                cplx pv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pv[jj] = valid[q] ? prow[q][uo + jj] : make_double2(0.0, 0.0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {   // conj(p) z = (pr zr + pi zi) + i (pr zi - pi zr)
                    re[q] = pj_mfma(pv[jj].x, zv[jj].x, re[q]);
                    re[q] = pj_mfma(pv[jj].y, zv[jj].y, re[q]);
                    im[q] = pj_mfma(pv[jj].x, zv[jj].y, im[q]);
                    im[q] = pj_mfma(-pv[jj].y, zv[jj].x, im[q]);
                }
            }
        }
        // D[row = lane / 16 + 4 r][col = lane % 16]: row = c within the block, col = i_T within the wave's 16
        const int i_t = rb * 16 + r16;
        const unsigned t_bits = a.off_t[i_t];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = (nb0 + q) * 16 + kg + 4 * r;
                if (c >= ncb) continue;
                const size_t v = vbase + (size_t)i_t + ((size_t)c << a.t);
                a.vy[v] = make_double2(re[q][r], im[q][r]);
                const bool one = (t_bits & a.cb_mask) == a.off_cb[c] && (t_bits & a.tf_mask) == ((unsigned)ebits & a.tf_mask);
                a.vm[v] = make_double2(one ? 1.0 : 0.0, 0.0);
            }
    }
}

hipError_t launch_project(const ProjArgs& a, hipStream_t s) {
    if (a.t < 4 || a.us < 4 || a.cb < 0 || a.cb > 12 || a.batch < 1) return hipErrorInvalidValue;
    const int rows = 1 << a.t;
    const dim3 grid((unsigned)((rows / 16 + 3) / 4), (unsigned)(2 * a.batch));
    if (a.cb <= 4) project_kernel<1><<<grid, 256, 0, s>>>(a);
    else if (a.cb == 5) project_kernel<2><<<grid, 256, 0, s>>>(a);
    else project_kernel<4><<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace aqc
