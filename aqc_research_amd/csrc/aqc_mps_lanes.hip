// Lockstep, device-resident lanes of the MPS engine (host side: aqc_mps_batch.cpp): every kernel serves all lanes of a batch (a grid
// dimension = lane; bond dimensions, thetas and tensors of a lane are read on the device), a truncated 2-qubit gate is ONE workgroup per lane.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "aqc_launch.h"
#include "aqc_mps_dev.h"

namespace aqc {

// ---- device-resident lanes (aqc_mps_batch.cpp).  blockIdx = lane; bond dimensions, thetas and tensors of the lane are read on the device.
struct GateRef { const cplx* m; };
__device__ __forceinline__ void lane_rot(const LaneRot& r, const double* __restrict__ th, cplx* o) {   // o = 2 x 2 of one rotation
    const double t = r.idx >= 0 ? r.scale * th[r.idx] : r.scale;
    double s, c;
    sincos(0.5 * t, &s, &c);
    if (r.kind == 1) { o[0] = make_double2(c, -s); o[1] = make_double2(0.0, 0.0); o[2] = make_double2(0.0, 0.0); o[3] = make_double2(c, s); }
    else if (r.kind == 2) { o[0] = make_double2(c, 0.0); o[1] = make_double2(-s, 0.0); o[2] = make_double2(s, 0.0); o[3] = make_double2(c, 0.0); }
    else { o[0] = make_double2(c, 0.0); o[1] = make_double2(0.0, -s); o[2] = make_double2(0.0, -s); o[3] = make_double2(c, 0.0); }
}
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
// u = r[0] r[1] r[2] of the lane
__device__ __forceinline__ void lane_gate1_matrix(const LaneGate1& g, const double* __restrict__ th, cplx* u) {
    lane_rot(g.r[0], th, u);
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (g.r[k].kind) {
            cplx v[4];
            lane_rot(g.r[k], th, v);
            const cplx p0 = cadd(cmul(u[0], v[0]), cmul(u[1], v[2])), p1 = cadd(cmul(u[0], v[1]), cmul(u[1], v[3]));
            const cplx p2 = cadd(cmul(u[2], v[0]), cmul(u[3], v[2])), p3 = cadd(cmul(u[2], v[1]), cmul(u[3], v[3]));
            u[0] = p0; u[1] = p1; u[2] = p2; u[3] = p3;
        }
}
__device__ __forceinline__ void lane_apply_gate1(cplx* __restrict__ t, int ne, const cplx* u, int first, int step) {   // T[2][ne] <- u T
    for (int i = first; i < ne; i += step) {
        const cplx a0 = t[i], a1 = t[ne + i];
        t[i] = cadd(cmul(u[0], a0), cmul(u[1], a1));
        t[ne + i] = cadd(cmul(u[2], a0), cmul(u[3], a1));
    }
}
__global__ __launch_bounds__(128) void lanes_gate1_kernel(LaneMps a, LaneMps b, const LaneOp1* __restrict__ ops, LaneOp1 one, const double* __restrict__ thetas,
                                                          int T, int lanes) {   // blockIdx.y = state and lane, blockIdx.z = gate of the table
    const int l = blockIdx.y % lanes;
    const LaneMps& m = blockIdx.y < (unsigned)lanes ? a : b;
    const LaneOp1 op = ops ? ops[blockIdx.z] : one;
    const int* dims = m.dims + (size_t)l * (m.n + 1);
    cplx u[4];
    lane_gate1_matrix(op.g, thetas + (size_t)l * T, u);
    lane_apply_gate1(static_cast<cplx*>(m.T) + ((size_t)l * m.n + op.q) * kLaneSite, dims[op.q] * dims[op.q + 1], u, blockIdx.x * blockDim.x + threadIdx.x,
                     gridDim.x * blockDim.x);
}

#ifdef AQC_TUNING   // in-kernel stamps (diagnostic builds only): where the time of lanes_gate2_kernel goes (workgroup 0)
__device__ unsigned long long g_gate2_stamps[16];
#define G2_STAMP(slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (slot) atomicAdd(&g_gate2_stamps[slot], now_ - last_); else atomicAdd(&g_gate2_stamps[0], 1ull); last_ = now_; } } while (0)
#define G2_COUNT(slot, v) atomicAdd(&g_gate2_stamps[slot], (unsigned long long)(v))
#else
#define G2_STAMP(slot) do { } while (0)
#define G2_COUNT(slot, v) do { } while (0)
#endif
// One truncated 2-qubit gate on the sites (q, q + 1) of every lane, the whole of it in ONE workgroup per lane: two-site tensor with the gate
// -> Jacobi work matrix in LDS, the sweeps, singular values, order / rank / truncation (the rule of gate_adjacent, aqc_mps_engine.cpp, on the
// lane's own values), new site tensors, Schmidt values and bond dimension.
__global__ __launch_bounds__(1024) void lanes_gate2_kernel(LaneMps m0, LaneMps m1, int lanes, const LaneOp2* __restrict__ ops, LaneOp2 one,
                                                           const double* __restrict__ thetas, int T, double trunc_thr, int max_bond, double tol, int max_sweeps,
                                                           int* __restrict__ status, int* __restrict__ peak, unsigned lds_elems,
                                                           unsigned long long* __restrict__ jstats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ cplx gm[16];
    __shared__ double sig[2 * kLaneCap];
    __shared__ int ord[2 * kLaneCap];
    __shared__ double sorted[2 * kLaneCap];
    __shared__ int k_sh;
    __shared__ double rescale_sh;
    const int l = blockIdx.x % lanes, tid = threadIdx.x;
    const LaneMps& m = blockIdx.x < (unsigned)lanes ? m0 : m1;   // (the two operands of the gradient walk take every gate in one launch)
    const LaneOp2 op = ops ? ops[blockIdx.y] : one;              // (blockIdx.y: the gates of one layer, on disjoint sites)
    const int n = m.n, q = op.q;
    const LaneGate2& g = op.g;
#ifdef AQC_TUNING
    unsigned long long last_ = 0;
#endif
    G2_STAMP(0);   // (tuning builds stamp workgroup 0 of the first gate)
    int* dims = m.dims + (size_t)l * (n + 1);
    const int chil = dims[q], chim = dims[q + 1], chir = dims[q + 2];
    const int rows = 2 * chil, cols = 2 * chir, mode = cols <= rows ? 0 : 1;
    const int wr = mode == 0 ? rows : cols, wc = mode == 0 ? cols : rows;
    // LDS: the work matrix, then V -- and, before the sweeps start, the two site tensors in V's place
    const int nq = 2 * chil * chim, nq1 = 2 * chim * chir;
    if ((unsigned)(wr * wc + max(wc * wc, nq + nq1)) > lds_elems) {   // the launch was sized for smaller bonds: the host repeats the evaluation at full size
        if (tid == 0) atomicOr(&status[l], kLaneLdsShort);
        return;
    }
    if (tid < 16) {   // the 4 x 4 gate, index 2 bit_q + bit_{q+1}
        const int i = tid >> 2, j = tid & 3;
        const int si = g.flip ? ((i & 1) << 1 | (i >> 1)) : i, sj = g.flip ? ((j & 1) << 1 | (j >> 1)) : j;
        cplx v = make_double2(0.0, 0.0);
        if (g.kind == 0) v.x = (si == sj && (si == 0 || si == 3)) || (si == 1 && sj == 2) || (si == 2 && sj == 1) ? 1.0 : 0.0;
        else if (g.kind == 1) v.x = (si == sj && si < 2) || (si == 2 && sj == 3) || (si == 3 && sj == 2) ? 1.0 : 0.0;
        else if (g.kind == 2) v.x = si == sj ? (si == 3 ? -1.0 : 1.0) : 0.0;
        else if (si == sj) {
            if (si < 3) v.x = 1.0;
            else { const double t = g.scale * thetas[(size_t)l * T + g.idx]; v = make_double2(cos(t), sin(t)); }
        }
        gm[tid] = v;
    }
    cplx* sw = reinterpret_cast<cplx*>(smem);   // [wc][wr]
    cplx* sv = sw + (size_t)wc * wr;             // [wc][wc]
    cplx* tq = static_cast<cplx*>(m.T) + ((size_t)l * n + q) * kLaneSite;
    cplx* tq1 = tq + kLaneSite;
    const int nb = n > 1 ? n - 1 : 1;
    // the two site tensors and the Schmidt values of the left bond come in once, coalesced (the dot products of the two-site tensor read
    // every element 2 chi times: from global memory, one dependent load after the other, that was a third of the kernel)
    __shared__ double lam_left[kLaneCap];
    if (tid < chil) lam_left[tid] = q > 0 ? m.lam[((size_t)l * nb + (q - 1)) * kLaneCap + tid] : 1.0;
    for (int i = tid; i < nq; i += blockDim.x) sv[i] = tq[i];
    for (int i = tid; i < nq1; i += blockDim.x) sv[nq + i] = tq1[i];
    __syncthreads();
    G2_STAMP(1);
    const GateRef gref{gm};
    for (int idx = tid; idx < chil * chir; idx += blockDim.x) mps_theta_fused_body(sv, sv + nq, lam_left, chil, chim, chir, gref, mode, sw, idx);
    __syncthreads();
    G2_STAMP(2);
    for (int i = tid; i < wc * wc; i += blockDim.x) sv[i] = make_double2((i / wc) == (i % wc) ? 1.0 : 0.0, 0.0);
    double fr = 0.0;
    for (int i = tid; i < wr * wc; i += blockDim.x) { const cplx v = sw[i]; fr += v.x * v.x + v.y * v.y; }
    const double fro2 = lds_block_total(fr);
    G2_STAMP(3);
    const int sweeps = jacobi_lds_core(sw, sv, wr, wc, tol, max_sweeps, fro2);
    G2_STAMP(4);
    if (jstats && tid == 0) {   // work of the sweeps that ran (bench.py: roofline of the lockstep lanes)
        const unsigned long long rot = (unsigned long long)sweeps * (unsigned long long)(wc * (wc - 1) / 2);
        atomicAdd(&jstats[0], rot * (unsigned long long)(36 * wr + 20 * wc));
        atomicAdd(&jstats[1], 1ull);
        atomicAdd(&jstats[2], (unsigned long long)sweeps);
        atomicAdd(&jstats[3], rot);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { G2_COUNT(8, sweeps); G2_COUNT(9, wc); G2_COUNT(10, wr); G2_COUNT(12, sweeps * (wc + (wc & 1) - 1)); }
    for (int c = tid; c < wc; c += blockDim.x) sig[c] = lds_column_norm(sw, wr, c);
    __syncthreads();
    G2_STAMP(5);
    for (int c = tid; c < wc; c += blockDim.x) {   // stable descending order by counting
        const double sc = sig[c];
        int r = 0;
        for (int j = 0; j < wc; ++j) { const double sj = sig[j]; r += (sj > sc || (sj == sc && j < c)) ? 1 : 0; }
        ord[r] = c;
        sorted[r] = sc;
    }
    __syncthreads();
    G2_STAMP(6);
    if (tid < 64) {   // rank and truncation by the rule of gate_adjacent (aqc_mps_engine.cpp), on the first wave: lane j holds the j-th largest
                      // singular value (at most 64); sums by butterflies and a suffix scan instead of one thread's loops over LDS (which were
                      // a quarter of the kernel) -- the sums may differ from the host's sequential ones in the last bit
        const int j = tid;
        const double v = j < wc ? sorted[j] : 0.0, v2 = v * v;
        const double smax = __shfl(v, 0, 64);
        int flags = sweeps >= max_sweeps ? kLaneNoConv : 0;
        int k = 1;
        double rescale = 1.0;
        if (!(smax > 0.0) || !isfinite(smax)) flags |= kLaneZero;
        else {
            double total = v2;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
            k = __popcll(__ballot(v > 1e-14 * smax));                   // values above the rank-deficiency floor (they are sorted)
            if (max_bond > 0) k = min(k, max_bond);
            if (trunc_thr > 0.0) {                                       // drop the tail while its weight stays below the threshold
                double tail = j < k ? v2 : 0.0;                          // -> sum of the values j .. k - 1
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_down(tail, off, 64); if (j + off < 64) tail += t; }
                k -= __popcll(__ballot(j >= 1 && j < k && tail < trunc_thr));
            }
            if (k > kLaneCap) { flags |= kLaneOverflow; k = kLaneCap; }
            double kept = j < k ? v2 : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
            rescale = kept > 0.0 ? sqrt(total / kept) : 1.0;
            if (j == 0) atomicAdd(&m.discarded[l], total - kept);   // (the gates of a layer run side by side: several may add to the lane's weight)
        }
        if (j == 0) {
            k_sh = k; rescale_sh = rescale;
            if (flags) atomicOr(&status[l], flags);
            atomicMax(&peak[l], k);
        }
    }
    __syncthreads();
    G2_STAMP(7);
    const int k = k_sh;
    const double rescale = rescale_sh;
    const size_t total_el = (size_t)rows * k + (size_t)k * cols;
    for (size_t idx = tid; idx < total_el; idx += blockDim.x)
        mps_split_body(sw, sv, ord, sig, lam_left, chil, chir, k, mode, rescale, tq, tq1, nullptr, nullptr, idx);
    double* lam_dst = m.lam + ((size_t)l * nb + q) * kLaneCap;
    for (int j = tid; j < k; j += blockDim.x) lam_dst[j] = sorted[j] * rescale;
    if (tid == 0) dims[q + 1] = k;
    G2_STAMP(11);
}
#ifdef AQC_TUNING
extern "C" void aqc_dbg_gate2_stamps() {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gate2_stamps), sizeof h) != hipSuccess || h[0] == 0) return;
    const double n = (double)h[0];
    fprintf(stderr, "aqc_hip stamps: lanes_gate2 workgroup 0, %.0f launches: mean work matrix %.1f x %.1f, %.2f sweeps; cycles: stage %.0f theta %.0f fro2 %.0f "
            "sweeps %.0f (%.0f per round) sigma %.0f sort %.0f decision %.0f split %.0f\n", n, h[10] / n, h[9] / n, h[8] / n, h[1] / n, h[2] / n, h[3] / n,
            h[4] / n, (double)h[4] / (double)std::max<unsigned long long>(h[12], 1), h[5] / n, h[6] / n, h[7] / n, h[11] / n);
}
#endif

__global__ __launch_bounds__(256) void lanes_env_left_kernel(LaneMps w, LaneMps z, int p, const cplx* __restrict__ in, size_t in_stride, cplx* __restrict__ out,
                                                             size_t out_stride, int has_op, Gate4c gh) {
    const int l = blockIdx.x, n = w.n;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    mps_env_left_body(in + (size_t)l * in_stride, static_cast<const cplx*>(w.T) + ((size_t)l * n + p) * kLaneSite,
                      static_cast<const cplx*>(z.T) + ((size_t)l * n + p) * kLaneSite, dw[p], dw[p + 1], dz[p], dz[p + 1], has_op, gh, out + (size_t)l * out_stride);
}
__global__ __launch_bounds__(256) void lanes_env_right_kernel(LaneMps w, LaneMps z, int p, const cplx* __restrict__ in, size_t in_stride, cplx* __restrict__ out,
                                                              size_t out_stride) {
    const int l = blockIdx.x, n = w.n;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    mps_env_right_body(in + (size_t)l * in_stride, static_cast<const cplx*>(w.T) + ((size_t)l * n + p) * kLaneSite,
                       static_cast<const cplx*>(z.T) + ((size_t)l * n + p) * kLaneSite, dw[p], dw[p + 1], dz[p], dz[p + 1], out + (size_t)l * out_stride);
}
// vals[lane][slot] = sum_i e[i] conj(rc[i]) over the dims[hi + 1] of both operands (fixed order)
__global__ __launch_bounds__(256) void lanes_env_dot_kernel(LaneMps w, LaneMps z, int hi, const cplx* __restrict__ e, size_t e_stride, const cplx* __restrict__ rc,
                                                            size_t rc_stride, cplx* __restrict__ vals, int nvals, int slot) {
    const int l = blockIdx.x, n = w.n;
    const int count = w.dims[(size_t)l * (n + 1) + hi + 1] * z.dims[(size_t)l * (n + 1) + hi + 1];
    __shared__ double sr[256], si[256];
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) {
        const cplx a = e[(size_t)l * e_stride + i], b = rc[(size_t)l * rc_stride + i];
        re += a.x * b.x + a.y * b.y;
        im += a.y * b.x - a.x * b.y;
    }
    sr[threadIdx.x] = re; si[threadIdx.x] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sr[threadIdx.x] += sr[threadIdx.x + s]; si[threadIdx.x] += si[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) vals[(size_t)l * nvals + slot] = make_double2(sr[0], si[0]);
}
// Up to three consecutive parameters of the gradient walk that sit on the SAME site, in ONE launch: per parameter the rotation on site q of
// both operands, then <P w|z> with its Pauli P on that site from the environments on either side (they do not involve site q and are shared):
// E = step(L[q], site q seen through P^H), vals[lane][slot + k] = sum E conj(R[q]).  (A parameter was three launches, then one.)
__global__ __launch_bounds__(256) void lanes_grad_step_kernel(LaneMps w, LaneMps z, int q, LaneSteps st, const double* __restrict__ thetas, int T,
                                                              const cplx* __restrict__ env_l, size_t l_stride, const cplx* __restrict__ env_r, size_t r_stride,
                                                              cplx* __restrict__ scratch, cplx* __restrict__ vals, int nvals, int slot) {
    const int l = blockIdx.x, n = w.n, tid = threadIdx.x;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    const int xa = dw[q], ua = dw[q + 1], yb = dz[q], vb = dz[q + 1];
    cplx* A = static_cast<cplx*>(w.T) + ((size_t)l * n + q) * kLaneSite;
    cplx* B = static_cast<cplx*>(z.T) + ((size_t)l * n + q) * kLaneSite;
    cplx* e = scratch + (size_t)l * kLaneEnv;
    const cplx* rc = env_r + (size_t)l * r_stride;
    __shared__ double sr[256], si[256];
    for (int k = 0; k < st.count; ++k) {
        {
            cplx u[4];
            lane_gate1_matrix(st.g[k], thetas + (size_t)l * T, u);
            lane_apply_gate1(A, xa * ua, u, tid, 256);
            lane_apply_gate1(B, yb * vb, u, tid, 256);
        }
        __syncthreads();   // (workgroup-scope release / acquire: the environment step below reads what other threads have just written)
        Gate4c gh;
#pragma unroll
        for (int i = 0; i < 4; ++i) gh.m[i] = make_double2(st.gh[k][2 * i], st.gh[k][2 * i + 1]);
        mps_env_left_body(env_l + (size_t)l * l_stride, A, B, xa, ua, yb, vb, 1, gh, e);
        double re = 0.0, im = 0.0;
        for (int i = tid; i < ua * vb; i += 256) {
            const cplx a = e[i], b = rc[i];
            re += a.x * b.x + a.y * b.y;
            im += a.y * b.x - a.x * b.y;
        }
        sr[tid] = re; si[tid] = im;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) { sr[tid] += sr[tid + s]; si[tid] += si[tid + s]; }
            __syncthreads();
        }
        if (tid == 0) vals[(size_t)l * nvals + slot + k] = make_double2(sr[0], si[0]);
        __syncthreads();   // (sr / si and the scratch are reused by the next parameter)
    }
}
// every lane <- the computational-basis state bits[lane][site]: site tensors [2][1][1], Schmidt values 1, bond dimensions 1
__global__ void lanes_basis_kernel(LaneMps m, const unsigned char* __restrict__ bits, int lanes) {
    const int l = blockIdx.x, n = m.n, nb = n > 1 ? n - 1 : 1;
    for (int q = threadIdx.x; q <= n; q += blockDim.x) {
        m.dims[(size_t)l * (n + 1) + q] = 1;
        if (q == n) break;
        cplx* t = static_cast<cplx*>(m.T) + ((size_t)l * n + q) * kLaneSite;
        const int bit = bits[(size_t)l * n + q] ? 1 : 0;
        t[0] = make_double2(bit ? 0.0 : 1.0, 0.0);
        t[1] = make_double2(bit ? 1.0 : 0.0, 0.0);
        if (q < n - 1) m.lam[((size_t)l * nb + q) * kLaneCap] = 1.0;
    }
    if (threadIdx.x == 0) m.discarded[l] = 0.0;
}
__global__ void lanes_env_init_kernel(cplx* env_l, size_t l_stride, cplx* env_r_last, size_t r_stride, int lanes) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    env_l[(size_t)l * l_stride] = make_double2(1.0, 0.0);
    env_r_last[(size_t)l * r_stride] = make_double2(1.0, 0.0);
}

hipError_t launch_lanes_gate1(const LaneMps& a, const LaneMps* b, const LaneOp1* ops, int nops, const LaneOp1& one, const double* thetas, int T, int lanes,
                              int bond_hint, hipStream_t s) {
    const int blocks = std::max(1, (bond_hint * bond_hint + 127) / 128);
    lanes_gate1_kernel<<<dim3(blocks, lanes * (b ? 2 : 1), ops ? nops : 1), 128, 0, s>>>(a, b ? *b : a, ops, one, thetas, T, lanes);
    return hipGetLastError();
}
hipError_t launch_lanes_gate2(const LaneMps& m, const LaneMps* m2, const LaneOp2* ops, int nops, const LaneOp2& one, const double* thetas, int T,
                              double trunc_thr, int max_bond, int* status, int* peak, int lanes, int bond_hint, hipStream_t s,
                              unsigned long long* jstats) {
    static bool attr_set[64] = {};   // per device: hipFuncSetAttribute applies to the current one
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev] || dev == 0) {   // (slot 0 is also the catch-all: always set there)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lanes_gate2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * kSmallMax * kSmallMax * (int)sizeof(cplx));
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int h = std::min(kLaneCap, std::max(2, bond_hint));
    const unsigned lds_elems = 2u * (2 * h) * (2 * h);
    // A half-wave per column pair of the largest work matrix the launch is sized for while the lanes do not fill the chip (shortest
    // rounds); half of that once there are several workgroups per CU (typical matrices are well below the largest, idle waves only cost
    // barrier time and occupancy: 6.4 k -> 7.4 k evals/s at 1024 lanes of the 32-qubit workload, 7.6 k -> 9.6 k at 4096)
    const int groups = lanes * (m2 ? 2 : 1), gates = ops ? nops : 1;
    const int per_bond = groups * gates > 512 ? 16 : 32;
    const int threads = std::min(1024, std::max(64, (per_bond * h + 63) & ~63));
    lanes_gate2_kernel<<<dim3(groups, gates), threads, lds_elems * sizeof(cplx), s>>>(m, m2 ? *m2 : m, lanes, ops, one, thetas, T, trunc_thr, max_bond, 1e-15, 60,
                                                                                      status, peak, lds_elems, jstats);
    return hipGetLastError();
}
hipError_t launch_lanes_env_left(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride,
                                 const double* gh8, int lanes, hipStream_t s) {
    Gate4c g;
    for (int i = 0; i < 4; ++i) g.m[i] = gh8 ? make_double2(gh8[2 * i], gh8[2 * i + 1]) : make_double2(0.0, 0.0);
    lanes_env_left_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, p, static_cast<const cplx*>(in), in_stride, static_cast<cplx*>(out), out_stride,
                                                                       gh8 ? 1 : 0, g);
    return hipGetLastError();
}
hipError_t launch_lanes_env_right(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride, int lanes,
                                  hipStream_t s) {
    lanes_env_right_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, p, static_cast<const cplx*>(in), in_stride, static_cast<cplx*>(out), out_stride);
    return hipGetLastError();
}
hipError_t launch_lanes_env_dot(const LaneMps& w, const LaneMps& z, int hi, const void* e, size_t e_stride, const void* rc, size_t rc_stride, void* vals,
                                int nvals, int slot, int lanes, hipStream_t s) {
    lanes_env_dot_kernel<<<lanes, 256, 0, s>>>(w, z, hi, static_cast<const cplx*>(e), e_stride, static_cast<const cplx*>(rc), rc_stride,
                                               static_cast<cplx*>(vals), nvals, slot);
    return hipGetLastError();
}
hipError_t launch_lanes_grad_step(const LaneMps& w, const LaneMps& z, int q, const LaneSteps& steps, const double* thetas, int T, const void* env_l,
                                  size_t l_stride, const void* env_r, size_t r_stride, void* scratch, void* vals, int nvals, int slot, int lanes, hipStream_t s) {
    lanes_grad_step_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, q, steps, thetas, T, static_cast<const cplx*>(env_l), l_stride,
                                                                        static_cast<const cplx*>(env_r), r_stride, static_cast<cplx*>(scratch),
                                                                        static_cast<cplx*>(vals), nvals, slot);
    return hipGetLastError();
}
hipError_t launch_lanes_basis(const LaneMps& m, const unsigned char* bits, int lanes, hipStream_t s) {
    lanes_basis_kernel<<<lanes, 64, 0, s>>>(m, bits, lanes);
    return hipGetLastError();
}
hipError_t launch_lanes_env_init(void* env_l, size_t l_stride, void* env_r_last, size_t r_stride, int lanes, hipStream_t s) {
    lanes_env_init_kernel<<<(lanes + 127) / 128, 128, 0, s>>>(static_cast<cplx*>(env_l), l_stride, static_cast<cplx*>(env_r_last), r_stride, lanes);
    return hipGetLastError();
}

}  // namespace aqc
