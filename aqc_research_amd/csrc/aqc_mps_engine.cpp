// Device-resident matrix-product states with truncated 2-qubit gates (C ABI: aqc_mps_*, aqc_svd).
//
// This is the arithmetic the reference delegates to qiskit-aer's matrix_product_state simulator
// (mps_operations.py:252-257, reached from mps_dot_objective.py:245-468): 1-qubit gates act on the physical index
// of one site tensor; a 2-qubit gate contracts two neighbouring sites, applies the 4x4 matrix, and splits the
// result with an SVD whose smallest singular values are discarded while the sum of their squares stays below
// `trunc_thr`; non-neighbouring qubits are brought together by swaps.  Aer itself is third party and absent from
// this image, so its truncation arithmetic is PARITY UNPINNED; with trunc_thr -> 0 every operation is exact and
// is tested against the dense state-vector oracle (the level the reference's own tests pin).
//
// Representation: site q holds T_q = Gamma_q . diag(lambda_q) as a [2][chi_l][chi_r] row-major complex128 tensor
// (the form _preprocess_mps builds, mps_operations.py:126-156) plus the Schmidt vector lambda_q of the bond to
// its right; |psi> = prod_q T_q.  A gate on (q, q+1) forms theta = diag(lambda_{q-1}) T_q T_{q+1}, so only one
// division by lambda (on the left bond) is needed when the new T_q is extracted.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/aqc_hip.h"
#include "aqc_launch.h"
#include "aqc_mps_host.h"

using namespace aqc;

namespace {

struct Scratch {   // grow-only device buffer
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) HIP_OK(hipFree(p));
        p = nullptr; cap = 0;
        HIP_OK(hipMalloc(&p, bytes));
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Round-robin tournament: rounds x (n2/2) pairs over n2 = even(cols) players; index >= cols is a bye (-1).
void tournament(int cols, std::vector<int>& pairs, int& rounds, int& per_round) {
    const int n2 = cols + (cols & 1);
    rounds = n2 - 1;
    per_round = n2 / 2;
    pairs.assign((size_t)std::max(rounds, 0) * per_round * 2, -1);
    std::vector<int> ring(n2);
    std::iota(ring.begin(), ring.end(), 0);
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < per_round; ++i) {
            int a = ring[i], b = ring[n2 - 1 - i];
            if (a > b) std::swap(a, b);
            if (b >= cols) { a = -1; b = -1; }
            pairs[((size_t)r * per_round + i) * 2] = a;
            pairs[((size_t)r * per_round + i) * 2 + 1] = b;
        }
        std::rotate(ring.begin() + 1, ring.end() - 1, ring.end());   // player 0 stays, the others move one seat
    }
}

struct SvdWork {
    Scratch pairs, flag, sigma;
    std::vector<int> h_pairs;
    int cached_cols = -1, cached_blocked = -1, rounds = 0, per_round = 0;
    void release() { pairs.release(); flag.release(); sigma.release(); }
};

// Tournament over column blocks: a block without a partner (odd number of blocks) still plays, alone (.y = -1).
void block_tournament(int nblocks, std::vector<int>& pairs, int& rounds, int& per_round) {
    const int n2 = nblocks + (nblocks & 1);
    rounds = std::max(n2 - 1, 1);
    per_round = std::max(n2 / 2, 1);
    pairs.assign((size_t)rounds * per_round * 2, -1);
    if (nblocks == 1) { pairs[0] = 0; return; }
    std::vector<int> ring(n2);
    std::iota(ring.begin(), ring.end(), 0);
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < per_round; ++i) {
            int a = ring[i], b = ring[n2 - 1 - i];
            if (a > b) std::swap(a, b);
            pairs[((size_t)r * per_round + i) * 2] = a;
            pairs[((size_t)r * per_round + i) * 2 + 1] = b < nblocks ? b : -1;
        }
        std::rotate(ring.begin() + 1, ring.end() - 1, ring.end());
    }
}

// Orthogonalises the columns of the column-major W (rows x cols) in place, accumulating V (cols x cols);
// returns the column norms in h_sigma.  `sweeps_out` reports the number of sweeps used.
int jacobi_svd(SvdWork& sw, void* W, int rows, void* V, int cols, hipStream_t st, std::vector<double>& h_sigma, int* sweeps_out) {
    const char* env_blocked = getenv("AQC_SVD_BLOCKED");   // 0: round-per-launch cross-check
    const bool blocked_ok = !(env_blocked && atoi(env_blocked) == 0);
    const bool small = svd_fits_small(rows, cols);
    const bool blocked = !small && blocked_ok && svd_fits_block(rows, cols);
    if (!small && (sw.cached_cols != cols || sw.cached_blocked != (int)blocked)) {   // (the one-workgroup kernel works its pairing out itself)
        if (blocked) block_tournament((cols + svd_block_size() - 1) / svd_block_size(), sw.h_pairs, sw.rounds, sw.per_round);
        else tournament(cols, sw.h_pairs, sw.rounds, sw.per_round);
        if (sw.pairs.reserve(std::max<size_t>(sw.h_pairs.size(), 2) * sizeof(int))) return 1;
        if (!sw.h_pairs.empty()) HIP_OK(hipMemcpyAsync(sw.pairs.p, sw.h_pairs.data(), sw.h_pairs.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
        sw.cached_cols = cols;
        sw.cached_blocked = (int)blocked;
    }
    constexpr int kFlagInts = 64 + 4;   // [0..63] rotations per sweep | [64] barrier | [65] sweeps used | [66] barrier timeout
    if (sw.flag.reserve(sizeof(int) * kFlagInts) || sw.sigma.reserve(sizeof(double) * (std::max(cols, 1) + 2))) return 1;   // sigma | fro2 | sweeps
    double* fro2 = static_cast<double*>(sw.sigma.p) + std::max(cols, 1);   // scale of the negligible-column rule (aqc_mps_dev.h: kNegligible2)
    int* flag = static_cast<int*>(sw.flag.p);
    const double tol = 1e-15;
    int sweeps = 0;
    int status[2] = {0, 0};
    if (small) {   // one launch: matrix and V live in the LDS of one workgroup
        HIP_OK(launch_jacobi_small(W, rows, V, cols, tol, 60, flag, static_cast<double*>(sw.sigma.p), st));
    } else if (blocked) {   // one cooperative launch: persistent workgroups, 16 columns at a time in LDS
        HIP_OK(launch_svd_identity(V, cols, st));
        HIP_OK(launch_svd_fro2(W, (size_t)rows * cols, fro2, st));
        HIP_OK(hipMemsetAsync(flag, 0, sizeof(int) * kFlagInts, st));
        HIP_OK(launch_jacobi_block(W, rows, V, cols, sw.pairs.p, sw.rounds, sw.per_round, tol, 60, fro2, flag, reinterpret_cast<unsigned*>(flag + 64), flag + 65, st));
        HIP_OK(hipMemcpyAsync(status, flag + 65, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    } else {
        HIP_OK(launch_svd_identity(V, cols, st));
        HIP_OK(launch_svd_fro2(W, (size_t)rows * cols, fro2, st));
    }
    for (; !small && !blocked && sweeps < 60 && cols > 1; ++sweeps) {
        HIP_OK(hipMemsetAsync(flag, 0, sizeof(int), st));
        for (int r = 0; r < sw.rounds; ++r)
            HIP_OK(launch_jacobi_round(W, rows, V, cols, static_cast<int*>(sw.pairs.p) + (size_t)r * sw.per_round * 2, sw.per_round, tol, fro2, flag, st));
        int rotations = 0;
        HIP_OK(hipMemcpyAsync(&rotations, flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (rotations == 0) { ++sweeps; break; }
    }
    h_sigma.resize(cols + 2);
    if (!small) HIP_OK(launch_svd_norms(W, rows, cols, static_cast<double*>(sw.sigma.p), st));   // (the one-launch kernel delivers them itself)
    HIP_OK(hipMemcpyAsync(h_sigma.data(), sw.sigma.p, sizeof(double) * (small ? cols + 2 : cols), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (small) sweeps = (int)h_sigma[cols + 1];   // the sweep count came with the singular values
    h_sigma.resize(cols);
    if (blocked) {
        if (status[1] != 0) return failf("Jacobi SVD: a grid barrier of the persistent kernel timed out (%d x %d)", rows, cols);
        sweeps = status[0];
    }
    if (sweeps_out) *sweeps_out = sweeps;   // the single-launch paths deliver their count with this synchronisation
    // every path stops at 60 sweeps; matrices of this engine converge in 6-14, so the cap means "did not converge"
    if (cols > 1 && sweeps >= 60) return failf("Jacobi SVD: no convergence within 60 sweeps (%d x %d)", rows, cols);
    return 0;
}

}  // namespace

struct aqc_mps {
    int device = 0, n = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    std::vector<int> dims;                    // n + 1 bond dimensions, dims[0] = dims[n] = 1
    std::vector<double2*> t;                  // per site: [2][dims[q]][dims[q+1]]
    std::vector<size_t> t_cap, lam_cap;       // allocated elements (buffers only grow)
    std::vector<std::vector<double>> lam;     // n - 1 Schmidt vectors (host copy)
    std::vector<double*> d_lam;               // the same on the device
    Scratch theta, work, vmat, ord, tmp;
    // Pinned staging of what the host decides per 2-qubit gate (column order, new Schmidt values): two buffers used in turn.
    // A gate's uploads are asynchronous; the NEXT gate synchronises the stream to read its singular values, so by the time a
    // buffer is written again (two gates later) the copy out of it has completed -- no synchronisation of its own.
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    unsigned stage_turn = 0;
    SvdWork svd;
    double discarded = 0.0;                   // accumulated discarded weight (sum of squared singular values)
    int last_sweeps = 0;
};

namespace {

size_t site_elems(const aqc_mps* m, int q) { return (size_t)2 * m->dims[q] * m->dims[q + 1]; }

int reserve_lambda(aqc_mps* m, int bond, size_t count) {
    if (count > m->lam_cap[bond]) {
        HIP_OK(hipStreamSynchronize(m->stream));
        if (m->d_lam[bond]) HIP_OK(hipFree(m->d_lam[bond]));
        m->d_lam[bond] = nullptr;
        HIP_OK(hipMalloc((void**)&m->d_lam[bond], sizeof(double) * count));
        m->lam_cap[bond] = count;
    }
    return 0;
}

int set_lambda(aqc_mps* m, int bond, const std::vector<double>& v) {
    m->lam[bond] = v;
    if (reserve_lambda(m, bond, v.size())) return 1;
    HIP_OK(hipMemcpyAsync(m->d_lam[bond], m->lam[bond].data(), sizeof(double) * v.size(), hipMemcpyHostToDevice, m->stream));
    HIP_OK(hipStreamSynchronize(m->stream));   // the host vector may be replaced by the next gate
    return 0;
}

// pinned staging buffer of this gate (see aqc_mps::stage)
int stage_buffer(aqc_mps* m, size_t bytes, void** out) {
    const unsigned i = m->stage_turn++ & 1u;
    if (bytes > m->stage_cap[i]) {
        HIP_OK(hipStreamSynchronize(m->stream));
        if (m->stage[i]) HIP_OK(hipHostFree(m->stage[i]));
        m->stage[i] = nullptr; m->stage_cap[i] = 0;
        const size_t cap = std::max<size_t>(bytes * 2, 4096);
        HIP_OK(hipHostMalloc(&m->stage[i], cap, hipHostMallocDefault));
        m->stage_cap[i] = cap;
    }
    *out = m->stage[i];
    return 0;
}

int reserve_site(aqc_mps* m, int q, size_t elems) {   // contents are NOT preserved
    if (elems <= m->t_cap[q]) return 0;
    HIP_OK(hipStreamSynchronize(m->stream));
    if (m->t[q]) HIP_OK(hipFree(m->t[q]));
    m->t[q] = nullptr;
    HIP_OK(hipMalloc((void**)&m->t[q], sizeof(double2) * elems));
    m->t_cap[q] = elems;
    return 0;
}

void destroy(aqc_mps* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (double2* p : m->t) if (p) (void)hipFree(p);
    for (double* p : m->d_lam) if (p) (void)hipFree(p);
    m->theta.release(); m->work.release(); m->vmat.release(); m->ord.release(); m->tmp.release(); m->svd.release();
    for (void* p : m->stage) if (p) (void)hipHostFree(p);
    if (m->stream && m->owns_stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int new_mps(int device, int n, aqc_mps** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return failf("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return failf("device out of range");
    if (n < 1 || n > 4096) return failf("number of qubits out of range");
    HIP_OK(hipSetDevice(device));
    aqc_mps* m = new aqc_mps();
    m->device = device; m->n = n;
    m->dims.assign(n + 1, 1);
    m->t.assign(n, nullptr);
    m->t_cap.assign(n, 0);
    m->lam_cap.assign(std::max(n - 1, 0), 0);
    m->lam.assign(std::max(n - 1, 0), {});
    m->d_lam.assign(std::max(n - 1, 0), nullptr);
    if (hipStreamCreate(&m->stream) != hipSuccess) { delete m; return failf("hipStreamCreate failed"); }
    *out = m;
    return 0;
}

// One 4x4 gate on the neighbouring sites (q, q+1); matrix index = 2 * bit_q + bit_{q+1}.
int gate_adjacent(aqc_mps* m, int q, const double* g16, double trunc_thr, int max_bond) {
    const int chil = m->dims[q], chim = m->dims[q + 1], chir = m->dims[q + 2];
    const int rows = 2 * chil, cols = 2 * chir;
    hipStream_t st = m->stream;
    // Jacobi runs on the side with fewer columns
    const int mode = cols <= rows ? 0 : 1;
    const int wrows = mode == 0 ? rows : cols, wcols = mode == 0 ? cols : rows;
    if (m->work.reserve(sizeof(double2) * (size_t)wrows * wcols) || m->vmat.reserve(sizeof(double2) * (size_t)wcols * wcols)) return 1;
    const double* lam_left = q > 0 ? m->d_lam[q - 1] : nullptr;
    if (chim <= 64 && (size_t)chil * chir <= 4096) {   // small bonds: product, scaling and gate in one launch
        HIP_OK(launch_mps_theta_fused(m->t[q], m->t[q + 1], lam_left, chil, chim, chir, g16, mode, m->work.p, st));
    } else {
        if (m->theta.reserve(sizeof(double2) * (size_t)rows * cols)) return 1;
        // theta0[(a,l), (b,r)] = sum_m T_q[(a,l), m] T_{q+1}[b][m][r]
        for (int b = 0; b < 2; ++b)
            HIP_OK(launch_zgemm(false, false, rows, chir, chim, m->t[q], chim, m->t[q + 1] + (size_t)b * chim * chir, chir,
                                static_cast<double2*>(m->theta.p) + (size_t)b * chir, cols, st));
        HIP_OK(launch_mps_theta(m->theta.p, lam_left, chil, chir, g16, mode, m->work.p, st));
    }
    std::vector<double> sigma;
    if (jacobi_svd(m->svd, m->work.p, wrows, m->vmat.p, wcols, st, sigma, &m->last_sweeps)) return 1;
    // order, rank and truncation (host: wcols numbers)
    std::vector<int> ord(wcols);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return sigma[a] > sigma[b]; });
    const double smax = sigma[ord[0]];
    if (!(smax > 0.0) || !std::isfinite(smax)) return failf("2-qubit gate produced a zero or non-finite state");
    int k = 0;
    double total = 0.0;
    for (int j = 0; j < wcols; ++j) {
        total += sigma[ord[j]] * sigma[ord[j]];
        if (sigma[ord[j]] > 1e-14 * smax) k = j + 1;          // drop numerically zero values (rank deficiency)
    }
    if (max_bond > 0) k = std::min(k, max_bond);
    double dropped = 0.0;
    if (trunc_thr > 0.0) {                                     // discard the tail while its weight stays below the threshold
        while (k > 1 && dropped + sigma[ord[k - 1]] * sigma[ord[k - 1]] < trunc_thr) { dropped += sigma[ord[k - 1]] * sigma[ord[k - 1]]; --k; }
    }
    double kept = 0.0;
    for (int j = 0; j < k; ++j) kept += sigma[ord[j]] * sigma[ord[j]];
    const double rescale = kept > 0.0 ? std::sqrt(total / kept) : 1.0;   // keep the norm of the state
    m->discarded += total - kept;
    // new tensors
    // theta has consumed the old site tensors: the new ones go into the same (grow-only) buffers
    if (reserve_site(m, q, (size_t)rows * k) || reserve_site(m, q + 1, (size_t)k * cols)) return 1;
    const size_t lam_off = (sizeof(int) * (size_t)wcols + 15) & ~(size_t)15;
    if (m->ord.reserve(lam_off + sizeof(double) * (size_t)wcols) || reserve_lambda(m, q, (size_t)k)) return 1;
    // column order and new Schmidt values go up through the pinned staging buffer of this gate: asynchronous, no wait (the
    // one synchronisation of a gate is the read-back of its singular values in jacobi_svd)
    void* stage = nullptr;
    if (stage_buffer(m, lam_off + sizeof(double) * (size_t)k, &stage)) return 1;
    memcpy(stage, ord.data(), sizeof(int) * wcols);
    double* h_lam = reinterpret_cast<double*>(static_cast<char*>(stage) + lam_off);
    std::vector<double>& lam = m->lam[q];
    lam.resize(k);
    for (int j = 0; j < k; ++j) h_lam[j] = lam[j] = sigma[ord[j]] * rescale;
    // ONE upload (column order | new Schmidt values); the split kernel copies the latter into the bond's vector
    HIP_OK(hipMemcpyAsync(m->ord.p, stage, lam_off + sizeof(double) * (size_t)k, hipMemcpyHostToDevice, st));
    HIP_OK(launch_mps_split(m->work.p, m->vmat.p, static_cast<int*>(m->ord.p), static_cast<double*>(m->svd.sigma.p), lam_left, chil, chir, k,
                            mode, rescale, m->t[q], m->t[q + 1], reinterpret_cast<const double*>(static_cast<const char*>(m->ord.p) + lam_off),
                            m->d_lam[q], st));
    m->dims[q + 1] = k;
    return 0;
}

int gate1_site(aqc_mps* m, int q, const double* g8) {
    HIP_OK(launch_gate1q(m->t[q], m->t[q], 1, (size_t)m->dims[q] * m->dims[q + 1], 0, g8, m->stream));
    return 0;
}

// 4x4 gate (index 2 * bit_ctrl + bit_targ) on any pair of qubits
int gate2_pair(aqc_mps* m, int ctrl, int targ, const double* gate, double trunc_thr, int max_bond) {
    static const double swap_gate[32] = {1, 0, 0, 0, 0, 0, 0, 0,  0, 0, 0, 0, 1, 0, 0, 0,  0, 0, 1, 0, 0, 0, 0, 0,  0, 0, 0, 0, 0, 0, 1, 0};
    const int lo = std::min(ctrl, targ), hi = std::max(ctrl, targ);
    // bring qubit `hi` down to position lo + 1 by swaps, apply, swap back (the route Aer takes as well)
    for (int p = hi - 1; p > lo; --p)
        if (gate_adjacent(m, p, swap_gate, trunc_thr, max_bond)) return 1;
    double g[32];
    permute_gate(gate, ctrl > targ, g);      // site lo carries the lower qubit: flip when ctrl is the upper one
    if (gate_adjacent(m, lo, g, trunc_thr, max_bond)) return 1;
    for (int p = lo + 1; p < hi; ++p)
        if (gate_adjacent(m, p, swap_gate, trunc_thr, max_bond)) return 1;
    return 0;
}

// ---- the ansatz, gate by gate (host side of aqc_mps_apply_circuit / aqc_mps_fast_dot_gradient) ----------
int gate1(aqc_mps* m, int q, const M2& g) { double g8[8]; pack(g, g8); return gate1_site(m, q, g8); }

int apply_circuit(aqc_mps* m, const aqc_circuit* c, const double* th, bool inverse, double trunc_thr, int max_bond) {
    const int n = m->n, tpb = c->entangler == AQC_CP ? 5 : 4;
    const double* t2 = th + 3 * n;
    const bool cx = c->entangler == AQC_CX;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    double ent[32];
    if (!inverse) {   // core_operations.py:671-708
        for (int q = 0; q < n; ++q)
            if (gate1(m, q, rz_m(th[3 * q]) * ry_m(th[3 * q + 1]) * rz_m(th[3 * q + 2]))) return 1;
        for (const BlockRef& b : blocks) {
            const double* p = t2 + (size_t)tpb * b.j;
            if (c->trotter && b.i % 3 == 0 && gate1(m, b.c, rz_m(-half_pi))) return 1;
            entangler_matrix(c->entangler, tpb == 5 ? p[4] : 0.0, ent);
            if (gate2_pair(m, b.c, b.t, ent, trunc_thr, max_bond)) return 1;
            if (gate1(m, b.c, rz_m(p[1]) * ry_m(p[0]))) return 1;
            if (gate1(m, b.t, (cx ? rx_m(p[3]) : rz_m(p[3])) * ry_m(p[2]))) return 1;
            if (c->trotter && b.i % 3 == 2 && gate1(m, b.t, rz_m(half_pi))) return 1;
        }
    } else {          // core_operations.py:787-818
        for (auto it = blocks.rbegin(); it != blocks.rend(); ++it) {
            const BlockRef& b = *it;
            const double* p = t2 + (size_t)tpb * b.j;
            if (c->trotter && b.i % 3 == 2 && gate1(m, b.t, rz_m(-half_pi))) return 1;
            if (gate1(m, b.t, ry_m(-p[2]) * (cx ? rx_m(-p[3]) : rz_m(-p[3])))) return 1;
            if (gate1(m, b.c, ry_m(-p[0]) * rz_m(-p[1]))) return 1;
            entangler_matrix(c->entangler, tpb == 5 ? -p[4] : 0.0, ent);
            if (gate2_pair(m, b.c, b.t, ent, trunc_thr, max_bond)) return 1;
            if (c->trotter && b.i % 3 == 0 && gate1(m, b.c, rz_m(half_pi))) return 1;
        }
        for (int q = 0; q < n; ++q)
            if (gate1(m, q, rz_m(-th[3 * q + 2]) * ry_m(-th[3 * q + 1]) * rz_m(-th[3 * q]))) return 1;
    }
    return 0;
}

// Environments of the pair (w, z) for <(ops) w|z>: L[q] = contraction of the sites < q ([chi_w][chi_z]), Rc[q] = the
// CONJUGATE of the contraction of the sites > q.  A gate on sites [lo, hi] leaves L[<= lo] and Rc[>= hi] valid, so
// the ~T inner products of one gradient cost a few site steps each instead of a full chain of n.
struct Environments {
    aqc_mps* w; aqc_mps* z;
    int n, valid_l, valid_r;
    std::vector<Scratch> L, R;
    Scratch t, e0, e1, bsite, vals;
    hipStream_t st;
    int init(aqc_mps* w_, aqc_mps* z_, int nvals) {
        w = w_; z = z_; n = w->n; st = w->stream;
        L.resize(n + 1); R.resize(n);
        valid_l = 0; valid_r = n - 1;
        const double one[2] = {1.0, 0.0};
        if (L[0].reserve(sizeof(double2)) || R[n - 1].reserve(sizeof(double2)) || vals.reserve(sizeof(double2) * std::max(nvals, 1))) return 1;
        HIP_OK(hipMemcpyAsync(L[0].p, one, sizeof one, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(R[n - 1].p, one, sizeof one, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
        return 0;
    }
    void release() { for (Scratch& x : L) x.release(); for (Scratch& x : R) x.release(); t.release(); e0.release(); e1.release(); bsite.release(); vals.release(); }
    void touched(int lo, int hi) { valid_l = std::min(valid_l, lo); valid_r = std::max(valid_r, hi); }
    // site p of z, seen through G^H when an operator G sits on w's side there (<G w|z> = <w|G^H z>)
    int z_site(int p, const M2* op, const double2** out) {
        *out = z->t[p];
        if (!op) return 0;
        const size_t ne = (size_t)z->dims[p] * z->dims[p + 1];
        if (bsite.reserve(sizeof(double2) * 2 * ne)) return 1;
        const M2 gh = {{std::conj(op->m[0]), std::conj(op->m[2]), std::conj(op->m[1]), std::conj(op->m[3])}};
        double g8[8]; pack(gh, g8);
        HIP_OK(launch_gate1q(z->t[p], bsite.p, 1, ne, 0, g8, st));
        *out = static_cast<const double2*>(bsite.p);
        return 0;
    }
    // out[u][v] = sum_bit sum_xy conj(A_p[bit][x][u]) in[x][y] B_p[bit][y][v]
    int step_left(int p, const void* in, const M2* op, Scratch& out) {
        const int xa = w->dims[p], ua = w->dims[p + 1], yb = z->dims[p], vb = z->dims[p + 1];
        if (mps_env_fits_small(xa, ua, yb, vb)) {   // small bonds: one launch, the operator folded in
            if (out.reserve(sizeof(double2) * (size_t)ua * vb)) return 1;
            double g8[8];
            if (op) { const M2 gh = {{std::conj(op->m[0]), std::conj(op->m[2]), std::conj(op->m[1]), std::conj(op->m[3])}}; pack(gh, g8); }
            HIP_OK(launch_mps_env_left(in, w->t[p], z->t[p], xa, ua, yb, vb, op ? g8 : nullptr, out.p, st));
            return 0;
        }
        const double2* bq = nullptr;
        if (z_site(p, op, &bq)) return 1;
        if (t.reserve(sizeof(double2) * (size_t)xa * vb) || out.reserve(sizeof(double2) * (size_t)ua * vb)) return 1;
        for (int bit = 0; bit < 2; ++bit) {
            HIP_OK(launch_zgemm(false, false, xa, vb, yb, in, yb, bq + (size_t)bit * yb * vb, vb, t.p, vb, st));
            HIP_OK(launch_zgemm(true, bit == 1, ua, vb, xa, w->t[p] + (size_t)bit * xa * ua, ua, t.p, vb, out.p, vb, st));
        }
        return 0;
    }
    // Rc[p-1][x][y] = sum_bit A_p[bit][x][u] (Rc[p] B_p[bit]^H)[u][y]
    int step_right(int p) {
        const int xa = w->dims[p], ua = w->dims[p + 1], yb = z->dims[p], vb = z->dims[p + 1];
        if (mps_env_fits_small(xa, ua, yb, vb)) {
            if (R[p - 1].reserve(sizeof(double2) * (size_t)xa * yb)) return 1;
            HIP_OK(launch_mps_env_right(R[p].p, w->t[p], z->t[p], xa, ua, yb, vb, R[p - 1].p, st));
            return 0;
        }
        if (t.reserve(sizeof(double2) * (size_t)ua * yb) || R[p - 1].reserve(sizeof(double2) * (size_t)xa * yb)) return 1;
        for (int bit = 0; bit < 2; ++bit) {
            HIP_OK(launch_zgemm_bh(false, false, ua, yb, vb, R[p].p, vb, z->t[p] + (size_t)bit * yb * vb, vb, t.p, yb, st));
            HIP_OK(launch_zgemm(false, bit == 1, xa, yb, ua, w->t[p] + (size_t)bit * xa * ua, ua, t.p, yb, R[p - 1].p, yb, st));
        }
        return 0;
    }
    // vals[slot] = <(G_1 on q_1)(G_2 on q_2) w | z>, q_1 < q_2 (nops = 1: only q_1)
    int dot(int slot, int nops, const int* q, const M2* const* g) {
        const int lo = q[0], hi = q[nops - 1];
        for (; valid_l < lo; ++valid_l)
            if (step_left(valid_l, L[valid_l].p, nullptr, L[valid_l + 1])) return 1;
        for (; valid_r > hi; --valid_r)
            if (step_right(valid_r)) return 1;
        const void* cur = L[lo].p;
        Scratch* pp[2] = {&e0, &e1};
        for (int p = lo; p <= hi; ++p) {
            const M2* op = p == q[0] ? g[0] : (nops > 1 && p == q[1] ? g[1] : nullptr);
            Scratch& out = *pp[(p - lo) & 1];
            if (step_left(p, cur, op, out)) return 1;
            cur = out.p;
        }
        HIP_OK(launch_mps_env_dot(cur, R[hi].p, (size_t)w->dims[hi + 1] * z->dims[hi + 1], static_cast<double2*>(vals.p) + slot, st));
        return 0;
    }
};

int fast_dot_gradient(const aqc_circuit* c, aqc_mps* w, aqc_mps* z, const double* th, double trunc_thr, int max_bond, int lo_blk, int hi_blk,
                      bool front_layer, double* grad) {
    const int n = w->n, tpb = c->entangler == AQC_CP ? 5 : 4, L = c->num_blocks, T = 3 * n + tpb * L;
    const bool cx = c->entangler == AQC_CX;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    // every recorded inner product: (theta index, factor); several may add into one theta (Trotter tail)
    std::vector<std::pair<int, cd>> rec;
    rec.reserve((size_t)3 * n + (size_t)tpb * blocks.size());
    Environments env;
    if (env.init(w, z, 3 * n + tpb * (int)blocks.size())) { env.release(); return 1; }
    auto both = [&](int q, const M2& g) -> int {
        if (gate1(w, q, g) || gate1(z, q, g)) return 1;
        env.touched(q, q);
        return 0;
    };
    auto record = [&](int tindex, cd factor, int nops, const int* q, const M2* const* g) -> int {
        if (env.dot((int)rec.size(), nops, q, g)) return 1;
        rec.emplace_back(tindex, factor);
        return 0;
    };
    int rc = 1;
    do {
        bool bad = false;
        // front layer: Rz(t2), Ry(t1), Rz(t0), rightmost first (core_operations.py:921-935)
        for (int q = 0; q < n && !bad; ++q) {
            const int slots[3] = {2, 1, 0};
            for (int k = 0; k < 3 && !bad; ++k) {
                const int slot = slots[k];
                const bool is_y = slot == 1;
                if (both(q, is_y ? ry_m(th[3 * q + slot]) : rz_m(th[3 * q + slot]))) { bad = true; break; }
                const M2* op = is_y ? &kPauliY : &kPauliZ;
                if (front_layer && record(3 * q + slot, cd(0, 0.5), 1, &q, &op)) bad = true;
            }
        }
        double ent[32];
        for (size_t bi = 0; bi < blocks.size() && !bad; ++bi) {
            const BlockRef& b = blocks[bi];
            const double* p = th + 3 * n + (size_t)tpb * b.j;
            const int base = 3 * n + tpb * b.j;
            const bool live = lo_blk <= b.j && b.j < hi_blk;
            if (c->trotter && b.i % 3 == 0 && both(b.c, rz_m(-half_pi))) { bad = true; break; }
            if (live && tpb == 5) {   // -1j <P11 w|z> before the gate (core_op_matrix.py:430-477)
                const int qq[2] = {std::min(b.c, b.t), std::max(b.c, b.t)};
                const M2* gg[2] = {&kProj1, &kProj1};
                if (record(base + 4, cd(0, -1.0), 2, qq, gg)) { bad = true; break; }
            }
            entangler_matrix(c->entangler, tpb == 5 ? p[4] : 0.0, ent);
            if (gate2_pair(z, b.c, b.t, ent, trunc_thr, max_bond) || gate2_pair(w, b.c, b.t, ent, trunc_thr, max_bond)) { bad = true; break; }
            env.touched(std::min(b.c, b.t), std::max(b.c, b.t));
            const int qs[4] = {b.c, b.c, b.t, b.t};
            const M2 gs[4] = {ry_m(p[0]), rz_m(p[1]), ry_m(p[2]), cx ? rx_m(p[3]) : rz_m(p[3])};
            const M2* ps[4] = {&kPauliY, &kPauliZ, &kPauliY, cx ? &kPauliX : &kPauliZ};
            for (int k = 0; k < 4 && !bad; ++k) {
                if (both(qs[k], gs[k])) { bad = true; break; }
                if (live && record(base + k, cd(0, 0.5), 1, &qs[k], &ps[k])) bad = true;
            }
            if (!bad && c->trotter && b.i % 3 == 2 && both(b.t, rz_m(half_pi))) bad = true;
        }
        if (bad) break;
        std::vector<cd> vals(rec.size());
        if (!rec.empty() && hipMemcpyAsync(vals.data(), env.vals.p, sizeof(cd) * rec.size(), hipMemcpyDeviceToHost, env.st) != hipSuccess) { failf("gradient download failed"); break; }
        if (hipStreamSynchronize(env.st) != hipSuccess) { failf("stream synchronisation failed"); break; }
        std::vector<cd> g(T, cd(0.0, 0.0));
        for (size_t i = 0; i < rec.size(); ++i) g[rec[i].first] += rec[i].second * vals[i];
        std::memcpy(grad, g.data(), sizeof(cd) * T);
        rc = 0;
    } while (false);
    env.release();
    return rc;
}

}  // namespace

int aqc::mps_peek(const aqc_mps* m, int q, const void** site, const double** lam) {
    if (!m || q < 0 || q >= m->n) return failf("mps_peek: invalid argument");
    HIP_OK(hipStreamSynchronize(m->stream));
    *site = m->t[q];
    *lam = q < m->n - 1 ? m->d_lam[q] : nullptr;
    return 0;
}

int aqc::mps_adopt(int device, int n, const int* dims, const void* const* sites, const double* const* lams, double discarded, aqc_mps** out) {
    aqc_mps* m = nullptr;
    if (new_mps(device, n, &m)) return 1;
    m->dims.assign(dims, dims + n + 1);
    m->discarded = discarded;
    for (int q = 0; q < n; ++q) {
        const size_t ne = site_elems(m, q);
        if (reserve_site(m, q, ne) || hipMemcpyAsync(m->t[q], sites[q], sizeof(double2) * ne, hipMemcpyDeviceToDevice, m->stream) != hipSuccess) {
            destroy(m);
            return failf("MPS copy failed");
        }
        if (q < n - 1) {
            std::vector<double> lam(dims[q + 1]);
            if (hipMemcpy(lam.data(), lams[q], sizeof(double) * lam.size(), hipMemcpyDeviceToHost) != hipSuccess || set_lambda(m, q, lam)) {
                destroy(m);
                return failf("MPS copy failed");
            }
        }
    }
    if (hipStreamSynchronize(m->stream) != hipSuccess) { destroy(m); return failf("MPS copy failed"); }
    *out = m;
    return 0;
}

extern "C" {

int aqc_mps_create(int device, int n, const int32_t* dims, const double* gammas, const double* lambdas, aqc_mps** out) {
    if (!dims || !gammas || !out || (n > 1 && !lambdas)) return failf("null MPS argument");
    if (n < 1) return failf("number of qubits out of range");
    if (dims[0] != 1 || dims[n] != 1) return failf("MPS boundary bond dimensions must be 1");
    for (int q = 0; q <= n; ++q) if (dims[q] < 1) return failf("MPS bond dimensions must be positive");
    aqc_mps* m = nullptr;
    if (new_mps(device, n, &m)) return 1;
    m->dims.assign(dims, dims + n + 1);
    size_t off = 0, loff = 0;
    for (int q = 0; q < n; ++q) {
        const size_t ne = site_elems(m, q);
        if (reserve_site(m, q, ne) ||
            hipMemcpyAsync(m->t[q], gammas + 2 * off, sizeof(double2) * ne, hipMemcpyHostToDevice, m->stream) != hipSuccess) {
            destroy(m);
            return failf("MPS upload failed");
        }
        off += ne;
        if (q < n - 1) {
            std::vector<double> lam(lambdas + loff, lambdas + loff + dims[q + 1]);
            loff += dims[q + 1];
            for (double v : lam) if (!(v > 0.0)) { destroy(m); return failf("MPS Schmidt coefficients must be positive"); }
            if (set_lambda(m, q, lam)) { destroy(m); return 1; }
            if (launch_mps_colscale(m->t[q], m->d_lam[q], (size_t)2 * dims[q], dims[q + 1], 1, m->stream) != hipSuccess) { destroy(m); return failf("MPS scale failed"); }
        }
    }
    if (hipStreamSynchronize(m->stream) != hipSuccess) { destroy(m); return failf("MPS upload failed"); }
    *out = m;
    return 0;
}

int aqc_mps_destroy(aqc_mps* m) {
    destroy(m);
    return 0;
}

int aqc_mps_clone(const aqc_mps* src, aqc_mps** out) {
    if (!src || !out) return failf("null argument");
    aqc_mps* m = nullptr;
    if (new_mps(src->device, src->n, &m)) return 1;
    m->dims = src->dims;
    m->discarded = src->discarded;
    for (int q = 0; q < src->n; ++q) {
        const size_t bytes = sizeof(double2) * site_elems(src, q);
        if (reserve_site(m, q, site_elems(src, q)) || hipMemcpy(m->t[q], src->t[q], bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
            destroy(m);
            return failf("MPS clone failed");
        }
        if (q < src->n - 1 && set_lambda(m, q, src->lam[q])) { destroy(m); return 1; }
    }
    *out = m;
    return 0;
}

int aqc_mps_num_qubits(const aqc_mps* m) { return m ? m->n : -1; }
int aqc_mps_device(const aqc_mps* m) { return m ? m->device : -1; }

int aqc_mps_dims(const aqc_mps* m, int32_t* dims) {
    if (!m || !dims) return failf("null argument");
    for (int q = 0; q <= m->n; ++q) dims[q] = m->dims[q];
    return 0;
}

double aqc_mps_discarded_weight(const aqc_mps* m) { return m ? m->discarded : -1.0; }

/* Gamma_q = T_q / lambda_q and the lambdas, packed like the inputs of aqc_mps_create */
int aqc_mps_export(aqc_mps* m, double* gammas, double* lambdas) {
    if (!m || !gammas || (m->n > 1 && !lambdas)) return failf("null argument");
    HIP_OK(hipSetDevice(m->device));
    size_t off = 0, loff = 0;
    for (int q = 0; q < m->n; ++q) {
        const size_t ne = site_elems(m, q);
        if (m->tmp.reserve(sizeof(double2) * ne)) return 1;
        HIP_OK(hipMemcpyAsync(m->tmp.p, m->t[q], sizeof(double2) * ne, hipMemcpyDeviceToDevice, m->stream));
        if (q < m->n - 1) {
            HIP_OK(launch_mps_colscale(m->tmp.p, m->d_lam[q], (size_t)2 * m->dims[q], m->dims[q + 1], 0, m->stream));
            std::memcpy(lambdas + loff, m->lam[q].data(), sizeof(double) * m->lam[q].size());
            loff += m->lam[q].size();
        }
        HIP_OK(hipMemcpyAsync(gammas + 2 * off, m->tmp.p, sizeof(double2) * ne, hipMemcpyDeviceToHost, m->stream));
        HIP_OK(hipStreamSynchronize(m->stream));
        off += ne;
    }
    return 0;
}

int aqc_mps_gate1(aqc_mps* m, int qubit, const double* gate) {
    if (!m || !gate) return failf("null argument");
    if (qubit < 0 || qubit >= m->n) return failf("qubit out of range");
    HIP_OK(hipSetDevice(m->device));
    return gate1_site(m, qubit, gate);
}

/* 4x4 gate (index 2 * bit_ctrl + bit_targ) on any pair of qubits */
int aqc_mps_gate2(aqc_mps* m, int ctrl, int targ, const double* gate, double trunc_thr, int max_bond) {
    if (!m || !gate) return failf("null argument");
    if (ctrl < 0 || ctrl >= m->n || targ < 0 || targ >= m->n || ctrl == targ) return failf("invalid qubit pair");
    if (!(trunc_thr >= 0.0)) return failf("trunc_thr must be non-negative");
    HIP_OK(hipSetDevice(m->device));
    return gate2_pair(m, ctrl, targ, gate, trunc_thr, max_bond);
}

/* V(thetas)|mps> (inverse = 0) or V(thetas)^H|mps> (inverse = 1) in place, the whole ansatz in one call
 * (mps_operations.py:326-371: v_mul_mps / v_dagger_mul_mps) */
int aqc_mps_apply_circuit(aqc_mps* m, const aqc_circuit* circ, const double* thetas, int inverse, double trunc_thr, int max_bond) {
    if (!m || !thetas) return failf("null argument");
    if (check_circuit(circ, m->n)) return 1;
    if (!(trunc_thr >= 0.0)) return failf("trunc_thr must be non-negative");
    HIP_OK(hipSetDevice(m->device));
    return apply_circuit(m, circ, thetas, inverse != 0, trunc_thr, max_bond);
}

/* complex gradient of <V lvec|phi> given vh_phi = V^H|phi>, the whole gate-by-gate walk in one call
 * (mps_dot_objective.py:41-242 fast_dot_gradient); block_from < 0: all blocks.  The operands are left intact. */
int aqc_mps_fast_dot_gradient(const aqc_circuit* circ, const aqc_mps* lvec, const aqc_mps* vh_phi, const double* thetas, double trunc_thr,
                              int max_bond, int block_from, int block_to, int front_layer, double* grad) {
    if (!lvec || !vh_phi || !thetas || !grad) return failf("null argument");
    if (lvec->n != vh_phi->n || lvec->device != vh_phi->device) return failf("MPS operands differ in size or device");
    if (check_circuit(circ, lvec->n)) return 1;
    if (!(trunc_thr >= 0.0)) return failf("trunc_thr must be non-negative");
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (block_from > block_to || block_to > circ->num_blocks) return failf("invalid block range");
    HIP_OK(hipSetDevice(lvec->device));
    aqc_mps *w = nullptr, *z = nullptr;
    if (aqc_mps_clone(lvec, &w)) return 1;
    if (aqc_mps_clone(vh_phi, &z)) { destroy(w); return 1; }
    (void)hipStreamDestroy(z->stream);   // one stream orders the gates on both operands and the environments
    z->stream = w->stream;
    z->owns_stream = false;
    const int rc = fast_dot_gradient(circ, w, z, thetas, trunc_thr, max_bond, block_from, block_to, front_layer != 0, grad);
    destroy(z);
    destroy(w);
    return rc;
}

/* <(prod_i G_i on qubit_i) a | b> by transfer matrices (mps_dot, mps_operations.py:192-213; with one Pauli this is
 * the 0.5j<P w|z> of mps_dot_objective.py:471-516 without building P.w).  G_i acts on a's side, so site qubit_i of b
 * is read through G_i^H: <G a|b> = <a|G^H b>.  nops = 0: plain <a|b>. */
int aqc_mps_dot_ops(aqc_mps* a, aqc_mps* b, int nops, const int32_t* qubits, const double* gates, double* out) {
    if (!a || !b || !out || nops < 0 || (nops > 0 && (!qubits || !gates))) return failf("invalid argument");
    if (a->n != b->n || a->device != b->device) return failf("MPS operands differ in size or device");
    HIP_OK(hipSetDevice(a->device));
    HIP_OK(hipStreamSynchronize(b->stream));
    hipStream_t st = a->stream;
    const int n = a->n;
    std::vector<int> op_of(n, -1);
    for (int i = 0; i < nops; ++i) {
        if (qubits[i] < 0 || qubits[i] >= n || op_of[qubits[i]] >= 0) return failf("operator qubits must be distinct and in range");
        op_of[qubits[i]] = i;
    }
    size_t need = 1, site_max = 1;
    for (int q = 0; q <= n; ++q) need = std::max(need, (size_t)a->dims[q] * b->dims[q]);
    for (int q = 0; q < n; ++q) {
        need = std::max(need, (size_t)a->dims[q] * b->dims[q + 1]);
        if (op_of[q] >= 0) site_max = std::max(site_max, site_elems(b, q));
    }
    if (a->tmp.reserve(sizeof(double2) * (3 * need + site_max))) return 1;
    double2* e = static_cast<double2*>(a->tmp.p);
    double2* en = e + need;
    double2* t = en + need;
    double2* bsite = t + need;
    auto b_site = [&](int q, const double2** ptr) -> int {   // site q of b, seen through G^H when an operator sits there
        *ptr = b->t[q];
        if (op_of[q] < 0) return 0;
        const double* g = gates + 8 * (size_t)op_of[q];
        const double gh[8] = {g[0], -g[1], g[4], -g[5], g[2], -g[3], g[6], -g[7]};   // conjugate transpose of the 2x2
        HIP_OK(launch_gate1q(b->t[q], bsite, 1, (size_t)b->dims[q] * b->dims[q + 1], 0, gh, st));
        *ptr = bsite;
        return 0;
    };
    const double2* bq = nullptr;
    if (b_site(0, &bq)) return 1;
    // E[x][y] = sum_bit conj(A_0[bit][0][x]) B_0[bit][0][y]
    HIP_OK(launch_zgemm(true, false, a->dims[1], b->dims[1], 2, a->t[0], a->dims[1], bq, b->dims[1], e, b->dims[1], st));
    for (int q = 1; q < n; ++q) {
        const int xa = a->dims[q], ua = a->dims[q + 1], yb = b->dims[q], vb = b->dims[q + 1];
        if (b_site(q, &bq)) return 1;
        for (int bit = 0; bit < 2; ++bit) {
            HIP_OK(launch_zgemm(false, false, xa, vb, yb, e, yb, bq + (size_t)bit * yb * vb, vb, t, vb, st));
            HIP_OK(launch_zgemm(true, bit == 1, ua, vb, xa, a->t[q] + (size_t)bit * xa * ua, ua, t, vb, en, vb, st));
        }
        std::swap(e, en);
    }
    HIP_OK(hipMemcpyAsync(out, e, sizeof(double2), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}

int aqc_mps_dot(aqc_mps* a, aqc_mps* b, double* out) { return aqc_mps_dot_ops(a, b, 0, nullptr, nullptr, out); }

/* A (m x n, row-major, host) = U diag(S) Vh with k = min(m, n), S descending; U (m x k), Vh (k x n) row-major.
 * The SVD kernel of the MPS engine, exposed for testing and for callers that need a device SVD. */
int aqc_svd(int device, int m, int n, const double* a_in, double* u_out, double* s_out, double* vh_out, int* sweeps) {
    if (!a_in || !u_out || !s_out || !vh_out || m < 1 || n < 1) return failf("invalid SVD arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return failf("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return failf("device out of range");
    HIP_OK(hipSetDevice(device));
    const int mode = n <= m ? 0 : 1, k = std::min(m, n);
    const int wrows = mode == 0 ? m : n, wcols = k;
    const size_t na = (size_t)m * n;
    Scratch da, dw, dv, du, dvh, dord, dss;
    SvdWork sw;
    std::vector<double> sigma;
    int rc = 1;
    hipStream_t st = nullptr;
    HIP_OK(hipStreamCreate(&st));   // the rounds are many short dependent launches: keep them off the legacy default stream
    do {
        if (da.reserve(sizeof(double2) * na) || dw.reserve(sizeof(double2) * na) || dv.reserve(sizeof(double2) * (size_t)wcols * wcols) ||
            du.reserve(sizeof(double2) * (size_t)m * k) || dvh.reserve(sizeof(double2) * (size_t)k * n) || dord.reserve(sizeof(int) * k) ||
            dss.reserve(sizeof(double) * k)) break;
        if (hipMemcpy(da.p, a_in, sizeof(double2) * na, hipMemcpyHostToDevice) != hipSuccess) { failf("SVD upload failed"); break; }
        if (launch_svd_load(da.p, m, n, mode, dw.p, st) != hipSuccess) { failf("SVD load kernel failed"); break; }
        if (jacobi_svd(sw, dw.p, wrows, dv.p, wcols, st, sigma, sweeps)) break;
        std::vector<int> ord(wcols);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return sigma[x] > sigma[y]; });
        if (hipMemcpy(dord.p, ord.data(), sizeof(int) * k, hipMemcpyHostToDevice) != hipSuccess) { failf("SVD upload failed"); break; }
        if (launch_svd_assemble(dw.p, dv.p, static_cast<int*>(dord.p), static_cast<double*>(sw.sigma.p), m, n, k, mode, du.p, dvh.p,
                                static_cast<double*>(dss.p), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { failf("SVD assemble kernel failed"); break; }
        if (hipMemcpy(u_out, du.p, sizeof(double2) * (size_t)m * k, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(vh_out, dvh.p, sizeof(double2) * (size_t)k * n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(s_out, dss.p, sizeof(double) * k, hipMemcpyDeviceToHost) != hipSuccess) { failf("SVD download failed"); break; }
        rc = 0;
    } while (false);
    da.release(); dw.release(); dv.release(); du.release(); dvh.release(); dord.release(); dss.release(); sw.release();
    (void)hipStreamDestroy(st);
    return rc;
}

}  // extern "C"
