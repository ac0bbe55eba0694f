// Lockstep lanes of the native MPS engine (aqc_mpsb_*): L independent problems that share ONE ansatz -- the seeds / restarts /
// targets of a horizon, mps_dot_objective.py:41 called once per job in the reference (job_executor.py:141) -- walk the circuit
// together.  The single-lane engine (aqc_mps_engine.cpp) is a chain of small dependent launches with one host decision (the
// truncation rank) per 2-qubit gate: one lane cannot fill the device and host threads saturate the runtime's launch path
// (DESIGN 6e.7).  Here every step of the walk is ONE launch for all lanes (grid dimension = lane, a lane's operands and bond
// dimensions come from its descriptor): a workgroup per lane forms the two-site tensor, a workgroup per lane runs the whole Jacobi
// SVD in LDS, ONE read-back brings every lane's singular values, the host takes the L rank decisions, ONE upload and ONE launch
// split the tensors.  The environments behind the ~T inner products of a gradient advance for all lanes per launch as well.
// Scope: bonds up to 32 (work matrices up to 64 x 64: the one-workgroup kernels); larger bonds stay with the single-lane engine.
// Arithmetic and truncation rule are the single-lane engine's (same device bodies, aqc_svd.hip), lane by lane.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <numeric>
#include <string>
#include <utility>

#include "../../include/aqc_hip.h"
#include "aqc_launch.h"
#include "aqc_mps_host.h"

using namespace aqc;

namespace {

constexpr int kCap = 32;                    // largest bond dimension of a lane
constexpr int kSite = 2 * kCap * kCap;      // complex elements reserved per site tensor
constexpr int kEnv = kCap * kCap;           // ... per environment
constexpr int kWork = 4 * kCap * kCap;      // ... per Jacobi work matrix / V (64 x 64)
constexpr int kSig = 2 * kCap + 2;          // doubles per lane: singular values | fro2 | sweeps
constexpr int kStage = 512;                 // bytes per lane: column order (64 ints) | new Schmidt values (32 doubles)
constexpr int kRing = 64;                   // descriptor-table slots

struct Lanes {   // L MPS of n sites, flat device storage with fixed strides; T_q = Gamma_q diag(lambda_q) like the single-lane engine
    int n = 0, L = 0;
    double2* T = nullptr;        // [L][n][kSite]
    double* lam = nullptr;       // [L][max(n - 1, 1)][kCap]
    std::vector<int> dims;       // [L][n + 1]
    std::vector<double> discarded;
    double2* site(int l, int q) const { return T + ((size_t)l * n + q) * kSite; }
    double* lambda(int l, int b) const { return lam + ((size_t)l * std::max(n - 1, 1) + b) * kCap; }
    int& dim(int l, int q) { return dims[(size_t)l * (n + 1) + q]; }
    int dim(int l, int q) const { return dims[(size_t)l * (n + 1) + q]; }
    size_t t_bytes() const { return sizeof(double2) * (size_t)L * n * kSite; }
    size_t lam_bytes() const { return sizeof(double) * (size_t)L * std::max(n - 1, 1) * kCap; }
    int alloc(int n_, int L_) {
        n = n_; L = L_;
        dims.assign((size_t)L * (n + 1), 1);
        discarded.assign(L, 0.0);
        HIP_OK(hipMalloc((void**)&T, t_bytes()));
        HIP_OK(hipMalloc((void**)&lam, lam_bytes()));
        return 0;
    }
    void release() { if (T) (void)hipFree(T); if (lam) (void)hipFree(lam); T = nullptr; lam = nullptr; }
};

}  // namespace

struct aqc_mpsb {
    int device = 0, n = 0, L = 0;
    hipStream_t st = nullptr;
    Lanes target, lhs, vh, w, z;
    bool have_target = false, have_lhs = false;
    double2* work = nullptr;     // [L][kWork]
    double2* vmat = nullptr;     // [L][kWork]
    // pinned host memory the kernels read and write directly: the singular values of a gate come back with the kernel itself (one
    // synchronisation, no copy call), the host's rank decision (column order | new Schmidt values) is read by the split kernel in place
    double* h_sigma = nullptr;   // [L][kSig]
    char* h_ordlam = nullptr;    // [L][kStage]
    char* h_ring = nullptr;      // descriptor tables: kRing slots
    size_t slot_bytes = 0;
    unsigned turn = 0;
    int pending = 0;             // tables handed to the stream since its last synchronisation
    void* d_pairs = nullptr;     // round-robin tournaments of 2 .. 64 columns, back to back (int2 units)
    std::vector<int> pairs_off, rounds, per_round;
    // environments of the pair (w, z), see aqc_mps_engine.cpp
    double2* env_l = nullptr;    // [L][n + 1][kEnv]
    double2* env_r = nullptr;    // [L][n][kEnv]
    double2* e0 = nullptr;       // [L][kEnv]
    double2* e1 = nullptr;
    double2* vals = nullptr;     // [L][nvals]
    int nvals = 0, valid_l = 0, valid_r = 0;
};

namespace {

void tournament(int cols, std::vector<int>& pairs, int& rounds, int& per_round) {   // as in aqc_mps_engine.cpp
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
            pairs[((size_t)r * per_round + i) * 2] = a;
            pairs[((size_t)r * per_round + i) * 2 + 1] = b < cols ? b : -1;
        }
        std::rotate(ring.begin() + 1, ring.end() - 1, ring.end());
    }
}

void destroy(aqc_mpsb* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->st) (void)hipStreamSynchronize(b->st);
    for (Lanes* s : {&b->target, &b->lhs, &b->vh, &b->w, &b->z}) s->release();
    for (void* p : {(void*)b->work, (void*)b->vmat, b->d_pairs, (void*)b->env_l, (void*)b->env_r,
                    (void*)b->e0, (void*)b->e1, (void*)b->vals})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)b->h_sigma, (void*)b->h_ordlam, (void*)b->h_ring})
        if (p) (void)hipHostFree(p);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
}

int sync(aqc_mpsb* b) {
    HIP_OK(hipStreamSynchronize(b->st));
    b->pending = 0;
    return 0;
}

// descriptor table of one launch, written into a slot of the pinned ring: the kernel's workgroups read their lane's entry straight from
// host memory (a few hundred bytes per lane over the bus; a copy of the table to the device cost a runtime call per launch: 1,014 ->
// 1,633 evals/s at 256 lanes of the 32-qubit workload).  A slot is written again kRing tables later; the stream is synchronised before that
// can overtake a kernel that still reads it.
template <typename D>
int push(aqc_mpsb* b, const std::vector<D>& descs, const D** out) {
    const size_t bytes = sizeof(D) * descs.size();
    if (bytes > b->slot_bytes) return failf("descriptor table larger than its slot");
    if (b->pending >= kRing - 2 && sync(b)) return 1;
    const unsigned i = b->turn++ % kRing;
    char* h = b->h_ring + (size_t)i * b->slot_bytes;
    memcpy(h, descs.data(), bytes);
    ++b->pending;
    *out = reinterpret_cast<const D*>(h);
    return 0;
}

int clone(aqc_mpsb* b, const Lanes& src, Lanes& dst) {
    HIP_OK(hipMemcpyAsync(dst.T, src.T, src.t_bytes(), hipMemcpyDeviceToDevice, b->st));
    HIP_OK(hipMemcpyAsync(dst.lam, src.lam, src.lam_bytes(), hipMemcpyDeviceToDevice, b->st));
    dst.dims = src.dims;
    dst.discarded = src.discarded;
    return 0;
}

// T_q <- g_l T_q for every lane of `s` (and of `s2`, when given: the two operands of the gradient walk take every 1-qubit gate together)
int gate1_all(aqc_mpsb* b, Lanes& s, int q, const std::vector<M2>& g, Lanes* s2 = nullptr) {
    const int states = s2 ? 2 : 1;
    std::vector<BGate1> tab((size_t)states * b->L);
    int max_ne = 1;
    for (int k = 0; k < states; ++k) {
        Lanes& x = k ? *s2 : s;
        for (int l = 0; l < b->L; ++l) {
            BGate1& d = tab[(size_t)k * b->L + l];
            d.t = x.site(l, q);
            d.ne = x.dim(l, q) * x.dim(l, q + 1);
            d.pad = 0;
            pack(g[l], d.g);
            max_ne = std::max(max_ne, d.ne);
        }
    }
    const BGate1* dev = nullptr;
    if (push(b, tab, &dev)) return 1;
    HIP_OK(launch_mpsb_gate1(dev, states * b->L, max_ne, b->st));
    return 0;
}

// one 4 x 4 gate (index 2 bit_q + bit_{q+1}; 32 doubles per lane, or one matrix for all) on the neighbouring sites (q, q + 1) of every
// lane: the steps of gate_adjacent (aqc_mps_engine.cpp), each as one launch / one copy for all lanes
int gate_adjacent_all(aqc_mpsb* b, Lanes& s, int q, const double* g32, bool per_lane_gate, double trunc_thr, int max_bond) {
    const int L = b->L;
    std::vector<BTheta> th(L);
    std::vector<BJacobi> jc(L);
    std::vector<int> mode(L), wrows(L), wcols(L);
    int max_lr = 1, max_pr = 1;
    size_t lds = 0;
    for (int l = 0; l < L; ++l) {
        const int chil = s.dim(l, q), chim = s.dim(l, q + 1), chir = s.dim(l, q + 2);
        const int rows = 2 * chil, cols = 2 * chir;
        mode[l] = cols <= rows ? 0 : 1;
        wrows[l] = mode[l] == 0 ? rows : cols;
        wcols[l] = mode[l] == 0 ? cols : rows;
        BTheta& t = th[l];
        t.tq = s.site(l, q); t.tq1 = s.site(l, q + 1);
        t.lam_left = q > 0 ? s.lambda(l, q - 1) : nullptr;
        t.work = b->work + (size_t)l * kWork;
        t.chil = chil; t.chim = chim; t.chir = chir; t.mode = mode[l];
        const double* g = g32 + (per_lane_gate ? (size_t)32 * l : 0);
        for (int i = 0; i < 16; ++i) t.m[i] = make_double2(g[2 * i], g[2 * i + 1]);
        max_lr = std::max(max_lr, chil * chir);
        BJacobi& j = jc[l];
        j.W = t.work; j.V = b->vmat + (size_t)l * kWork; j.sigma = b->h_sigma + (size_t)l * kSig;
        j.rows = wrows[l]; j.cols = wcols[l];
        j.rounds = b->rounds[wcols[l]]; j.per_round = b->per_round[wcols[l]]; j.pairs_off = b->pairs_off[wcols[l]]; j.pad = 0;
        max_pr = std::max(max_pr, j.per_round);
        lds = std::max(lds, sizeof(double2) * ((size_t)wcols[l] * wrows[l] + (size_t)wcols[l] * wcols[l]));
    }
    const BTheta* d_th = nullptr;
    const BJacobi* d_jc = nullptr;
    if (push(b, th, &d_th)) return 1;
    HIP_OK(launch_mpsb_theta(d_th, L, max_lr, b->st));
    if (push(b, jc, &d_jc)) return 1;
    HIP_OK(launch_mpsb_jacobi(d_jc, b->d_pairs, L, max_pr, lds, 1e-15, 60, b->st));
    if (sync(b)) return 1;
    // order, rank and truncation per lane (the rule of gate_adjacent)
    std::vector<BSplit> sp(L);
    size_t max_total = 1;
    for (int l = 0; l < L; ++l) {
        const int wc = wcols[l], chil = s.dim(l, q), chir = s.dim(l, q + 2);
        const double* sigma = b->h_sigma + (size_t)l * kSig;
        if (wc > 1 && sigma[wc + 1] >= 60.0) return failf("Jacobi SVD: no convergence within 60 sweeps (lane %d, %d x %d)", l, wrows[l], wc);
        int* ord = reinterpret_cast<int*>(b->h_ordlam + (size_t)l * kStage);
        double* lam = reinterpret_cast<double*>(b->h_ordlam + (size_t)l * kStage + 256);
        std::iota(ord, ord + wc, 0);
        std::stable_sort(ord, ord + wc, [&](int x, int y) { return sigma[x] > sigma[y]; });
        const double smax = sigma[ord[0]];
        if (!(smax > 0.0) || !std::isfinite(smax)) return failf("2-qubit gate produced a zero or non-finite state (lane %d)", l);
        int k = 0;
        double total = 0.0;
        for (int j = 0; j < wc; ++j) {
            total += sigma[ord[j]] * sigma[ord[j]];
            if (sigma[ord[j]] > 1e-14 * smax) k = j + 1;
        }
        if (max_bond > 0) k = std::min(k, max_bond);
        double dropped = 0.0;
        if (trunc_thr > 0.0)
            while (k > 1 && dropped + sigma[ord[k - 1]] * sigma[ord[k - 1]] < trunc_thr) { dropped += sigma[ord[k - 1]] * sigma[ord[k - 1]]; --k; }
        if (k > kCap)   // never a silent extra truncation: the caller repeats the batch on the single-lane engine
            return failf("lane %d: bond %d grows to %d, beyond the %d of the lockstep lanes (set max_bond <= %d or use the single-lane engine)", l, q, k,
                         kCap, kCap);
        double kept = 0.0;
        for (int j = 0; j < k; ++j) kept += sigma[ord[j]] * sigma[ord[j]];
        const double rescale = kept > 0.0 ? std::sqrt(total / kept) : 1.0;
        s.discarded[l] += total - kept;
        for (int j = 0; j < k; ++j) lam[j] = sigma[ord[j]] * rescale;
        BSplit& d = sp[l];
        d.W = b->work + (size_t)l * kWork; d.V = b->vmat + (size_t)l * kWork;
        const char* stage = b->h_ordlam + (size_t)l * kStage;
        d.ord = reinterpret_cast<const int*>(stage);
        d.sigma = b->h_sigma + (size_t)l * kSig;
        d.lam_left = q > 0 ? s.lambda(l, q - 1) : nullptr;
        d.tq = s.site(l, q); d.tq1 = s.site(l, q + 1);
        d.lam_new = reinterpret_cast<const double*>(stage + 256);
        d.lam_dst = s.lambda(l, q);
        d.rescale = rescale; d.chil = chil; d.chir = chir; d.k = k; d.mode = mode[l];
        max_total = std::max(max_total, (size_t)2 * chil * k + (size_t)k * 2 * chir);
        s.dim(l, q + 1) = k;
    }
    const BSplit* d_sp = nullptr;
    if (push(b, sp, &d_sp)) return 1;
    HIP_OK(launch_mpsb_split(d_sp, L, max_total, b->st));
    return 0;
}

// 4 x 4 gate (index 2 bit_ctrl + bit_targ) on any pair of qubits: swaps bring the upper one next to the lower one and back
int gate2_pair_all(aqc_mpsb* b, Lanes& s, int ctrl, int targ, const double* gates, bool per_lane_gate, double trunc_thr, int max_bond) {
    static const double swap_gate[32] = {1, 0, 0, 0, 0, 0, 0, 0,  0, 0, 0, 0, 1, 0, 0, 0,  0, 0, 1, 0, 0, 0, 0, 0,  0, 0, 0, 0, 0, 0, 1, 0};
    const int lo = std::min(ctrl, targ), hi = std::max(ctrl, targ);
    for (int p = hi - 1; p > lo; --p)
        if (gate_adjacent_all(b, s, p, swap_gate, false, trunc_thr, max_bond)) return 1;
    const int ng = per_lane_gate ? b->L : 1;
    std::vector<double> g((size_t)32 * ng);
    for (int l = 0; l < ng; ++l) permute_gate(gates + (size_t)32 * l, ctrl > targ, g.data() + (size_t)32 * l);
    if (gate_adjacent_all(b, s, lo, g.data(), per_lane_gate, trunc_thr, max_bond)) return 1;
    for (int p = lo + 1; p < hi; ++p)
        if (gate_adjacent_all(b, s, p, swap_gate, false, trunc_thr, max_bond)) return 1;
    return 0;
}

// M2 of every lane from its thetas
template <typename F>
std::vector<M2> per_lane(int L, F f) {
    std::vector<M2> g(L);
    for (int l = 0; l < L; ++l) g[l] = f(l);
    return g;
}

// V(theta_l) or V(theta_l)^H on every lane (apply_circuit of aqc_mps_engine.cpp; core_operations.py:671-708, :787-818)
int apply_circuit_all(aqc_mpsb* b, Lanes& s, const aqc_circuit* c, const double* thetas, int T, bool inverse, double trunc_thr, int max_bond) {
    const int n = b->n, L = b->L, tpb = c->entangler == AQC_CP ? 5 : 4;
    const bool cx = c->entangler == AQC_CX, cp = c->entangler == AQC_CP;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    auto th = [&](int l, int i) { return thetas[(size_t)l * T + i]; };
    std::vector<double> ent((size_t)32 * (cp ? L : 1));
    auto entangle = [&](const BlockRef& blk, double sign) -> int {
        for (int l = 0; l < (cp ? L : 1); ++l) entangler_matrix(c->entangler, cp ? sign * th(l, 3 * n + tpb * blk.j + 4) : 0.0, ent.data() + (size_t)32 * l);
        return gate2_pair_all(b, s, blk.c, blk.t, ent.data(), cp, trunc_thr, max_bond);
    };
    const std::vector<M2> pre(L, rz_m(-half_pi)), post(L, rz_m(half_pi));
    if (!inverse) {
        for (int q = 0; q < n; ++q)
            if (gate1_all(b, s, q, per_lane(L, [&](int l) { return rz_m(th(l, 3 * q)) * ry_m(th(l, 3 * q + 1)) * rz_m(th(l, 3 * q + 2)); }))) return 1;
        for (const BlockRef& blk : blocks) {
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 0 && gate1_all(b, s, blk.c, pre)) return 1;
            if (entangle(blk, 1.0)) return 1;
            if (gate1_all(b, s, blk.c, per_lane(L, [&](int l) { return rz_m(th(l, p + 1)) * ry_m(th(l, p)); }))) return 1;
            if (gate1_all(b, s, blk.t, per_lane(L, [&](int l) { return (cx ? rx_m(th(l, p + 3)) : rz_m(th(l, p + 3))) * ry_m(th(l, p + 2)); }))) return 1;
            if (c->trotter && blk.i % 3 == 2 && gate1_all(b, s, blk.t, post)) return 1;
        }
    } else {
        for (auto it = blocks.rbegin(); it != blocks.rend(); ++it) {
            const BlockRef& blk = *it;
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 2 && gate1_all(b, s, blk.t, pre)) return 1;
            if (gate1_all(b, s, blk.t, per_lane(L, [&](int l) { return ry_m(-th(l, p + 2)) * (cx ? rx_m(-th(l, p + 3)) : rz_m(-th(l, p + 3))); }))) return 1;
            if (gate1_all(b, s, blk.c, per_lane(L, [&](int l) { return ry_m(-th(l, p)) * rz_m(-th(l, p + 1)); }))) return 1;
            if (entangle(blk, -1.0)) return 1;
            if (c->trotter && blk.i % 3 == 0 && gate1_all(b, s, blk.c, post)) return 1;
        }
        for (int q = 0; q < n; ++q)
            if (gate1_all(b, s, q, per_lane(L, [&](int l) { return rz_m(-th(l, 3 * q + 2)) * ry_m(-th(l, 3 * q + 1)) * rz_m(-th(l, 3 * q)); }))) return 1;
    }
    return 0;
}

// ---- environments of the pair (w, z), all lanes per launch (struct Environments of aqc_mps_engine.cpp) -------------------------
double2* env_l(const aqc_mpsb* b, int l, int q) { return b->env_l + ((size_t)l * (b->n + 1) + q) * kEnv; }
double2* env_r(const aqc_mpsb* b, int l, int q) { return b->env_r + ((size_t)l * b->n + q) * kEnv; }

int env_init(aqc_mpsb* b) {
    std::vector<double2> ones(b->L, make_double2(1.0, 0.0));
    if (sync(b)) return 1;   // (the host vector below must outlive its copies; nothing of an earlier walk is in flight either)
    HIP_OK(hipMemcpy2DAsync(b->env_l, sizeof(double2) * (size_t)(b->n + 1) * kEnv, ones.data(), sizeof(double2), sizeof(double2), b->L,
                            hipMemcpyHostToDevice, b->st));
    HIP_OK(hipMemcpy2DAsync(env_r(b, 0, b->n - 1), sizeof(double2) * (size_t)b->n * kEnv, ones.data(), sizeof(double2), sizeof(double2), b->L,
                            hipMemcpyHostToDevice, b->st));
    if (sync(b)) return 1;
    b->valid_l = 0;
    b->valid_r = b->n - 1;
    return 0;
}
void env_touched(aqc_mpsb* b, int lo, int hi) { b->valid_l = std::min(b->valid_l, lo); b->valid_r = std::max(b->valid_r, hi); }

// out_l[u][v] = sum_bit sum_xy conj(A_p[bit][x][u]) in_l[x][y] B_p[bit][y][v]; `op` (may be null) sits on w's side of site p
template <typename In, typename Out>
int step_left_all(aqc_mpsb* b, int p, In in, const M2* op, Out out) {
    std::vector<BEnv> tab(b->L);
    size_t lds = 16;
    M2 gh{};
    if (op) gh = {{std::conj(op->m[0]), std::conj(op->m[2]), std::conj(op->m[1]), std::conj(op->m[3])}};
    for (int l = 0; l < b->L; ++l) {
        BEnv& d = tab[l];
        d.in = in(l); d.A = b->w.site(l, p); d.B = b->z.site(l, p); d.out = out(l);
        d.xa = b->w.dim(l, p); d.ua = b->w.dim(l, p + 1); d.yb = b->z.dim(l, p); d.vb = b->z.dim(l, p + 1);
        d.has_op = op ? 1 : 0; d.pad = 0;
        for (int i = 0; i < 4; ++i) d.m[i] = make_double2(gh.m[i].real(), gh.m[i].imag());
        lds = std::max(lds, sizeof(double2) * (size_t)d.xa * d.vb);
    }
    const BEnv* dev = nullptr;
    if (push(b, tab, &dev)) return 1;
    HIP_OK(launch_mpsb_env_left(dev, b->L, lds, b->st));
    return 0;
}
// Rc[p - 1][x][y] = sum_bit A_p[bit][x][u] (Rc[p] B_p[bit]^H)[u][y]
int step_right_all(aqc_mpsb* b, int p) {
    std::vector<BEnv> tab(b->L);
    size_t lds = 16;
    for (int l = 0; l < b->L; ++l) {
        BEnv& d = tab[l];
        d.in = env_r(b, l, p); d.A = b->w.site(l, p); d.B = b->z.site(l, p); d.out = env_r(b, l, p - 1);
        d.xa = b->w.dim(l, p); d.ua = b->w.dim(l, p + 1); d.yb = b->z.dim(l, p); d.vb = b->z.dim(l, p + 1);
        d.has_op = 0; d.pad = 0;
        for (int i = 0; i < 4; ++i) d.m[i] = make_double2(0.0, 0.0);
        lds = std::max(lds, sizeof(double2) * (size_t)d.ua * d.yb);
    }
    const BEnv* dev = nullptr;
    if (push(b, tab, &dev)) return 1;
    HIP_OK(launch_mpsb_env_right(dev, b->L, lds, b->st));
    return 0;
}
// vals[l][slot] = <(G_1 on q_1)(G_2 on q_2) w_l | z_l>, q_1 < q_2 (nops = 1: only q_1)
int dot_all(aqc_mpsb* b, int slot, int nops, const int* q, const M2* const* g) {
    if (slot >= b->nvals) return failf("inner-product slot out of range");
    const int lo = q[0], hi = q[nops - 1];
    for (; b->valid_l < lo; ++b->valid_l) {
        const int p = b->valid_l;
        if (step_left_all(b, p, [&](int l) { return env_l(b, l, p); }, nullptr, [&](int l) { return env_l(b, l, p + 1); })) return 1;
    }
    for (; b->valid_r > hi; --b->valid_r)
        if (step_right_all(b, b->valid_r)) return 1;
    double2* cur = nullptr;   // null: the left environment of site lo
    double2* pp[2] = {b->e0, b->e1};
    for (int p = lo; p <= hi; ++p) {
        const M2* op = p == q[0] ? g[0] : (nops > 1 && p == q[1] ? g[1] : nullptr);
        double2* out = pp[(p - lo) & 1];
        double2* in_base = cur;
        if (step_left_all(b, p, [&](int l) { return in_base ? in_base + (size_t)l * kEnv : env_l(b, l, lo); }, op,
                          [&](int l) { return out + (size_t)l * kEnv; })) return 1;
        cur = out;
    }
    std::vector<BDot> tab(b->L);
    for (int l = 0; l < b->L; ++l) {
        BDot& d = tab[l];
        d.e = cur + (size_t)l * kEnv; d.rc = env_r(b, l, hi);
        d.out = b->vals + (size_t)l * b->nvals + slot;
        d.count = b->w.dim(l, hi + 1) * b->z.dim(l, hi + 1); d.pad = 0;
    }
    const BDot* dev = nullptr;
    if (push(b, tab, &dev)) return 1;
    HIP_OK(launch_mpsb_env_dot(dev, b->L, b->st));
    return 0;
}

// the gate-by-gate gradient walk of mps_dot_objective.py:41-242 on every lane: w = lhs, z = vh (both consumed)
int gradient_all(aqc_mpsb* b, const aqc_circuit* c, const double* thetas, int T, double trunc_thr, int max_bond, int lo_blk, int hi_blk,
                 bool front_layer, std::complex<double>* grad /* [L][T] */) {
    const int n = b->n, L = b->L, tpb = c->entangler == AQC_CP ? 5 : 4;
    const bool cx = c->entangler == AQC_CX, cp = c->entangler == AQC_CP;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    auto th = [&](int l, int i) { return thetas[(size_t)l * T + i]; };
    std::vector<std::pair<int, cd>> rec;
    if (env_init(b)) return 1;
    auto both = [&](int q, const std::vector<M2>& g) -> int {
        if (gate1_all(b, b->w, q, g, &b->z)) return 1;
        env_touched(b, q, q);
        return 0;
    };
    auto record = [&](int tindex, cd factor, int nops, const int* q, const M2* const* g) -> int {
        if (dot_all(b, (int)rec.size(), nops, q, g)) return 1;
        rec.emplace_back(tindex, factor);
        return 0;
    };
    for (int q = 0; q < n; ++q) {
        const int slots[3] = {2, 1, 0};
        for (int k = 0; k < 3; ++k) {
            const int slot = slots[k];
            const bool is_y = slot == 1;
            if (both(q, per_lane(L, [&](int l) { return is_y ? ry_m(th(l, 3 * q + slot)) : rz_m(th(l, 3 * q + slot)); }))) return 1;
            const M2* op = is_y ? &kPauliY : &kPauliZ;
            if (front_layer && record(3 * q + slot, cd(0, 0.5), 1, &q, &op)) return 1;
        }
    }
    std::vector<double> ent((size_t)32 * (cp ? L : 1));
    const std::vector<M2> pre(L, rz_m(-half_pi)), post(L, rz_m(half_pi));
    for (const BlockRef& blk : blocks) {
        const int base = 3 * n + tpb * blk.j;
        const bool live = lo_blk <= blk.j && blk.j < hi_blk;
        if (c->trotter && blk.i % 3 == 0 && both(blk.c, pre)) return 1;
        if (live && cp) {
            const int qq[2] = {std::min(blk.c, blk.t), std::max(blk.c, blk.t)};
            const M2* gg[2] = {&kProj1, &kProj1};
            if (record(base + 4, cd(0, -1.0), 2, qq, gg)) return 1;
        }
        for (int l = 0; l < (cp ? L : 1); ++l) entangler_matrix(c->entangler, cp ? th(l, base + 4) : 0.0, ent.data() + (size_t)32 * l);
        if (gate2_pair_all(b, b->z, blk.c, blk.t, ent.data(), cp, trunc_thr, max_bond) ||
            gate2_pair_all(b, b->w, blk.c, blk.t, ent.data(), cp, trunc_thr, max_bond)) return 1;
        env_touched(b, std::min(blk.c, blk.t), std::max(blk.c, blk.t));
        const int qs[4] = {blk.c, blk.c, blk.t, blk.t};
        const M2* ps[4] = {&kPauliY, &kPauliZ, &kPauliY, cx ? &kPauliX : &kPauliZ};
        for (int k = 0; k < 4; ++k) {
            if (both(qs[k], per_lane(L, [&](int l) {
                    const double a = th(l, base + k);
                    return k == 0 ? ry_m(a) : k == 1 ? rz_m(a) : k == 2 ? ry_m(a) : (cx ? rx_m(a) : rz_m(a));
                }))) return 1;
            if (live && record(base + k, cd(0, 0.5), 1, &qs[k], &ps[k])) return 1;
        }
        if (c->trotter && blk.i % 3 == 2 && both(blk.t, post)) return 1;
    }
    std::vector<cd> vals((size_t)L * b->nvals);
    if (!rec.empty()) HIP_OK(hipMemcpyAsync(vals.data(), b->vals, sizeof(cd) * vals.size(), hipMemcpyDeviceToHost, b->st));
    if (sync(b)) return 1;
    for (int l = 0; l < L; ++l) {
        cd* g = grad + (size_t)l * T;
        std::fill(g, g + T, cd(0.0, 0.0));
        for (size_t i = 0; i < rec.size(); ++i) g[rec[i].first] += rec[i].second * vals[(size_t)l * b->nvals + i];
    }
    return 0;
}

// <lhs_l | vh_l> for every lane: the full chain of left steps, closed with the trivial right boundary
int overlap_all(aqc_mpsb* b, std::complex<double>* h) {
    // w = lhs, z = vh are in place; environments from scratch
    if (env_init(b)) return 1;
    const int q = b->n - 1;
    const M2 eye = {{1.0, 0.0, 0.0, 1.0}};
    const M2* g = &eye;
    if (dot_all(b, 0, 1, &q, &g)) return 1;
    std::vector<cd> v((size_t)b->L);
    HIP_OK(hipMemcpy2DAsync(v.data(), sizeof(cd), b->vals, sizeof(cd) * (size_t)b->nvals, sizeof(cd), b->L, hipMemcpyDeviceToHost, b->st));
    if (sync(b)) return 1;
    for (int l = 0; l < b->L; ++l) h[l] = v[l];
    return 0;
}

int load_lanes(aqc_mpsb* b, Lanes& dst, aqc_mps* const* src, int shared) {
    if (!src) return failf("null MPS list");
    for (int l = 0; l < b->L; ++l) {
        const aqc_mps* m = src[shared ? 0 : l];
        if (!m) return failf("null MPS handle (lane %d)", l);
        if (aqc_mps_num_qubits(m) != b->n) return failf("lane %d: the MPS has %d qubits, the batch %d", l, aqc_mps_num_qubits(m), b->n);
        std::vector<int32_t> dims(b->n + 1);
        if (aqc_mps_dims(m, dims.data())) return 1;
        for (int q = 0; q <= b->n; ++q) {
            if (dims[q] > kCap) return failf("lane %d: bond dimension %d exceeds the %d of the lockstep lanes (use the single-lane engine)", l, dims[q], kCap);
            dst.dim(l, q) = dims[q];
        }
        dst.discarded[l] = aqc_mps_discarded_weight(m);
        for (int q = 0; q < b->n; ++q) {
            const void* site = nullptr;
            const double* lam = nullptr;
            if (mps_peek(m, q, &site, &lam)) return 1;
            HIP_OK(hipMemcpyAsync(dst.site(l, q), site, sizeof(double2) * 2 * dims[q] * dims[q + 1], hipMemcpyDeviceToDevice, b->st));
            if (q < b->n - 1) HIP_OK(hipMemcpyAsync(dst.lambda(l, q), lam, sizeof(double) * dims[q + 1], hipMemcpyDeviceToDevice, b->st));
        }
    }
    return sync(b);
}

}  // namespace

extern "C" {

int aqc_mpsb_create(int device, int num_qubits, int lanes, aqc_mpsb** out) {
    if (!out) return failf("null output");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return failf("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return failf("device out of range");
    if (num_qubits < 2 || num_qubits > 4096 || lanes < 1 || lanes > 4096) return failf("number of qubits / lanes out of range");
    HIP_OK(hipSetDevice(device));
    aqc_mpsb* b = new aqc_mpsb();
    b->device = device; b->n = num_qubits; b->L = lanes;
    auto bad = [&]() { destroy(b); return 1; };
    if (hipStreamCreate(&b->st) != hipSuccess) { failf("hipStreamCreate failed"); return bad(); }
    for (Lanes* s : {&b->target, &b->lhs, &b->vh, &b->w, &b->z})
        if (s->alloc(num_qubits, lanes)) return bad();
    const size_t L = lanes;
    b->slot_bytes = L * std::max({sizeof(BTheta), sizeof(BSplit), sizeof(BEnv), sizeof(BJacobi), sizeof(BGate1), sizeof(BDot)});
    b->nvals = 3 * num_qubits + 5 * 1;   // grown per circuit in aqc_mpsb_eval
    if (hipMalloc((void**)&b->work, sizeof(double2) * L * kWork) != hipSuccess || hipMalloc((void**)&b->vmat, sizeof(double2) * L * kWork) != hipSuccess ||
        hipMalloc((void**)&b->env_l, sizeof(double2) * L * (num_qubits + 1) * kEnv) != hipSuccess ||
        hipMalloc((void**)&b->env_r, sizeof(double2) * L * num_qubits * kEnv) != hipSuccess ||
        hipMalloc((void**)&b->e0, sizeof(double2) * L * kEnv) != hipSuccess || hipMalloc((void**)&b->e1, sizeof(double2) * L * kEnv) != hipSuccess ||
        hipHostMalloc((void**)&b->h_sigma, sizeof(double) * L * kSig, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&b->h_ordlam, L * kStage, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&b->h_ring, b->slot_bytes * kRing, hipHostMallocDefault) != hipSuccess) {
        failf("allocation of the lockstep lanes failed (%d lanes, %d qubits)", lanes, num_qubits);
        return bad();
    }
    // tournaments of 2 .. 64 columns
    std::vector<int> all, one;
    b->pairs_off.assign(2 * kCap + 1, 0); b->rounds.assign(2 * kCap + 1, 0); b->per_round.assign(2 * kCap + 1, 1);
    for (int cols = 2; cols <= 2 * kCap; ++cols) {
        tournament(cols, one, b->rounds[cols], b->per_round[cols]);
        b->pairs_off[cols] = (int)(all.size() / 2);
        all.insert(all.end(), one.begin(), one.end());
    }
    if (hipMalloc(&b->d_pairs, sizeof(int) * all.size()) != hipSuccess ||
        hipMemcpy(b->d_pairs, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice) != hipSuccess) { failf("tournament upload failed"); return bad(); }
    *out = b;
    return 0;
}

int aqc_mpsb_destroy(aqc_mpsb* b) {
    destroy(b);
    return 0;
}

int aqc_mpsb_set_targets(aqc_mpsb* b, aqc_mps* const* targets, int shared) {
    if (!b) return failf("null batch");
    HIP_OK(hipSetDevice(b->device));
    if (load_lanes(b, b->target, targets, shared)) return 1;
    b->have_target = true;
    return 0;
}

int aqc_mpsb_set_lhs(aqc_mpsb* b, aqc_mps* const* lhs, int shared) {
    if (!b) return failf("null batch");
    HIP_OK(hipSetDevice(b->device));
    if (load_lanes(b, b->lhs, lhs, shared)) return 1;
    b->have_lhs = true;
    return 0;
}

int aqc_mpsb_eval(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int block_from, int block_to,
                  int front_layer, double* h_out, double* grad_out, double* discarded_out, int32_t* max_bond_out) {
    if (!b || !circ || !thetas || !h_out || !grad_out) return failf("null argument");
    if (!b->have_target || !b->have_lhs) return failf("set the targets and the lhs states of the lanes first");
    if (check_circuit(circ, b->n)) return 1;
    if (max_bond > kCap) return failf("the lockstep lanes keep bonds up to %d", kCap);
    HIP_OK(hipSetDevice(b->device));
    const int tpb = circ->entangler == AQC_CP ? 5 : 4, T = 3 * b->n + tpb * circ->num_blocks;
    const int nblk = (int)blocks_of(circ).size();
    const int need = std::max(1, 3 * b->n + tpb * nblk);
    if (need > b->nvals || !b->vals) {
        if (sync(b)) return 1;
        if (b->vals) HIP_OK(hipFree(b->vals));
        b->vals = nullptr;
        HIP_OK(hipMalloc((void**)&b->vals, sizeof(double2) * (size_t)b->L * need));
        b->nvals = need;
    }
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (circ->num_blocks > 0 && !(0 <= block_from && block_from <= block_to && block_to <= circ->num_blocks)) return failf("invalid block range");
    // vh = V^H target
    if (clone(b, b->target, b->vh)) return 1;
    if (apply_circuit_all(b, b->vh, circ, thetas, T, true, trunc_thr, max_bond)) return 1;
    // h = <lhs | vh>
    if (clone(b, b->lhs, b->w) || clone(b, b->vh, b->z)) return 1;
    if (overlap_all(b, reinterpret_cast<std::complex<double>*>(h_out))) return 1;
    // gradient walk on (w, z) = (lhs, vh)
    if (gradient_all(b, circ, thetas, T, trunc_thr, max_bond, block_from, block_to, front_layer != 0, reinterpret_cast<std::complex<double>*>(grad_out)))
        return 1;
    for (int l = 0; l < b->L; ++l) {
        if (discarded_out) discarded_out[l] = b->vh.discarded[l];
        if (max_bond_out) {
            int mx = 1;
            for (int q = 0; q <= b->n; ++q) mx = std::max(mx, b->vh.dim(l, q));
            max_bond_out[l] = mx;
        }
    }
    return 0;
}

}  // extern "C"
