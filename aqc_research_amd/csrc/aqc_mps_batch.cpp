// Lockstep lanes of the native MPS engine (aqc_mpsb_*): L independent problems that share ONE ansatz -- the seeds / restarts /
// targets of a horizon, mps_dot_objective.py:41 called once per job in the reference (job_executor.py:141) -- walk the circuit
// together, device-resident from the first gate to the last inner product.
//
// The single-lane engine (aqc_mps_engine.cpp) is a chain of small dependent launches with one host decision (the truncation rank)
// per 2-qubit gate: one lane cannot fill the device and host threads saturate the runtime's launch path.  Here
//   * every step of the walk is ONE launch for all lanes (grid dimension = lane);
//   * a lane's bond dimensions live on the device: a truncated 2-qubit gate is one workgroup per lane that forms the two-site
//     tensor in LDS, runs the Jacobi sweeps there, orders the singular values, takes the rank / truncation decision of
//     gate_adjacent (aqc_mps_engine.cpp) on its own values and writes the new tensors, Schmidt values and bond dimension;
//   * gate matrices are worked out by the kernels from thetas[lane][index], so a launch's arguments do not depend on the lane,
//     the parameters or the data: the host enqueues the whole evaluation (V^H|target>, <lhs|.>, the gate-by-gate gradient with
//     its cached environments) without waiting once, and reads objective values, gradients' inner products, status words and
//     bond statistics in one transfer at the end.
// Workgroups are sized for the bonds seen so far (`hint`: LDS and threads for bonds <= 4 / 6 / 8 / ... / 32); a lane that outgrows
// the launch raises a status bit and the evaluation is repeated at full size -- never a wrong result.  Bonds above 32 (a work
// matrix larger than 64 x 64 does not fit one workgroup's LDS) are refused: the caller goes to the single-lane engine.
// Arithmetic and truncation rule are the single-lane engine's (same device bodies, aqc_svd.hip), lane by lane.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <string>
#include <utility>

#include "../../include/aqc_hip.h"
#include "aqc_launch.h"
#include "aqc_mps_host.h"

using namespace aqc;

namespace {

struct Lanes {   // L MPS of n sites, flat device storage with fixed strides; T_q = Gamma_q diag(lambda_q) like the single-lane engine
    int n = 0, L = 0, nb = 1;
    char* base = nullptr;        // one allocation: T | lam | discarded | dims, so that a state is cloned by ONE copy
    size_t bytes = 0;
    LaneMps dev{};
    int max_dim_in = 1;          // largest bond of what was loaded (host side)
    int alloc(int n_, int L_) {
        n = n_; L = L_; nb = std::max(n - 1, 1);
        const size_t t = sizeof(double2) * (size_t)L * n * kLaneSite, lam = sizeof(double) * (size_t)L * nb * kLaneCap;
        const size_t disc = sizeof(double) * (size_t)L, dims = sizeof(int) * (size_t)L * (n + 1);
        bytes = t + lam + disc + dims;
        HIP_OK(hipMalloc((void**)&base, bytes));
        dev.T = base; dev.lam = reinterpret_cast<double*>(base + t); dev.discarded = reinterpret_cast<double*>(base + t + lam);
        dev.dims = reinterpret_cast<int*>(base + t + lam + disc);
        dev.n = n; dev.pad = 0;
        return 0;
    }
    double2* site(int l, int q) const { return static_cast<double2*>(dev.T) + ((size_t)l * n + q) * kLaneSite; }
    double* lambda(int l, int b) const { return dev.lam + ((size_t)l * nb + b) * kLaneCap; }
    void release() { if (base) (void)hipFree(base); base = nullptr; }
};

}  // namespace

struct aqc_mpsb {
    int device = 0, n = 0, L = 0;
    hipStream_t st = nullptr;
    Lanes target, lhs, vh, w, z;
    bool have_target = false, have_lhs = false;
    double* thetas = nullptr;    // [L][T] on the device
    double* h_thetas = nullptr;  // pinned staging of the same
    int T_cap = 0;
    int* status = nullptr;       // [L] status bits | [L] largest bond a gate has produced
    // environments of the pair (w, z), see aqc_mps_engine.cpp
    double2* env_l = nullptr;    // [L][n + 1][kLaneEnv]
    double2* env_r = nullptr;    // [L][n][kLaneEnv]
    double2* e0 = nullptr;       // [L][kLaneEnv]
    double2* e1 = nullptr;
    double2* vals = nullptr;     // [L][nvals]
    char* h_out = nullptr;       // pinned: vals | status+peak | discarded | dims of vh
    size_t h_out_bytes = 0;
    int nvals = 0, valid_l = 0, valid_r = 0;
    int hint = kLaneCap;         // bonds the launches are sized for
    int peak_seen = 0;           // largest bond a gate of the previous evaluation produced (0: none yet)
};

namespace {

void destroy(aqc_mpsb* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->st) (void)hipStreamSynchronize(b->st);
    for (Lanes* s : {&b->target, &b->lhs, &b->vh, &b->w, &b->z}) s->release();
    for (void* p : {(void*)b->thetas, (void*)b->status, (void*)b->env_l, (void*)b->env_r, (void*)b->e0, (void*)b->e1, (void*)b->vals})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)b->h_thetas, (void*)b->h_out})
        if (p) (void)hipHostFree(p);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
}

int clone(aqc_mpsb* b, const Lanes& src, Lanes& dst) {
    HIP_OK(hipMemcpyAsync(dst.base, src.base, src.bytes, hipMemcpyDeviceToDevice, b->st));
    return 0;
}

LaneRot rot(int kind, int idx, double scale) { return LaneRot{kind, idx, scale}; }
LaneGate1 g1(LaneRot a, LaneRot b = LaneRot{0, -1, 0.0}, LaneRot c = LaneRot{0, -1, 0.0}) { return LaneGate1{{a, b, c}}; }
constexpr int RZ = 1, RY = 2, RX = 3;

int gate1_all(aqc_mpsb* b, Lanes& s, int q, const LaneGate1& g, int T, Lanes* s2 = nullptr) {
    HIP_OK(launch_lanes_gate1(s.dev, s2 ? &s2->dev : nullptr, q, g, b->thetas, T, b->L, b->hint, b->st));
    return 0;
}

int gate_adjacent_all(aqc_mpsb* b, Lanes& s, Lanes* s2, int q, const LaneGate2& g, int T, double trunc_thr, int max_bond) {
    HIP_OK(launch_lanes_gate2(s.dev, s2 ? &s2->dev : nullptr, q, g, b->thetas, T, trunc_thr, max_bond, b->status, b->status + b->L, b->L, b->hint, b->st));
    return 0;
}

// entangler of a block (control c, target t) on any pair of qubits of `s` (and of `s2`, when given: the two operands of the gradient walk
// take every gate in one launch): swaps bring the upper one next to the lower one and back
int gate2_pair_all(aqc_mpsb* b, Lanes& s, Lanes* s2, int ctrl, int targ, int kind, int idx, double scale, int T, double trunc_thr, int max_bond) {
    const LaneGate2 swap{0, -1, 0, 0, 0.0};
    const int lo = std::min(ctrl, targ), hi = std::max(ctrl, targ);
    for (int p = hi - 1; p > lo; --p)
        if (gate_adjacent_all(b, s, s2, p, swap, T, trunc_thr, max_bond)) return 1;
    const LaneGate2 g{kind, idx, ctrl > targ ? 1 : 0, 0, scale};
    if (gate_adjacent_all(b, s, s2, lo, g, T, trunc_thr, max_bond)) return 1;
    for (int p = lo + 1; p < hi; ++p)
        if (gate_adjacent_all(b, s, s2, p, swap, T, trunc_thr, max_bond)) return 1;
    return 0;
}

int entangler_kind(const aqc_circuit* c) { return c->entangler == AQC_CX ? 1 : (c->entangler == AQC_CZ ? 2 : 3); }

// V(theta_l) or V(theta_l)^H on every lane (apply_circuit of aqc_mps_engine.cpp; core_operations.py:671-708, :787-818)
int apply_circuit_all(aqc_mpsb* b, Lanes& s, const aqc_circuit* c, int T, bool inverse, double trunc_thr, int max_bond) {
    const int n = b->n, tpb = c->entangler == AQC_CP ? 5 : 4;
    const bool cx = c->entangler == AQC_CX, cp = c->entangler == AQC_CP;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    const int ek = entangler_kind(c), rt = cx ? RX : RZ;
    const LaneGate1 pre = g1(rot(RZ, -1, -half_pi)), post = g1(rot(RZ, -1, half_pi));
    if (!inverse) {
        for (int q = 0; q < n; ++q)
            if (gate1_all(b, s, q, g1(rot(RZ, 3 * q, 1.0), rot(RY, 3 * q + 1, 1.0), rot(RZ, 3 * q + 2, 1.0)), T)) return 1;
        for (const BlockRef& blk : blocks) {
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 0 && gate1_all(b, s, blk.c, pre, T)) return 1;
            if (gate2_pair_all(b, s, nullptr, blk.c, blk.t, ek, cp ? p + 4 : -1, 1.0, T, trunc_thr, max_bond)) return 1;
            if (gate1_all(b, s, blk.c, g1(rot(RZ, p + 1, 1.0), rot(RY, p, 1.0)), T)) return 1;
            if (gate1_all(b, s, blk.t, g1(rot(rt, p + 3, 1.0), rot(RY, p + 2, 1.0)), T)) return 1;
            if (c->trotter && blk.i % 3 == 2 && gate1_all(b, s, blk.t, post, T)) return 1;
        }
    } else {
        for (auto it = blocks.rbegin(); it != blocks.rend(); ++it) {
            const BlockRef& blk = *it;
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 2 && gate1_all(b, s, blk.t, pre, T)) return 1;
            if (gate1_all(b, s, blk.t, g1(rot(RY, p + 2, -1.0), rot(rt, p + 3, -1.0)), T)) return 1;
            if (gate1_all(b, s, blk.c, g1(rot(RY, p, -1.0), rot(RZ, p + 1, -1.0)), T)) return 1;
            if (gate2_pair_all(b, s, nullptr, blk.c, blk.t, ek, cp ? p + 4 : -1, -1.0, T, trunc_thr, max_bond)) return 1;
            if (c->trotter && blk.i % 3 == 0 && gate1_all(b, s, blk.c, post, T)) return 1;
        }
        for (int q = 0; q < n; ++q)
            if (gate1_all(b, s, q, g1(rot(RZ, 3 * q + 2, -1.0), rot(RY, 3 * q + 1, -1.0), rot(RZ, 3 * q, -1.0)), T)) return 1;
    }
    return 0;
}

// ---- environments of the pair (w, z), all lanes per launch (struct Environments of aqc_mps_engine.cpp) -------------------------
constexpr size_t kEnvL(int n) { return (size_t)(n + 1) * kLaneEnv; }
constexpr size_t kEnvR(int n) { return (size_t)n * kLaneEnv; }

int env_init(aqc_mpsb* b) {
    HIP_OK(launch_lanes_env_init(b->env_l, kEnvL(b->n), b->env_r + (size_t)(b->n - 1) * kLaneEnv, kEnvR(b->n), b->L, b->st));
    b->valid_l = 0;
    b->valid_r = b->n - 1;
    return 0;
}
void env_touched(aqc_mpsb* b, int lo, int hi) { b->valid_l = std::min(b->valid_l, lo); b->valid_r = std::max(b->valid_r, hi); }

// out[u][v] = sum_bit sum_xy conj(A_p[bit][x][u]) in[x][y] B_p[bit][y][v]; `op` (may be null) sits on w's side of site p
int step_left_all(aqc_mpsb* b, int p, const double2* in, size_t in_stride, const M2* op, double2* out, size_t out_stride) {
    double g8[8];
    if (op) { const M2 gh = {{std::conj(op->m[0]), std::conj(op->m[2]), std::conj(op->m[1]), std::conj(op->m[3])}}; pack(gh, g8); }
    HIP_OK(launch_lanes_env_left(b->w.dev, b->z.dev, p, in, in_stride, out, out_stride, op ? g8 : nullptr, b->L, b->st));
    return 0;
}
// left environments up to site lo, right environments down to site hi
int env_advance(aqc_mpsb* b, int lo, int hi) {
    const int n = b->n;
    for (; b->valid_l < lo; ++b->valid_l) {
        const int p = b->valid_l;
        if (step_left_all(b, p, b->env_l + (size_t)p * kLaneEnv, kEnvL(n), nullptr, b->env_l + (size_t)(p + 1) * kLaneEnv, kEnvL(n))) return 1;
    }
    for (; b->valid_r > hi; --b->valid_r) {   // Rc[p - 1][x][y] = sum_bit A_p[bit][x][u] (Rc[p] B_p[bit]^H)[u][y]
        const int p = b->valid_r;
        HIP_OK(launch_lanes_env_right(b->w.dev, b->z.dev, p, b->env_r + (size_t)p * kLaneEnv, kEnvR(n), b->env_r + (size_t)(p - 1) * kLaneEnv, kEnvR(n), b->L,
                                      b->st));
    }
    return 0;
}
// vals[l][slot] = <(G_1 on q_1)(G_2 on q_2) w_l | z_l>, q_1 < q_2 (nops = 1: only q_1)
int dot_all(aqc_mpsb* b, int slot, int nops, const int* q, const M2* const* g) {
    if (slot >= b->nvals) return failf("inner-product slot out of range");
    const int lo = q[0], hi = q[nops - 1], n = b->n;
    if (env_advance(b, lo, hi)) return 1;
    const double2* cur = b->env_l + (size_t)lo * kLaneEnv;
    size_t cur_stride = kEnvL(n);
    double2* pp[2] = {b->e0, b->e1};
    for (int p = lo; p <= hi; ++p) {
        const M2* op = p == q[0] ? g[0] : (nops > 1 && p == q[1] ? g[1] : nullptr);
        double2* out = pp[(p - lo) & 1];
        if (step_left_all(b, p, cur, cur_stride, op, out, kLaneEnv)) return 1;
        cur = out;
        cur_stride = kLaneEnv;
    }
    HIP_OK(launch_lanes_env_dot(b->w.dev, b->z.dev, hi, cur, cur_stride, b->env_r + (size_t)hi * kLaneEnv, kEnvR(n), b->vals, b->nvals, slot, b->L, b->st));
    return 0;
}

// the gate-by-gate gradient walk of mps_dot_objective.py:41-242 on every lane: w = lhs, z = vh (both consumed); the inner products go
// to slots 1, 2, ... of vals (slot 0 holds <lhs|vh>); rec = (theta index, factor) per slot
int gradient_all(aqc_mpsb* b, const aqc_circuit* c, int T, double trunc_thr, int max_bond, int lo_blk, int hi_blk, bool front_layer,
                 std::vector<std::pair<int, cd>>& rec) {
    const int n = b->n, tpb = c->entangler == AQC_CP ? 5 : 4;
    const bool cx = c->entangler == AQC_CX, cp = c->entangler == AQC_CP;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    const int ek = entangler_kind(c);
    rec.clear();
    auto both = [&](int q, const LaneGate1& g) -> int {
        if (gate1_all(b, b->w, q, g, T, &b->z)) return 1;
        env_touched(b, q, q);
        return 0;
    };
    auto record = [&](int tindex, cd factor, int nops, const int* q, const M2* const* g) -> int {
        if (dot_all(b, 1 + (int)rec.size(), nops, q, g)) return 1;
        rec.emplace_back(tindex, factor);
        return 0;
    };
    // rotation on site q of both operands + its inner product 0.5j <P w|z>: one launch (the environments do not involve site q)
    auto rotate_and_record = [&](int tindex, int q, const LaneGate1& g, const M2* op) -> int {
        const int slot = 1 + (int)rec.size();
        if (slot >= b->nvals) return failf("inner-product slot out of range");
        if (env_advance(b, q, q)) return 1;
        const M2 gh = {{std::conj(op->m[0]), std::conj(op->m[2]), std::conj(op->m[1]), std::conj(op->m[3])}};
        double g8[8];
        pack(gh, g8);
        HIP_OK(launch_lanes_grad_step(b->w.dev, b->z.dev, q, g, b->thetas, T, b->env_l + (size_t)q * kLaneEnv, kEnvL(n), b->env_r + (size_t)q * kLaneEnv, kEnvR(n),
                                      g8, b->e0, b->vals, b->nvals, slot, b->L, b->st));
        env_touched(b, q, q);
        rec.emplace_back(tindex, cd(0, 0.5));
        return 0;
    };
    for (int q = 0; q < n; ++q) {
        const int slots[3] = {2, 1, 0};
        for (int k = 0; k < 3; ++k) {
            const int slot = slots[k];
            const bool is_y = slot == 1;
            const LaneGate1 g = g1(rot(is_y ? RY : RZ, 3 * q + slot, 1.0));
            if (front_layer ? rotate_and_record(3 * q + slot, q, g, is_y ? &kPauliY : &kPauliZ) : both(q, g)) return 1;
        }
    }
    const LaneGate1 pre = g1(rot(RZ, -1, -half_pi)), post = g1(rot(RZ, -1, half_pi));
    for (const BlockRef& blk : blocks) {
        const int base = 3 * n + tpb * blk.j;
        const bool live = lo_blk <= blk.j && blk.j < hi_blk;
        if (c->trotter && blk.i % 3 == 0 && both(blk.c, pre)) return 1;
        if (live && cp) {
            const int qq[2] = {std::min(blk.c, blk.t), std::max(blk.c, blk.t)};
            const M2* gg[2] = {&kProj1, &kProj1};
            if (record(base + 4, cd(0, -1.0), 2, qq, gg)) return 1;
        }
        if (gate2_pair_all(b, b->z, &b->w, blk.c, blk.t, ek, cp ? base + 4 : -1, 1.0, T, trunc_thr, max_bond)) return 1;
        env_touched(b, std::min(blk.c, blk.t), std::max(blk.c, blk.t));
        const int qs[4] = {blk.c, blk.c, blk.t, blk.t};
        const int kinds[4] = {RY, RZ, RY, cx ? RX : RZ};
        const M2* ps[4] = {&kPauliY, &kPauliZ, &kPauliY, cx ? &kPauliX : &kPauliZ};
        for (int k = 0; k < 4; ++k) {
            const LaneGate1 g = g1(rot(kinds[k], base + k, 1.0));
            if (live ? rotate_and_record(base + k, qs[k], g, ps[k]) : both(qs[k], g)) return 1;
        }
        if (c->trotter && blk.i % 3 == 2 && both(blk.t, post)) return 1;
    }
    return 0;
}

int load_lanes(aqc_mpsb* b, Lanes& dst, aqc_mps* const* src, int shared) {
    if (!src) return failf("null MPS list");
    const int n = b->n, L = b->L, distinct = shared ? 1 : L;
    std::vector<int> dims_all((size_t)L * (n + 1));
    std::vector<double> disc(L);
    dst.max_dim_in = 1;
    for (int l = 0; l < distinct; ++l) {
        const aqc_mps* m = src[l];
        if (!m) return failf("null MPS handle (lane %d)", l);
        if (aqc_mps_num_qubits(m) != n) return failf("lane %d: the MPS has %d qubits, the batch %d", l, aqc_mps_num_qubits(m), n);
        std::vector<int32_t> dims(n + 1);
        if (aqc_mps_dims(m, dims.data())) return 1;
        for (int q = 0; q <= n; ++q) {
            if (dims[q] > kLaneCap)
                return failf("lane %d: bond dimension %d exceeds the %d of the lockstep lanes (use the single-lane engine)", l, dims[q], kLaneCap);
            dims_all[(size_t)l * (n + 1) + q] = dims[q];
            dst.max_dim_in = std::max(dst.max_dim_in, (int)dims[q]);
        }
        disc[l] = aqc_mps_discarded_weight(m);
        for (int q = 0; q < n; ++q) {
            const void* site = nullptr;
            const double* lam = nullptr;
            if (mps_peek(m, q, &site, &lam)) return 1;
            HIP_OK(hipMemcpyAsync(dst.site(l, q), site, sizeof(double2) * 2 * dims[q] * dims[q + 1], hipMemcpyDeviceToDevice, b->st));
            if (q < n - 1) HIP_OK(hipMemcpyAsync(dst.lambda(l, q), lam, sizeof(double) * dims[q + 1], hipMemcpyDeviceToDevice, b->st));
        }
    }
    if (shared) {   // one state for all lanes: lane 0 is replicated by doubling (lanes [0, c) -> [c, 2c)), 2 log2(L) copies instead of 2 n L
        for (int l = 1; l < L; ++l) {
            std::copy(dims_all.begin(), dims_all.begin() + (n + 1), dims_all.begin() + (size_t)l * (n + 1));
            disc[l] = disc[0];
        }
        const size_t t_lane = sizeof(double2) * (size_t)n * kLaneSite, lam_lane = sizeof(double) * (size_t)dst.nb * kLaneCap;
        for (int c = 1; c < L; c *= 2) {
            const int cnt = std::min(c, L - c);
            HIP_OK(hipMemcpyAsync(static_cast<char*>(dst.dev.T) + t_lane * c, dst.dev.T, t_lane * cnt, hipMemcpyDeviceToDevice, b->st));
            HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(dst.dev.lam) + lam_lane * c, dst.dev.lam, lam_lane * cnt, hipMemcpyDeviceToDevice, b->st));
        }
    }
    HIP_OK(hipMemcpyAsync(dst.dev.dims, dims_all.data(), sizeof(int) * dims_all.size(), hipMemcpyHostToDevice, b->st));
    HIP_OK(hipMemcpyAsync(dst.dev.discarded, disc.data(), sizeof(double) * disc.size(), hipMemcpyHostToDevice, b->st));
    HIP_OK(hipStreamSynchronize(b->st));   // (the host vectors above must outlive their copies)
    return 0;
}

// launches are sized (LDS, threads) for one of a few bond limits: what the loaded states need exactly, what the gates of the previous
// evaluation produced plus a quarter of head room
int hint_for(int loaded, int produced) {
    const int want = std::max(loaded, produced + (produced + 3) / 4);
    for (int h : {4, 6, 8, 12, 16, 20, 24}) if (want <= h) return h;
    return kLaneCap;
}

}  // namespace

extern "C" {

int aqc_mpsb_create(int device, int num_qubits, int lanes, aqc_mpsb** out) {
    if (!out) return failf("null output");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return failf("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return failf("device out of range");
    if (num_qubits < 2 || num_qubits > 4096 || lanes < 1 || lanes > 32767) return failf("number of qubits / lanes out of range");
    HIP_OK(hipSetDevice(device));
    aqc_mpsb* b = new aqc_mpsb();
    b->device = device; b->n = num_qubits; b->L = lanes;
    auto bad = [&]() { destroy(b); return 1; };
    if (hipStreamCreate(&b->st) != hipSuccess) { failf("hipStreamCreate failed"); return bad(); }
    for (Lanes* s : {&b->target, &b->lhs, &b->vh, &b->w, &b->z})
        if (s->alloc(num_qubits, lanes)) return bad();
    const size_t L = lanes;
    if (hipMalloc((void**)&b->status, sizeof(int) * 2 * L) != hipSuccess ||
        hipMalloc((void**)&b->env_l, sizeof(double2) * L * kEnvL(num_qubits)) != hipSuccess ||
        hipMalloc((void**)&b->env_r, sizeof(double2) * L * kEnvR(num_qubits)) != hipSuccess ||
        hipMalloc((void**)&b->e0, sizeof(double2) * L * kLaneEnv) != hipSuccess || hipMalloc((void**)&b->e1, sizeof(double2) * L * kLaneEnv) != hipSuccess) {
        failf("allocation of the lockstep lanes failed (%d lanes, %d qubits)", lanes, num_qubits);
        return bad();
    }
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
    if (!(trunc_thr >= 0.0)) return failf("trunc_thr must be non-negative");
    if (max_bond > kLaneCap) return failf("the lockstep lanes keep bonds up to %d", kLaneCap);
    HIP_OK(hipSetDevice(b->device));
    const int n = b->n, L = b->L, tpb = circ->entangler == AQC_CP ? 5 : 4, T = 3 * n + tpb * circ->num_blocks;
    const int nblk = (int)blocks_of(circ).size();
    const int need = 1 + 3 * n + tpb * nblk;
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (block_from > block_to || block_to > circ->num_blocks) return failf("invalid block range");
    const size_t out_bytes = sizeof(double2) * (size_t)L * need + sizeof(int) * 2 * (size_t)L + sizeof(double) * (size_t)L + sizeof(int) * (size_t)L * (n + 1);
    if (need > b->nvals || T > b->T_cap || out_bytes > b->h_out_bytes) {
        HIP_OK(hipStreamSynchronize(b->st));
        if (b->vals) HIP_OK(hipFree(b->vals));
        if (b->thetas) HIP_OK(hipFree(b->thetas));
        if (b->h_thetas) HIP_OK(hipHostFree(b->h_thetas));
        if (b->h_out) HIP_OK(hipHostFree(b->h_out));
        b->vals = nullptr; b->thetas = nullptr; b->h_thetas = nullptr; b->h_out = nullptr; b->nvals = 0; b->T_cap = 0; b->h_out_bytes = 0;
        const int nv = std::max(need, b->nvals), tc = std::max(T, 1);
        const size_t ob = sizeof(double2) * (size_t)L * nv + sizeof(int) * 2 * (size_t)L + sizeof(double) * (size_t)L + sizeof(int) * (size_t)L * (n + 1);
        HIP_OK(hipMalloc((void**)&b->vals, sizeof(double2) * (size_t)L * nv));
        HIP_OK(hipMalloc((void**)&b->thetas, sizeof(double) * (size_t)L * tc));
        HIP_OK(hipHostMalloc((void**)&b->h_thetas, sizeof(double) * (size_t)L * tc, hipHostMallocDefault));
        HIP_OK(hipHostMalloc((void**)&b->h_out, ob, hipHostMallocDefault));
        b->nvals = nv; b->T_cap = tc; b->h_out_bytes = ob;
    }
    memcpy(b->h_thetas, thetas, sizeof(double) * (size_t)L * T);
    HIP_OK(hipMemcpyAsync(b->thetas, b->h_thetas, sizeof(double) * (size_t)L * T, hipMemcpyHostToDevice, b->st));
    char* h_vals = b->h_out;
    char* h_status = h_vals + sizeof(double2) * (size_t)L * b->nvals;
    char* h_disc = h_status + sizeof(int) * 2 * (size_t)L;
    char* h_dims = h_disc + sizeof(double) * (size_t)L;
    std::vector<std::pair<int, cd>> rec;
    const int in_peak = std::max(b->target.max_dim_in, b->lhs.max_dim_in);
    b->hint = b->peak_seen == 0 ? kLaneCap : hint_for(in_peak, b->peak_seen);   // first evaluation: full size
    for (int attempt = 0;; ++attempt) {
        HIP_OK(hipMemsetAsync(b->status, 0, sizeof(int) * 2 * (size_t)L, b->st));
        // vh = V^H target
        if (clone(b, b->target, b->vh)) return 1;
        if (apply_circuit_all(b, b->vh, circ, T, true, trunc_thr, max_bond)) return 1;
        // h = <lhs | vh> (slot 0), then the gradient walk on (w, z) = (lhs, vh)
        if (clone(b, b->lhs, b->w) || clone(b, b->vh, b->z)) return 1;
        if (env_init(b)) return 1;
        {
            const int q = n - 1;
            const M2 eye = {{1.0, 0.0, 0.0, 1.0}};
            const M2* g = &eye;
            if (dot_all(b, 0, 1, &q, &g)) return 1;
        }
        if (gradient_all(b, circ, T, trunc_thr, max_bond, block_from, block_to, front_layer != 0, rec)) return 1;
        HIP_OK(hipMemcpyAsync(h_vals, b->vals, sizeof(double2) * (size_t)L * b->nvals, hipMemcpyDeviceToHost, b->st));
        HIP_OK(hipMemcpyAsync(h_status, b->status, sizeof(int) * 2 * (size_t)L, hipMemcpyDeviceToHost, b->st));
        HIP_OK(hipMemcpyAsync(h_disc, b->vh.dev.discarded, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, b->st));
        HIP_OK(hipMemcpyAsync(h_dims, b->vh.dev.dims, sizeof(int) * (size_t)L * (n + 1), hipMemcpyDeviceToHost, b->st));
        HIP_OK(hipStreamSynchronize(b->st));
        const int* st = reinterpret_cast<const int*>(h_status);
        int flags = 0, peak = 1;
        for (int l = 0; l < L; ++l) { flags |= st[l]; peak = std::max(peak, st[L + l]); }
        if ((flags & kLaneLdsShort) && attempt == 0 && b->hint < kLaneCap) {   // a lane outgrew the launch size: once more at full size
            b->hint = kLaneCap;
            continue;
        }
        b->peak_seen = std::max(peak, 1);
        for (int l = 0; l < L; ++l) {
            if (st[l] & kLaneOverflow)
                return failf("lane %d: a bond grows beyond the %d of the lockstep lanes (set max_bond <= %d or use the single-lane engine)", l, kLaneCap, kLaneCap);
            if (st[l] & kLaneNoConv) return failf("Jacobi SVD: no convergence within 60 sweeps (lane %d of the lockstep lanes)", l);
            if (st[l] & kLaneZero) return failf("2-qubit gate produced a zero or non-finite state (lane %d of the lockstep lanes)", l);
            if (st[l] & kLaneLdsShort) return failf("internal: workgroup of the lockstep lanes too small at full size (lane %d)", l);
        }
        break;
    }
    const cd* vals = reinterpret_cast<const cd*>(h_vals);
    cd* hh = reinterpret_cast<cd*>(h_out);
    cd* grad = reinterpret_cast<cd*>(grad_out);
    const double* disc = reinterpret_cast<const double*>(h_disc);
    const int* dims = reinterpret_cast<const int*>(h_dims);
    for (int l = 0; l < L; ++l) {
        hh[l] = vals[(size_t)l * b->nvals];
        cd* g = grad + (size_t)l * T;
        std::fill(g, g + T, cd(0.0, 0.0));
        for (size_t i = 0; i < rec.size(); ++i) g[rec[i].first] += rec[i].second * vals[(size_t)l * b->nvals + 1 + i];
        if (discarded_out) discarded_out[l] = disc[l];
        if (max_bond_out) {
            int mx = 1;
            for (int q = 0; q <= n; ++q) mx = std::max(mx, dims[(size_t)l * (n + 1) + q]);
            max_bond_out[l] = mx;
        }
    }
    return 0;
}

}  // extern "C"
