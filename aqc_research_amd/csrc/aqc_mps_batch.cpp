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
// Arithmetic and truncation rule are the single-lane engine's (same device bodies, aqc_mps_dev.h), lane by lane.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <map>
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

// the gates of V or V^H in levels of pairwise disjoint gates (apply_circuit_all below), cached per circuit
struct Level { int off1, n1, off2, n2; };
struct Schedule {
    std::vector<Level> levels;
    LaneOp1* ops1 = nullptr;   // device tables, level after level
    LaneOp2* ops2 = nullptr;
};

}  // namespace

struct aqc_mpsb {
    int device = 0, n = 0, L = 0;
    hipStream_t st = nullptr;
    Lanes target, lhs, vh, w, z;
    std::map<std::string, Schedule> schedules;
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
    int peak_vh = 0, peak_grad = 0;   // largest bond the gates of the last V^H / gradient phase produced (0: none yet)
    int active = 0;              // lanes the gate launches cover (all, or the first half while V^H runs for lanes that repeat it)
    int cur_T = 0, cur_max_bond = 0;   // what the last aqc_mpsb_vh / eval ran with (the gradient phase continues from it)
    double cur_trunc = 0.0;
    bool vh_ready = false;
    // statistics of the truncated 2-qubit gates (aqc_mpsb_gate2_stats): work of the Jacobi sweeps on the device, and -- while
    // profiling is on -- the duration of every lanes_gate2 launch from a pool of event pairs (read back when the pool is full
    // and when the figures are asked for)
    unsigned long long* jstats = nullptr;   // device [4]
    bool prof = false;
    std::vector<hipEvent_t> ev;             // pairs
    size_t ev_used = 0;
    double gate2_ms = 0.0;
    long long gate2_launches = 0;
    uint8_t* bits = nullptr;     // basis-state patterns of aqc_mpsb_set_lhs_basis and their pinned staging
    uint8_t* h_bits = nullptr;
    size_t bits_cap = 0;
};

namespace {

void destroy(aqc_mpsb* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->st) (void)hipStreamSynchronize(b->st);
    for (Lanes* s : {&b->target, &b->lhs, &b->vh, &b->w, &b->z}) s->release();
    for (auto& kv : b->schedules) { if (kv.second.ops1) (void)hipFree(kv.second.ops1); if (kv.second.ops2) (void)hipFree(kv.second.ops2); }
    for (void* p : {(void*)b->thetas, (void*)b->status, (void*)b->env_l, (void*)b->env_r, (void*)b->e0, (void*)b->e1, (void*)b->vals, (void*)b->bits,
                    (void*)b->jstats})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    for (void* p : {(void*)b->h_thetas, (void*)b->h_out, (void*)b->h_bits})
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
    HIP_OK(launch_lanes_gate1(s.dev, s2 ? &s2->dev : nullptr, nullptr, 1, LaneOp1{q, 0, g}, b->thetas, T, b->active, b->hint, b->st));
    return 0;
}

int gate2_events_flush(aqc_mpsb* b) {   // sums the recorded pairs (waits for the stream)
    if (b->ev_used == 0) return 0;
    HIP_OK(hipStreamSynchronize(b->st));
    for (size_t i = 0; i + 1 < b->ev_used; i += 2) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]));
        b->gate2_ms += ms;
        b->gate2_launches += 1;
    }
    b->ev_used = 0;
    return 0;
}
struct Gate2Scope {   // brackets one lanes_gate2 launch with an event pair while profiling is on
    aqc_mpsb* b;
    bool on;
    explicit Gate2Scope(aqc_mpsb* b_) : b(b_), on(b_->prof) {
        if (!on) return;
        if (b->ev_used + 2 > b->ev.size() && gate2_events_flush(b)) { on = false; return; }
        if (hipEventRecord(b->ev[b->ev_used], b->st) != hipSuccess) on = false;
    }
    ~Gate2Scope() {
        if (on && hipEventRecord(b->ev[b->ev_used + 1], b->st) == hipSuccess) b->ev_used += 2;
    }
};

int gate_adjacent_all(aqc_mpsb* b, Lanes& s, Lanes* s2, int q, const LaneGate2& g, int T, double trunc_thr, int max_bond) {
    Gate2Scope scope(b);
    HIP_OK(launch_lanes_gate2(s.dev, s2 ? &s2->dev : nullptr, nullptr, 1, LaneOp2{q, 0, g}, b->thetas, T, trunc_thr, max_bond, b->status, b->status + b->L,
                              b->active, b->hint, b->st, b->jstats));
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

// V(theta_l) or V(theta_l)^H on every lane (apply_circuit of aqc_mps_engine.cpp; core_operations.py:671-708, :787-818), layer by layer:
// the gates in program order (long-range entanglers already routed by swaps) are levelled as soon as possible -- a gate goes one level
// above the last gate on any of its sites.  The gates of a level act on pairwise disjoint site tensors, Schmidt vectors and bond
// dimensions, so they run in ONE launch per kind (1-qubit, 2-qubit) and leave exactly the bits the one-after-the-other order leaves:
// a brickwork layer of n / 2 entanglers is one launch of n / 2 x lanes workgroups instead of n / 2 dependent launches.
void emit_circuit(const aqc_circuit* c, int n, bool inverse, std::vector<std::pair<int, LaneOp1>>& o1, std::vector<std::pair<int, LaneOp2>>& o2) {
    // (first = position in program order, shared by both lists)
    const int tpb = c->entangler == AQC_CP ? 5 : 4;
    const bool cx = c->entangler == AQC_CX, cp = c->entangler == AQC_CP;
    const std::vector<BlockRef> blocks = blocks_of(c);
    const double half_pi = 1.5707963267948966;
    const int ek = entangler_kind(c), rt = cx ? RX : RZ;
    const LaneGate1 pre = g1(rot(RZ, -1, -half_pi)), post = g1(rot(RZ, -1, half_pi));
    int pos = 0;
    auto one = [&](int q, const LaneGate1& g) { o1.emplace_back(pos++, LaneOp1{q, 0, g}); };
    auto two = [&](int ctrl, int targ, int idx, double scale) {   // swaps bring the upper qubit next to the lower one and back
        const LaneGate2 swap{0, -1, 0, 0, 0.0};
        const int lo = std::min(ctrl, targ), hi = std::max(ctrl, targ);
        for (int p = hi - 1; p > lo; --p) o2.emplace_back(pos++, LaneOp2{p, 0, swap});
        o2.emplace_back(pos++, LaneOp2{lo, 0, LaneGate2{ek, idx, ctrl > targ ? 1 : 0, 0, scale}});
        for (int p = lo + 1; p < hi; ++p) o2.emplace_back(pos++, LaneOp2{p, 0, swap});
    };
    if (!inverse) {
        for (int q = 0; q < n; ++q) one(q, g1(rot(RZ, 3 * q, 1.0), rot(RY, 3 * q + 1, 1.0), rot(RZ, 3 * q + 2, 1.0)));
        for (const BlockRef& blk : blocks) {
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 0) one(blk.c, pre);
            two(blk.c, blk.t, cp ? p + 4 : -1, 1.0);
            one(blk.c, g1(rot(RZ, p + 1, 1.0), rot(RY, p, 1.0)));
            one(blk.t, g1(rot(rt, p + 3, 1.0), rot(RY, p + 2, 1.0)));
            if (c->trotter && blk.i % 3 == 2) one(blk.t, post);
        }
    } else {
        for (auto it = blocks.rbegin(); it != blocks.rend(); ++it) {
            const BlockRef& blk = *it;
            const int p = 3 * n + tpb * blk.j;
            if (c->trotter && blk.i % 3 == 2) one(blk.t, pre);
            one(blk.t, g1(rot(RY, p + 2, -1.0), rot(rt, p + 3, -1.0)));
            one(blk.c, g1(rot(RY, p, -1.0), rot(RZ, p + 1, -1.0)));
            two(blk.c, blk.t, cp ? p + 4 : -1, -1.0);
            if (c->trotter && blk.i % 3 == 0) one(blk.c, post);
        }
        for (int q = 0; q < n; ++q) one(q, g1(rot(RZ, 3 * q + 2, -1.0), rot(RY, 3 * q + 1, -1.0), rot(RZ, 3 * q, -1.0)));
    }
}

int build_schedule(aqc_mpsb* b, const aqc_circuit* c, bool inverse, Schedule& out) {
    const int n = b->n;
    std::vector<std::pair<int, LaneOp1>> o1;
    std::vector<std::pair<int, LaneOp2>> o2;
    emit_circuit(c, n, inverse, o1, o2);
    // as-soon-as-possible levels in program order
    std::vector<int> site_level(n, -1), lev1(o1.size()), lev2(o2.size());
    size_t i1 = 0, i2 = 0;
    int depth = 0;
    while (i1 < o1.size() || i2 < o2.size()) {
        if (i2 >= o2.size() || (i1 < o1.size() && o1[i1].first < o2[i2].first)) {
            const int q = o1[i1].second.q;
            lev1[i1] = ++site_level[q];
            depth = std::max(depth, lev1[i1] + 1);
            ++i1;
        } else {
            const int q = o2[i2].second.q;
            const int lv = std::max(site_level[q], site_level[q + 1]) + 1;
            site_level[q] = site_level[q + 1] = lev2[i2] = lv;
            depth = std::max(depth, lv + 1);
            ++i2;
        }
    }
    std::vector<LaneOp1> t1;
    std::vector<LaneOp2> t2;
    out.levels.assign(depth, Level{0, 0, 0, 0});
    for (int lv = 0; lv < depth; ++lv) {
        Level& L = out.levels[lv];
        L.off1 = (int)t1.size(); L.off2 = (int)t2.size();
        for (size_t i = 0; i < o1.size(); ++i) if (lev1[i] == lv) t1.push_back(o1[i].second);
        for (size_t i = 0; i < o2.size(); ++i) if (lev2[i] == lv) t2.push_back(o2[i].second);
        L.n1 = (int)t1.size() - L.off1; L.n2 = (int)t2.size() - L.off2;
    }
    if (!t1.empty()) {
        HIP_OK(hipMalloc((void**)&out.ops1, sizeof(LaneOp1) * t1.size()));
        HIP_OK(hipMemcpy(out.ops1, t1.data(), sizeof(LaneOp1) * t1.size(), hipMemcpyHostToDevice));
    }
    if (!t2.empty()) {
        HIP_OK(hipMalloc((void**)&out.ops2, sizeof(LaneOp2) * t2.size()));
        HIP_OK(hipMemcpy(out.ops2, t2.data(), sizeof(LaneOp2) * t2.size(), hipMemcpyHostToDevice));
    }
    return 0;
}

std::string circuit_key(const aqc_circuit* c, bool inverse) {
    std::string k(reinterpret_cast<const char*>(&c->num_qubits), sizeof(int32_t) * 5);
    k.append(reinterpret_cast<const char*>(c->blocks), sizeof(int32_t) * 2 * (size_t)std::max(c->num_blocks, 0));
    k.push_back(inverse ? 1 : 0);
    return k;
}

int apply_circuit_all(aqc_mpsb* b, Lanes& s, const aqc_circuit* c, int T, bool inverse, double trunc_thr, int max_bond) {
    const std::string key = circuit_key(c, inverse);
    auto it = b->schedules.find(key);
    if (it == b->schedules.end()) {
        if (b->schedules.size() >= 16) {   // (a driver walks a few horizons; nothing keeps more than a handful of circuits alive)
            HIP_OK(hipStreamSynchronize(b->st));
            for (auto& kv : b->schedules) { if (kv.second.ops1) (void)hipFree(kv.second.ops1); if (kv.second.ops2) (void)hipFree(kv.second.ops2); }
            b->schedules.clear();
        }
        Schedule sch;
        if (build_schedule(b, c, inverse, sch)) return 1;
        it = b->schedules.emplace(key, std::move(sch)).first;
    }
    const Schedule& sch = it->second;
    for (const Level& lv : sch.levels) {
        if (lv.n1)
            HIP_OK(launch_lanes_gate1(s.dev, nullptr, sch.ops1 + lv.off1, lv.n1, LaneOp1{}, b->thetas, T, b->active, b->hint, b->st));
        if (lv.n2) {
            Gate2Scope scope(b);
            HIP_OK(launch_lanes_gate2(s.dev, nullptr, sch.ops2 + lv.off2, lv.n2, LaneOp2{}, b->thetas, T, trunc_thr, max_bond, b->status, b->status + b->L,
                                      b->active, b->hint, b->st, b->jstats));
        }
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
// to slots slot0, slot0 + 1, ... of vals; rec = (theta index, factor) per slot
int gradient_all(aqc_mpsb* b, const aqc_circuit* c, int T, double trunc_thr, int max_bond, int lo_blk, int hi_blk, bool front_layer, int slot0,
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
        if (dot_all(b, slot0 + (int)rec.size(), nops, q, g)) return 1;
        rec.emplace_back(tindex, factor);
        return 0;
    };
    // consecutive parameters on one site q: per parameter the rotation on both operands + its inner product 0.5j <P w|z>, up to three of them
    // in one launch (the environments do not involve site q: they are advanced once, in front)
    auto rotate_and_record = [&](int q, int count, const int* tindex, const LaneGate1* g, const M2* const* op) -> int {
        const int slot = slot0 + (int)rec.size();
        if (slot + count > b->nvals) return failf("inner-product slot out of range");
        if (env_advance(b, q, q)) return 1;
        LaneSteps st{};
        st.count = count;
        for (int k = 0; k < count; ++k) {
            st.g[k] = g[k];
            const M2 gh = {{std::conj(op[k]->m[0]), std::conj(op[k]->m[2]), std::conj(op[k]->m[1]), std::conj(op[k]->m[3])}};
            pack(gh, st.gh[k]);
        }
        HIP_OK(launch_lanes_grad_step(b->w.dev, b->z.dev, q, st, b->thetas, T, b->env_l + (size_t)q * kLaneEnv, kEnvL(n), b->env_r + (size_t)q * kLaneEnv, kEnvR(n),
                                      b->e0, b->vals, b->nvals, slot, b->L, b->st));
        env_touched(b, q, q);
        for (int k = 0; k < count; ++k) rec.emplace_back(tindex[k], cd(0, 0.5));
        return 0;
    };
    for (int q = 0; q < n; ++q) {   // front layer: Rz(t2), Ry(t1), Rz(t0), rightmost first (core_operations.py:921-935)
        const int tix[3] = {3 * q + 2, 3 * q + 1, 3 * q};
        const LaneGate1 gs[3] = {g1(rot(RZ, tix[0], 1.0)), g1(rot(RY, tix[1], 1.0)), g1(rot(RZ, tix[2], 1.0))};
        const M2* ops[3] = {&kPauliZ, &kPauliY, &kPauliZ};
        if (front_layer) {
            if (rotate_and_record(q, 3, tix, gs, ops)) return 1;
        } else {
            for (int k = 0; k < 3; ++k)
                if (both(q, gs[k])) return 1;
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
        const int tix[4] = {base, base + 1, base + 2, base + 3};
        const LaneGate1 gs[4] = {g1(rot(kinds[0], base, 1.0)), g1(rot(kinds[1], base + 1, 1.0)), g1(rot(kinds[2], base + 2, 1.0)), g1(rot(kinds[3], base + 3, 1.0))};
        if (live) {   // the two rotations of the control, then the two of the target: one launch per qubit
            if (rotate_and_record(blk.c, 2, tix, gs, ps) || rotate_and_record(blk.t, 2, tix + 2, gs + 2, ps + 2)) return 1;
        } else {
            for (int k = 0; k < 4; ++k)
                if (both(qs[k], gs[k])) return 1;
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


// validates, sizes the per-evaluation buffers for `circ` (inner-product slots: amplitudes, then one per parameter) and uploads thetas
int begin(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int num_amps, int* T_out) {
    if (check_circuit(circ, b->n)) return 1;
    if (!(trunc_thr >= 0.0)) return failf("trunc_thr must be non-negative");
    if (max_bond > kLaneCap) return failf("the lockstep lanes keep bonds up to %d", kLaneCap);
    HIP_OK(hipSetDevice(b->device));
    const int n = b->n, L = b->L, tpb = circ->entangler == AQC_CP ? 5 : 4, T = 3 * n + tpb * circ->num_blocks;
    const int need = num_amps + 3 * n + tpb * (int)blocks_of(circ).size();
    const auto out_size = [&](int nv) { return sizeof(double2) * (size_t)L * nv + sizeof(int) * 2 * (size_t)L + sizeof(double) * (size_t)L + sizeof(int) * (size_t)L * (n + 1); };
    if (need > b->nvals || T > b->T_cap) {
        HIP_OK(hipStreamSynchronize(b->st));
        if (b->vals) HIP_OK(hipFree(b->vals));
        if (b->thetas) HIP_OK(hipFree(b->thetas));
        if (b->h_thetas) HIP_OK(hipHostFree(b->h_thetas));
        if (b->h_out) HIP_OK(hipHostFree(b->h_out));
        b->vals = nullptr; b->thetas = nullptr; b->h_thetas = nullptr; b->h_out = nullptr;
        const int nv = std::max(need, b->nvals), tc = std::max({T, b->T_cap, 1});
        b->nvals = 0; b->T_cap = 0;
        HIP_OK(hipMalloc((void**)&b->vals, sizeof(double2) * (size_t)L * nv));
        HIP_OK(hipMalloc((void**)&b->thetas, sizeof(double) * (size_t)L * tc));
        HIP_OK(hipHostMalloc((void**)&b->h_thetas, sizeof(double) * (size_t)L * tc, hipHostMallocDefault));
        HIP_OK(hipHostMalloc((void**)&b->h_out, out_size(nv), hipHostMallocDefault));
        b->nvals = nv; b->T_cap = tc;
    }
    HIP_OK(hipStreamSynchronize(b->st));   // (the staging buffer may still feed the previous upload)
    memcpy(b->h_thetas, thetas, sizeof(double) * (size_t)L * T);
    HIP_OK(hipMemcpyAsync(b->thetas, b->h_thetas, sizeof(double) * (size_t)L * T, hipMemcpyHostToDevice, b->st));
    b->cur_T = T; b->cur_trunc = trunc_thr; b->cur_max_bond = max_bond;
    b->vh_ready = false;
    b->hint = b->peak_vh + b->peak_grad == 0 ? kLaneCap   // first evaluation: full size
                                             : hint_for(std::max(b->target.max_dim_in, b->lhs.max_dim_in), std::max(b->peak_vh, b->peak_grad));
    *T_out = T;
    return 0;
}

// lanes [L/2, L) of `s` <- lanes [0, L/2)
int replicate_half(aqc_mpsb* b, Lanes& s) {
    const size_t h = (size_t)b->L / 2, n = s.n;
    HIP_OK(hipMemcpyAsync(static_cast<double2*>(s.dev.T) + h * n * kLaneSite, s.dev.T, sizeof(double2) * h * n * kLaneSite, hipMemcpyDeviceToDevice, b->st));
    HIP_OK(hipMemcpyAsync(s.dev.lam + h * s.nb * kLaneCap, s.dev.lam, sizeof(double) * h * s.nb * kLaneCap, hipMemcpyDeviceToDevice, b->st));
    HIP_OK(hipMemcpyAsync(s.dev.discarded + h, s.dev.discarded, sizeof(double) * h, hipMemcpyDeviceToDevice, b->st));
    HIP_OK(hipMemcpyAsync(s.dev.dims + h * (n + 1), s.dev.dims, sizeof(int) * h * (n + 1), hipMemcpyDeviceToDevice, b->st));
    return 0;
}

// vh = V^H target on every lane (the first half when the second repeats it), (w, z) = (lhs, vh) with fresh environments, and the
// amplitudes <lhs|vh> (slot 0), <X_q lhs|vh> (slots 1 + q)
int enqueue_vh(aqc_mpsb* b, const aqc_circuit* circ, int T, bool half, int num_amps) {
    const int n = b->n, L = b->L;
    HIP_OK(hipMemsetAsync(b->status, 0, sizeof(int) * 2 * (size_t)L, b->st));
    if (clone(b, b->target, b->vh)) return 1;
    b->active = half ? L / 2 : L;
    const int rc = apply_circuit_all(b, b->vh, circ, T, true, b->cur_trunc, b->cur_max_bond);
    b->active = L;
    if (rc) return 1;
    if (half && replicate_half(b, b->vh)) return 1;
    if (clone(b, b->lhs, b->w) || clone(b, b->vh, b->z)) return 1;
    if (env_init(b)) return 1;
    for (int k = 1; k < num_amps; ++k) {   // flips first: site 0 upwards, the right environments are built once on the way down to site 0
        const int q = k - 1;
        const M2* g = &kPauliX;
        if (dot_all(b, k, 1, &q, &g)) return 1;
    }
    const int q = n - 1;
    const M2 eye = {{1.0, 0.0, 0.0, 1.0}};
    const M2* g = &eye;
    return dot_all(b, 0, 1, &q, &g);
}

// results of the enqueued work -> pinned host memory; 0 ok, 2 = a lane outgrew the launch size (the caller repeats at full size), 1 error
int finish(aqc_mpsb* b, bool may_retry, int* peak_out) {
    const int n = b->n, L = b->L;
    char* h_vals = b->h_out;
    char* h_status = h_vals + sizeof(double2) * (size_t)L * b->nvals;
    char* h_disc = h_status + sizeof(int) * 2 * (size_t)L;
    char* h_dims = h_disc + sizeof(double) * (size_t)L;
    HIP_OK(hipMemcpyAsync(h_vals, b->vals, sizeof(double2) * (size_t)L * b->nvals, hipMemcpyDeviceToHost, b->st));
    HIP_OK(hipMemcpyAsync(h_status, b->status, sizeof(int) * 2 * (size_t)L, hipMemcpyDeviceToHost, b->st));
    HIP_OK(hipMemcpyAsync(h_disc, b->vh.dev.discarded, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, b->st));
    HIP_OK(hipMemcpyAsync(h_dims, b->vh.dev.dims, sizeof(int) * (size_t)L * (n + 1), hipMemcpyDeviceToHost, b->st));
    HIP_OK(hipStreamSynchronize(b->st));
    const int* st = reinterpret_cast<const int*>(h_status);
    int flags = 0, peak = 1;
    for (int l = 0; l < L; ++l) { flags |= st[l]; peak = std::max(peak, st[L + l]); }
    if ((flags & kLaneLdsShort) && may_retry && b->hint < kLaneCap) {
        b->hint = kLaneCap;
        return 2;
    }
    *peak_out = peak;
    for (int l = 0; l < L; ++l) {
        if (st[l] & kLaneOverflow)
            return failf("lane %d: a bond grows beyond the %d of the lockstep lanes (set max_bond <= %d or use the single-lane engine)", l, kLaneCap, kLaneCap);
        if (st[l] & kLaneNoConv) return failf("Jacobi SVD: no convergence within 60 sweeps (lane %d of the lockstep lanes)", l);
        if (st[l] & kLaneZero) return failf("2-qubit gate produced a zero or non-finite state (lane %d of the lockstep lanes)", l);
        if (st[l] & kLaneLdsShort) return failf("internal: workgroup of the lockstep lanes too small at full size (lane %d)", l);
    }
    return 0;
}

struct Readback {
    const cd* vals; const double* disc; const int* dims;
    int max_dim(int l, int n) const {
        int mx = 1;
        for (int q = 0; q <= n; ++q) mx = std::max(mx, dims[(size_t)l * (n + 1) + q]);
        return mx;
    }
};
Readback readback(const aqc_mpsb* b) {
    const size_t L = b->L;
    const char* h_vals = b->h_out;
    const char* h_disc = h_vals + sizeof(double2) * L * b->nvals + sizeof(int) * 2 * L;
    return Readback{reinterpret_cast<const cd*>(h_vals), reinterpret_cast<const double*>(h_disc), reinterpret_cast<const int*>(h_disc + sizeof(double) * L)};
}
// grad[lane][T] from the recorded inner products (slots slot0, slot0 + 1, ...)
void assemble(const aqc_mpsb* b, const Readback& r, const std::vector<std::pair<int, cd>>& rec, int slot0, int T, cd* grad) {
    for (int l = 0; l < b->L; ++l) {
        cd* g = grad + (size_t)l * T;
        std::fill(g, g + T, cd(0.0, 0.0));
        for (size_t i = 0; i < rec.size(); ++i) g[rec[i].first] += rec[i].second * r.vals[(size_t)l * b->nvals + slot0 + i];
    }
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
    b->device = device; b->n = num_qubits; b->L = lanes; b->active = lanes;
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

/* Work and time of the truncated 2-qubit gates since the last reset.  enable: 1 starts counting (Jacobi work on the device; every
 * lanes_gate2 launch bracketed by an event pair), 0 stops, -1 leaves the switch alone.  out (may be null): [0] fp64 flops of the
 * Jacobi rotations that ran, [1] SVDs, [2] sweeps, [3] rotations, [4] lanes_gate2 launches timed, [5] their total duration in ms.
 * reset != 0 clears the figures after reading them. */
int aqc_mpsb_gate2_stats(aqc_mpsb* b, int enable, double* out, int reset) {
    if (!b) return failf("null batch");
    HIP_OK(hipSetDevice(b->device));
    if (!b->jstats) {
        HIP_OK(hipMalloc((void**)&b->jstats, 4 * sizeof(unsigned long long)));
        HIP_OK(hipMemset(b->jstats, 0, 4 * sizeof(unsigned long long)));
    }
    if (gate2_events_flush(b)) return 1;
    if (enable == 1 && b->ev.empty()) {
        b->ev.resize(2048);
        for (hipEvent_t& e : b->ev) HIP_OK(hipEventCreate(&e));
    }
    if (enable >= 0) b->prof = enable != 0;
    if (out) {
        unsigned long long h[4];
        HIP_OK(hipStreamSynchronize(b->st));
        HIP_OK(hipMemcpy(h, b->jstats, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) out[i] = (double)h[i];
        out[4] = (double)b->gate2_launches;
        out[5] = b->gate2_ms;
    }
    if (reset) {
        HIP_OK(hipMemset(b->jstats, 0, 4 * sizeof(unsigned long long)));
        b->gate2_ms = 0.0;
        b->gate2_launches = 0;
    }
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

/* lhs states of all lanes = computational-basis states, built on the device: bits[lane][n] (0 / 1), bit q = qubit q = site q */
int aqc_mpsb_set_lhs_basis(aqc_mpsb* b, const uint8_t* bits) {
    if (!b || !bits) return failf("null argument");
    HIP_OK(hipSetDevice(b->device));
    const size_t bytes = (size_t)b->L * b->n;
    if (bytes > b->bits_cap) {
        HIP_OK(hipStreamSynchronize(b->st));
        if (b->bits) HIP_OK(hipFree(b->bits));
        if (b->h_bits) HIP_OK(hipHostFree(b->h_bits));
        b->bits = nullptr; b->h_bits = nullptr; b->bits_cap = 0;
        HIP_OK(hipMalloc((void**)&b->bits, bytes));
        HIP_OK(hipHostMalloc((void**)&b->h_bits, bytes, hipHostMallocDefault));
        b->bits_cap = bytes;
    }
    HIP_OK(hipStreamSynchronize(b->st));   // (the staging buffer may still feed the previous upload)
    memcpy(b->h_bits, bits, bytes);
    HIP_OK(hipMemcpyAsync(b->bits, b->h_bits, bytes, hipMemcpyHostToDevice, b->st));
    HIP_OK(launch_lanes_basis(b->lhs.dev, b->bits, b->L, b->st));
    b->lhs.max_dim_in = 1;
    b->have_lhs = true;
    return 0;
}

/* v_mul_mps / v_dagger_mul_mps (mps_operations.py:326-371) for every lane: the working state of the batch <- V(theta_l)|phi_l> (inverse = 0) or
 * V(theta_l)^H|phi_l> (inverse = 1), layer by layer (apply_circuit_all); aqc_mpsb_export hands a lane's result out.  No lhs states needed. */
int aqc_mpsb_apply_circuit(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, int inverse, double trunc_thr, int max_bond, double* discarded_out,
                           int32_t* max_bond_out) {
    if (!b || !circ || !thetas) return failf("null argument");
    if (!b->have_target) return failf("set the states of the lanes first (aqc_mpsb_set_targets)");
    int T = 0;
    if (begin(b, circ, thetas, trunc_thr, max_bond, 1, &T)) return 1;
    for (int attempt = 0;; ++attempt) {
        HIP_OK(hipMemsetAsync(b->status, 0, sizeof(int) * 2 * (size_t)b->L, b->st));
        if (clone(b, b->target, b->vh)) return 1;
        if (apply_circuit_all(b, b->vh, circ, T, inverse != 0, trunc_thr, max_bond)) return 1;
        const int rc = finish(b, attempt == 0, &b->peak_vh);
        if (rc == 2) continue;
        if (rc) return 1;
        break;
    }
    const Readback r = readback(b);
    for (int l = 0; l < b->L; ++l) {
        if (discarded_out) discarded_out[l] = r.disc[l];
        if (max_bond_out) max_bond_out[l] = r.max_dim(l, b->n);
    }
    b->vh_ready = inverse != 0;
    return 0;
}

/* lane `lane` of the working state (the result of aqc_mpsb_apply_circuit / aqc_mpsb_vh) as a single-lane MPS of its own */
int aqc_mpsb_export(aqc_mpsb* b, int lane, aqc_mps** out) {
    if (!b || !out) return failf("null argument");
    if (lane < 0 || lane >= b->L) return failf("lane out of range");
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipStreamSynchronize(b->st));
    const int n = b->n;
    std::vector<int> dims(n + 1);
    double discarded = 0.0;
    HIP_OK(hipMemcpy(dims.data(), b->vh.dev.dims + (size_t)lane * (n + 1), sizeof(int) * (n + 1), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&discarded, b->vh.dev.discarded + lane, sizeof(double), hipMemcpyDeviceToHost));
    std::vector<const void*> sites(n);
    std::vector<const double*> lams(std::max(n - 1, 1), nullptr);
    for (int q = 0; q < n; ++q) {
        sites[q] = b->vh.site(lane, q);
        if (q < n - 1) lams[q] = b->vh.lambda(lane, q);
    }
    return mps_adopt(b->device, n, dims.data(), sites.data(), lams.data(), discarded, out);
}

/* Phase 1 of an evaluation: vh_l = V(theta_l)^H|phi_l> for every lane, kept in the batch, and the amplitudes
 * amps[lane][0] = <lhs_l|vh_l>, amps[lane][1 + q] = <X_q lhs_l|vh_l> (num_amps = 1 or 1 + n: with a basis state as lhs these are the
 * amplitudes of the flip states of objective_lhs_sur_max.py:82-117).  half != 0: lanes [L/2, L) repeat the targets and thetas of lanes
 * [0, L/2) (the same problems with another lhs state): V^H runs on the first half only and is copied. */
int aqc_mpsb_vh(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int half, int num_amps, double* amps,
                double* discarded_out, int32_t* max_bond_out) {
    if (!b || !circ || !thetas || !amps) return failf("null argument");
    if (!b->have_target || !b->have_lhs) return failf("set the targets and the lhs states of the lanes first");
    if (num_amps != 1 && num_amps != 1 + b->n) return failf("num_amps: 1 (<lhs|vh>) or 1 + n (and the single-flip amplitudes)");
    if (half && (b->L & 1)) return failf("half: the batch needs an even number of lanes");
    int T = 0;
    if (begin(b, circ, thetas, trunc_thr, max_bond, num_amps, &T)) return 1;
    const int L = b->L;
    for (int attempt = 0;; ++attempt) {
        if (enqueue_vh(b, circ, T, half != 0, num_amps)) return 1;
        const int rc = finish(b, attempt == 0, &b->peak_vh);
        if (rc == 2) continue;
        if (rc) return 1;
        break;
    }
    const Readback r = readback(b);
    cd* out = reinterpret_cast<cd*>(amps);
    for (int l = 0; l < L; ++l) {
        for (int k = 0; k < num_amps; ++k) out[(size_t)l * num_amps + k] = r.vals[(size_t)l * b->nvals + k];
        if (discarded_out) discarded_out[l] = r.disc[l];
        if (max_bond_out) max_bond_out[l] = r.max_dim(l, b->n);
    }
    b->vh_ready = true;
    return 0;
}

/* Phase 2: the gate-by-gate gradient walk of every lane from its CURRENT lhs state (it may have been replaced since phase 1) and the
 * vh, thetas, trunc_thr and max_bond of the last aqc_mpsb_vh (same circuit). */
int aqc_mpsb_grad(aqc_mpsb* b, const aqc_circuit* circ, int block_from, int block_to, int front_layer, double* grad_out) {
    if (!b || !circ || !grad_out) return failf("null argument");
    if (!b->vh_ready) return failf("aqc_mpsb_grad follows aqc_mpsb_vh");
    if (!b->have_lhs) return failf("set the lhs states of the lanes first");
    if (check_circuit(circ, b->n)) return 1;
    HIP_OK(hipSetDevice(b->device));
    const int n = b->n, L = b->L, tpb = circ->entangler == AQC_CP ? 5 : 4, T = 3 * n + tpb * circ->num_blocks;
    if (T != b->cur_T || 3 * n + tpb * (int)blocks_of(circ).size() > b->nvals) return failf("aqc_mpsb_grad: not the circuit of the last aqc_mpsb_vh");
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (block_from > block_to || block_to > circ->num_blocks) return failf("invalid block range");
    std::vector<std::pair<int, cd>> rec;
    b->hint = b->peak_vh + b->peak_grad == 0 ? kLaneCap : hint_for(std::max(b->target.max_dim_in, b->lhs.max_dim_in), std::max(b->peak_vh, b->peak_grad));
    for (int attempt = 0;; ++attempt) {
        HIP_OK(hipMemsetAsync(b->status, 0, sizeof(int) * 2 * (size_t)L, b->st));
        if (clone(b, b->lhs, b->w) || clone(b, b->vh, b->z)) return 1;
        if (env_init(b)) return 1;
        if (gradient_all(b, circ, T, b->cur_trunc, b->cur_max_bond, block_from, block_to, front_layer != 0, 0, rec)) return 1;
        const int rc = finish(b, attempt == 0, &b->peak_grad);
        if (rc == 2) continue;
        if (rc) return 1;
        break;
    }
    const Readback r = readback(b);
    assemble(b, r, rec, 0, T, reinterpret_cast<cd*>(grad_out));
    return 0;
}

/* fast_dot_gradient(circ, thetas, lvec, vh_phi, trunc_thr, block_range, front_layer) (mps_dot_objective.py:41-242) for every lane, with
 * vh_phi_l ALREADY formed by the caller: the states set by aqc_mpsb_set_targets are taken as vh_phi_l, those of aqc_mpsb_set_lhs as lvec_l. */
int aqc_mpsb_gradient_of(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int block_from, int block_to,
                         int front_layer, double* grad_out) {
    if (!b || !circ || !thetas || !grad_out) return failf("null argument");
    if (!b->have_target || !b->have_lhs) return failf("set the vh_phi states (aqc_mpsb_set_targets) and the lhs states of the lanes first");
    int T = 0;
    if (begin(b, circ, thetas, trunc_thr, max_bond, 0, &T)) return 1;
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (block_from > block_to || block_to > circ->num_blocks) return failf("invalid block range");
    std::vector<std::pair<int, cd>> rec;
    for (int attempt = 0;; ++attempt) {
        HIP_OK(hipMemsetAsync(b->status, 0, sizeof(int) * 2 * (size_t)b->L, b->st));
        if (clone(b, b->lhs, b->w) || clone(b, b->target, b->z)) return 1;
        if (env_init(b)) return 1;
        if (gradient_all(b, circ, T, trunc_thr, max_bond, block_from, block_to, front_layer != 0, 0, rec)) return 1;
        const int rc = finish(b, attempt == 0, &b->peak_grad);
        if (rc == 2) continue;
        if (rc) return 1;
        break;
    }
    assemble(b, readback(b), rec, 0, T, reinterpret_cast<cd*>(grad_out));
    return 0;
}

/* both phases in one call, one wait: h[lane] = <lhs_l|vh_l> and the gradient */
int aqc_mpsb_eval(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int block_from, int block_to,
                  int front_layer, double* h_out, double* grad_out, double* discarded_out, int32_t* max_bond_out) {
    if (!b || !circ || !thetas || !h_out || !grad_out) return failf("null argument");
    if (!b->have_target || !b->have_lhs) return failf("set the targets and the lhs states of the lanes first");
    int T = 0;
    if (begin(b, circ, thetas, trunc_thr, max_bond, 1, &T)) return 1;
    const int L = b->L;
    if (block_from < 0) { block_from = 0; block_to = circ->num_blocks; }
    if (block_from > block_to || block_to > circ->num_blocks) return failf("invalid block range");
    std::vector<std::pair<int, cd>> rec;
    for (int attempt = 0;; ++attempt) {
        if (enqueue_vh(b, circ, T, false, 1)) return 1;   // leaves (w, z) = (lhs, vh) with their environments started
        if (gradient_all(b, circ, T, trunc_thr, max_bond, block_from, block_to, front_layer != 0, 1, rec)) return 1;
        const int rc = finish(b, attempt == 0, &b->peak_vh);
        if (rc == 2) continue;
        if (rc) return 1;
        break;
    }
    b->peak_grad = 0;   // (peak_vh covers the whole evaluation here)
    b->vh_ready = true;
    const Readback r = readback(b);
    cd* hh = reinterpret_cast<cd*>(h_out);
    for (int l = 0; l < L; ++l) {
        hh[l] = r.vals[(size_t)l * b->nvals];
        if (discarded_out) discarded_out[l] = r.disc[l];
        if (max_bond_out) max_bond_out[l] = r.max_dim(l, b->n);
    }
    assemble(b, r, rec, 1, T, reinterpret_cast<cd*>(grad_out));
    return 0;
}

}  // extern "C"
