// C ABI (include/aqc_hip.h): host runtime around the gfx950 kernels.
//
// A context (aqc_ctx) is the immutable gate program of one ansatz.  A workspace (aqc_ws) binds
// it to one HIP device + one stream and keeps `batch` independent evaluations resident in HBM:
//   thetas[B][T] -> coef[B][n+L+1][24]         (coef_kernel + sign_kernel, once per theta upload)
//   Y, Z, X, W, ZW, X2 : [B][2^n][pitch] complex128 (pitch = columns padded to a power of two)
//   partial[B][5*G][ntiles], grads[B][T]        (inner-product partials and their fixed-order sum)
// Nothing below ever falls back to host arithmetic: if HIP is unusable every call fails loudly.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aqc_hip.h"
#include "aqc_device.h"
#include "aqc_launch.h"
#include "aqc_plan.h"

using namespace aqc;

namespace {

thread_local std::string g_error;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int ceil_log2(int v) {
    int b = 0;
    while ((1 << b) < v) ++b;
    return b;
}

struct DevPlan {
    Plan plan;
    std::vector<DevStage> h_stages;
    std::vector<DevOp> h_ops;
    std::vector<DevSub> h_subs;   // register-blocked kernels
    std::vector<DevMop> h_mops;
    DevStage* d_stages = nullptr;
    DevOp* d_ops = nullptr;
    DevSub* d_subs = nullptr;
    DevMop* d_mops = nullptr;
    int k = 0, ntiles = 0, reg_bits = 0;
    bool v2 = false;              // run the register-blocked kernels
    // matrix-core kernels (family 3)
    bool v3 = false;
    std::vector<DevSub3> h_subs3;
    std::vector<DevGrp> h_grps;
    DevSub3* d_subs3 = nullptr;
    DevGrp* d_grps = nullptr;
    double* d_umat = nullptr;     // [batch][nsubs][12][64]
    double2* d_rpart = nullptr;   // sweep plan only: [batch][nsubs][ntiles][256]
    bool u_valid = false;         // d_umat matches the coefficients in use
    int family() const { return v3 ? 3 : (v2 ? 2 : 1); }
};

unsigned swz3_host(unsigned l) { return l ^ ((l >> 4) & 15u) ^ ((l >> 8) & 15u); }

// Micro-ops of one gate group on register bits (pc, pt); forward or conjugate-transposed order.
// with_dots: the sweep; every group ends with a MOP_REDUCE that folds its inner products.
void emit_mops(const Program& prog, int gi, int pc, int pt, bool inverse, bool with_dots, std::vector<DevMop>& out) {
    const GateGroup& g = prog.groups[gi];
    const int rec = g.coef * kCoefStride;
    const int konst = (prog.n + prog.num_blocks) * kCoefStride + kLiftOffset;  // Rz(pi/2) in lifting form
    const int neg = inverse ? MOPF_NEG_S : 0;
    std::vector<std::pair<int, int>> dots;  // (slot, producer kind), oldest first
    auto rot = [&](int kind, int p, int pair, int flags, int slot) {
        const int coef = pair < 0 ? konst : rec + kLiftOffset + 2 * pair;
        const bool has = with_dots && slot >= 0;
        out.push_back({kind, p, 0, flags, coef, has ? slot : -1, g.jblock, pair < 0 ? -1 : g.theta0 + pair});
        if (has) dots.push_back({slot, kind});
    };
    auto reduce = [&]() {
        if (dots.empty()) return;
        DevMop m = {MOP_REDUCE, -1, -1, 0, -1, -1, g.jblock, -2};
        int kinds = 0;
        for (size_t j = 0; j < dots.size(); ++j) {  // newest first
            const auto& d = dots[dots.size() - 1 - j];
            kinds |= d.second << (4 * j);
            (j == 0 ? m.slot : j == 1 ? m.p : j == 2 ? m.p2 : m.coef) = d.first;
        }
        m.flags = kinds;
        out.push_back(m);
        dots.clear();
    };
    const int slot0 = gi * kSlotsPerGroup;
    if (g.type == GROUP_FRONT) {
        if (!inverse) {  // Rz(t2), Ry(t1), Rz(t0), rightmost first (core_operations.py:671-677,921-935)
            rot(MOP_RZ, pc, 2, 0, slot0 + 0);
            rot(MOP_RY, pc, 1, 0, slot0 + 1);
            rot(MOP_RZ, pc, 0, 0, slot0 + 2);
            reduce();
        } else {         // (Rz Ry Rz)^H (core_operations.py:812-818)
            rot(MOP_RZ, pc, 0, neg, -1);
            rot(MOP_RY, pc, 1, neg, -1);
            rot(MOP_RZ, pc, 2, neg, -1);
        }
        return;
    }
    const int ekind = prog.entangler == 0 ? MOP_CX : (prog.entangler == 1 ? MOP_CZ : MOP_CP);
    const int rs = prog.entangler == 0 ? MOP_RX : MOP_RZ;
    if (!inverse) {  // core_operations.py:956-1017
        if (g.flags & FLAG_PRE_RZ) rot(MOP_RZ, pc, -1, MOPF_NEG_S, -1);   // Rz(-pi/2) on control
        const bool cpdot = with_dots && prog.entangler == 2;
        out.push_back({ekind, pc, pt, 0, rec + 8, cpdot ? slot0 + 4 : -1, g.jblock, prog.entangler == 2 ? g.theta0 + 4 : -2});
        if (cpdot) { dots.push_back({slot0 + 4, MOP_CP}); reduce(); }
        rot(MOP_RY, pc, 0, 0, slot0 + 0);
        rot(MOP_RZ, pc, 1, 0, slot0 + 1);
        rot(MOP_RY, pt, 2, 0, slot0 + 2);
        rot(rs, pt, 3, 0, slot0 + 3);
        reduce();
        if (g.flags & FLAG_POST_RZ) rot(MOP_RZ, pt, -1, 0, -1);           // Rz(+pi/2) on target
    } else {         // core_operations.py:787-809
        if (g.flags & FLAG_POST_RZ) rot(MOP_RZ, pt, -1, MOPF_NEG_S, -1);
        rot(rs, pt, 3, neg, -1);
        rot(MOP_RY, pt, 2, neg, -1);
        rot(MOP_RZ, pc, 1, neg, -1);
        rot(MOP_RY, pc, 0, neg, -1);
        out.push_back({ekind, pc, pt, neg, rec + 8, -1, g.jblock, prog.entangler == 2 ? g.theta0 + 4 : -2});
        if (g.flags & FLAG_PRE_RZ) rot(MOP_RZ, pc, -1, 0, -1);
    }
}

void lower_plan(const Program& prog, const Plan& plan, DevPlan& out, int reg_bits, bool with_dots, bool mfma = false) {
    out.plan = plan;
    out.h_stages.clear();
    out.h_ops.clear();
    out.h_subs.clear();
    out.h_mops.clear();
    out.h_subs3.clear();
    out.h_grps.clear();
    out.reg_bits = reg_bits;
    out.v3 = mfma && reg_bits == 4 && (int)plan.stages.front().bits.size() >= 8;
    out.v2 = !out.v3 && reg_bits > 0 && (int)plan.stages.front().bits.size() >= reg_bits;
    if (out.v3) {
        split_substages(prog, out.plan, 4, 1 << 30);   // a sub-stage is one 16 x 16 unitary: any number of groups
    } else if (out.v2) {
        int max_ops = reg_bits == 4 ? kMaxOpsPerSub : kMaxOpsPerSub / 2;
        if (prog.entangler == 2) max_ops /= 2;   // CP: two reductions per block
        split_substages(prog, out.plan, reg_bits, max_ops);
    }
    out.k = (int)plan.stages.front().bits.size();
    out.ntiles = 1 << (plan.nbits - out.k);
    for (const Stage& st : out.plan.stages) {
        DevStage ds;
        memset(&ds, 0, sizeof ds);
        ds.k = (int)st.bits.size();
        ds.nops = (int)st.ops.size();
        ds.op_begin = (int)out.h_ops.size();
        ds.ntiles = 1 << (plan.nbits - ds.k);
        std::vector<int> local_of(plan.nbits, -1);
        for (int j = 0; j < ds.k; ++j) local_of[st.bits[j]] = j;
        for (int b = 0; b < plan.nbits; ++b)
            if (local_of[b] < 0) ds.ubits[ds.nub++] = b;
        for (unsigned i = 0; i < 64; ++i) {
            uint32_t off = 0;
            for (int j = 0; j < 6 && j < ds.k; ++j)
                if (i >> j & 1) off |= 1u << st.bits[j];
            ds.dlo[i] = off;
        }
        for (unsigned i = 0; i < 256; ++i) {
            uint32_t off = 0;
            for (int j = 0; j < 8 && 6 + j < ds.k; ++j)
                if (i >> j & 1) off |= 1u << st.bits[6 + j];
            ds.dhi[i] = off;
        }
        for (int gi : st.ops) {
            const GateGroup& g = prog.groups[gi];
            DevOp op;
            op.type = g.type;
            op.p0 = local_of[plan.col_bits + g.q0];
            op.p1 = g.q1 >= 0 ? local_of[plan.col_bits + g.q1] : 0;
            op.flags = g.flags;
            op.coef = g.coef;
            op.slot = gi * kSlotsPerGroup;
            op.jblock = g.jblock;
            op.pad = 0;
            out.h_ops.push_back(op);
        }
        ds.sub_begin = (int)out.h_subs.size();
        ds.nsubs = 0;
        if (out.v2 || out.v3) {
            for (const SubStage& sub : st.subs) {
                DevSub dsub;
                memset(&dsub, 0, sizeof dsub);
                dsub.nbits = (int)sub.bits.size();
                std::vector<int> reg_of(ds.k, -1);
                for (int j = 0; j < dsub.nbits; ++j) { dsub.bits[j] = sub.bits[j]; reg_of[sub.bits[j]] = j; }
                dsub.mop_begin = (int)out.h_mops.size();
                const int grp_begin = (int)out.h_grps.size();
                for (int gi : sub.ops) {
                    const GateGroup& g = prog.groups[gi];
                    const int pc = reg_of[local_of[plan.col_bits + g.q0]];
                    const int pt = g.q1 >= 0 ? reg_of[local_of[plan.col_bits + g.q1]] : 0;
                    emit_mops(prog, gi, pc, pt, plan.inverse, with_dots, out.h_mops);
                    if (out.v3)   // front groups get a dummy second bit so that one code path serves both types
                        out.h_grps.push_back({g.type, pc, g.q1 >= 0 ? pt : (pc == 0 ? 1 : 0), g.flags, g.theta0,
                                              with_dots ? gi * kSlotsPerGroup : -1, g.jblock, 0});
                }
                dsub.nmops = (int)out.h_mops.size() - dsub.mop_begin;
                out.h_subs.push_back(dsub);
                ++ds.nsubs;
                if (out.v3) {   // slot tables: amplitude bits = register bits, chunk bits = the other local bits, ascending
                    DevSub3 d3;
                    memset(&d3, 0, sizeof d3);
                    d3.mop_begin = dsub.mop_begin;
                    d3.nmops = dsub.nmops;
                    d3.grp_begin = grp_begin;
                    d3.ngrp = (int)out.h_grps.size() - grp_begin;
                    std::vector<int> cbits;
                    for (int j = 0; j < ds.k; ++j)
                        if (reg_of[j] < 0) cbits.push_back(j);
                    for (int j = 0; j < 4; ++j) d3.bits[j] = dsub.bits[j];
                    auto deposit = [](unsigned v, const int* bits, int nb) {
                        unsigned o = 0;
                        for (int j = 0; j < nb; ++j)
                            if (v >> j & 1) o |= 1u << bits[j];
                        return o;
                    };
                    const int nc = (int)cbits.size();
                    for (unsigned v = 0; v < 16; ++v) {
                        d3.dep_a[v] = (uint16_t)swz3_host(deposit(v, dsub.bits, 4));
                        d3.dep_clo[v] = (uint16_t)swz3_host(deposit(v, cbits.data(), std::min(nc, 4)));
                    }
                    for (unsigned g = 0; g < 64; ++g)
                        d3.dep_chi[g] = nc > 4 && g < (1u << (nc - 4)) ? (uint16_t)swz3_host(deposit(g, cbits.data() + 4, nc - 4)) : 0;
                    for (unsigned l = 0; l < 64; ++l)
                        d3.lane12[l] = (uint32_t)(d3.dep_clo[l & 15] ^ d3.dep_a[l >> 4]) | ((uint32_t)(d3.dep_a[l & 15] ^ d3.dep_clo[l >> 4]) << 16);
                    for (unsigned g = 0; g < 16; ++g)
                        for (unsigned s = 0; s < 4; ++s) {
                            d3.kk[g][s] = (uint32_t)(d3.dep_a[4 * s] ^ d3.dep_chi[g]) << 4;
                            d3.kk[g][4 + s] = (uint32_t)(d3.dep_clo[4 * s] ^ d3.dep_chi[g]) << 4;
                        }
                    out.h_subs3.push_back(d3);
                }
            }
        }
        out.h_stages.push_back(ds);
    }
}

}  // namespace

namespace aqc { int set_error(const std::string& msg) { g_error = msg; return 1; } }  // for the other translation units

struct aqc_ctx {
    Program prog;
    std::mutex mu;
    // lowered plans (host side: stages, sub-stages, micro-ops) by (which, col_bits, tile bits, low bits, family): workspaces
    // of the same shape -- one per batch of jobs in the drivers -- share the planning work (the sub-stage search of a
    // deep Trotter ansatz takes a few tenths of a second)
    std::map<std::vector<int>, DevPlan> plan_cache;
    std::map<int, aqc_ws*> oneshot;  // ncols -> batch-1 workspace used by the host-pointer entry points
};

struct aqc_ws {
    aqc_ctx* ctx = nullptr;
    int device = 0, batch = 1, ncols = 1, pitch = 1, col_bits = 0, nbits = 0, threads = 256;
    size_t lane_elems = 0;  // 2^nbits
    hipStream_t stream = nullptr;
    DevPlan fwd, inv, sweep;
    double* d_thetas = nullptr;       // parameters in use (own buffer or a slice of the bank)
    double* d_thetas_own = nullptr;
    double* d_theta_bank = nullptr;
    int bank_sets = 0, gather_count = 0;
    double* d_coef = nullptr;
    double2* bufs[AQC_NUM_BUFS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* h_pin = nullptr;           // pinned staging: thetas | grads | gathered
    size_t pin_thetas = 0, pin_grads = 0, pin_small = 0;
    double2* d_partial = nullptr;
    double2* d_grads = nullptr;
    double* mirror_grads = nullptr;    // set by aqc_ws_eval around its launches: pinned host copies written by the kernels
    double* mirror_small = nullptr;
    double2* d_small = nullptr;  // gather / vdot results
    double2* d_vdot_part = nullptr;
    double2* d_vdot_out = nullptr;
    long long* d_index = nullptr;
    long long* d_tmp_index = nullptr;   // one-shot gather / vdot: never disturb the persistent gather set-up
    double2* d_tmp_small = nullptr;
    size_t tmp_index_cap = 0, tmp_small_cap = 0;
    long long* d_basis_index = nullptr;   // [batch], set_basis only (keeps the gather set-up intact)
    long long* d_combo_prev[AQC_NUM_BUFS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // set_combo: positions written last time
    bool combo_valid[AQC_NUM_BUFS] = {false, false, false, false, false, false};   // buffer holds exactly that sparse pattern
    // aqc_ws_surrogate_eval: device block [f B | fidelity B | weight B | hs 2 B S | max_no B ints], its pinned mirror
    void* d_sur = nullptr;
    void* h_sur = nullptr;
    int sur_states = 0;
    double* d_sur_real = nullptr;     // real parts of the gradient when only those are asked for
    size_t sur_real_cap = 0;
    long long* d_combo_index = nullptr;   // [batch][2] staging of set_combo
    double2* d_combo_coef = nullptr;      // [batch][2]
    size_t small_cap = 0, index_cap = 0;
    int* d_theta_slots = nullptr;
    int* d_slot_theta = nullptr;       // slot -> theta when every theta has exactly one slot (grads_direct), see rgrad_kernel
    bool grads_direct = false;
    int* d_slot_ntiles = nullptr;
    int nslots = 0, vdot_parts = 0;
    bool coef_valid = false;
    bool need_coef = false;           // something besides the stage kernels reads d_coef (coordinate descent)
    // aqc_ws_eval as a HIP graph: the whole chain (thetas H2D, U builder, V^H stages, gather, sweep stages, gradient walk,
    // D2H copies) captured once per call signature and replayed -- one launch instead of ~11 host calls per evaluation
    std::map<std::vector<long long>, hipGraphExec_t> graphs;
    bool capturing = false;
    UJob* d_ujobs = nullptr;          // family 3: [V^H subs | sweep subs | V subs]
    struct MpsSlot {
        std::vector<int> dims;          // n + 1 bond dimensions
        std::vector<size_t> offset;     // element offset of site q inside d_t
        double2* d_t = nullptr;         // [q][2][dims[q]][dims[q+1]], lambda folded in
        size_t cap = 0;                 // capacity of d_t (grow-only: re-uploads of the same shape allocate nothing)
    } mps[AQC_MPS_SLOTS];
    double* d_mps_lam = nullptr;        // staging of the packed Schmidt vectors (grow-only)
    size_t mps_lam_cap = 0;
    double2* d_mps_scratch = nullptr;
    size_t mps_scratch_cap = 0;
    // device pointer tables of the batched MPS -> dense contraction: a few resident sets, found again by their contents (an
    // optimisation converts the same operands into the same lanes evaluation after evaluation: no upload, no synchronisation)
    struct MpsTabs { std::vector<const void*> host; const void** dev = nullptr; size_t cap = 0; unsigned long long tick = 0; };
    // coordinate descent as one persistent launch: the walk's step list, thetas [batch][T] and objective values on the device
    void* d_cd_prog = nullptr;
    int cd_nsteps = 0;
    double* d_cd_thetas = nullptr;
    double* d_cd_fobj = nullptr;
    size_t cd_fobj_cap = 0;
    MpsTabs mps_tabs[32];   // resident pointer-table sets (one per distinct chain: operands x lanes x bond dimensions)
    unsigned long long mps_tabs_tick = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, pev0 = nullptr, pev1 = nullptr;
    hipStream_t copy_stream = nullptr;        // aqc_ws_results_async: result copies run beside the next evaluation's kernels
    hipEvent_t ev_ready = nullptr, ev_copied = nullptr;
    hipStream_t mps_stream = nullptr;         // batched MPS -> dense: the right half's chain runs beside the left half's
    hipEvent_t ev_mps_fork = nullptr, ev_mps_join = nullptr;
    bool copy_pending = false;                // the producers of the next evaluation wait for ev_copied before they overwrite the results
    const double* theta_host = nullptr;      // aqc_ws_eval: pinned thetas the next U build reads directly (and copies to d_thetas)
    bool gather_rides = false;               // aqc_ws_eval: the next gradient walk also performs the registered gather (see there)
    bool profile = false;
    int64_t prof_count[AQC_NUM_KINDS] = {0, 0, 0, 0, 0};
    double prof_ms[AQC_NUM_KINDS] = {0, 0, 0, 0, 0};
};

namespace {

int upload_plan(DevPlan& p) {
    HIP_OK(hipMalloc((void**)&p.d_stages, p.h_stages.size() * sizeof(DevStage)));
    HIP_OK(hipMemcpy(p.d_stages, p.h_stages.data(), p.h_stages.size() * sizeof(DevStage), hipMemcpyHostToDevice));
    const size_t nops = std::max<size_t>(p.h_ops.size(), 1);
    HIP_OK(hipMalloc((void**)&p.d_ops, nops * sizeof(DevOp)));
    HIP_OK(hipMemset(p.d_ops, 0, nops * sizeof(DevOp)));
    if (!p.h_ops.empty())
        HIP_OK(hipMemcpy(p.d_ops, p.h_ops.data(), p.h_ops.size() * sizeof(DevOp), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&p.d_subs, std::max<size_t>(p.h_subs.size(), 1) * sizeof(DevSub)));
    HIP_OK(hipMalloc((void**)&p.d_mops, std::max<size_t>(p.h_mops.size(), 1) * sizeof(DevMop)));
    if (!p.h_subs.empty())
        HIP_OK(hipMemcpy(p.d_subs, p.h_subs.data(), p.h_subs.size() * sizeof(DevSub), hipMemcpyHostToDevice));
    if (!p.h_mops.empty())
        HIP_OK(hipMemcpy(p.d_mops, p.h_mops.data(), p.h_mops.size() * sizeof(DevMop), hipMemcpyHostToDevice));
    if (p.v3 && !p.h_subs3.empty()) {
        HIP_OK(hipMalloc((void**)&p.d_subs3, p.h_subs3.size() * sizeof(DevSub3)));
        HIP_OK(hipMemcpy(p.d_subs3, p.h_subs3.data(), p.h_subs3.size() * sizeof(DevSub3), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc((void**)&p.d_grps, std::max<size_t>(p.h_grps.size(), 1) * sizeof(DevGrp)));
        if (!p.h_grps.empty()) HIP_OK(hipMemcpy(p.d_grps, p.h_grps.data(), p.h_grps.size() * sizeof(DevGrp), hipMemcpyHostToDevice));
    }
    return 0;
}

struct ProfScope {  // brackets one launch with events when profiling is on
    aqc_ws* ws;
    int kind;
    ProfScope(aqc_ws* w, int k) : ws(w), kind(k) {
        if (ws->profile) (void)hipEventRecord(ws->pev0, ws->stream);
    }
    ~ProfScope() {
        if (!ws->profile) return;
        float ms = 0.f;
        if (hipEventRecord(ws->pev1, ws->stream) == hipSuccess && hipEventSynchronize(ws->pev1) == hipSuccess &&
            hipEventElapsedTime(&ms, ws->pev0, ws->pev1) == hipSuccess) {
            ws->prof_count[kind] += 1;
            ws->prof_ms[kind] += ms;
        }
    }
};

int check_buf(const aqc_ws* ws, int buf) {
    if (!ws) return fail("null workspace");
    if (buf < 0 || buf >= AQC_NUM_BUFS) return fail("invalid buffer id %d", buf);
    return 0;
}

int ensure_small(aqc_ws* ws, size_t n_cplx) {
    if (n_cplx <= ws->small_cap) return 0;
    if (ws->d_small) HIP_OK(hipFree(ws->d_small));
    ws->d_small = nullptr;
    HIP_OK(hipMalloc((void**)&ws->d_small, n_cplx * sizeof(double2)));
    ws->small_cap = n_cplx;
    return 0;
}

int ensure_tmp(aqc_ws* ws, size_t n_index, size_t n_cplx) {
    if (n_index > ws->tmp_index_cap) {
        if (ws->d_tmp_index) HIP_OK(hipFree(ws->d_tmp_index));
        ws->d_tmp_index = nullptr;
        HIP_OK(hipMalloc((void**)&ws->d_tmp_index, n_index * sizeof(long long)));
        ws->tmp_index_cap = n_index;
    }
    if (n_cplx > ws->tmp_small_cap) {
        if (ws->d_tmp_small) HIP_OK(hipFree(ws->d_tmp_small));
        ws->d_tmp_small = nullptr;
        HIP_OK(hipMalloc((void**)&ws->d_tmp_small, n_cplx * sizeof(double2)));
        ws->tmp_small_cap = n_cplx;
    }
    return 0;
}

int ensure_index(aqc_ws* ws, size_t n) {
    if (n <= ws->index_cap) return 0;
    if (ws->d_index) HIP_OK(hipFree(ws->d_index));
    ws->d_index = nullptr;
    HIP_OK(hipMalloc((void**)&ws->d_index, n * sizeof(long long)));
    ws->index_cap = n;
    return 0;
}

int ensure_coef(aqc_ws* ws) {
    if (ws->coef_valid) return 0;
    return fail("thetas have not been uploaded (aqc_ws_set_thetas)");
}

// host <-> device copy of [rows][ncols] <-> [rows][pitch]
int copy_in(aqc_ws* ws, double2* dst, const double* src, size_t rows) {
    if (ws->pitch == ws->ncols) {
        HIP_OK(hipMemcpyAsync(dst, src, rows * ws->ncols * sizeof(double2), hipMemcpyHostToDevice, ws->stream));
    } else {
        HIP_OK(hipMemsetAsync(dst, 0, rows * ws->pitch * sizeof(double2), ws->stream));
        HIP_OK(hipMemcpy2DAsync(dst, (size_t)ws->pitch * sizeof(double2), src, (size_t)ws->ncols * sizeof(double2),
                                (size_t)ws->ncols * sizeof(double2), rows, hipMemcpyHostToDevice, ws->stream));
    }
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}
int copy_out(aqc_ws* ws, double* dst, const double2* src, size_t rows) {
    if (ws->pitch == ws->ncols) {
        HIP_OK(hipMemcpyAsync(dst, src, rows * ws->ncols * sizeof(double2), hipMemcpyDeviceToHost, ws->stream));
    } else {
        HIP_OK(hipMemcpy2DAsync(dst, (size_t)ws->ncols * sizeof(double2), src, (size_t)ws->pitch * sizeof(double2),
                                (size_t)ws->ncols * sizeof(double2), rows, hipMemcpyDeviceToHost, ws->stream));
    }
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

// family 3: the 16 x 16 unitaries of the plan's sub-stages for the coefficients in use
// Jobs are laid out [V^H | sweep | V]: the objective+gradient path (V^H then the sweep) is built by one launch.
int ensure_umat(aqc_ws* ws, DevPlan& p) {
    if (!p.v3 || p.u_valid) return 0;
    const int T = ws->ctx->prog.num_thetas();
    const int ninv = ws->inv.v3 ? (int)ws->inv.h_subs3.size() : 0, nsw = ws->sweep.v3 ? (int)ws->sweep.h_subs3.size() : 0;
    const int nfwd = ws->fwd.v3 ? (int)ws->fwd.h_subs3.size() : 0;
    ProfScope ps(ws, AQC_K_COEF);
    if (&p == &ws->fwd) {
        HIP_OK(launch_ubuild(ws->d_ujobs + ninv + nsw, nfwd, ws->d_thetas, T, ws->batch, ws->stream));
        p.u_valid = true;
    } else {
        // aqc_ws_eval (small batches): the thetas are read from its pinned staging buffer and land in HBM through this kernel
        HIP_OK(launch_ubuild(ws->d_ujobs, ninv + nsw, ws->theta_host ? ws->theta_host : ws->d_thetas, T, ws->batch, ws->stream,
                             ws->theta_host ? ws->d_thetas : nullptr));
        ws->theta_host = nullptr;
        ws->inv.u_valid = ws->sweep.u_valid = true;
    }
    return 0;
}

int run_coef(aqc_ws* ws) {
    const Program& prog = ws->ctx->prog;
    ws->fwd.u_valid = ws->inv.u_valid = ws->sweep.u_valid = false;
    ws->coef_valid = true;
    if (ws->fwd.v3 && ws->inv.v3 && ws->sweep.v3 && !ws->need_coef) return 0;   // the matrix-core path reads the thetas directly
    ProfScope ps(ws, AQC_K_COEF);
    HIP_OK(launch_coef(ws->d_thetas, ws->d_coef, prog.n, prog.num_blocks, prog.tpb, prog.tail_blocks, ws->batch, ws->stream));
    return 0;
}

int run_apply(aqc_ws* ws, bool inverse, int src_buf, int dst_buf) {
    DevPlan& p = inverse ? ws->inv : ws->fwd;
    const Program& prog = ws->ctx->prog;
    if (p.v3) {
        if (ensure_umat(ws, p)) return 1;
        for (size_t s = 0; s < p.h_stages.size(); ++s) {
            Stage3Args a;
            memset(&a, 0, sizeof a);
            a.stage = p.h_stages[s];
            a.subs = p.d_subs3;
            a.umat = p.d_umat;
            a.nsubs_total = (int)p.h_subs3.size();
            a.in0 = s == 0 ? ws->bufs[src_buf] : ws->bufs[dst_buf];
            a.out0 = ws->bufs[dst_buf];
            a.lane_stride = ws->lane_elems;
            a.ntiles = p.ntiles;
            a.batch = ws->batch;
#ifdef AQC_TUNING   // AQC_STAMPS=1: mean cycles per phase of the V / V^H workgroups of this launch, on stderr
            static unsigned long long* d_stamps_a = nullptr;
            const size_t nwg = (size_t)p.ntiles * ws->batch;
            if (env_int("AQC_STAMPS", 0) != 0 && nwg <= 65536) {
                if (!d_stamps_a) HIP_OK(hipMalloc((void**)&d_stamps_a, sizeof(unsigned long long) * 65536 * kStampSlots));
                HIP_OK(hipMemsetAsync(d_stamps_a, 0, sizeof(unsigned long long) * nwg * kStampSlots, ws->stream));
                a.stamps = d_stamps_a;
            }
#endif
            {
                ProfScope ps(ws, AQC_K_APPLY);
                HIP_OK(launch_apply3(p.ntiles, ws->batch, p.k, ws->stream, a));
            }
#ifdef AQC_TUNING
            if (a.stamps) {
                std::vector<unsigned long long> h(nwg * kStampSlots);
                HIP_OK(hipStreamSynchronize(ws->stream));
                HIP_OK(hipMemcpy(h.data(), d_stamps_a, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
                double load = 0, loop = 0, store = 0, bar = 0;
                const int ns = p.h_stages[s].nsubs;
                for (size_t w = 0; w < nwg; ++w) {
                    const unsigned long long* t = h.data() + w * kStampSlots;
                    load += (double)(t[1] - t[0]); loop += (double)(t[2] - t[1]); store += (double)(t[3] - t[2]);
                    for (int i = 0; i < ns && 5 + i < kStampSlots; ++i) bar += (double)(t[5 + i] - t[4 + i]);
                }
                fprintf(stderr, "aqc_hip stamps: V/V^H stage %zu (%d sub-stages, %zu workgroups): load %.0f + sub-stage loop %.0f (%.0f per sub-stage, of which "
                        "waiting at its barrier %.0f) + store %.0f cycles per workgroup\n", s, ns, nwg, load / nwg, loop / nwg, loop / nwg / std::max(ns, 1),
                        bar / nwg / std::max(ns, 1), store / nwg);
            }
#endif
        }
        return 0;
    }
    for (size_t s = 0; s < p.h_stages.size(); ++s) {
        StageArgs a;
        memset(&a, 0, sizeof a);
        a.stage = p.d_stages + s;
        a.ops = p.d_ops;
        a.subs = p.d_subs;
        a.mops = p.d_mops;
        a.coef = ws->d_coef;
        a.ncoef = prog.n + prog.num_blocks + 1;
        a.in0 = s == 0 ? ws->bufs[src_buf] : ws->bufs[dst_buf];
        a.out0 = ws->bufs[dst_buf];
        a.lane_stride = ws->lane_elems;
        a.final_stage = (s + 1 == p.h_stages.size()) ? 1 : 0;
        ProfScope ps(ws, AQC_K_APPLY);
        if (p.v2) HIP_OK(launch_apply2(prog.entangler, p.ntiles, ws->batch, p.k, ws->stream, a));
        else HIP_OK(launch_apply(prog.entangler, inverse, p.ntiles, ws->batch, ws->threads, p.k, ws->stream, a));
    }
    return 0;
}

}  // namespace

extern "C" {

const char* aqc_version(void) { return "aqc_hip 0.1.0 (gfx950)"; }
const char* aqc_last_error(void) { return g_error.c_str(); }

int aqc_device_count(void) {
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess ? ndev : 0;
}

int aqc_create(int num_qubits, int entangler, const int32_t* blocks, int num_blocks, int trotter, int second_order,
               aqc_ctx** out) {
    if (!out) return fail("out pointer is null");
    *out = nullptr;
    aqc_ctx* ctx = new aqc_ctx();
    const std::string err = build_program(num_qubits, entangler, blocks, num_blocks, trotter != 0, second_order != 0, ctx->prog);
    if (!err.empty()) {
        delete ctx;
        return fail("%s", err.c_str());
    }
    *out = ctx;
    return 0;
}

int aqc_destroy(aqc_ctx* ctx) {
    if (!ctx) return 0;
    for (auto& kv : ctx->oneshot) aqc_ws_destroy(kv.second);
    delete ctx;
    return 0;
}

int aqc_num_thetas(const aqc_ctx* ctx) { return ctx ? ctx->prog.num_thetas() : -1; }
int aqc_num_gate_groups(const aqc_ctx* ctx) { return ctx ? (int)ctx->prog.groups.size() : -1; }

int aqc_plan_query(aqc_ctx* ctx, int ncols, int which, int tile_bits, int low_bits, int stage, int* num_stages,
                   int* bits_out, int* num_bits, int* ops_out, int* num_ops) {
    if (!ctx) return fail("null context");
    if (ncols < 1) return fail("ncols must be positive");
    const int col_bits = ceil_log2(ncols);
    if (tile_bits <= 0) tile_bits = which == 1 ? 12 : 13;
    if (low_bits < 0) low_bits = 3;
    const Plan plan = make_plan(ctx->prog, col_bits, tile_bits, low_bits, which == 0);
    const std::string err = check_plan(ctx->prog, plan);
    if (!err.empty()) return fail("planner produced an invalid plan: %s", err.c_str());
    if (num_stages) *num_stages = (int)plan.stages.size();
    if (stage >= 0) {
        if (stage >= (int)plan.stages.size()) return fail("stage index out of range");
        const Stage& st = plan.stages[stage];
        if (num_bits) *num_bits = (int)st.bits.size();
        if (num_ops) *num_ops = (int)st.ops.size();
        if (bits_out) for (size_t i = 0; i < st.bits.size(); ++i) bits_out[i] = st.bits[i];
        if (ops_out) for (size_t i = 0; i < st.ops.size(); ++i) ops_out[i] = st.ops[i];
    }
    return 0;
}

// host-only: register bits (local positions inside the stage's tile) and number of gate groups of every sub-stage of stage
// `stage` as the matrix-core kernels run it (4 register bits, unlimited groups); subs_out receives [num_subs][5] ints
int aqc_plan_substages(aqc_ctx* ctx, int ncols, int which, int tile_bits, int low_bits, int stage, int* num_subs, int* subs_out, int max_subs) {
    if (!ctx || !num_subs) return fail("null argument");
    if (ncols < 1) return fail("ncols must be positive");
    const int col_bits = ceil_log2(ncols);
    if (tile_bits <= 0) tile_bits = 12;
    if (low_bits < 0) low_bits = 3;
    Plan plan = make_plan(ctx->prog, col_bits, tile_bits, low_bits, which == 0);
    split_substages(ctx->prog, plan, 4, 1 << 20);
    const std::string err = check_plan(ctx->prog, plan);
    if (!err.empty()) return fail("planner produced an invalid plan: %s", err.c_str());
    if (stage < 0 || stage >= (int)plan.stages.size()) return fail("stage index out of range");
    const Stage& st = plan.stages[stage];
    *num_subs = (int)st.subs.size();
    if (subs_out)
        for (int i = 0; i < (int)st.subs.size() && i < max_subs; ++i) {
            for (int j = 0; j < 4; ++j) subs_out[5 * i + j] = j < (int)st.subs[i].bits.size() ? st.subs[i].bits[j] : -1;
            subs_out[5 * i + 4] = (int)st.subs[i].ops.size();
        }
    return 0;
}

int aqc_ws_create(aqc_ctx* ctx, int device, int batch, int ncols, int tile_bits_apply, int tile_bits_sweep, aqc_ws** out) {
    if (!out) return fail("out pointer is null");
    *out = nullptr;
    if (!ctx) return fail("null context");
    if (batch < 1 || batch > 65535) return fail("batch must be in [1, 65535] (lanes are the y dimension of the launch grid)");
    if (ncols < 1) return fail("ncols must be >= 1");
    const Program& prog = ctx->prog;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail("no HIP device available (%s): the aqc_hip path needs an AMD GPU and has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
    HIP_OK(hipSetDevice(device));
    HIP_OK(init_kernels());
    HIP_OK(init_kernels2());
    HIP_OK(init_kernels3());

    aqc_ws* ws = new aqc_ws();
    ws->ctx = ctx;
    ws->device = device;
    ws->batch = batch;
    ws->ncols = ncols;
    ws->col_bits = ceil_log2(ncols);
    ws->pitch = 1 << ws->col_bits;
    ws->nbits = ws->col_bits + prog.n;
    if (ws->nbits > kMaxBits) { delete ws; return fail("2^%d elements per lane is beyond this build's limit", ws->nbits); }
    ws->lane_elems = (size_t)1 << ws->nbits;
    // Kernel family and tile size.  Throughput regime (enough tiles x lanes to give every CU >= 1 workgroup of
    // the largest tile): register-blocked kernels on 2^12 / 2^13 tiles.  Latency regime (few lanes): the
    // per-gate-group kernels on small tiles, where all threads of a workgroup share every gate group and the
    // serial chain per launch is short.  AQC_KERNEL_V2 = 0 / 1 forces a family, AQC_TILE_BITS_* a tile size.
    const int low_bits = env_int("AQC_LOW_BITS", 3);
    int ka = tile_bits_apply > 0 ? tile_bits_apply : env_int("AQC_TILE_BITS_APPLY", 0);
    int ks = tile_bits_sweep > 0 ? tile_bits_sweep : env_int("AQC_TILE_BITS_SWEEP", 0);
    // Kernel family.  Default: the matrix-core kernels (family 3) whenever the lane has at least 2^8 elements (16 chunks
    // x 16 register-bit values per MFMA group); they won every measured workload, throughput and single-evaluation
    // latency alike (profiles/r02_family_comparison.txt).  Smaller registers run the per-group kernels.
    // AQC_KERNEL_FAMILY = 1 (per-group) | 2 (register-blocked VALU) | 3 forces a family (tests run every family);
    // AQC_KERNEL_V2 = 0 / 1 is the older spelling of 1 / 2.
    const int family = env_int("AQC_KERNEL_FAMILY", env_int("AQC_KERNEL_V2", -1) >= 0 ? env_int("AQC_KERNEL_V2", -1) + 1 : 0);
    if (family < 0 || family > 3) { delete ws; return fail("AQC_KERNEL_FAMILY must be 1 (per-group), 2 (register-blocked) or 3 (matrix cores)"); }
    const bool want_v3 = (family == 3 || family == 0) && ws->nbits >= 8;
    const int force_v2 = family == 3 ? 0 : (family ? family - 1 : -1);
    // Measured on MI355X (tools/tune.py mid / b1k): the register-blocked kernels pay off once 2^12-amplitude tiles x
    // lanes give every CU two workgroups (>= 512); below that the per-group kernels win, best with ~512 workgroups but
    // never with tiles under 2^10 (every extra stage is an extra launch and an extra HBM round trip).
    const size_t big_tiles = (size_t)batch << std::max(0, ws->nbits - 12);
    const bool want_v2 = force_v2 >= 0 ? force_v2 != 0 : big_tiles >= 512;
    auto pick = [&](int kmax) {
        if (want_v3) {   // 2^12 tiles when they fill the chip (>= 256 workgroups), else down to 2^10: with few lanes the
                         // latency of an evaluation is the serial chain inside one workgroup (tools/lat_probe.py)
            int k = std::min(12, ws->nbits);
            while (k > 10 && ((size_t)batch << (ws->nbits - k)) < 256) --k;
            // matrices: column bits are pure batch bits, so a tile may shrink to "all qubit bits + 3 column bits" without
            // adding a stage; with few lanes that spreads one lane over several CUs (config 1 at 64 lanes: 2^8 tiles, +15 %)
            const int kmin = std::max(8, prog.n + std::min(ws->col_bits, 3));
            while (ws->col_bits > 0 && k > kmin && ((size_t)batch << (ws->nbits - k)) < 256) --k;
            return k;
        }
        if (want_v2) return std::min(kmax, ws->nbits);
        int k = std::min(11, ws->nbits);
        while (k > 10 && ((size_t)batch << (ws->nbits - k)) < 512) --k;
        return k;
    };
    if (ka <= 0) ka = pick(13);
    if (ks <= 0) ks = pick(12);
    ka = std::min(std::min(ka, 13), ws->nbits);
    ks = std::min(std::min(ks, 12), ws->nbits);
    ws->threads = env_int("AQC_THREADS", 0);
    if (ws->threads <= 0) ws->threads = std::min(256, std::max(64, 1 << (std::min(ka, ks) - 2)));
    if (ws->threads < 64 || ws->threads > 512 || ws->threads % 64) { delete ws; return fail("AQC_THREADS must be a multiple of 64 in [64, 512]"); }

    // fewest launches wins; among equals prefer the longer contiguous HBM runs (more forced low bits)
    auto best_plan = [&](int k, bool inverse) {
        Plan best = make_plan(prog, ws->col_bits, k, low_bits, inverse);
        for (int lb = low_bits - 1; lb >= 2; --lb) {
            Plan cand = make_plan(prog, ws->col_bits, k, lb, inverse);
            if (cand.stages.size() < best.stages.size()) best = cand;
        }
        return best;
    };
    if (want_v3) { ka = std::min(std::max(ka, 8), 12); ks = std::max(ks, 8); }   // MFMA tiles: 2^8 .. 2^12 amplitudes
    auto cached_plan = [&](int which, int k, bool inverse, int reg_bits, bool dots, DevPlan& out) {
        const std::vector<int> key = {which, ws->col_bits, k, low_bits, reg_bits, (int)dots, (int)want_v3};
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto it = ctx->plan_cache.find(key);
        if (it == ctx->plan_cache.end()) {
            DevPlan fresh;
            lower_plan(prog, best_plan(k, inverse), fresh, reg_bits, dots, want_v3);
            it = ctx->plan_cache.emplace(key, std::move(fresh)).first;
        }
        out = it->second;   // host vectors copied; device pointers are null in the cache
    };
    cached_plan(2, ka, false, (want_v2 || want_v3) ? 4 : 0, false, ws->fwd);
    cached_plan(0, ka, true, (want_v2 || want_v3) ? 4 : 0, false, ws->inv);
    cached_plan(1, ks, false, want_v3 ? 4 : (want_v2 ? (env_int("AQC_SWEEP_REG_BITS", 4) == 3 ? 3 : 4) : 0), true, ws->sweep);
    for (DevPlan* p : {&ws->fwd, &ws->inv, &ws->sweep}) {
        const std::string err = check_plan(prog, p->plan);
        if (!err.empty()) { delete ws; return fail("planner produced an invalid plan: %s", err.c_str()); }
    }

    const int T = prog.num_thetas();
    const int G = (int)prog.groups.size();
    ws->nslots = G * kSlotsPerGroup;
    const int partial_tiles = ws->sweep.v3 ? 1 : ws->sweep.ntiles;   // family 3: rgrad_kernel writes one value per slot
    std::vector<int> theta_slots(2 * (size_t)std::max(T, 1), -1), slot_ntiles((size_t)std::max(ws->nslots, 1), partial_tiles);
    auto feed = [&](int theta, int slot) {
        if (theta_slots[2 * theta] < 0) theta_slots[2 * theta] = slot; else theta_slots[2 * theta + 1] = slot;
    };
    for (int gi = 0; gi < G; ++gi) {
        const GateGroup& g = prog.groups[gi];
        if (g.type == GROUP_FRONT) {
            feed(g.theta0 + 2, gi * kSlotsPerGroup + 0);
            feed(g.theta0 + 1, gi * kSlotsPerGroup + 1);
            feed(g.theta0 + 0, gi * kSlotsPerGroup + 2);
        } else {
            for (int d = 0; d < prog.tpb; ++d) feed(g.theta0 + d, gi * kSlotsPerGroup + d);
        }
    }

    std::vector<int> slot_theta((size_t)std::max(ws->nslots, 1), -1);
    ws->grads_direct = ws->sweep.v3 && T > 0 && env_int("AQC_GRADS_DIRECT", 1) != 0;
    for (int t = 0; t < T; ++t) {
        if (theta_slots[2 * t] < 0 || theta_slots[2 * t + 1] >= 0) { ws->grads_direct = false; break; }
        slot_theta[theta_slots[2 * t]] = t;
    }
    if (env_int("AQC_VERBOSE", 0) && want_v3)
        fprintf(stderr, "aqc_hip: matrix-core kernels, tiles 2^%d (V / V^H, %d workgroups per CU) / 2^%d (sweep, %d per CU), sub-stages %zu / %zu / %zu\n",
                ws->inv.k, mfma_occupancy(ws->inv.k, false), ws->sweep.k, mfma_occupancy(ws->sweep.k, true), ws->fwd.h_subs3.size(),
                ws->inv.h_subs3.size(), ws->sweep.h_subs3.size());
#define WS_TRY(x) do { if ((x) != 0) { aqc_ws_destroy(ws); return 1; } } while (0)
#define WS_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail("%s failed: %s", #x, hipGetErrorString(e_)); aqc_ws_destroy(ws); return 1; } } while (0)
    WS_HIP(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    WS_HIP(hipEventCreate(&ws->ev0)); WS_HIP(hipEventCreate(&ws->ev1));
    WS_HIP(hipEventCreate(&ws->pev0)); WS_HIP(hipEventCreate(&ws->pev1));
    WS_TRY(upload_plan(ws->fwd)); WS_TRY(upload_plan(ws->inv)); WS_TRY(upload_plan(ws->sweep));
    WS_HIP(hipMalloc((void**)&ws->d_thetas_own, sizeof(double) * (size_t)batch * std::max(T, 1)));
    ws->d_thetas = ws->d_thetas_own;
    WS_HIP(hipMalloc((void**)&ws->d_coef, sizeof(double) * (size_t)batch * (prog.n + prog.num_blocks + 1) * kCoefStride));
    for (int b = 0; b < AQC_NUM_BUFS; ++b) {
        WS_HIP(hipMalloc((void**)&ws->bufs[b], sizeof(double2) * (size_t)batch * ws->lane_elems));
        WS_HIP(hipMemsetAsync(ws->bufs[b], 0, sizeof(double2) * (size_t)batch * ws->lane_elems, ws->stream));
    }
    WS_HIP(hipMalloc((void**)&ws->d_partial, sizeof(double2) * (size_t)batch * std::max(ws->nslots, 1) * partial_tiles));
    for (DevPlan* p : {&ws->fwd, &ws->inv, &ws->sweep})
        if (p->v3) WS_HIP(hipMalloc((void**)&p->d_umat, sizeof(double) * (size_t)batch * std::max<size_t>(p->h_subs3.size(), 1) * 12 * 64));
    {
        std::vector<UJob> jobs;
        for (DevPlan* p : {&ws->inv, &ws->sweep, &ws->fwd})
            if (p->v3)
                for (size_t i = 0; i < p->h_subs3.size(); ++i)
                    jobs.push_back({p->d_subs3 + i, p->d_grps, p->d_umat, (int)i, (int)p->h_subs3.size(), p->plan.inverse ? 1 : 0, prog.entangler});
        if (!jobs.empty()) {
            WS_HIP(hipMalloc((void**)&ws->d_ujobs, sizeof(UJob) * jobs.size()));
            WS_HIP(hipMemcpy(ws->d_ujobs, jobs.data(), sizeof(UJob) * jobs.size(), hipMemcpyHostToDevice));
        }
    }
    if (ws->sweep.v3)
        WS_HIP(hipMalloc((void**)&ws->sweep.d_rpart, sizeof(double2) * (size_t)batch * std::max<size_t>(ws->sweep.h_subs3.size(), 1) *
                                                        sweep3_nparts(ws->sweep.ntiles, batch, ws->sweep.k) * 256));
    WS_HIP(hipMalloc((void**)&ws->d_grads, sizeof(double2) * (size_t)batch * std::max(T, 1)));
    WS_HIP(hipMalloc((void**)&ws->d_theta_slots, sizeof(int) * theta_slots.size()));
    WS_HIP(hipMalloc((void**)&ws->d_slot_ntiles, sizeof(int) * slot_ntiles.size()));
    WS_HIP(hipMemcpy(ws->d_theta_slots, theta_slots.data(), sizeof(int) * theta_slots.size(), hipMemcpyHostToDevice));
    WS_HIP(hipMalloc((void**)&ws->d_slot_theta, sizeof(int) * slot_theta.size()));
    WS_HIP(hipMemcpy(ws->d_slot_theta, slot_theta.data(), sizeof(int) * slot_theta.size(), hipMemcpyHostToDevice));
    WS_HIP(hipMemcpy(ws->d_slot_ntiles, slot_ntiles.data(), sizeof(int) * slot_ntiles.size(), hipMemcpyHostToDevice));
    ws->pin_thetas = (size_t)batch * std::max(T, 1);
    ws->pin_grads = 2 * (size_t)batch * std::max(T, 1);
    ws->pin_small = 2 * (size_t)batch * 64;
    WS_HIP(hipHostMalloc((void**)&ws->h_pin, sizeof(double) * (ws->pin_thetas + ws->pin_grads + ws->pin_small), hipHostMallocDefault));
    ws->vdot_parts = (int)std::min<size_t>(1024, std::max<size_t>(1, ws->lane_elems / 1024));
    WS_HIP(hipMalloc((void**)&ws->d_vdot_part, sizeof(double2) * (size_t)batch * ws->vdot_parts));
    WS_HIP(hipStreamSynchronize(ws->stream));
#undef WS_TRY
#undef WS_HIP
    *out = ws;
    return 0;
}

static void drop_graphs(aqc_ws* ws);

int aqc_ws_destroy(aqc_ws* ws) {
    if (!ws) return 0;
    (void)hipSetDevice(ws->device);
    if (ws->stream) (void)hipStreamSynchronize(ws->stream);
    drop_graphs(ws);
    for (DevPlan* p : {&ws->fwd, &ws->inv, &ws->sweep}) {
        if (p->d_stages) (void)hipFree(p->d_stages);
        if (p->d_ops) (void)hipFree(p->d_ops);
        if (p->d_subs) (void)hipFree(p->d_subs);
        if (p->d_mops) (void)hipFree(p->d_mops);
        if (p->d_subs3) (void)hipFree(p->d_subs3);
        if (p->d_grps) (void)hipFree(p->d_grps);
        if (p->d_umat) (void)hipFree(p->d_umat);
        if (p->d_rpart) (void)hipFree(p->d_rpart);
    }
    void* ptrs[] = {ws->d_thetas_own, ws->d_theta_bank, ws->d_coef, ws->d_partial, ws->d_grads, ws->d_small, ws->d_vdot_part, ws->d_index, ws->d_tmp_index, ws->d_tmp_small,
                    ws->d_theta_slots, ws->d_slot_theta, ws->d_slot_ntiles, ws->d_basis_index, ws->d_vdot_out, ws->d_ujobs};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int b = 0; b < AQC_NUM_BUFS; ++b) if (ws->bufs[b]) (void)hipFree(ws->bufs[b]);
    for (int b = 0; b < AQC_NUM_BUFS; ++b) if (ws->d_combo_prev[b]) (void)hipFree(ws->d_combo_prev[b]);
    if (ws->d_sur) (void)hipFree(ws->d_sur);
    if (ws->d_sur_real) (void)hipFree(ws->d_sur_real);
    if (ws->h_sur) (void)hipHostFree(ws->h_sur);
    if (ws->d_combo_index) (void)hipFree(ws->d_combo_index);
    if (ws->d_combo_coef) (void)hipFree(ws->d_combo_coef);
    if (ws->h_pin) (void)hipHostFree(ws->h_pin);
    for (auto& m : ws->mps) if (m.d_t) (void)hipFree(m.d_t);
    if (ws->d_mps_scratch) (void)hipFree(ws->d_mps_scratch);
    for (auto& t : ws->mps_tabs) if (t.dev) (void)hipFree(t.dev);
    if (ws->d_cd_prog) (void)hipFree(ws->d_cd_prog);
    if (ws->d_cd_thetas) (void)hipFree(ws->d_cd_thetas);
    if (ws->d_cd_fobj) (void)hipFree(ws->d_cd_fobj);
    if (ws->d_mps_lam) (void)hipFree(ws->d_mps_lam);
    for (hipEvent_t ev : {ws->ev0, ws->ev1, ws->pev0, ws->pev1, ws->ev_ready, ws->ev_copied}) if (ev) (void)hipEventDestroy(ev);
    if (ws->copy_stream) { (void)hipStreamSynchronize(ws->copy_stream); (void)hipStreamDestroy(ws->copy_stream); }
    if (ws->mps_stream) { (void)hipStreamSynchronize(ws->mps_stream); (void)hipStreamDestroy(ws->mps_stream); }
    for (hipEvent_t ev : {ws->ev_mps_fork, ws->ev_mps_join}) if (ev) (void)hipEventDestroy(ev);
    if (ws->stream) (void)hipStreamDestroy(ws->stream);
    delete ws;
    return 0;
}

int aqc_ws_set_thetas(aqc_ws* ws, const double* thetas) {
    if (!ws || !thetas) return fail("null argument");
    HIP_OK(hipSetDevice(ws->device));
    const Program& prog = ws->ctx->prog;
    const int T = prog.num_thetas();
    ws->d_thetas = ws->d_thetas_own;
    HIP_OK(hipMemcpyAsync(ws->d_thetas, thetas, sizeof(double) * (size_t)ws->batch * T, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));  // the host buffer may be reused right away
    return run_coef(ws);
}

int aqc_ws_upload(aqc_ws* ws, int buf, const double* src) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    if (!src) return fail("null source");
    HIP_OK(hipSetDevice(ws->device));
    return copy_in(ws, ws->bufs[buf], src, (size_t)ws->batch << ws->ctx->prog.n);
}

int aqc_ws_upload_lane(aqc_ws* ws, int buf, int lane, const double* src) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    if (!src) return fail("null source");
    if (lane < 0 || lane >= ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(ws->device));
    return copy_in(ws, ws->bufs[buf] + (size_t)lane * ws->lane_elems, src, (size_t)1 << ws->ctx->prog.n);
}

int aqc_ws_broadcast(aqc_ws* ws, int buf, const double* src) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    if (!src) return fail("null source");
    HIP_OK(hipSetDevice(ws->device));
    if (copy_in(ws, ws->bufs[buf], src, (size_t)1 << ws->ctx->prog.n)) return 1;
    for (int b = 1; b < ws->batch; ++b)
        HIP_OK(hipMemcpyAsync(ws->bufs[buf] + (size_t)b * ws->lane_elems, ws->bufs[buf], ws->lane_elems * sizeof(double2),
                              hipMemcpyDeviceToDevice, ws->stream));
    return 0;
}

// dst_ws.buf[dst_lane] <- src_ws.buf[src_lane], device to device (both workspaces on the same device, same lane size): how a
// driver hands device-resident targets (e.g. synthesised by another workspace, model_sp_lhs/trotter) to an objective's lanes
int aqc_ws_copy_lane(aqc_ws* dst_ws, int dst_buf, int dst_lane, aqc_ws* src_ws, int src_buf, int src_lane) {
    if (check_buf(dst_ws, dst_buf) || check_buf(src_ws, src_buf)) return 1;
    if (dst_ws->device != src_ws->device) return fail("copy_lane: the two workspaces live on different devices");
    if (dst_ws->lane_elems != src_ws->lane_elems) return fail("copy_lane: lane sizes differ");
    if (dst_lane < 0 || dst_lane >= dst_ws->batch || src_lane < 0 || src_lane >= src_ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(dst_ws->device));
    dst_ws->combo_valid[dst_buf] = false;
    if (src_ws->stream != dst_ws->stream) HIP_OK(hipStreamSynchronize(src_ws->stream));   // the source is complete
    HIP_OK(hipMemcpyAsync(dst_ws->bufs[dst_buf] + (size_t)dst_lane * dst_ws->lane_elems, src_ws->bufs[src_buf] + (size_t)src_lane * src_ws->lane_elems,
                          sizeof(double2) * dst_ws->lane_elems, hipMemcpyDeviceToDevice, dst_ws->stream));
    return 0;
}

int aqc_ws_download(aqc_ws* ws, int buf, double* dst) {
    if (check_buf(ws, buf)) return 1;
    if (!dst) return fail("null destination");
    HIP_OK(hipSetDevice(ws->device));
    return copy_out(ws, dst, ws->bufs[buf], (size_t)ws->batch << ws->ctx->prog.n);
}

int aqc_ws_download_lane(aqc_ws* ws, int buf, int lane, double* dst) {
    if (check_buf(ws, buf)) return 1;
    if (!dst) return fail("null destination");
    if (lane < 0 || lane >= ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(ws->device));
    return copy_out(ws, dst, ws->bufs[buf] + (size_t)lane * ws->lane_elems, (size_t)1 << ws->ctx->prog.n);
}

int aqc_ws_set_basis(aqc_ws* ws, int buf, const int64_t* index) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    if (!index) return fail("null index");
    HIP_OK(hipSetDevice(ws->device));
    const int64_t dim = (int64_t)1 << ws->ctx->prog.n;
    std::vector<long long> elem(ws->batch);
    for (int b = 0; b < ws->batch; ++b) {
        if (index[b] < 0 || index[b] >= dim) return fail("basis index out of range");
        elem[b] = (long long)index[b] << ws->col_bits;
    }
    if (!ws->d_basis_index) HIP_OK(hipMalloc((void**)&ws->d_basis_index, sizeof(long long) * ws->batch));
    HIP_OK(hipMemcpyAsync(ws->d_basis_index, elem.data(), sizeof(long long) * ws->batch, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    HIP_OK(hipMemsetAsync(ws->bufs[buf], 0, sizeof(double2) * (size_t)ws->batch * ws->lane_elems, ws->stream));
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_scatter_one(ws->bufs[buf], ws->lane_elems, ws->batch, ws->d_basis_index, ws->stream));
    return 0;
}

int aqc_ws_set_identity(aqc_ws* ws, int buf) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    const int dim = 1 << ws->ctx->prog.n;
    if (ws->ncols != dim) return fail("identity needs a square workspace (ncols == 2^n)");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipMemsetAsync(ws->bufs[buf], 0, sizeof(double2) * (size_t)ws->batch * ws->lane_elems, ws->stream));
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_set_identity(ws->bufs[buf], ws->lane_elems, dim, ws->pitch, ws->batch, ws->stream));
    return 0;
}

// buffer[lane] = coef[lane][0] |index[lane][0]> + coef[lane][1] |index[lane][1]>   (index[lane][1] < 0: one term).
// The gradient of <V x|y> is conjugate-linear in x, so the surrogate objective's two sweeps -- from |state_0> and from
// the leading flip state, combined as c_0 g_0 + c_max g_max (objective_lhs_sur_max.py:147-191) -- are ONE sweep from
// x = conj(c_0) |state_0> + conj(c_max) |state_max>.  Only the positions written by the previous call are cleared
// (the buffer is 2^n amplitudes per lane); any other writer of the buffer makes the next call clear all of it.
int aqc_ws_set_combo(aqc_ws* ws, int buf, const int64_t* index, const double* coef) {
    if (check_buf(ws, buf)) return 1;
    if (!index || !coef) return fail("null argument");
    if (buf == AQC_BUF_W || buf == AQC_BUF_ZW || buf == AQC_BUF_Z) return fail("set_combo targets an lhs buffer (X, X2) or Y");
    HIP_OK(hipSetDevice(ws->device));
    const int64_t dim = (int64_t)1 << ws->ctx->prog.n;
    const int B = ws->batch;
    std::vector<long long> elem(2 * (size_t)B);
    for (int b = 0; b < B; ++b) {
        const int64_t i0 = index[2 * b], i1 = index[2 * b + 1];
        if (i0 < 0 || i0 >= dim || i1 >= dim) return fail("basis index out of range");
        if (i1 == i0) return fail("the two basis states of a lane must differ");
        elem[2 * b] = (long long)i0 << ws->col_bits;
        elem[2 * b + 1] = i1 < 0 ? -1 : (long long)i1 << ws->col_bits;
    }
    if (!ws->d_combo_index) HIP_OK(hipMalloc((void**)&ws->d_combo_index, sizeof(long long) * 2 * B));
    if (!ws->d_combo_coef) HIP_OK(hipMalloc((void**)&ws->d_combo_coef, sizeof(double2) * 2 * B));
    if (!ws->d_combo_prev[buf]) { HIP_OK(hipMalloc((void**)&ws->d_combo_prev[buf], sizeof(long long) * 2 * B)); ws->combo_valid[buf] = false; }
    HIP_OK(hipMemcpyAsync(ws->d_combo_index, elem.data(), sizeof(long long) * 2 * B, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipMemcpyAsync(ws->d_combo_coef, coef, sizeof(double2) * 2 * B, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));   // `elem` and the caller's array may go away
    if (!ws->combo_valid[buf]) {
        HIP_OK(hipMemsetAsync(ws->bufs[buf], 0, sizeof(double2) * (size_t)B * ws->lane_elems, ws->stream));
        HIP_OK(hipMemsetAsync(ws->d_combo_prev[buf], 0xff, sizeof(long long) * 2 * B, ws->stream));   // -1: nothing to clear
    }
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_scatter_two(ws->bufs[buf], ws->lane_elems, B, ws->d_combo_index, ws->d_combo_coef, ws->d_combo_prev[buf], ws->stream));
    ws->combo_valid[buf] = true;
    return 0;
}

int aqc_ws_apply(aqc_ws* ws, int inverse, int src_buf, int dst_buf) {
    if (check_buf(ws, src_buf) || check_buf(ws, dst_buf)) return 1;
    ws->combo_valid[dst_buf] = false;
    if (ensure_coef(ws)) return 1;
    HIP_OK(hipSetDevice(ws->device));
    return run_apply(ws, inverse != 0, src_buf, dst_buf);
}

int aqc_ws_grad(aqc_ws* ws, int block_from, int block_to, int front_layer) {
    return aqc_ws_grad_from(ws, AQC_BUF_X, block_from, block_to, front_layer);
}

static int results_guard(aqc_ws* ws);

int aqc_ws_grad_from(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer) {
    if (check_buf(ws, x_buf)) return 1;
    if (results_guard(ws)) return 1;
    if (x_buf == AQC_BUF_W || x_buf == AQC_BUF_ZW || x_buf == AQC_BUF_Z) return fail("lhs buffer must not be Z, W or ZW");
    if (ensure_coef(ws)) return 1;
    const Program& prog = ws->ctx->prog;
    if (block_from < 0) { block_from = 0; block_to = prog.num_blocks; }
    if (prog.num_blocks > 0 && !(0 <= block_from && block_from < block_to && block_to <= prog.num_blocks))
        return fail("invalid block_range [%d, %d)", block_from, block_to);
    HIP_OK(hipSetDevice(ws->device));
    DevPlan& p = ws->sweep;
    if (p.v3) {
        if (ensure_umat(ws, p)) return 1;
        const int nsubs = (int)p.h_subs3.size();
        for (size_t s = 0; s < p.h_stages.size(); ++s) {
            Stage3Args a;
            memset(&a, 0, sizeof a);
            a.stage = p.h_stages[s];
            a.subs = p.d_subs3;
            a.umat = p.d_umat;
            a.nsubs_total = nsubs;
            a.in0 = s == 0 ? ws->bufs[x_buf] : ws->bufs[AQC_BUF_W];
            a.in1 = s == 0 ? ws->bufs[AQC_BUF_Z] : ws->bufs[AQC_BUF_ZW];
            a.out0 = ws->bufs[AQC_BUF_W];
            a.out1 = ws->bufs[AQC_BUF_ZW];
            a.lane_stride = ws->lane_elems;
            a.rpart = p.d_rpart;
            a.ntiles = p.ntiles;
            a.batch = ws->batch;
            a.chunk = sweep3_chunk(p.ntiles, ws->batch, p.k);
            a.nparts = sweep3_nparts(p.ntiles, ws->batch, p.k);
            a.store_out = s + 1 < p.h_stages.size() ? 1 : 0;
            if (a.stage.nsubs > 0) stage3_first_offsets(a, p.h_subs3[a.stage.sub_begin]);
#ifdef AQC_TUNING   // AQC_STAMPS=1: mean cycles per phase of the sweep workgroups of this launch, on stderr
            static unsigned long long* d_stamps = nullptr;
            const size_t nwg = (size_t)p.ntiles * ws->batch;
            const bool stamps = env_int("AQC_STAMPS", 0) != 0;
            a.debug = env_int("AQC_DEBUG_SKIP", 0);
            if (stamps) {
                if (!d_stamps) HIP_OK(hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 65536 * kStampSlots));
                if (nwg <= 65536) { HIP_OK(hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * nwg * kStampSlots, ws->stream)); a.stamps = d_stamps; }
            }
#endif
            {
                ProfScope ps(ws, AQC_K_SWEEP);
                HIP_OK(launch_sweep3(p.ntiles, ws->batch, p.k, ws->stream, a));
            }
#ifdef AQC_TUNING
            if (a.stamps) {
                std::vector<unsigned long long> h(nwg * kStampSlots);
                HIP_OK(hipStreamSynchronize(ws->stream));
                HIP_OK(hipMemcpy(h.data(), d_stamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
                const int ns = p.h_stages[s].nsubs;
                // the 2^12 sweep is persistent: a workgroup's per-sub-stage stamps are those of its LAST item, slot S-4 its end,
                // slots S-6 / S-5 bracket its last hand-over to a prefetched tile
                double load = 0, store = 0, total = 0, mf = 0, bar = 0, red = 0, top = 0, turn = 0;
                size_t live = 0;
                unsigned long long first_start = ~0ull, last_start = 0, first_end = ~0ull, last_end = 0, wg_min = ~0ull, wg_max = 0;
                for (size_t w = 0; w < nwg; ++w) {
                    const unsigned long long* t = h.data() + w * kStampSlots;
                    if (t[kStampSlots - 4] == 0) continue;   // no workgroup with this index (persistent grid)
                    ++live;
                    first_start = std::min(first_start, t[0]); last_start = std::max(last_start, t[0]);
                    first_end = std::min(first_end, t[kStampSlots - 4]); last_end = std::max(last_end, t[kStampSlots - 4]);
                    wg_min = std::min(wg_min, t[kStampSlots - 4] - t[0]); wg_max = std::max(wg_max, t[kStampSlots - 4] - t[0]);
                    load += (double)(t[1] - t[0]);
                    store += (double)(t[kStampSlots - 1] - t[kStampSlots - 2]);
                    total += (double)(t[kStampSlots - 4] - t[0]);
                    if (t[kStampSlots - 5]) turn += (double)(t[kStampSlots - 5] - t[kStampSlots - 6]);
                    for (int i = 0; i < ns && 5 + 4 * i < kStampSlots - 6; ++i) {
                        if (i) top += (double)(t[2 + 4 * i] - t[5 + 4 * (i - 1)]);
                        mf += (double)(t[3 + 4 * i] - t[2 + 4 * i]);
                        bar += (double)(t[4 + 4 * i] - t[3 + 4 * i]);
                        red += (double)(t[5 + 4 * i] - t[4 + 4 * i]);
                    }
                }
                const double n = (double)std::max<size_t>(live, 1), items = (double)nwg / n;
                fprintf(stderr, "aqc_hip stamps: stage %zu (%d sub-stages, %zu workgroups x %.1f items): total %.0f cycles per item = first load %.0f/items + "
                        "per sub-stage [top %.0f + mfma loop %.0f + scratch/barrier %.0f + reduce %.0f] + store %.0f + hand-over %.0f\n", s, ns, live, items,
                        total / n / items, load / n, top / n / std::max(ns - 1, 1), mf / n / ns, bar / n / ns, red / n / ns, store / n, turn / n);
                fprintf(stderr, "aqc_hip stamps: stage %zu workgroup lifetimes (s_memtime ticks): min %llu max %llu; starts spread over %llu, ends over %llu; "
                        "first start -> last end %llu\n", s, wg_min, wg_max, last_start - first_start, last_end - first_end, last_end - first_start);
            }
#endif
        }
        ProfScope ps(ws, AQC_K_FINALIZE);
        HIP_OK(launch_rgrad(p.d_subs3, p.d_grps, prog.entangler, ws->d_thetas, prog.num_thetas(), p.d_rpart, p.ntiles, nsubs, ws->d_partial,
                            ws->nslots, block_from, block_to, front_layer ? 1 : 0, ws->batch, ws->stream,
                            ws->grads_direct ? ws->d_slot_theta : nullptr, ws->d_grads, ws->mirror_grads,
                            ws->gather_rides ? GatherJob{ws->bufs[AQC_BUF_Z], ws->lane_elems, ws->d_index, ws->gather_count, ws->d_small, ws->mirror_small}
                                             : GatherJob{nullptr, 0, nullptr, 0, nullptr, nullptr},
                            sweep3_nparts(p.ntiles, ws->batch, p.k), sweep3_chunk(p.ntiles, ws->batch, p.k)));
#ifdef AQC_TUNING
        if (env_int("AQC_STAMPS", 0) != 0) { HIP_OK(hipStreamSynchronize(ws->stream)); rgrad_print_stamps(nsubs); }
#endif
        if (!ws->grads_direct)   // some theta collects two slots (2nd-order Trotter half-layers, core_operations.py:966-968)
            HIP_OK(launch_finalize(ws->d_partial, ws->d_theta_slots, ws->d_slot_ntiles, ws->d_grads, prog.num_thetas(), ws->nslots,
                                   1, prog.n, prog.tpb, block_from, block_to, front_layer ? 1 : 0, ws->batch, ws->stream, ws->mirror_grads));
        return 0;
    }
    for (size_t s = 0; s < p.h_stages.size(); ++s) {
        StageArgs a;
        memset(&a, 0, sizeof a);
        a.stage = p.d_stages + s;
        a.ops = p.d_ops;
        a.subs = p.d_subs;
        a.mops = p.d_mops;
        a.coef = ws->d_coef;
        a.ncoef = prog.n + prog.num_blocks + 1;
        a.in0 = s == 0 ? ws->bufs[x_buf] : ws->bufs[AQC_BUF_W];
        a.in1 = s == 0 ? ws->bufs[AQC_BUF_Z] : ws->bufs[AQC_BUF_ZW];
        a.out0 = ws->bufs[AQC_BUF_W];
        a.out1 = ws->bufs[AQC_BUF_ZW];
        a.lane_stride = ws->lane_elems;
        a.partial = ws->d_partial;
        a.nslots = ws->nslots;
        a.ntiles_max = p.ntiles;
#ifdef AQC_TUNING   // timing experiments only (tools/tune.py); never part of the shipped library
        a.debug = env_int("AQC_DEBUG_SKIP", 0);
#endif
        a.from = block_from;
        a.to = block_to;
        a.front = front_layer ? 1 : 0;
        ProfScope ps(ws, AQC_K_SWEEP);
        if (p.v2) HIP_OK(launch_sweep2(prog.entangler, p.ntiles, ws->batch, p.k, p.reg_bits, ws->stream, a));
        else HIP_OK(launch_sweep(prog.entangler, p.ntiles, ws->batch, ws->threads, p.k, ws->stream, a));
    }
    ProfScope ps(ws, AQC_K_FINALIZE);
    HIP_OK(launch_finalize(ws->d_partial, ws->d_theta_slots, ws->d_slot_ntiles, ws->d_grads, prog.num_thetas(), ws->nslots,
                           p.ntiles, prog.n, prog.tpb, block_from, block_to, front_layer ? 1 : 0, ws->batch, ws->stream, ws->mirror_grads));
    return 0;
}

static void drop_graphs(aqc_ws* ws) {
    for (auto& kv : ws->graphs) (void)hipGraphExecDestroy(kv.second);
    ws->graphs.clear();
}

int aqc_ws_eval(aqc_ws* ws, const double* thetas, int do_vdag, double* gathered, int x_buf, int block_from, int block_to,
                int front_layer, double* grads) {
    if (!ws) return fail("null workspace");
    HIP_OK(hipSetDevice(ws->device));
    if (ws->copy_pending) {   // result copies of an earlier aqc_ws_results_async: this call reuses the pinned buffer and may replay a graph
        HIP_OK(hipStreamSynchronize(ws->copy_stream));
        ws->copy_pending = false;
    }
    const Program& prog = ws->ctx->prog;
    const size_t nth = (size_t)ws->batch * prog.num_thetas();
    double* pin_th = ws->h_pin;
    double* pin_gr = ws->h_pin + ws->pin_thetas;
    double* pin_sm = pin_gr + ws->pin_grads;
    size_t nsm = 0;
    if (gathered) {
        if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called");
        nsm = (size_t)ws->batch * ws->gather_count;
        if (2 * nsm > ws->pin_small) return fail("too many gathered amplitudes for the staging buffer");
    }
    if (!thetas && (do_vdag || grads) && ensure_coef(ws)) return 1;
    if (check_buf(ws, x_buf)) return 1;
    // Small results skip the device-to-host copy nodes: the producing kernels write a second copy straight into the pinned
    // staging buffer (two nodes and their dependencies less on the single-evaluation critical path).
    const bool zero_copy = sizeof(double2) * (nth + nsm) <= 65536;
    struct MirrorScope {
        aqc_ws* w;
        MirrorScope(aqc_ws* w_, double* g, double* s) : w(w_) { w->mirror_grads = g; w->mirror_small = s; }
        ~MirrorScope() {   // also on the error paths of enqueue(): no stale pinned thetas / riding gather in the next call
            w->mirror_grads = nullptr; w->mirror_small = nullptr; w->theta_host = nullptr; w->gather_rides = false;
        }
    } mirror_scope(ws, zero_copy ? pin_gr : nullptr, zero_copy ? pin_sm : nullptr);
    auto enqueue = [&]() -> int {   // everything between the host copy of the thetas and the final synchronisation
        if (thetas) {
            ws->d_thetas = ws->d_thetas_own;
            // matrix-core path, small batch: no copy node -- the U builder (first kernel of V^H or of the sweep) reads the pinned
            // thetas over the bus and stores them to HBM for the gradient walk
            const bool direct_thetas = zero_copy && (do_vdag || grads) && ws->fwd.v3 && ws->inv.v3 && ws->sweep.v3 && !ws->need_coef &&
                                       (do_vdag ? ws->inv.v3 : true);
            if (!direct_thetas) HIP_OK(hipMemcpyAsync(ws->d_thetas, pin_th, sizeof(double) * nth, hipMemcpyHostToDevice, ws->stream));
            if (run_coef(ws)) return 1;
            ws->theta_host = direct_thetas ? pin_th : nullptr;
        }
        if (do_vdag && run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;
        // with a gradient in the same call the gather (it only reads Z, which the sweep leaves intact) rides along as one
        // extra workgroup per lane of the gradient-walk kernel: one node less on the single-evaluation critical path
        const bool ride = gathered && grads && zero_copy && ws->sweep.v3;
        if (gathered && !ride) {
            if (aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
            if (!zero_copy) HIP_OK(hipMemcpyAsync(pin_sm, ws->d_small, sizeof(double2) * nsm, hipMemcpyDeviceToHost, ws->stream));
        }
        ws->gather_rides = ride;
        if (grads) {
            if (aqc_ws_grad_from(ws, x_buf, block_from, block_to, front_layer)) return 1;
            if (!zero_copy) HIP_OK(hipMemcpyAsync(pin_gr, ws->d_grads, sizeof(double2) * nth, hipMemcpyDeviceToHost, ws->stream));
        }
        ws->gather_rides = false;
        ws->theta_host = nullptr;
        return 0;
    };
    if (thetas) memcpy(pin_th, thetas, sizeof(double) * nth);
    static const bool graphs_on = env_int("AQC_GRAPH", 1) != 0;
    if (thetas && graphs_on && !ws->profile) {
        const std::vector<long long> key = {do_vdag, gathered ? 1 : 0, grads ? 1 : 0, x_buf, block_from, block_to, front_layer,
                                            (long long)ws->gather_count, (long long)(size_t)ws->d_small, (long long)(size_t)ws->h_pin};
        auto it = ws->graphs.find(key);
        if (it == ws->graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            HIP_OK(hipStreamSynchronize(ws->stream));
            HIP_OK(hipStreamBeginCapture(ws->stream, hipStreamCaptureModeThreadLocal));
            ws->capturing = true;
            const int rc = enqueue();
            ws->capturing = false;
            const hipError_t e = hipStreamEndCapture(ws->stream, &graph);
            if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return 1; }
            if (e != hipSuccess || !graph) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(ei));
            if (ws->graphs.size() >= 16) drop_graphs(ws);
            it = ws->graphs.emplace(key, exec).first;
        }
        ws->d_thetas = ws->d_thetas_own;   // host-side state that enqueue() would have set
        ws->coef_valid = true;
        ws->fwd.u_valid = false;
        ws->inv.u_valid = ws->sweep.u_valid = (do_vdag || grads) && ws->inv.v3 && ws->sweep.v3;
        HIP_OK(hipGraphLaunch(it->second, ws->stream));
    } else if (enqueue()) {
        return 1;
    }
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (gathered) memcpy(gathered, pin_sm, sizeof(double2) * nsm);
    if (grads) memcpy(grads, pin_gr, sizeof(double2) * nth);
    return 0;
}

// ---- device-resident multi-start L-BFGS on the lane-batched surrogate objective (aqc_lbfgs.hip) ---------------
// Preconditions (what BatchedSurrogateObjective sets up): targets in Y, |state_0> one-hot in X, the flip-state indices
// registered with aqc_ws_gather_setup (state 0 first).  thetas, gradients and the history stay in HBM; per evaluation
// the host reads one flag word, per line-search trial another.
int aqc_ws_lbfgs(aqc_ws* ws, const double* x0, int maxiter, int memory, double gtol, double ftol, double fid_thr, int max_backtracks,
                 int block_from, int block_to, int front_layer, double* x_out, double* f_out, double* fidelity_out, int64_t* nit_out,
                 int64_t* nfev_out, double* weight_out, int64_t* max_no_out) {
    if (!ws || !x0 || !x_out || !f_out) return fail("null argument");
    if (block_from < 0) { block_from = 0; block_to = ws->ctx->prog.num_blocks; }
    if (ws->ctx->prog.num_blocks > 0 && !(0 <= block_from && block_from < block_to && block_to <= ws->ctx->prog.num_blocks))
        return fail("invalid block_range [%d, %d)", block_from, block_to);
    if (ws->ncols != 1) return fail("the L-BFGS driver works on state-vector workspaces");
    if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called (flip-state indices, state 0 first)");
    if (memory < 1 || memory > 32 || maxiter < 1 || max_backtracks < 1) return fail("invalid L-BFGS parameters");
    HIP_OK(hipSetDevice(ws->device));
    const Program& prog = ws->ctx->prog;
    const int B = ws->batch, T = prog.num_thetas(), S = ws->gather_count;
    const size_t BT = (size_t)B * T, BS = (size_t)B * S;
    hipStream_t st_ = ws->stream;
    HIP_OK(hipStreamSynchronize(st_));
    // one allocation for all double arrays, one for the complex ones, one for the integers
    const size_t nd = BT * (7 + 2 * (size_t)memory) + (size_t)B * (6 + memory + 2 + 2);
    double* dd = nullptr;
    double2* dc = nullptr;
    int* di = nullptr;
    long long* dl = nullptr;
    int* h_flags = nullptr;
    auto cleanup = [&]() {
        if (dd) (void)hipFree(dd);
        if (dc) (void)hipFree(dc);
        if (di) (void)hipFree(di);
        if (dl) (void)hipFree(dl);
        if (h_flags) (void)hipHostFree(h_flags);
    };
#define LB_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define LB_TRY(expr) do { if ((expr) != 0) { cleanup(); return 1; } } while (0)
    LB_OK(hipMalloc((void**)&dd, nd * sizeof(double)));
    LB_OK(hipMalloc((void**)&dc, (3 * BT + 3 * BS) * sizeof(double2)));
    LB_OK(hipMalloc((void**)&di, (size_t)(3 * B + 8) * sizeof(int)));
    LB_OK(hipMalloc((void**)&dl, (size_t)(3 * B) * sizeof(long long)));
    LB_OK(hipHostMalloc((void**)&h_flags, 8 * sizeof(int), hipHostMallocDefault));
    LB_OK(hipMemsetAsync(dd, 0, nd * sizeof(double), st_));
    LB_OK(hipMemsetAsync(di, 0, (size_t)(3 * B + 8) * sizeof(int), st_));
    LbState L;
    double* p = dd;
    auto take = [&](size_t n) { double* r = p; p += n; return r; };
    L.B = B; L.T = T; L.S = S; L.memory = memory;
    L.x = take(BT); L.g = take(BT); L.d = take(BT); L.x_new = take(BT);
    double* gt = take(BT);        // gradient at the trial points
    double* g_acc = take(BT);     // gradient at the accepted points under the new state
    double* spare = take(BT); (void)spare;
    L.Smem = take(BT * memory); L.Ymem = take(BT * memory);
    L.f = take(B); L.slope = take(B); L.step = take(B); L.weight = take(B); L.fidelity = take(B);
    double* ft = take(B);
    L.rho = take((size_t)B * memory);
    (void)take(2 * (size_t)B);
    double* f_acc = take(2 * (size_t)B);
    L.cur_g0 = dc; L.acc_g0 = dc + BT;
    double2* raw_g0_t = dc + 2 * BT;
    L.cur_hs = dc + 3 * BT; L.acc_hs = L.cur_hs + BS;
    double2* raw_hs_t = L.acc_hs + BS;
    L.active = di; L.done = di + B; L.max_no = di + 2 * B;
    int* d_flags = di + 3 * B;
    L.nit = dl;
    long long* d_prev = dl + B;   // [B][2]: positions of X2 written by the previous evaluation
    {   // weight = 1, max_no = 0, active = 1, X2 empty
        std::vector<double> ones(B, 1.0);
        std::vector<int> one_i(B, 1);
        std::vector<long long> neg(3 * (size_t)B, 0);
        for (int b = 0; b < 2 * B; ++b) neg[B + b] = -1;
        LB_OK(hipMemcpyAsync(L.weight, ones.data(), sizeof(double) * B, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemcpyAsync(L.active, one_i.data(), sizeof(int) * B, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemcpyAsync(dl, neg.data(), sizeof(long long) * 3 * B, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemcpyAsync(L.x, x0, sizeof(double) * BT, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemsetAsync(ws->bufs[AQC_BUF_X2], 0, sizeof(double2) * (size_t)B * ws->lane_elems, st_));
        ws->combo_valid[AQC_BUF_X2] = false;
        LB_OK(hipStreamSynchronize(st_));
    }
    int64_t nfev = 0;
    auto read_flags = [&]() -> int {
        HIP_OK(hipMemcpyAsync(h_flags, d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, st_));
        HIP_OK(hipStreamSynchronize(st_));
        return 0;
    };
    // f, g at the point in the workspace's theta buffer; raw results to (raw_hs, raw_g).  V^H, the amplitudes, the lane's
    // combined lhs state (lb_prepare) and ONE sweep from it -- no host round trip inside an evaluation.
    auto evaluate = [&](int update, double* f_o, double* g_o, double2* raw_hs, double2* raw_g) -> int {
        ws->d_thetas = ws->d_thetas_own;
        if (run_coef(ws)) return 1;
        if (run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;
        if (aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
        HIP_OK(lb_prepare(L, ws->d_small, update, f_o, raw_hs, ws->bufs[AQC_BUF_X2], ws->lane_elems, ws->d_index, d_prev, st_));
        if (aqc_ws_grad_from(ws, AQC_BUF_X2, block_from, block_to, front_layer)) return 1;
        HIP_OK(lb_take(L, ws->d_grads, g_o, raw_g, st_));
        ++nfev;
        return 0;
    };
    LB_OK(hipMemcpyAsync(ws->d_thetas_own, L.x, sizeof(double) * BT, hipMemcpyDeviceToDevice, st_));
    LB_TRY(evaluate(1, L.f, L.g, L.cur_hs, L.cur_g0));
    int count = 0;
    for (int it = 0; it < maxiter; ++it) {
        LB_OK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), st_));
        LB_OK(lb_active(L, gtol, fid_thr, d_flags, st_));
        LB_TRY(read_flags());
        if (!h_flags[2]) break;
        LB_OK(lb_direction(L, count, st_));
        LB_OK(lb_copy_raw(L, st_));
        for (int bt = 0; bt < max_backtracks; ++bt) {
            LB_OK(lb_trial(L, ws->d_thetas_own, st_));
            LB_TRY(evaluate(0, ft, gt, raw_hs_t, raw_g0_t));
            LB_OK(hipMemsetAsync(d_flags + 3, 0, sizeof(int), st_));
            LB_OK(lb_armijo(L, 1e-4, ws->d_thetas_own, ft, raw_hs_t, raw_g0_t, d_flags, st_));
            // the probe of the state update rides on the same read of the flags (it is only used once no lane backtracks any more)
            LB_OK(hipMemsetAsync(d_flags + 1, 0, sizeof(int), st_));
            LB_OK(lb_probe(L, L.acc_hs, d_flags, st_));
            LB_TRY(read_flags());
            if (!h_flags[3]) break;
        }
        // state update at the accepted points: from their raw results when no lane would lead with a flip state,
        // else by a device evaluation at x_new (the second sweep depends on the state chosen now)
        if (h_flags[1]) {   // (the last round's probe: nothing has touched the accepted points since)
            LB_OK(hipMemcpyAsync(ws->d_thetas_own, L.x_new, sizeof(double) * BT, hipMemcpyDeviceToDevice, st_));
            LB_TRY(evaluate(1, f_acc, g_acc, L.acc_hs, L.acc_g0));
        } else {
            LB_OK(lb_commit0(L, L.acc_hs, L.acc_g0, f_acc, g_acc, st_));
        }
        LB_OK(lb_history(L, count, ftol, f_acc, g_acc, st_));
        ++count;
    }
    LB_OK(hipMemcpyAsync(x_out, L.x, sizeof(double) * BT, hipMemcpyDeviceToHost, st_));
    LB_OK(hipMemcpyAsync(f_out, L.f, sizeof(double) * B, hipMemcpyDeviceToHost, st_));
    if (fidelity_out) LB_OK(hipMemcpyAsync(fidelity_out, L.fidelity, sizeof(double) * B, hipMemcpyDeviceToHost, st_));
    if (nit_out) LB_OK(hipMemcpyAsync(nit_out, L.nit, sizeof(long long) * B, hipMemcpyDeviceToHost, st_));
    if (weight_out) LB_OK(hipMemcpyAsync(weight_out, L.weight, sizeof(double) * B, hipMemcpyDeviceToHost, st_));
    std::vector<int> h_max_no(B, 0);
    if (max_no_out) LB_OK(hipMemcpyAsync(h_max_no.data(), L.max_no, sizeof(int) * B, hipMemcpyDeviceToHost, st_));
    LB_OK(hipStreamSynchronize(st_));
    if (max_no_out) for (int b = 0; b < B; ++b) max_no_out[b] = h_max_no[b];
    if (nfev_out) *nfev_out = nfev;
    cleanup();
#undef LB_OK
#undef LB_TRY
    return 0;
}

// One evaluation of the lane-batched surrogate objective without the host inside it: V^H, the flip-state amplitudes, the
// optional state update (hysteresis + weight smoothing, objective_lhs_sur_max.py:113-117,186), the value, the combined lhs
// state of every lane and ONE sweep from it (see aqc_ws_set_combo) -- the evaluate step of aqc_ws_lbfgs as a call of its own.
// Same preconditions: targets in Y, flip-state indices registered (state 0 first), X2 is used for the lhs states.
int aqc_ws_surrogate_eval(aqc_ws* ws, const double* thetas, int update_state, double* weight_io, int64_t* max_no_io, int block_from,
                          int block_to, int front_layer, double* f_out, double* fidelity_out, double* hs_out, double* grads_out,
                          double* grad_real_out) {
    if (!ws || !thetas || !weight_io || !max_no_io || !f_out || !(grads_out || grad_real_out)) return fail("null argument");
    if (ws->ncols != 1) return fail("the surrogate objective works on state-vector workspaces");
    if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called (flip-state indices, state 0 first)");
    if (update_state < 0 || update_state > 2) return fail("update_state is 0 (none), 1 (hysteresis and weight) or 2 (hysteresis only)");
    HIP_OK(hipSetDevice(ws->device));
    if (ws->copy_pending) {   // as in aqc_ws_eval: the pinned staging buffer is reused
        HIP_OK(hipStreamSynchronize(ws->copy_stream));
        ws->copy_pending = false;
    }
    const Program& prog = ws->ctx->prog;
    const int B = ws->batch, T = prog.num_thetas(), S = ws->gather_count;
    const size_t nth = (size_t)B * T;
    for (int b = 0; b < B; ++b)
        if (max_no_io[b] < 0 || max_no_io[b] >= S) return fail("leading state %lld of lane %d out of range", (long long)max_no_io[b], b);
    hipStream_t st = ws->stream;
    const size_t ndbl = (size_t)B * (3 + 2 * (size_t)S), bytes = ndbl * sizeof(double) + (size_t)B * sizeof(int);
    if (ws->sur_states != S) {
        HIP_OK(hipStreamSynchronize(st));
        if (ws->d_sur) { HIP_OK(hipFree(ws->d_sur)); ws->d_sur = nullptr; }
        if (ws->h_sur) { HIP_OK(hipHostFree(ws->h_sur)); ws->h_sur = nullptr; }
        ws->sur_states = 0;
        HIP_OK(hipMalloc(&ws->d_sur, bytes));
        HIP_OK(hipHostMalloc(&ws->h_sur, bytes, hipHostMallocDefault));
        ws->sur_states = S;
    }
    double* hd = static_cast<double*>(ws->h_sur);
    // Small problems (single evaluations above all): no copy nodes -- the kernels read the thetas and the objective state from
    // pinned host memory and write the state block and a second copy of the gradient straight back into it (as aqc_ws_eval does)
    const bool zero_copy = sizeof(double2) * (nth + (size_t)B * S) <= 65536;
    const bool direct_thetas = zero_copy && ws->fwd.v3 && ws->inv.v3 && ws->sweep.v3 && !ws->need_coef;
    double* dd = zero_copy ? hd : static_cast<double*>(ws->d_sur);
    struct Scope {
        aqc_ws* w;
        Scope(aqc_ws* w_, double* g) : w(w_) { w->mirror_grads = g; w->mirror_small = nullptr; }
        ~Scope() { w->mirror_grads = nullptr; w->mirror_small = nullptr; w->theta_host = nullptr; w->gather_rides = false; }
    } scope(ws, zero_copy ? ws->h_pin + ws->pin_thetas : nullptr);
    LbState L;
    memset(&L, 0, sizeof L);
    L.B = B; L.T = T; L.S = S;
    double* d_f = dd;
    L.fidelity = dd + B;
    L.weight = dd + 2 * (size_t)B;
    double2* d_hs = reinterpret_cast<double2*>(dd + 3 * (size_t)B);
    L.max_no = reinterpret_cast<int*>(dd + ndbl);
    int* h_max = reinterpret_cast<int*>(hd + ndbl);
    // state in: weight and leading state of every lane
    memcpy(hd + 2 * (size_t)B, weight_io, sizeof(double) * B);
    for (int b = 0; b < B; ++b) h_max[b] = (int)max_no_io[b];
    double* pin_th = ws->h_pin;
    double* pin_gr = ws->h_pin + ws->pin_thetas;
    memcpy(pin_th, thetas, sizeof(double) * nth);
    if (!ws->d_combo_prev[AQC_BUF_X2]) {
        HIP_OK(hipMalloc((void**)&ws->d_combo_prev[AQC_BUF_X2], sizeof(long long) * 2 * B));
        ws->combo_valid[AQC_BUF_X2] = false;
    }
    if (!ws->combo_valid[AQC_BUF_X2]) {   // (outside the replayed part: a whole-buffer clear is a one-off)
        HIP_OK(hipMemsetAsync(ws->bufs[AQC_BUF_X2], 0, sizeof(double2) * (size_t)B * ws->lane_elems, st));
        HIP_OK(hipMemsetAsync(ws->d_combo_prev[AQC_BUF_X2], 0xff, sizeof(long long) * 2 * B, st));   // -1: nothing to clear
    }
    const bool real_only = !zero_copy && !grads_out;
    if (real_only && ws->sur_real_cap < nth) {
        HIP_OK(hipStreamSynchronize(st));
        if (ws->d_sur_real) { HIP_OK(hipFree(ws->d_sur_real)); ws->d_sur_real = nullptr; ws->sur_real_cap = 0; }
        HIP_OK(hipMalloc((void**)&ws->d_sur_real, sizeof(double) * nth));
        ws->sur_real_cap = nth;
    }
    auto enqueue = [&]() -> int {   // everything between the host copies of the inputs and the final synchronisation
        if (!zero_copy) {
            HIP_OK(hipMemcpyAsync(L.weight, hd + 2 * (size_t)B, sizeof(double) * B, hipMemcpyHostToDevice, st));
            HIP_OK(hipMemcpyAsync(L.max_no, h_max, sizeof(int) * B, hipMemcpyHostToDevice, st));
        }
        ws->d_thetas = ws->d_thetas_own;
        if (!direct_thetas) HIP_OK(hipMemcpyAsync(ws->d_thetas, pin_th, sizeof(double) * nth, hipMemcpyHostToDevice, st));
        if (run_coef(ws)) return 1;
        ws->theta_host = direct_thetas ? pin_th : nullptr;   // the U builder reads the pinned thetas and stores them to HBM
        if (run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;
        if (aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
        {
            ProfScope ps(ws, AQC_K_MISC);
            HIP_OK(lb_prepare(L, ws->d_small, update_state, d_f, d_hs, ws->bufs[AQC_BUF_X2], ws->lane_elems, ws->d_index,
                              ws->d_combo_prev[AQC_BUF_X2], st));
        }
        // (update_state == 0 leaves weight / max_no / fidelity as they came in; fidelity is only written by an update)
        if (aqc_ws_grad_from(ws, AQC_BUF_X2, block_from, block_to, front_layer)) return 1;
        ws->theta_host = nullptr;
        if (!zero_copy) {
            if (real_only) {   // the surrogate's gradient is the real part: half the bytes over the bus, no pass over them on the host
                HIP_OK(lb_take(L, ws->d_grads, ws->d_sur_real, nullptr, st));
                HIP_OK(hipMemcpyAsync(pin_gr, ws->d_sur_real, sizeof(double) * nth, hipMemcpyDeviceToHost, st));
            } else {
                HIP_OK(hipMemcpyAsync(pin_gr, ws->d_grads, sizeof(double2) * nth, hipMemcpyDeviceToHost, st));
            }
            HIP_OK(hipMemcpyAsync(hd, dd, bytes, hipMemcpyDeviceToHost, st));
        }
        return 0;
    };
    static const bool graphs_on = env_int("AQC_GRAPH", 1) != 0;
    if (graphs_on && !ws->profile) {   // the launch sequence is replayed as a graph, as in aqc_ws_eval
        const std::vector<long long> key = {1000 + update_state + (zero_copy ? 10 : 0) + (real_only ? 20 : 0), block_from, block_to, front_layer,
                                            (long long)S, (long long)(size_t)ws->d_sur_real,
                                            (long long)(size_t)ws->d_sur, (long long)(size_t)ws->h_sur, (long long)(size_t)ws->h_pin,
                                            (long long)(size_t)ws->d_small, (long long)(size_t)ws->d_combo_prev[AQC_BUF_X2]};
        auto it = ws->graphs.find(key);
        if (it == ws->graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            HIP_OK(hipStreamSynchronize(st));
            HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            ws->capturing = true;
            const int rc = enqueue();
            ws->capturing = false;
            const hipError_t e = hipStreamEndCapture(st, &graph);
            if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return 1; }
            if (e != hipSuccess || !graph) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(ei));
            if (ws->graphs.size() >= 16) drop_graphs(ws);
            it = ws->graphs.emplace(key, exec).first;
        }
        ws->d_thetas = ws->d_thetas_own;   // host-side state that enqueue() would have set
        ws->coef_valid = true;
        ws->fwd.u_valid = false;
        ws->inv.u_valid = ws->sweep.u_valid = ws->inv.v3 && ws->sweep.v3;
        HIP_OK(hipGraphLaunch(it->second, st));
    } else if (enqueue()) {
        return 1;
    }
    ws->combo_valid[AQC_BUF_X2] = true;   // (grad_from does not write its lhs buffer)
    HIP_OK(hipStreamSynchronize(st));
    if (real_only) {
        memcpy(grad_real_out, pin_gr, sizeof(double) * nth);
    } else {
        if (grads_out) memcpy(grads_out, pin_gr, sizeof(double2) * nth);
        if (grad_real_out)
            for (size_t i = 0; i < nth; ++i) grad_real_out[i] = pin_gr[2 * i];
    }
    memcpy(f_out, hd, sizeof(double) * B);
    if (update_state) {
        if (fidelity_out) memcpy(fidelity_out, hd + B, sizeof(double) * B);
        memcpy(weight_io, hd + 2 * (size_t)B, sizeof(double) * B);
        for (int b = 0; b < B; ++b) max_no_io[b] = h_max[b];
    }
    if (hs_out) memcpy(hs_out, hd + 3 * (size_t)B, sizeof(double2) * (size_t)B * S);
    return 0;
}

int aqc_ws_get_grads(aqc_ws* ws, double* grads) {
    if (!ws || !grads) return fail("null argument");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipMemcpyAsync(grads, ws->d_grads, sizeof(double2) * (size_t)ws->batch * ws->ctx->prog.num_thetas(),
                          hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_gather(aqc_ws* ws, int buf, const int64_t* index, int count, double* out) {
    if (check_buf(ws, buf)) return 1;
    if (!index || !out || count < 1) return fail("invalid gather arguments");
    HIP_OK(hipSetDevice(ws->device));
    const int64_t dim = (int64_t)1 << ws->ctx->prog.n;
    std::vector<long long> elem(count);
    for (int i = 0; i < count; ++i) {
        if (index[i] < 0 || index[i] >= dim) return fail("gather index out of range");
        elem[i] = (long long)index[i] << ws->col_bits;
    }
    HIP_OK(hipStreamSynchronize(ws->stream));   // the temporaries may be re-allocated
    if (ensure_tmp(ws, count, (size_t)ws->batch * count)) return 1;
    HIP_OK(hipMemcpyAsync(ws->d_tmp_index, elem.data(), sizeof(long long) * count, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    {
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_gather(ws->bufs[buf], ws->lane_elems, ws->d_tmp_index, count, ws->batch, ws->d_tmp_small, ws->stream));
    }
    HIP_OK(hipMemcpyAsync(out, ws->d_tmp_small, sizeof(double2) * (size_t)ws->batch * count, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_vdot(aqc_ws* ws, int buf_a, int buf_b, double* out) {
    if (check_buf(ws, buf_a) || check_buf(ws, buf_b)) return 1;
    if (!out) return fail("null output");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (ensure_tmp(ws, 0, ws->batch)) return 1;
    {
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_vdot(ws->bufs[buf_a], ws->bufs[buf_b], ws->lane_elems, ws->lane_elems, ws->batch, ws->d_vdot_part,
                           ws->vdot_parts, ws->d_tmp_small, ws->stream));
    }
    HIP_OK(hipMemcpyAsync(out, ws->d_tmp_small, sizeof(double2) * (size_t)ws->batch, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_theta_bank(aqc_ws* ws, const double* thetas, int nsets) {
    if (!ws || !thetas || nsets < 1) return fail("invalid theta bank arguments");
    HIP_OK(hipSetDevice(ws->device));
    const size_t bytes = sizeof(double) * (size_t)nsets * ws->batch * std::max(ws->ctx->prog.num_thetas(), 1);
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (ws->d_theta_bank) HIP_OK(hipFree(ws->d_theta_bank));
    ws->d_theta_bank = nullptr;
    ws->bank_sets = 0;
    ws->d_thetas = ws->d_thetas_own;
    HIP_OK(hipMalloc((void**)&ws->d_theta_bank, bytes));
    HIP_OK(hipMemcpyAsync(ws->d_theta_bank, thetas, bytes, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    ws->bank_sets = nsets;
    return 0;
}

int aqc_ws_use_theta_set(aqc_ws* ws, int set_index) {
    if (!ws) return fail("null workspace");
    if (set_index < 0 || set_index >= ws->bank_sets) return fail("theta set %d out of range (%d loaded)", set_index, ws->bank_sets);
    HIP_OK(hipSetDevice(ws->device));
    const Program& prog = ws->ctx->prog;
    ws->d_thetas = ws->d_theta_bank + (size_t)set_index * ws->batch * prog.num_thetas();
    return run_coef(ws);
}

int aqc_ws_gather_setup(aqc_ws* ws, const int64_t* index, int count) {
    if (!ws || !index || count < 1) return fail("invalid gather arguments");
    HIP_OK(hipSetDevice(ws->device));
    const int64_t dim = (int64_t)1 << ws->ctx->prog.n;
    std::vector<long long> elem(count);
    for (int i = 0; i < count; ++i) {
        if (index[i] < 0 || index[i] >= dim) return fail("gather index out of range");
        elem[i] = (long long)index[i] << ws->col_bits;
    }
    HIP_OK(hipStreamSynchronize(ws->stream));
    drop_graphs(ws);   // captured evaluations hold the old index / staging pointers
    if (ensure_index(ws, count) || ensure_small(ws, (size_t)ws->batch * count)) return 1;
    if (2 * (size_t)ws->batch * count > ws->pin_small) {   // aqc_ws_eval stages the gathered amplitudes in pinned memory
        double* pin = nullptr;
        const size_t want = 2 * (size_t)ws->batch * count;
        HIP_OK(hipHostMalloc((void**)&pin, sizeof(double) * (ws->pin_thetas + ws->pin_grads + want), hipHostMallocDefault));
        if (ws->h_pin) HIP_OK(hipHostFree(ws->h_pin));
        ws->h_pin = pin;
        ws->pin_small = want;
    }
    HIP_OK(hipMemcpyAsync(ws->d_index, elem.data(), sizeof(long long) * count, hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    ws->gather_count = count;
    return 0;
}

int aqc_ws_gather_launch(aqc_ws* ws, int buf) {
    if (check_buf(ws, buf)) return 1;
    if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called");
    HIP_OK(hipSetDevice(ws->device));
    if (results_guard(ws)) return 1;
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_gather(ws->bufs[buf], ws->lane_elems, ws->d_index, ws->gather_count, ws->batch, ws->d_small, ws->stream, ws->mirror_small));
    return 0;
}

int aqc_ws_gather_fetch(aqc_ws* ws, double* out) {
    if (!ws || !out) return fail("null argument");
    if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipMemcpyAsync(out, ws->d_small, sizeof(double2) * (size_t)ws->batch * ws->gather_count, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

static int mps_scratch(aqc_ws* ws, size_t n_cplx);

// ---- dense zgemm with host pointers ---------------------------------------------------------------

int aqc_zgemm(int device, int conj_trans_a, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
              double* C, int ldc) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return fail("invalid zgemm arguments");
    const int a_rows = conj_trans_a ? K : M, a_cols = conj_trans_a ? M : K;
    if (lda < a_cols || ldb < N || ldc < N) return fail("invalid leading dimension");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device out of range");
    HIP_OK(hipSetDevice(device));
    double2 *dA = nullptr, *dB = nullptr, *dC = nullptr;
    const size_t na = (size_t)a_rows * lda, nb = (size_t)K * ldb, nc = (size_t)M * ldc;
    int rc = 0;
    hipError_t e = hipMalloc((void**)&dA, na * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&dB, nb * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&dC, nc * sizeof(double2));
    if (e == hipSuccess) e = hipMemcpy(dA, A, na * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, B, nb * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess && ldc != N) e = hipMemcpy(dC, C, nc * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_zgemm(conj_trans_a != 0, false, M, N, K, dA, lda, dB, ldb, dC, ldc, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(C, dC, nc * sizeof(double2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail("aqc_zgemm failed: %s", hipGetErrorString(e));
    if (dA) (void)hipFree(dA);
    if (dB) (void)hipFree(dB);
    if (dC) (void)hipFree(dC);
    return rc;
}

// ---- gate-level building blocks (one-shot, host pointers) ---------------------------------------

namespace {

struct DevBuf {   // RAII for the one-shot calls
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

int gate_args_ok(int device, int n, int64_t ncols) {
    if (n < 1 || n > 30 || ncols < 1 || ((size_t)ncols << n) > ((size_t)1 << kMaxBits)) return fail("invalid array shape");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: the aqc_hip path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device out of range");
    return 0;
}

}  // namespace

int aqc_gate_1q(int device, int n, int64_t ncols, int qubit, const double* gate, const double* src, double* dst) {
    if (!gate || !src || !dst) return fail("null argument");
    if (gate_args_ok(device, n, ncols)) return 1;
    if (qubit < 0 || qubit >= n) return fail("qubit out of range");
    HIP_OK(hipSetDevice(device));
    const size_t bytes = sizeof(double2) * ((size_t)ncols << n);
    DevBuf d;
    HIP_OK(d.alloc(bytes));
    HIP_OK(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
    HIP_OK(launch_gate1q(d.p, d.p, n, (size_t)ncols, qubit, gate, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(dst, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_gate_2q(int device, int n, int64_t ncols, int ctrl, int targ, const double* gate, const double* src, double* dst) {
    if (!gate || !src || !dst) return fail("null argument");
    if (gate_args_ok(device, n, ncols)) return 1;
    if (n < 2 || ctrl < 0 || ctrl >= n || targ < 0 || targ >= n || ctrl == targ) return fail("invalid qubit pair");
    HIP_OK(hipSetDevice(device));
    const size_t bytes = sizeof(double2) * ((size_t)ncols << n);
    DevBuf d;
    HIP_OK(d.alloc(bytes));
    HIP_OK(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
    HIP_OK(launch_gate2q(d.p, d.p, n, (size_t)ncols, ctrl, targ, gate, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(dst, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_gate_dot(int device, int n, int64_t ncols, int kind, int q0, int q1, const double* w, const double* z, double* out) {
    if (!w || !z || !out) return fail("null argument");
    if (gate_args_ok(device, n, ncols)) return 1;
    if (kind < 0 || kind > 3 || q0 < 0 || q0 >= n) return fail("invalid inner-product kind or qubit");
    if (kind == 3 && (n < 2 || q1 < 0 || q1 >= n || q1 == q0)) return fail("invalid qubit pair");
    HIP_OK(hipSetDevice(device));
    const size_t bytes = sizeof(double2) * ((size_t)ncols << n);
    DevBuf dw, dz, dp;
    HIP_OK(dw.alloc(bytes));
    HIP_OK(dz.alloc(bytes));
    HIP_OK(dp.alloc(sizeof(double2) * (size_t)(gate_dot_parts(n, (size_t)ncols, kind) + 1)));
    HIP_OK(hipMemcpy(dw.p, w, bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dz.p, z, bytes, hipMemcpyHostToDevice));
    double2* parts = static_cast<double2*>(dp.p);
    HIP_OK(launch_gate_dot(dw.p, dz.p, n, (size_t)ncols, kind, q0, q1, parts + 1, parts, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, parts, sizeof(double2), hipMemcpyDeviceToHost));
    return 0;
}

// ---- coordinate descent ------------------------------------------------------------------------

static int cd_checks(const aqc_ws* ws) {
    const Program& prog = ws->ctx->prog;
    if (ws->ncols != (1 << prog.n)) return fail("coordinate descent needs a square workspace (ncols == 2^n)");
    if (prog.entangler == AQC_CP) return fail("CPhase entangler is not supported yet");
    if (prog.trotter) return fail("matrix path does not support the Trotter ansatz");
    return 0;
}

int aqc_ws_cd_fits_one_launch(const aqc_ws* ws) {
    if (!ws) return 0;
    return aqc::cd_persistent_lds_bytes(ws->nbits, ws->ctx->prog.num_thetas()) <= (size_t)160 * 1024 ? 1 : 0;
}

int aqc_ws_cd_sweeps(aqc_ws* ws, double* thetas_io, double* fobj, int nsweeps, int max_steps) {
    if (!ws || !thetas_io || !fobj) return fail("null argument");
    if (nsweeps < 1) return fail("nsweeps must be positive");
    if (cd_checks(ws)) return 1;
    if (!aqc_ws_cd_fits_one_launch(ws))
        return fail("the operands of this coordinate descent (2 x %zu KiB) do not fit one workgroup's LDS: use aqc_ws_cd_sweep (launch chain, one lane)",
                    (ws->lane_elems * sizeof(double2)) >> 10);
    const Program& prog = ws->ctx->prog;
    const int T = prog.num_thetas();
    HIP_OK(hipSetDevice(ws->device));
    if (!ws->d_cd_prog) {   // the walk of core_op_matrix.py:852-912 cut into segments (address bits of this workspace)
        std::vector<aqc::CdSegHost> segs;
        for (const GateGroup& g : prog.groups) {
            aqc::CdSegHost sg{};
            if (g.type == GROUP_FRONT) {   // Rz(t2), Ry(t1), Rz(t0) on one qubit; the second bit of the 4-element groups: any other qubit
                sg.ha = ws->col_bits + g.q0; sg.hb = ws->col_bits + (g.q0 + 1) % prog.n; sg.ent = 0; sg.nrot = 3;
                const int kinds[3] = {1, 0, 1}, tix[3] = {g.theta0 + 2, g.theta0 + 1, g.theta0};
                for (int r = 0; r < 3; ++r) { sg.kind[r] = kinds[r]; sg.on_b[r] = 0; sg.tindex[r] = tix[r]; }
            } else {                       // entangler, Ry(t0) Rz(t1) on the control, Ry(t2) Rs(t3) on the target
                sg.ha = ws->col_bits + g.q0; sg.hb = ws->col_bits + g.q1; sg.ent = prog.entangler == AQC_CX ? 1 : 2; sg.nrot = 4;
                const int kinds[4] = {0, 1, 0, prog.entangler == AQC_CX ? 2 : 1};
                for (int r = 0; r < 4; ++r) { sg.kind[r] = kinds[r]; sg.on_b[r] = r >= 2; sg.tindex[r] = g.theta0 + r; }
            }
            segs.push_back(sg);
        }
        HIP_OK(hipMalloc(&ws->d_cd_prog, segs.size() * sizeof(aqc::CdSegHost)));
        HIP_OK(hipMemcpy(ws->d_cd_prog, segs.data(), segs.size() * sizeof(aqc::CdSegHost), hipMemcpyHostToDevice));
        ws->cd_nsteps = (int)segs.size();
        HIP_OK(hipMalloc((void**)&ws->d_cd_thetas, sizeof(double) * (size_t)ws->batch * T));
    }
    const size_t nf = (size_t)ws->batch * nsweeps;
    if (nf > ws->cd_fobj_cap) {
        if (ws->d_cd_fobj) HIP_OK(hipFree(ws->d_cd_fobj));
        ws->d_cd_fobj = nullptr;
        HIP_OK(hipMalloc((void**)&ws->d_cd_fobj, sizeof(double) * nf));
        ws->cd_fobj_cap = nf;
    }
    HIP_OK(hipMemcpyAsync(ws->d_cd_thetas, thetas_io, sizeof(double) * (size_t)ws->batch * T, hipMemcpyHostToDevice, ws->stream));
    {
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(aqc::launch_cd_persistent(ws->d_cd_prog, ws->cd_nsteps, ws->nbits, ws->col_bits, ws->bufs[AQC_BUF_Y], ws->lane_elems, ws->d_cd_thetas,
                                         T, ws->d_cd_fobj, nsweeps, max_steps, ws->batch, ws->stream));
    }
    HIP_OK(hipMemcpyAsync(thetas_io, ws->d_cd_thetas, sizeof(double) * (size_t)ws->batch * T, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipMemcpyAsync(fobj, ws->d_cd_fobj, sizeof(double) * nf, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_cd_sweep(aqc_ws* ws, double* thetas_io, double* fobj) {
    if (!ws || !thetas_io || !fobj) return fail("null argument");
    if (cd_checks(ws)) return 1;
    const char* chain = getenv("AQC_CD_CHAIN");   // "1": the launch chain below even where one launch would do (cross-check, timing)
    if (aqc_ws_cd_fits_one_launch(ws) && !(chain && chain[0] == '1')) return aqc_ws_cd_sweeps(ws, thetas_io, fobj, 1, -1);
    const Program& prog = ws->ctx->prog;
    const int dim = 1 << prog.n;
    if (ws->batch != 1) return fail("the launch-chain coordinate descent (operands beyond one workgroup's LDS) runs one lane");
    HIP_OK(hipSetDevice(ws->device));
    const int T = prog.num_thetas();
    if (aqc_ws_set_thetas(ws, thetas_io)) return 1;                  // theta_in = d_thetas_own
    if (aqc_ws_apply(ws, 1, AQC_BUF_Y, AQC_BUF_Z)) return 1;          // z = V^H U      (core_op_matrix.py:806-810)
    if (aqc_ws_set_identity(ws, AQC_BUF_X)) return 1;                 // w = I
    double* d_theta_out = nullptr;
    HIP_OK(hipMalloc((void**)&d_theta_out, sizeof(double) * T));
    HIP_OK(hipMemcpyAsync(d_theta_out, ws->d_thetas_own, sizeof(double) * T, hipMemcpyDeviceToDevice, ws->stream));
    double2* w = ws->bufs[AQC_BUF_X];
    double2* z = ws->bufs[AQC_BUF_Z];
    const size_t npairs = ws->lane_elems >> 1, ngroups = ws->lane_elems >> 2;
    const int nparts = cd_num_parts(npairs);
    if (mps_scratch(ws, 2 * (size_t)nparts)) { (void)hipFree(d_theta_out); return 1; }
    double2* part = ws->d_mps_scratch;
    int rc = 0;
    auto step = [&](int qubit, int kind, int tindex) -> int {
        const int hbit = ws->col_bits + qubit;
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_cd_dot(w, z, npairs, hbit, kind, part, ws->stream));
        HIP_OK(launch_cd_update(w, z, npairs, hbit, kind, part, nparts, ws->d_thetas_own, d_theta_out, tindex, (double)dim, ws->stream));
        return 0;
    };
    for (const GateGroup& g : prog.groups) {
        if (g.type == GROUP_FRONT) {
            rc = step(g.q0, 1, g.theta0 + 2) || step(g.q0, 0, g.theta0 + 1) || step(g.q0, 1, g.theta0 + 0);
        } else {
            hipError_t e = launch_cd_entangle(w, z, ngroups, ws->col_bits + g.q0, ws->col_bits + g.q1, prog.entangler, ws->stream);
            if (e != hipSuccess) { rc = fail("cd_entangle launch failed: %s", hipGetErrorString(e)); }
            else rc = step(g.q0, 0, g.theta0) || step(g.q0, 1, g.theta0 + 1) || step(g.q1, 0, g.theta0 + 2) ||
                      step(g.q1, prog.entangler == AQC_CX ? 2 : 1, g.theta0 + 3);
        }
        if (rc) break;
    }
    if (!rc) {
        double prod[2] = {0, 0};
        rc = aqc_ws_vdot(ws, AQC_BUF_X, AQC_BUF_Z, prod);
        if (!rc) {
            const double a = std::hypot(prod[0], prod[1]) / dim;
            *fobj = 1.0 - a * a;
            hipError_t e = hipMemcpyAsync(thetas_io, d_theta_out, sizeof(double) * T, hipMemcpyDeviceToHost, ws->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ws->stream);
            if (e != hipSuccess) rc = fail("theta download failed: %s", hipGetErrorString(e));
        }
    }
    (void)hipStreamSynchronize(ws->stream);
    (void)hipFree(d_theta_out);
    return rc;
}

// ---- MPS helpers ------------------------------------------------------------------------------

static int mps_scratch(aqc_ws* ws, size_t n_cplx) {
    if (n_cplx <= ws->mps_scratch_cap) return 0;
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (ws->d_mps_scratch) HIP_OK(hipFree(ws->d_mps_scratch));
    ws->d_mps_scratch = nullptr;
    ws->mps_scratch_cap = 0;
    HIP_OK(hipMalloc((void**)&ws->d_mps_scratch, n_cplx * sizeof(double2)));
    ws->mps_scratch_cap = n_cplx;
    return 0;
}

static int check_mps_slot(const aqc_ws* ws, int slot, bool need_data) {
    if (!ws) return fail("null workspace");
    if (ws->ncols != 1) return fail("MPS helpers need a state-vector workspace (ncols == 1)");
    if (slot < 0 || slot >= AQC_MPS_SLOTS) return fail("MPS slot %d out of range", slot);
    if (need_data && !ws->mps[slot].d_t) return fail("MPS slot %d is empty", slot);
    return 0;
}

int aqc_ws_mps_upload(aqc_ws* ws, int slot, const int32_t* dims, const double* gammas, const double* lambdas) {
    if (check_mps_slot(ws, slot, false)) return 1;
    const int n = ws->ctx->prog.n;
    if (n > 64) return fail("MPS helpers of the workspace serve up to 64 qubits");
    if (!dims || !gammas || (n > 1 && !lambdas)) return fail("null MPS argument");
    if (dims[0] != 1 || dims[n] != 1) return fail("MPS boundary bond dimensions must be 1");
    HIP_OK(hipSetDevice(ws->device));
    aqc_ws::MpsSlot& m = ws->mps[slot];
    m.dims.assign(dims, dims + n + 1);
    m.offset.assign(n + 1, 0);
    MpsSites sites;
    memset(&sites, 0, sizeof sites);
    sites.n = n;
    size_t total = 0, lam_total = 0;
    for (int q = 0; q < n; ++q) {
        if (dims[q] < 1 || dims[q + 1] < 1) return fail("MPS bond dimensions must be positive");
        m.offset[q] = total;
        sites.offset[q] = total;
        sites.cols[q] = dims[q + 1];
        sites.lam_offset[q] = (int)lam_total;
        total += (size_t)2 * dims[q] * dims[q + 1];
        if (q < n - 1) lam_total += dims[q + 1];
    }
    m.offset[n] = total;
    sites.offset[n] = total;
    sites.total = total;
    if (total > m.cap || lam_total > ws->mps_lam_cap) {   // grow-only device buffers
        HIP_OK(hipStreamSynchronize(ws->stream));
        if (total > m.cap) {
            if (m.d_t) HIP_OK(hipFree(m.d_t));
            m.d_t = nullptr; m.cap = 0;
            HIP_OK(hipMalloc((void**)&m.d_t, total * sizeof(double2)));
            m.cap = total;
        }
        if (lam_total > ws->mps_lam_cap) {
            if (ws->d_mps_lam) HIP_OK(hipFree(ws->d_mps_lam));
            ws->d_mps_lam = nullptr; ws->mps_lam_cap = 0;
            HIP_OK(hipMalloc((void**)&ws->d_mps_lam, std::max<size_t>(lam_total, 1) * sizeof(double)));
            ws->mps_lam_cap = std::max<size_t>(lam_total, 1);
        }
    }
    HIP_OK(hipMemcpyAsync(m.d_t, gammas, total * sizeof(double2), hipMemcpyHostToDevice, ws->stream));
    if (lam_total) {
        HIP_OK(hipMemcpyAsync(ws->d_mps_lam, lambdas, lam_total * sizeof(double), hipMemcpyHostToDevice, ws->stream));
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_mps_scale_all(m.d_t, ws->d_mps_lam, sites, ws->stream));   // _preprocess_mps: lambda on the right bond
    }
    HIP_OK(hipStreamSynchronize(ws->stream));   // the host arrays (and the shared lambda staging) may be reused right away
    return 0;
}

// MPS -> dense state (mps_to_vector, mps_operations.py:159-189; index bit q <-> site q), contracted from both ends:
//   L[lo][chi]   = sites 0 .. h-1       (rows grow by appending the next site as the next HIGHER bit)
//   Rt[chi][i]   = sites n-1 .. h       (columns grow likewise, so i is the BIT-REVERSED high part of the index)
//   G = L Rt, then out[(rev(i) << h) + lo] = G[lo][i].
// O(2^(n/2) chi^2 + 2^n chi) flops instead of the O(2^n chi^2) of a one-sided sweep; both values of the site's bit go
// through one batched GEMM launch.
int aqc_ws_mps_to_vec(aqc_ws* ws, int slot, int buf, int lane) {
    if (check_mps_slot(ws, slot, true) || check_buf(ws, buf)) return 1;
    if (lane < 0 || lane >= ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(ws->device));
    const int n = ws->ctx->prog.n;
    if (n >= 2) {   // one (slot, lane) pair of the batched chain below
        const int32_t s1 = slot, l1 = lane;
        return aqc_ws_mps_to_vec_batch(ws, 1, &s1, buf, &l1);
    }
    // n == 1: the state is the site tensor itself, [b][1][1]
    ws->combo_valid[buf] = false;
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(hipMemcpyAsync(ws->bufs[buf] + (size_t)lane * ws->lane_elems, ws->mps[slot].d_t, sizeof(double2) * 2, hipMemcpyDeviceToDevice, ws->stream));
    return 0;
}

// MPS -> dense (mps_operations.py:159-189) for `count` (slot, lane) pairs at once: MPS slots[i] -> lane lanes[i] of `buf`.
// When all the slots have the same bond dimensions (the lanes of a batched objective) every step of the chain is ONE launch
// for all lanes (zgemm over device pointer tables): (n/2 - 1) + (n - n/2 - 1) + 1 launches whatever the number of lanes, ONE
// for product states; otherwise the pairs are served one after the other.
static int mps_to_vec_batch_uniform(aqc_ws* ws, int count, const int32_t* slots, int buf, const int32_t* lanes);

int aqc_ws_mps_to_vec_batch(aqc_ws* ws, int count, const int32_t* slots, int buf, const int32_t* lanes) {
    if (!ws || !slots || !lanes || count < 1) return fail("invalid batched MPS arguments");
    if (check_buf(ws, buf)) return 1;
    for (int i = 0; i < count; ++i) {
        if (check_mps_slot(ws, slots[i], true)) return 1;
        if (lanes[i] < 0 || lanes[i] >= ws->batch) return fail("lane out of range");
    }
    if (ws->ctx->prog.n / 2 == 0) {
        for (int i = 0; i < count; ++i)
            if (aqc_ws_mps_to_vec(ws, slots[i], buf, lanes[i])) return 1;
        return 0;
    }
    // lanes whose operands share their bond dimensions share every launch of the contraction chain: one chain per distinct
    // dimension vector (truncated canonical tensors -- the reference's trunc_thr = 1e-6 -- differ from target to target by a
    // few bond entries; taking every such lane through a chain of its own made a 64-lane step 14x slower than equal bonds)
    std::vector<int> group(count, -1);
    int ngroups = 0;
    for (int i = 0; i < count; ++i) {
        if (group[i] >= 0) continue;
        group[i] = ngroups;
        for (int j = i + 1; j < count; ++j)
            if (group[j] < 0 && ws->mps[slots[j]].dims == ws->mps[slots[i]].dims) group[j] = ngroups;
        ++ngroups;
    }
    if (ngroups == 1) return mps_to_vec_batch_uniform(ws, count, slots, buf, lanes);
    std::vector<int32_t> gs, gl;
    for (int g = 0; g < ngroups; ++g) {
        gs.clear(); gl.clear();
        for (int i = 0; i < count; ++i)
            if (group[i] == g) { gs.push_back(slots[i]); gl.push_back(lanes[i]); }
        if (mps_to_vec_batch_uniform(ws, (int)gs.size(), gs.data(), buf, gl.data())) return 1;
    }
    return 0;
}

static int mps_to_vec_batch_uniform(aqc_ws* ws, int count, const int32_t* slots, int buf, const int32_t* lanes) {
    const int n = ws->ctx->prog.n;
    const int h = n / 2, mh = n - h;
    HIP_OK(hipSetDevice(ws->device));
    const std::vector<int>& dims = ws->mps[slots[0]].dims;
    const std::vector<size_t>& off = ws->mps[slots[0]].offset;
    ws->combo_valid[buf] = false;
    std::vector<const void*> tabs;
    auto table = [&](auto fn) { const size_t at = tabs.size(); for (int i = 0; i < count; ++i) tabs.push_back(fn(i)); return at; };
    auto out_lane = [&](int i) { return (const void*)(ws->bufs[buf] + (size_t)lanes[i] * ws->lane_elems); };
    const void* const* T = nullptr;
    auto upload_tables = [&]() -> int {   // pointer tables of every launch of the chain: a resident set, or one copy
        aqc_ws::MpsTabs* hit = nullptr;
        aqc_ws::MpsTabs* lru = &ws->mps_tabs[0];
        for (auto& t : ws->mps_tabs) {
            if (t.dev && t.host == tabs) hit = &t;
            if (t.tick < lru->tick) lru = &t;
        }
        if (!hit) {
            HIP_OK(hipStreamSynchronize(ws->stream));   // launches in flight may still read the set that is recycled
            if (tabs.size() > lru->cap) {
                if (lru->dev) HIP_OK(hipFree(lru->dev));
                lru->dev = nullptr; lru->cap = 0; lru->host.clear();
                HIP_OK(hipMalloc((void**)&lru->dev, tabs.size() * sizeof(void*)));
                lru->cap = tabs.size();
            }
            lru->host = tabs;   // (stays alive next to the device copy: nothing to wait for after the upload)
            HIP_OK(hipMemcpyAsync(lru->dev, lru->host.data(), tabs.size() * sizeof(void*), hipMemcpyHostToDevice, ws->stream));
            hit = lru;
        }
        hit->tick = ++ws->mps_tabs_tick;
        T = hit->dev;
        return 0;
    };
    bool product = true;
    for (int q = 0; q <= n; ++q) product = product && dims[q] == 1;
    if (product) {   // product states (|0>, the Neel state, ...: the usual lhs operand): one launch, no chain
        const size_t ta = table([&](int i) { return (const void*)ws->mps[slots[i]].d_t; });
        const size_t tc = table(out_lane);
        if (upload_tables()) return 1;
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_mps_product(T + ta, (void* const*)(T + tc), n, count, ws->stream));
        return 0;
    }
    // Left half L[l][chi] (sites 0 .. h-1, site 0 the lowest bit of l), right half transposed Rt[r][chi] (sites n-1 .. h, every
    // new site becoming the LOWEST bit of r, so that site h ends up there), and out[r 2^h + l] = sum_chi Rt[r][chi] L[l][chi]
    // written by the last product straight into the lane's buffer in the workspace's bit order (bit q = site q): no
    // scratch copy of the dense state, no permutation pass.
    size_t need_l = 2, need_r = 2;
    for (int q = 0; q < h; ++q) need_l = std::max(need_l, ((size_t)2 << q) * dims[q + 1]);
    for (int q = n - 1; q >= h; --q) need_r = std::max(need_r, ((size_t)2 << (n - 1 - q)) * dims[q]);
    const size_t per = 2 * need_l + 2 * need_r;
    if (mps_scratch(ws, per * (size_t)count)) return 1;
    struct Step { int kind, q; size_t a, b, c; };   // offsets (in pointers) of the three tables inside the upload
    std::vector<Step> steps;
    auto lb = [&](int i, int k) { return (const void*)(ws->d_mps_scratch + per * (size_t)i + need_l * (size_t)k); };
    auto rb = [&](int i, int k) { return (const void*)(ws->d_mps_scratch + per * (size_t)i + 2 * need_l + need_r * (size_t)k); };
    auto site = [&](int i, int q) { return (const void*)(ws->mps[slots[i]].d_t + off[q]); };
    for (int q = 1; q < h; ++q) {       // left part: L_q = L_{q-1} T_q, both values of the site's bit (inner = 2)
        Step st{0, q, 0, 0, 0};
        st.a = q == 1 ? table([&](int i) { return site(i, 0); }) : table([&](int i) { return lb(i, (q - 1) & 1); });
        st.b = table([&](int i) { return site(i, q); });
        st.c = table([&](int i) { return lb(i, q & 1); });
        steps.push_back(st);
    }
    // right part: Rt_0 = the last site as it is stored ([2][chi][1] = [r][chi]); Rt_j[2 c + b] = Rt_{j-1}[c] T_q[b]^T
    for (int q = n - 2; q >= h; --q) {
        const int j = n - 1 - q;
        Step st{2, q, 0, 0, 0};
        st.a = j == 1 ? table([&](int i) { return site(i, n - 1); }) : table([&](int i) { return rb(i, (j - 1) & 1); });
        st.b = table([&](int i) { return site(i, q); });
        st.c = table([&](int i) { return rb(i, j & 1); });
        steps.push_back(st);
    }
    Step fin{3, 0, 0, 0, 0};
    fin.a = mh == 1 ? table([&](int i) { return site(i, n - 1); }) : table([&](int i) { return rb(i, (mh - 1) & 1); });
    fin.b = h == 1 ? table([&](int i) { return site(i, 0); }) : table([&](int i) { return lb(i, (h - 1) & 1); });
    fin.c = table(out_lane);
    steps.push_back(fin);
    if (upload_tables()) return 1;
    ProfScope ps(ws, AQC_K_MISC);
    // the two halves are independent chains of small launches (latency-bound at small bond dimensions): the right half runs
    // on a second stream, forked after everything queued so far and joined before the last product
    const bool fork = !ws->profile && !ws->capturing && h > 1 && mh > 1;
    if (fork) {
        if (!ws->mps_stream) {
            HIP_OK(hipStreamCreateWithFlags(&ws->mps_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&ws->ev_mps_fork, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&ws->ev_mps_join, hipEventDisableTiming));
        }
        HIP_OK(hipEventRecord(ws->ev_mps_fork, ws->stream));
        HIP_OK(hipStreamWaitEvent(ws->mps_stream, ws->ev_mps_fork, 0));
    }
    for (const Step& st : steps) {
        hipStream_t sst = (fork && st.kind == 2) ? ws->mps_stream : ws->stream;
        if (fork && st.kind == 3) {
            HIP_OK(hipEventRecord(ws->ev_mps_join, ws->mps_stream));
            HIP_OK(hipStreamWaitEvent(ws->stream, ws->ev_mps_join, 0));
        }
        if (st.kind == 0) {
            const int q = st.q, rows = 1 << q, kk = dims[q], nn = dims[q + 1];
            HIP_OK(launch_zgemm_tables(rows, nn, kk, T + st.a, kk, T + st.b, nn, (void* const*)(T + st.c), nn, 0, (size_t)kk * nn, (size_t)rows * nn,
                                       count, 2, sst));
        } else if (st.kind == 2) {   // C rows 2 c + b: ldc = 2 chil, the bit's block starts chil further; B = T_q[b] stored [chil][chir], used transposed
            const int q = st.q, j = n - 1 - q, cols = 1 << j, chil = dims[q], chir = dims[q + 1];
            HIP_OK(launch_zgemm_tables(cols, chil, chir, T + st.a, chir, T + st.b, chir, (void* const*)(T + st.c), 2 * chil, 0, (size_t)chil * chir,
                                       (size_t)chil, count, 2, sst, 1));
        } else {                     // out [2^mh][2^h] = Rt [2^mh][chi] . L^T, L stored [2^h][chi]
            const int chi = dims[h];
            HIP_OK(launch_zgemm_tables(1 << mh, 1 << h, chi, T + st.a, chi, T + st.b, chi, (void* const*)(T + st.c), 1 << h, 0, 0, 0, count, 1,
                                       sst, 1));
        }
    }
    return 0;
}

int aqc_ws_mps_dot(aqc_ws* ws, int slot_a, int slot_b, double* out) {
    if (check_mps_slot(ws, slot_a, true) || check_mps_slot(ws, slot_b, true)) return 1;
    if (!out) return fail("null output");
    HIP_OK(hipSetDevice(ws->device));
    const aqc_ws::MpsSlot& a = ws->mps[slot_a];
    const aqc_ws::MpsSlot& b = ws->mps[slot_b];
    const int n = ws->ctx->prog.n;
    size_t need = 1;
    for (int q = 0; q <= n; ++q) need = std::max(need, (size_t)a.dims[q] * b.dims[q]);
    for (int q = 0; q < n; ++q) need = std::max(need, (size_t)a.dims[q] * b.dims[q + 1]);
    if (mps_scratch(ws, 3 * need)) return 1;
    double2* e0 = ws->d_mps_scratch;
    double2* e1 = e0 + need;
    double2* t = e1 + need;
    {   // E[x][y] = sum_b conj(A0[b][x]) B0[b][y]
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_zgemm(true, false, a.dims[1], b.dims[1], 2, a.d_t, a.dims[1], b.d_t, b.dims[1], e0, b.dims[1], ws->stream));
    }
    double2* e = e0;
    double2* en = e1;
    for (int q = 1; q < n; ++q) {
        const int xa = a.dims[q], ua = a.dims[q + 1], yb = b.dims[q], vb = b.dims[q + 1];
        for (int bit = 0; bit < 2; ++bit) {
            ProfScope ps(ws, AQC_K_MISC);
            // T = E B_q[bit]            (xa x vb)
            HIP_OK(launch_zgemm(false, false, xa, vb, yb, e, yb, b.d_t + b.offset[q] + (size_t)bit * yb * vb, vb, t, vb, ws->stream));
            // E' (+)= A_q[bit]^H T      (ua x vb)
            HIP_OK(launch_zgemm(true, bit == 1, ua, vb, xa, a.d_t + a.offset[q] + (size_t)bit * xa * ua, ua, t, vb, en, vb, ws->stream));
        }
        std::swap(e, en);
    }
    HIP_OK(hipMemcpyAsync(out, e, sizeof(double2), hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_vdot_launch(aqc_ws* ws, int buf_a, int buf_b) {
    if (check_buf(ws, buf_a) || check_buf(ws, buf_b)) return 1;
    HIP_OK(hipSetDevice(ws->device));
    if (!ws->d_vdot_out) HIP_OK(hipMalloc((void**)&ws->d_vdot_out, sizeof(double2) * ws->batch));
    if (results_guard(ws)) return 1;
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_vdot(ws->bufs[buf_a], ws->bufs[buf_b], ws->lane_elems, ws->lane_elems, ws->batch, ws->d_vdot_part,
                       ws->vdot_parts, ws->d_vdot_out, ws->stream));
    return 0;
}

int aqc_ws_vdot_fetch(aqc_ws* ws, double* out) {
    if (!ws || !out) return fail("null argument");
    if (!ws->d_vdot_out) return fail("aqc_ws_vdot_launch has not been called");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipMemcpyAsync(out, ws->d_vdot_out, sizeof(double2) * (size_t)ws->batch, hipMemcpyDeviceToHost, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

// Results of the evaluation just enqueued -> pinned host memory, asynchronously on the workspace's stream (what an optimizer
// on the host consumes every evaluation: gradients, gathered amplitudes, <A|B>); aqc_ws_results_fetch waits and hands them out.
// The copies run on a second stream: they wait for the producing kernels (event) and overlap with the kernels of the NEXT
// evaluation; the kernels that overwrite the results (gather, <A|B>, gradient walk) wait for the copies in turn (results_guard).
static int results_guard(aqc_ws* ws) {
    if (ws->copy_pending && !ws->capturing) {
        HIP_OK(hipStreamWaitEvent(ws->stream, ws->ev_copied, 0));
        ws->copy_pending = false;
    }
    return 0;
}

int aqc_ws_results_async(aqc_ws* ws) {
    if (!ws) return fail("null workspace");
    HIP_OK(hipSetDevice(ws->device));
    if (!ws->copy_stream) {
        HIP_OK(hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&ws->ev_ready, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&ws->ev_copied, hipEventDisableTiming));
    }
    const size_t nth = (size_t)ws->batch * ws->ctx->prog.num_thetas();
    double* pin_gr = ws->h_pin + ws->pin_thetas;
    double* pin_sm = pin_gr + ws->pin_grads;
    HIP_OK(hipEventRecord(ws->ev_ready, ws->stream));
    HIP_OK(hipStreamWaitEvent(ws->copy_stream, ws->ev_ready, 0));
    HIP_OK(hipMemcpyAsync(pin_gr, ws->d_grads, sizeof(double2) * nth, hipMemcpyDeviceToHost, ws->copy_stream));
    if (ws->gather_count > 0 && 2 * (size_t)ws->batch * ws->gather_count <= ws->pin_small)
        HIP_OK(hipMemcpyAsync(pin_sm, ws->d_small, sizeof(double2) * (size_t)ws->batch * ws->gather_count, hipMemcpyDeviceToHost, ws->copy_stream));
    else if (ws->d_vdot_out && 2 * (size_t)ws->batch <= ws->pin_small)
        HIP_OK(hipMemcpyAsync(pin_sm, ws->d_vdot_out, sizeof(double2) * (size_t)ws->batch, hipMemcpyDeviceToHost, ws->copy_stream));
    HIP_OK(hipEventRecord(ws->ev_copied, ws->copy_stream));
    ws->copy_pending = true;
    return 0;
}

int aqc_ws_results_fetch(aqc_ws* ws, double* small_out, double* grads_out) {
    if (!ws) return fail("null workspace");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (ws->copy_stream) HIP_OK(hipStreamSynchronize(ws->copy_stream));
    ws->copy_pending = false;
    const size_t nth = (size_t)ws->batch * ws->ctx->prog.num_thetas();
    const double* pin_gr = ws->h_pin + ws->pin_thetas;
    const double* pin_sm = pin_gr + ws->pin_grads;
    if (grads_out) memcpy(grads_out, pin_gr, sizeof(double2) * nth);
    if (small_out) {
        const size_t count = ws->gather_count > 0 ? (size_t)ws->gather_count : 1;
        if (2 * (size_t)ws->batch * count > ws->pin_small) return fail("gathered amplitudes do not fit the staging buffer");
        memcpy(small_out, pin_sm, sizeof(double2) * (size_t)ws->batch * count);
    }
    return 0;
}

int aqc_ws_sync(aqc_ws* ws) {
    if (!ws) return fail("null workspace");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipStreamSynchronize(ws->stream));
    return 0;
}

int aqc_ws_timer_start(aqc_ws* ws) {
    if (!ws) return fail("null workspace");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipEventRecord(ws->ev0, ws->stream));
    return 0;
}

int aqc_ws_timer_stop(aqc_ws* ws, float* elapsed_ms) {
    if (!ws || !elapsed_ms) return fail("null argument");
    HIP_OK(hipSetDevice(ws->device));
    HIP_OK(hipEventRecord(ws->ev1, ws->stream));
    HIP_OK(hipEventSynchronize(ws->ev1));
    HIP_OK(hipEventElapsedTime(elapsed_ms, ws->ev0, ws->ev1));
    return 0;
}

int aqc_ws_profile_enable(aqc_ws* ws, int on) {
    if (!ws) return fail("null workspace");
    ws->profile = on != 0;
    return 0;
}

int aqc_ws_profile_get(aqc_ws* ws, int kind, int64_t* launches, double* total_ms) {
    if (!ws || kind < 0 || kind >= AQC_NUM_KINDS) return fail("invalid argument");
    if (launches) *launches = ws->prof_count[kind];
    if (total_ms) *total_ms = ws->prof_ms[kind];
    return 0;
}

int aqc_ws_profile_reset(aqc_ws* ws) {
    if (!ws) return fail("null workspace");
    for (int i = 0; i < AQC_NUM_KINDS; ++i) { ws->prof_count[i] = 0; ws->prof_ms[i] = 0.0; }
    return 0;
}

int aqc_ws_plan_info(aqc_ws* ws, int which, int* num_stages, int* tile_bits, int* num_tiles) {
    if (!ws) return fail("null workspace");
    const DevPlan& p = which == 0 ? ws->inv : (which == 1 ? ws->sweep : ws->fwd);
    if (num_stages) *num_stages = (int)p.h_stages.size();
    if (tile_bits) *tile_bits = p.k;
    if (num_tiles) *num_tiles = p.ntiles;
    return 0;
}

int aqc_ws_plan_substages(aqc_ws* ws, int which) {
    if (!ws) return -1;
    const DevPlan& p = which == 0 ? ws->inv : (which == 1 ? ws->sweep : ws->fwd);
    return p.v3 ? (int)p.h_subs3.size() : (int)p.h_subs.size();
}

int aqc_ws_kernel_family(aqc_ws* ws, int which) {
    if (!ws) return -1;
    const DevPlan& p = which == 0 ? ws->inv : (which == 1 ? ws->sweep : ws->fwd);
    return p.family();
}

// ---- one-shot host-pointer entry points -------------------------------------------------------

static int oneshot_ws(aqc_ctx* ctx, int ncols, aqc_ws** out) {
    if (!ctx) return fail("null context");
    auto it = ctx->oneshot.find(ncols);
    if (it != ctx->oneshot.end()) { *out = it->second; return 0; }
    aqc_ws* ws = nullptr;
    if (aqc_ws_create(ctx, env_int("AQC_DEVICE", 0), 1, ncols, 0, 0, &ws)) return 1;
    ctx->oneshot[ncols] = ws;
    *out = ws;
    return 0;
}

static int oneshot_apply(aqc_ctx* ctx, const double* thetas, const double* src, double* dst, int ncols, int inverse) {
    if (!thetas || !src || !dst) return fail("null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    aqc_ws* ws = nullptr;
    if (oneshot_ws(ctx, ncols, &ws)) return 1;
    if (aqc_ws_set_thetas(ws, thetas)) return 1;
    if (aqc_ws_upload(ws, AQC_BUF_Y, src)) return 1;
    if (aqc_ws_apply(ws, inverse, AQC_BUF_Y, AQC_BUF_Z)) return 1;
    return aqc_ws_download(ws, AQC_BUF_Z, dst);
}

int aqc_v_mul_vec(aqc_ctx* ctx, const double* thetas, const double* vec, double* out) {
    if (!ctx) return fail("null context");
    return oneshot_apply(ctx, thetas, vec, out, 1, 0);
}
int aqc_vdag_mul_vec(aqc_ctx* ctx, const double* thetas, const double* vec, double* out) {
    if (!ctx) return fail("null context");
    return oneshot_apply(ctx, thetas, vec, out, 1, 1);
}
int aqc_v_mul_mat(aqc_ctx* ctx, const double* thetas, double* mat, int ncols) {
    if (!ctx) return fail("null context");
    if (ctx->prog.trotter) return fail("matrix path does not support the Trotter ansatz (core_op_matrix.py:480)");
    return oneshot_apply(ctx, thetas, mat, mat, ncols, 0);
}
int aqc_vdag_mul_mat(aqc_ctx* ctx, const double* thetas, double* mat, int ncols) {
    if (!ctx) return fail("null context");
    if (ctx->prog.trotter) return fail("matrix path does not support the Trotter ansatz (core_op_matrix.py:562)");
    return oneshot_apply(ctx, thetas, mat, mat, ncols, 1);
}

static int oneshot_grad(aqc_ctx* ctx, const double* thetas, const double* x, const double* vh_y, int ncols, int from, int to,
                        int front, double* grad) {
    if (!thetas || !x || !vh_y || !grad) return fail("null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    aqc_ws* ws = nullptr;
    if (oneshot_ws(ctx, ncols, &ws)) return 1;
    if (aqc_ws_set_thetas(ws, thetas)) return 1;
    if (aqc_ws_upload(ws, AQC_BUF_X, x)) return 1;
    if (aqc_ws_upload(ws, AQC_BUF_Z, vh_y)) return 1;
    if (aqc_ws_grad(ws, from, to, front)) return 1;
    return aqc_ws_get_grads(ws, grad);
}

int aqc_grad_dot_vec(aqc_ctx* ctx, const double* thetas, const double* x, const double* vh_y, int block_from, int block_to,
                     int front_layer, double* grad) {
    if (!ctx) return fail("null context");
    return oneshot_grad(ctx, thetas, x, vh_y, 1, block_from, block_to, front_layer, grad);
}
int aqc_grad_dot_mat(aqc_ctx* ctx, const double* thetas, const double* x_mat, const double* vh_y_mat, int ncols, double* grad) {
    if (!ctx) return fail("null context");
    if (ctx->prog.trotter) return fail("matrix path does not support the Trotter ansatz (core_op_matrix.py:645)");
    return oneshot_grad(ctx, thetas, x_mat, vh_y_mat, ncols, -1, -1, 1, grad);
}

}  // extern "C"
