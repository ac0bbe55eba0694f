// Internal declarations shared by the translation units behind the C ABI (include/aqc_hip.h):
//   aqc_api.cpp        contexts, workspaces, buffers, thetas, small results, one-shot entry points
//   aqc_ws_plan.cpp    lowering of stage plans to device tables (micro-ops, sub-stage slot tables), mirrored V^H plans
//   aqc_ws_sweep.cpp   V / V^H launches, the w/z sweep (dense and sparse-lhs routes), aqc_ws_eval
//   aqc_ws_optim.cpp   device-resident L-BFGS and the one-call surrogate evaluation
//   aqc_ws_extra.cpp   zgemm, gate-level building blocks, coordinate descent, MPS helpers
#pragma once
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

namespace aqc {

int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));   // sets the thread's error message; returns 1

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ::aqc::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

inline int ceil_log2(int v) {
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

// Projected route of the sparse-lhs sweep (aqc_ws_project.cpp): the dense stages run on a virtual register
struct ProjRoute {
    bool ok = false;
    int t = 0, cb = 0, us = 0, nv = 0, nvp = 0, kv = 0, ntiles_v = 0;
    int first_subs = 0;            // sub-stages of the real sweep plan that still run on the real register (its first stage)
    Program vprog;                 // the gate groups of the dense stages on virtual qubits (group indices, thetas, slots: the real ones)
    std::vector<int> rest;         // their indices, in execution order
    DevPlan vsw;                   // sweep plan of the virtual register
    DevPlan vinv;                  // ... walked backwards (the objective's V^H by projection: Y_0 = (later stages)^H proj(y))
    double2* vm = nullptr;         // [batch][2][2^nvp]: the virtual lhs pattern M_0 ...
    double2* vy = nullptr;         // ... the virtual z ...
    double2* vme = nullptr;        // ... and M after the later stages' gates (objective by projection)
    double2* cpart = nullptr;      // partial tile products of the fused pass when its walk is split (few lanes)
    int cpart_shares = 0;
    int vy_copies = 1;             // vy holds this many copies of the virtual z (fused pass over more than 256 summed values: partial sums)
    unsigned l0_mask = 0;          // address bits local to the first stage
    unsigned* d_tab = nullptr;     // off_t | off_usblk | off_cb
    std::vector<unsigned> h_tab;
    unsigned ff_mask = 0, cb_mask = 0, tf_mask = 0;
    TileItem* d_items = nullptr;   // [2 batch ntiles_v]
    int* d_count = nullptr;
    int* d_lane_parts = nullptr;   // [batch]
    int init_buf = -1;             // vm holds the pattern M_0 (and d_items the list) of this lhs buffer ...
    unsigned long long init_version = 0;   // ... at this version of its support (single virtual stage: nothing overwrites vm)
};

// aqc_ws_plan.cpp
void lower_plan(const Program& prog, const Plan& plan, DevPlan& out, int reg_bits, bool with_dots, bool mfma = false, bool presplit = false, int beam_width = 64);
int upload_plan(DevPlan& p);
Plan mirror_plan(const Plan& plan);   // the same stages and sub-stages walked backwards: the plan of V^H whose intermediate states are the sweep's

}  // namespace aqc

// the two handle types of the C ABI live in the global namespace (include/aqc_hip.h)
using aqc::DevPlan; using aqc::Program; using aqc::UJob;

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
    std::vector<long long> combo_last_elem[AQC_NUM_BUFS];   // host copy of the pattern (positions, coefficients) the HOST wrote last; device-side
    std::vector<double> combo_last_coef[AQC_NUM_BUFS];      // writers (lb_prepare) clear it
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
    UJob* d_ujobs = nullptr;          // family 3: [V^H subs | sweep subs | V subs]; ujobs_mirror: no V^H jobs, the sweep's write both operand sets
    bool ujobs_mirror = false;
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
    // mirrored V^H plan, its checkpoint, the sparse-lhs sweep (aqc_ws_sweep.cpp)
    bool inv_mirrored = false;     // inv = the sweep plan walked backwards: V^H into Z leaves the state before its last stage in ZW ...
    bool ckpt_valid = false;       // ... and ZW holds it for the thetas in use and the present contents of Z
    bool w_clean = true;           // W is zero outside the tiles named in d_sw_prev_tiles
    bool sparse_enabled = true;    // AQC_SPARSE_SWEEP=0: always the dense route
    bool r_only_enabled = true;    // AQC_R_ONLY_LAST=0: the sweep's very last sub-stage runs its U products like every other
    int r_only_max_subs = 12;      // ... and so it does when the last stage has more sub-stages than this (AQC_R_ONLY_MAX_SUBS)
    bool skipw_enabled = false;    // AQC_SKIP_ZERO_W=1: skip zero groups / K-steps of w inside a stage (exact; measured slower, off by default)
    long sparse_min_items = 512;   // the sparse route pays from this many (tile, lane) items per stage launch (AQC_SPARSE_MIN_ITEMS)
    unsigned long long supp_version[AQC_NUM_BUFS] = {0, 0, 0, 0, 0, 0};   // bumped whenever d_combo_prev[buf] (the support of a sparse lhs) changes
    aqc::TileItem* d_sw_items = nullptr;    // first-stage items of the sparse sweep [2 batch], and the tiles to clear in W
    aqc::TileItem* d_sw_clear = nullptr;
    int* d_sw_counts = nullptr;             // [0] items, [1] tiles to clear
    int* d_sw_lane_parts = nullptr;         // items (= partial-R slots in use) per lane
    int* d_sw_prev_tiles = nullptr;         // [batch][2] tiles of W written by the list in use
    int sw_lists_built = 0;                 // bit 0: the sweep's list has been built at least once, bit 1: V^H's
    int sw_items_buf = -1;                  // the list in d_sw_items belongs to this lhs buffer ...
    unsigned long long sw_items_version = 0;   // ... at this version of its support
    double2* w2 = nullptr;                  // second scratch pair of the sparse route (plans of >= 3 stages)
    double2* zw2 = nullptr;
    // "objective" V^H: the last stage of the mirrored V^H runs only over the tiles its readers touch -- the registered gather
    // indices and the support of the lhs state -- and Z is completed on demand (ensure_z_full) while the checkpoint is valid
    bool lazy_z_enabled = true;             // AQC_LAZY_Z=0: V^H always writes all of Z
    bool z_full = true;                     // Z holds V^H y everywhere (false: on the tiles of d_vd_items only)
    aqc::TileItem* d_vd_items = nullptr;    // [batch][2 + gather_count]
    size_t vd_items_cap = 0;
    unsigned long long gather_gen = 0;      // bumped by aqc_ws_gather_setup
    unsigned long long vd_key[3] = {~0ull, ~0ull, ~0ull};   // what d_vd_items was built from: lhs buffer (or -1: gather set alone), its support version, gather_gen
    unsigned long long z_gather_gen = 0;    // what the tiles of a partial Z were chosen for: the gather set ...
    int z_x_buf = -1;                       // ... and the support of this lhs buffer at this version
    unsigned long long z_x_version = 0;
    aqc::ProjRoute proj;                    // dense stages of the sparse route on a virtual register (AQC_PROJECTED=0: off)
    bool proj_vdag_enabled = true;          // AQC_PROJECTED_VDAG=0: V^H of a one-call evaluation always by its stages
    long long proj_vdag_min_elems = 1ll << 24;   // ... and from this many amplitudes per batch (fewer: its extra launches cost more than V^H's stages; AQC_PROJECTED_VDAG_MIN_ELEMS)
    bool proj_fused_enabled = true;         // AQC_PROJECTED_FUSED=0: its two products as two launches (two fetches of the target)
    bool proj_y0_ready = false;             // the virtual z (proj.vy) holds Y_0 for the sweep that follows in the same call (run_vdag_projected)
    bool z_from_y = false;                  // a partial Z without a checkpoint: completed by a full V^H from Y (thetas and Y unchanged since)
    std::vector<long long> h_gather;        // host copy of the registered gather indices (elements)
    unsigned long long projb_key[3] = {~0ull, ~0ull, ~0ull};   // (lhs buffer, its support version, gather_gen) the verdict below was taken for
    bool projb_ok = false;
    bool profile = false;
    int64_t prof_count[AQC_NUM_KINDS] = {};
    double prof_ms[AQC_NUM_KINDS] = {};
    std::vector<std::pair<int, float>> prof_log;   // (kind, ms) of every profiled launch, in order (bounded)
};

namespace aqc {

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
            if (ws->prof_log.size() < 100000) ws->prof_log.emplace_back(kind, ms);
        }
    }
};

// aqc_api.cpp
int check_buf(const aqc_ws* ws, int buf);
int ensure_small(aqc_ws* ws, size_t n_cplx);
int ensure_tmp(aqc_ws* ws, size_t n_index, size_t n_cplx);
int ensure_index(aqc_ws* ws, size_t n);
int ensure_coef(aqc_ws* ws);
int copy_in(aqc_ws* ws, double2* dst, const double* src, size_t rows);
int copy_out(aqc_ws* ws, double* dst, const double2* src, size_t rows);
int results_guard(aqc_ws* ws);
// aqc_ws_sweep.cpp
void touch_buf(aqc_ws* ws, int buf);   // somebody other than the V^H / sweep pair is about to write the whole buffer
int ensure_z_full(aqc_ws* ws, bool reader);   // before anybody reads Z (or writes a part of it): complete a partial V^H y
bool vdag_route_restricted(const aqc_ws* ws, int x_buf);
int run_vdag_restricted(aqc_ws* ws, int x_buf, bool support_in_gather_set = false);
void vdag_restricted_state_after(aqc_ws* ws, int x_buf);
int grad_from_impl(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer, bool support_in_gather_set);
bool sweep_route_sparse(const aqc_ws* ws, int x_buf, bool will_vdag);
bool sweep_skips_zero_w(const aqc_ws* ws, int x_buf);
int sweep_r_only_sub(const aqc_ws* ws);   // the lhs state is a combination of basis states the device knows: zero groups of w are skipped
int sweep_sparse_prepare(aqc_ws* ws);
void apply_state_after(aqc_ws* ws, bool inverse, int src_buf, int dst_buf);
void sweep_state_after(aqc_ws* ws, bool sparse, bool replayed);
int ensure_umat(aqc_ws* ws, DevPlan& p);
int run_coef(aqc_ws* ws);
int run_apply(aqc_ws* ws, bool inverse, int src_buf, int dst_buf);
void drop_graphs(aqc_ws* ws);
// aqc_ws_project.cpp
void proj_plan(aqc_ws* ws, int low_bits);   // decides the route and lowers the virtual plan (host only)
int proj_alloc(aqc_ws* ws);                 // its device side (plan tables, buffers, offset tables)
void proj_free(aqc_ws* ws);
bool sweep_route_projected(const aqc_ws* ws, bool sparse);
int run_projected_stages(aqc_ws* ws);       // projection + the virtual stage launches (after the sweep's first stage)
aqc::RgradSecond projected_rgrad_plan(aqc_ws* ws);   // the virtual plan's gradient walk, to ride in the real plan's launch
bool vdag_route_projected(aqc_ws* ws, int x_buf);   // the objective's V^H by two passes over y instead of its stages (host-known single basis state)
int run_vdag_projected(aqc_ws* ws, int x_buf);      // Y -> Z on the lhs tiles, the virtual z for the sweep that follows
int proj_fix_amplitudes(aqc_ws* ws, int x_buf);     // after the gather: the amplitudes outside the lhs tiles, from the virtual z
int ensure_sweep_items(aqc_ws* ws, int x_buf);      // aqc_ws_sweep.cpp: the first-stage item list of the sparse sweep, rebuilt when the support changed

}  // namespace aqc
