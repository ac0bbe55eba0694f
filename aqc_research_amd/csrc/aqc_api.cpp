// C ABI (include/aqc_hip.h): host runtime around the gfx950 kernels.
//
// A context (aqc_ctx) is the immutable gate program of one ansatz.  A workspace (aqc_ws) binds
// it to one HIP device + one stream and keeps `batch` independent evaluations resident in HBM:
//   thetas[B][T] -> coef[B][n+L+1][24]         (coef_kernel + sign_kernel, once per theta upload)
//   Y, Z, X, W, ZW, X2 : [B][2^n][pitch] complex128 (pitch = columns padded to a power of two)
//   partial[B][5*G][ntiles], grads[B][T]        (inner-product partials and their fixed-order sum)
// Nothing below ever falls back to host arithmetic: if HIP is unusable every call fails loudly.
// This file: contexts, workspaces, buffers, thetas, small results, profiling, one-shot entry points (see aqc_ws.h for the
// other translation units).
#include "aqc_ws.h"

using namespace aqc;

namespace {
thread_local std::string g_error;
}

namespace aqc {

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}
int set_error(const std::string& msg) { g_error = msg; return 1; }  // for the other translation units

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

// Results of the evaluation just enqueued -> pinned host memory, asynchronously on the workspace's stream (what an optimizer
// on the host consumes every evaluation: gradients, gathered amplitudes, <A|B>); aqc_ws_results_fetch waits and hands them out.
// The copies run on a second stream: they wait for the producing kernels (event) and overlap with the kernels of the NEXT
// evaluation; the kernels that overwrite the results (gather, <A|B>, gradient walk) wait for the copies in turn (results_guard).
int results_guard(aqc_ws* ws) {
    if (ws->copy_pending && !ws->capturing) {
        HIP_OK(hipStreamWaitEvent(ws->stream, ws->ev_copied, 0));
        ws->copy_pending = false;
    }
    return 0;
}


}  // namespace aqc

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
    Plan plan = make_plan(ctx->prog, col_bits, tile_bits, low_bits, which == 0);
    if (which == 3) {   // V^H as the matrix-core workspaces run it: the sweep's plan (sub-stages included) walked backwards
        split_substages(ctx->prog, plan, 4, 1 << 20);
        plan = mirror_plan(plan);
    }
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

// host-only: the projected route a state-vector workspace of this tiling would take (same fields as aqc_ws_projected_info)
int aqc_plan_projected(aqc_ctx* ctx, int tile_bits, int low_bits, int32_t* info) {
    if (!ctx || !info) return fail("null argument");
    if (tile_bits <= 0) tile_bits = 12;
    if (low_bits < 0) low_bits = 3;
    const Program& prog = ctx->prog;
    tile_bits = std::min(std::max(tile_bits, 8), std::min(12, prog.n));
    aqc_ws tmp;   // (never touches a device: proj_plan only reads the plan and the switches)
    tmp.ctx = ctx;
    tmp.ncols = 1; tmp.col_bits = 0; tmp.nbits = prog.n;
    Plan best = make_plan(prog, 0, tile_bits, low_bits, false);
    for (int lb = low_bits - 1; lb >= 2; --lb) {
        Plan cand = make_plan(prog, 0, tile_bits, lb, false);
        if (cand.stages.size() < best.stages.size()) best = cand;
    }
    lower_plan(prog, best, tmp.sweep, 4, true, true);
    tmp.sparse_enabled = true;
    tmp.inv_mirrored = tmp.sweep.v3 && tmp.sweep.plan.stages.size() >= 2;
    proj_plan(&tmp, low_bits);
    const int rc = aqc_ws_projected_info(&tmp, info);
    tmp.ctx = nullptr;
    return rc;
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
    if (which == 3) plan = mirror_plan(plan);
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
    cached_plan(1, ks, false, want_v3 ? 4 : (want_v2 ? (env_int("AQC_SWEEP_REG_BITS", 4) == 3 ? 3 : 4) : 0), true, ws->sweep);
    // V^H: on the matrix-core path with equal tile sizes, the SWEEP's plan walked backwards (same stages, same sub-stages, same
    // cost), so that the states between its stages are the states z takes between the sweep's stages -- see aqc_ws_sweep.cpp
    ws->inv_mirrored = want_v3 && ka == ks && ws->sweep.v3 && ws->sweep.plan.stages.size() >= 2 && env_int("AQC_MIRROR_PLAN", 1) != 0;
    if (ws->inv_mirrored) {
        const std::vector<int> key = {3, ws->col_bits, ks, low_bits, 4, 0, 1};
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto it = ctx->plan_cache.find(key);
        if (it == ctx->plan_cache.end()) {
            DevPlan fresh;
            lower_plan(prog, mirror_plan(ws->sweep.plan), fresh, 4, false, true, true);
            it = ctx->plan_cache.emplace(key, std::move(fresh)).first;
        }
        ws->inv = it->second;
    } else {
        cached_plan(0, ka, true, (want_v2 || want_v3) ? 4 : 0, false, ws->inv);
    }
    ws->sparse_enabled = env_int("AQC_SPARSE_SWEEP", 1) != 0;
    ws->sparse_min_items = env_int("AQC_SPARSE_MIN_ITEMS", 512);
    ws->lazy_z_enabled = env_int("AQC_LAZY_Z", 1) != 0;
    ws->r_only_enabled = env_int("AQC_R_ONLY_LAST", 1) != 0;
    ws->r_only_max_subs = env_int("AQC_R_ONLY_MAX_SUBS", 12);
    ws->proj_vdag_enabled = env_int("AQC_PROJECTED_VDAG", 1) != 0;
    ws->proj_fused_enabled = env_int("AQC_PROJECTED_FUSED", 1) != 0;
    ws->proj_vdag_min_elems = (long long)env_int("AQC_PROJECTED_VDAG_MIN_ELEMS", 1 << 24);
    ws->skipw_enabled = env_int("AQC_SKIP_ZERO_W", 0) != 0;   // (measured slower than multiplying the zeros: opt-in, see sweep_mfma_kernel)
    for (DevPlan* p : {&ws->fwd, &ws->inv, &ws->sweep}) {
        const std::string err = check_plan(prog, p->plan);
        if (!err.empty()) { delete ws; return fail("planner produced an invalid plan: %s", err.c_str()); }
    }
    proj_plan(ws, low_bits);   // the sweep's later stages on a virtual register, where the plan allows (aqc_ws_project.cpp)

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
    WS_TRY(proj_alloc(ws));
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
        // [V^H | sweep | virtual sweep | V]; with the mirrored V^H plan the sweep's jobs write V^H's operands as well (sub-stage s of the
        // sweep is sub-stage nsubs - 1 - s of V^H, conjugate-transposed) and V^H gets no jobs of its own
        std::vector<UJob> jobs;
        DevPlan* vsw = ws->proj.ok ? &ws->proj.vsw : nullptr;
        const int nsw = ws->sweep.v3 ? (int)ws->sweep.h_subs3.size() : 0;
        ws->ujobs_mirror = ws->inv_mirrored && ws->inv.v3 && (int)ws->inv.h_subs3.size() == nsw && env_int("AQC_UBUILD_MIRROR", 1) != 0;
        for (DevPlan* p : {&ws->inv, &ws->sweep, vsw, &ws->fwd})
            if (p && p->v3)
                for (size_t i = 0; i < p->h_subs3.size(); ++i) {
                    if (p == &ws->inv && ws->ujobs_mirror) continue;
                    UJob j{p->d_subs3 + i, p->d_grps, p->d_umat, (int)i, (int)p->h_subs3.size(), p->plan.inverse ? 1 : 0, prog.entangler, nullptr, 0, 0};
                    if (p == &ws->sweep && ws->ujobs_mirror) { j.umat_mirror = ws->inv.d_umat; j.mirror_index = nsw - 1 - (int)i; j.mirror_nsubs = nsw; }
                    if (p == vsw) {   // the virtual plan walked backwards takes its operands from the same jobs
                        const int nvs = (int)vsw->h_subs3.size();
                        j.umat_mirror = ws->proj.vinv.d_umat; j.mirror_index = nvs - 1 - (int)i; j.mirror_nsubs = nvs;
                    }
                    jobs.push_back(j);
                }
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


int aqc_ws_destroy(aqc_ws* ws) {
    if (!ws) return 0;
    (void)hipSetDevice(ws->device);
    if (ws->stream) (void)hipStreamSynchronize(ws->stream);
    drop_graphs(ws);
    proj_free(ws);
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
    for (void* q : {(void*)ws->d_sw_items, (void*)ws->d_sw_clear, (void*)ws->d_sw_counts, (void*)ws->d_sw_lane_parts, (void*)ws->d_sw_prev_tiles,
                    (void*)ws->w2, (void*)ws->zw2, (void*)ws->d_vd_items})
        if (q) (void)hipFree(q);
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
    touch_buf(ws, buf);
    if (!src) return fail("null source");
    HIP_OK(hipSetDevice(ws->device));
    return copy_in(ws, ws->bufs[buf], src, (size_t)ws->batch << ws->ctx->prog.n);
}

int aqc_ws_upload_lane(aqc_ws* ws, int buf, int lane, const double* src) {
    if (check_buf(ws, buf)) return 1;
    if (buf == AQC_BUF_Z && ensure_z_full(ws, false)) return 1;
    ws->combo_valid[buf] = false;
    touch_buf(ws, buf);
    if (!src) return fail("null source");
    if (lane < 0 || lane >= ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(ws->device));
    return copy_in(ws, ws->bufs[buf] + (size_t)lane * ws->lane_elems, src, (size_t)1 << ws->ctx->prog.n);
}

int aqc_ws_broadcast(aqc_ws* ws, int buf, const double* src) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    touch_buf(ws, buf);
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
    if (src_buf == AQC_BUF_Z && ensure_z_full(src_ws, true)) return 1;
    if (dst_buf == AQC_BUF_Z && ensure_z_full(dst_ws, false)) return 1;
    HIP_OK(hipSetDevice(dst_ws->device));
    dst_ws->combo_valid[dst_buf] = false;
    touch_buf(dst_ws, dst_buf);
    if (src_ws->stream != dst_ws->stream) HIP_OK(hipStreamSynchronize(src_ws->stream));   // the source is complete
    HIP_OK(hipMemcpyAsync(dst_ws->bufs[dst_buf] + (size_t)dst_lane * dst_ws->lane_elems, src_ws->bufs[src_buf] + (size_t)src_lane * src_ws->lane_elems,
                          sizeof(double2) * dst_ws->lane_elems, hipMemcpyDeviceToDevice, dst_ws->stream));
    return 0;
}

int aqc_ws_download(aqc_ws* ws, int buf, double* dst) {
    if (check_buf(ws, buf)) return 1;
    if (!dst) return fail("null destination");
    HIP_OK(hipSetDevice(ws->device));
    if (buf == AQC_BUF_Z && ensure_z_full(ws, true)) return 1;
    return copy_out(ws, dst, ws->bufs[buf], (size_t)ws->batch << ws->ctx->prog.n);
}

int aqc_ws_download_lane(aqc_ws* ws, int buf, int lane, double* dst) {
    if (check_buf(ws, buf)) return 1;
    if (!dst) return fail("null destination");
    if (lane < 0 || lane >= ws->batch) return fail("lane out of range");
    HIP_OK(hipSetDevice(ws->device));
    if (buf == AQC_BUF_Z && ensure_z_full(ws, true)) return 1;
    return copy_out(ws, dst, ws->bufs[buf] + (size_t)lane * ws->lane_elems, (size_t)1 << ws->ctx->prog.n);
}

int aqc_ws_set_basis(aqc_ws* ws, int buf, const int64_t* index) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    touch_buf(ws, buf);
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
    {
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_scatter_one(ws->bufs[buf], ws->lane_elems, ws->batch, ws->d_basis_index, ws->stream));
    }
    // the buffer now holds exactly this sparse pattern: the same record aqc_ws_set_combo keeps (positions written, second one
    // absent), so that a later set_combo clears one element per lane instead of the buffer and the sweep knows the support
    std::vector<long long> supp(2 * (size_t)ws->batch, -1);
    for (int b = 0; b < ws->batch; ++b) supp[2 * (size_t)b] = elem[b];
    if (!ws->d_combo_prev[buf]) HIP_OK(hipMalloc((void**)&ws->d_combo_prev[buf], sizeof(long long) * 2 * ws->batch));
    HIP_OK(hipMemcpyAsync(ws->d_combo_prev[buf], supp.data(), sizeof(long long) * supp.size(), hipMemcpyHostToDevice, ws->stream));
    HIP_OK(hipStreamSynchronize(ws->stream));
    ws->combo_valid[buf] = true;
    ws->combo_last_elem[buf] = supp;
    ws->combo_last_coef[buf].assign(4 * (size_t)ws->batch, 0.0);
    for (int b = 0; b < ws->batch; ++b) ws->combo_last_coef[buf][4 * (size_t)b] = 1.0;
    ++ws->supp_version[buf];
    return 0;
}

int aqc_ws_set_identity(aqc_ws* ws, int buf) {
    if (check_buf(ws, buf)) return 1;
    ws->combo_valid[buf] = false;
    touch_buf(ws, buf);
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
    // the same combination as the buffer already holds (an objective hands the same lhs state in on every call): nothing to do
    std::vector<double> cf(coef, coef + 4 * (size_t)B);
    if (ws->combo_valid[buf] && ws->combo_last_elem[buf] == elem && ws->combo_last_coef[buf] == cf) return 0;
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
    ws->combo_last_elem[buf] = elem;
    ws->combo_last_coef[buf].swap(cf);
    ++ws->supp_version[buf];
    touch_buf(ws, buf);
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
    if (buf == AQC_BUF_Z && ensure_z_full(ws, true)) return 1;
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
    if ((buf_a == AQC_BUF_Z || buf_b == AQC_BUF_Z) && ensure_z_full(ws, true)) return 1;
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
    ws->h_gather = elem;
    ++ws->gather_gen;
    return 0;
}

int aqc_ws_gather_launch(aqc_ws* ws, int buf) {
    if (check_buf(ws, buf)) return 1;
    if (ws->gather_count < 1) return fail("aqc_ws_gather_setup has not been called");
    HIP_OK(hipSetDevice(ws->device));
    if (results_guard(ws)) return 1;
    if (buf == AQC_BUF_Z && !ws->z_full && ws->z_gather_gen != ws->gather_gen && ensure_z_full(ws, true)) return 1;   // a partial Z covers the
                                                                                                                    // set it was computed for
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
int aqc_ws_vdot_launch(aqc_ws* ws, int buf_a, int buf_b) {
    if (check_buf(ws, buf_a) || check_buf(ws, buf_b)) return 1;
    HIP_OK(hipSetDevice(ws->device));
    if ((buf_a == AQC_BUF_Z || buf_b == AQC_BUF_Z) && ensure_z_full(ws, true)) return 1;
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
    ws->prof_log.clear();
    return 0;
}

int aqc_ws_profile_log(aqc_ws* ws, int32_t* kinds, double* ms, int cap, int* count) {
    if (!ws || !count || cap < 0 || (cap > 0 && (!kinds || !ms))) return fail("invalid argument");
    *count = (int)ws->prof_log.size();
    for (int i = 0; i < cap && i < *count; ++i) { kinds[i] = ws->prof_log[i].first; ms[i] = ws->prof_log[i].second; }
    return 0;
}

int aqc_ws_plan_stage(aqc_ws* ws, int which, int stage, int* num_subs, int* num_groups, int* bits_out) {
    if (!ws) return fail("null workspace");
    const DevPlan& p = which == 0 ? ws->inv : (which == 1 ? ws->sweep : ws->fwd);
    if (stage < 0 || stage >= (int)p.plan.stages.size()) return fail("stage index out of range");
    const Stage& st = p.plan.stages[stage];
    if (num_subs) *num_subs = (int)st.subs.size();
    if (num_groups) *num_groups = (int)st.ops.size();
    if (bits_out) for (size_t i = 0; i < st.bits.size(); ++i) bits_out[i] = st.bits[i];
    return 0;
}

int aqc_ws_plan_skips(aqc_ws* ws, int which, int stage, int* out, int max_subs) {
    if (!ws || !out) return fail("null argument");
    const DevPlan& p = which == 0 ? ws->inv : (which == 1 ? ws->sweep : ws->fwd);
    if (stage < 0 || stage >= (int)p.h_stages.size()) return fail("stage index out of range");
    const DevStage& ds = p.h_stages[stage];
    for (int i = 0; i < ds.nsubs && i < max_subs; ++i) {
        const uint32_t info = p.v3 && ws->skipw_enabled ? p.h_subs3[ds.sub_begin + i].skipinfo : 0u;
        out[2 * i] = -__builtin_popcount(info & 15u);
        out[2 * i + 1] = -__builtin_popcount((info >> 6) & 3u);
    }
    return 0;
}

int aqc_ws_sparse_counts(aqc_ws* ws, int64_t* counts) {
    if (!ws || !counts) return fail("null argument");
    counts[0] = counts[1] = counts[2] = -1;
    if (!ws->d_sw_counts) return 0;
    HIP_OK(hipSetDevice(ws->device));
    int h[4] = {0, 0, 0, 0};
    HIP_OK(hipStreamSynchronize(ws->stream));
    HIP_OK(hipMemcpy(h, ws->d_sw_counts, sizeof h, hipMemcpyDeviceToHost));
    if (ws->sw_lists_built & 1) { counts[0] = h[0]; counts[1] = h[1]; }
    if (ws->sw_lists_built & 2) counts[2] = h[2];
    return 0;
}

int aqc_ws_projected_info(aqc_ws* ws, int32_t* info) {
    if (!ws || !info) return fail("null argument");
    const aqc::ProjRoute& pr = ws->proj;
    for (int i = 0; i < 16; ++i) info[i] = 0;
    if (!pr.ok) return 0;
    info[0] = 1; info[1] = pr.nv; info[2] = pr.t; info[3] = pr.cb; info[4] = (int32_t)pr.vsw.h_stages.size();
    info[5] = (int32_t)pr.vsw.h_subs3.size(); info[6] = pr.kv; info[7] = pr.first_subs; info[8] = pr.us; info[9] = pr.nvp;
    for (size_t s = 0; s < pr.vsw.h_stages.size() && s < 6; ++s) info[10 + s] = pr.vsw.h_stages[s].nsubs;
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
