// C ABI (include/aqc_hip.h): device-resident multi-start L-BFGS and the one-call surrogate evaluation.
#include "aqc_ws.h"

using namespace aqc;

extern "C" {

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
    LB_OK(hipMalloc((void**)&dl, (size_t)B * sizeof(long long)));
    if (!ws->d_combo_prev[AQC_BUF_X2]) LB_OK(hipMalloc((void**)&ws->d_combo_prev[AQC_BUF_X2], sizeof(long long) * 2 * B));
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
    long long* d_prev = ws->d_combo_prev[AQC_BUF_X2];   // [B][2]: positions of X2 written by the previous evaluation (the support of the lhs states)
    {   // weight = 1, max_no = 0, active = 1, X2 empty
        std::vector<double> ones(B, 1.0);
        std::vector<int> one_i(B, 1);
        LB_OK(hipMemcpyAsync(L.weight, ones.data(), sizeof(double) * B, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemcpyAsync(L.active, one_i.data(), sizeof(int) * B, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemsetAsync(dl, 0, sizeof(long long) * B, st_));
        LB_OK(hipMemsetAsync(d_prev, 0xff, sizeof(long long) * 2 * B, st_));   // -1: nothing written yet
        LB_OK(hipMemcpyAsync(L.x, x0, sizeof(double) * BT, hipMemcpyHostToDevice, st_));
        LB_OK(hipMemsetAsync(ws->bufs[AQC_BUF_X2], 0, sizeof(double2) * (size_t)B * ws->lane_elems, st_));
        ws->combo_valid[AQC_BUF_X2] = true;   // the buffer holds exactly the pattern its record names (nothing, so far)
        ws->combo_last_elem[AQC_BUF_X2].clear();
        ++ws->supp_version[AQC_BUF_X2];
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
        const bool sparse = sweep_route_sparse(ws, AQC_BUF_X2, true);
        if (sparse && sweep_sparse_prepare(ws)) return 1;
        if (sparse && vdag_route_restricted(ws, AQC_BUF_X2)) { if (run_vdag_restricted(ws, AQC_BUF_X2, true)) return 1; }   // V^H where the gather and
        else if (run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;                                                   // the sweep read it
        if (aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
        HIP_OK(lb_prepare(L, ws->d_small, update, f_o, raw_hs, ws->bufs[AQC_BUF_X2], ws->lane_elems, ws->d_index, d_prev, st_));
        ++ws->supp_version[AQC_BUF_X2];   // (the leading flip state is chosen on the device: the support may have moved)
        ws->combo_last_elem[AQC_BUF_X2].clear();
        if (grad_from_impl(ws, AQC_BUF_X2, block_from, block_to, front_layer, true)) return 1;
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
        ws->combo_valid[AQC_BUF_X2] = true;   // the buffer holds exactly the pattern its record names
        ws->combo_last_elem[AQC_BUF_X2].clear();
    }
    const bool sparse = sweep_route_sparse(ws, AQC_BUF_X2, true);   // (decided here: part of the captured graph's key)
    if (sparse && sweep_sparse_prepare(ws)) return 1;
    const bool lazy = sparse && vdag_route_restricted(ws, AQC_BUF_X2);
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
        if (lazy ? run_vdag_restricted(ws, AQC_BUF_X2, true) : run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;
        if (aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
        {
            ProfScope ps(ws, AQC_K_MISC);
            HIP_OK(lb_prepare(L, ws->d_small, update_state, d_f, d_hs, ws->bufs[AQC_BUF_X2], ws->lane_elems, ws->d_index,
                              ws->d_combo_prev[AQC_BUF_X2], st));
            ++ws->supp_version[AQC_BUF_X2];
            ws->combo_last_elem[AQC_BUF_X2].clear();
        }
        // (update_state == 0 leaves weight / max_no / fidelity as they came in; fidelity is only written by an update)
        if (grad_from_impl(ws, AQC_BUF_X2, block_from, block_to, front_layer, true)) return 1;
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
                                            (long long)(size_t)ws->d_small, (long long)(size_t)ws->d_combo_prev[AQC_BUF_X2], (sparse ? 1 : 0) + (lazy ? 2 : 0) + (sweep_skips_zero_w(ws, AQC_BUF_X2) ? 4 : 0),
                                            (long long)(size_t)ws->d_vd_items};
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
        ws->ckpt_valid = false;
        if (lazy) vdag_restricted_state_after(ws, AQC_BUF_X2); else apply_state_after(ws, true, AQC_BUF_Y, AQC_BUF_Z);
        ++ws->supp_version[AQC_BUF_X2];
        ws->combo_last_elem[AQC_BUF_X2].clear();
        sweep_state_after(ws, sparse, true);
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

}  // extern "C"
