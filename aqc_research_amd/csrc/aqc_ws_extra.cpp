// C ABI (include/aqc_hip.h): dense zgemm, gate-level building blocks, coordinate descent, MPS helpers.
#include "aqc_ws.h"

using namespace aqc;

extern "C" {

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
    if (ws->batch != 1) return fail("aqc_ws_cd_sweep takes thetas_io[T] and one objective value: a single-lane workspace (aqc_ws_cd_sweeps serves lanes)");
    if (aqc_ws_cd_fits_one_launch(ws) && !(chain && chain[0] == '1')) return aqc_ws_cd_sweeps(ws, thetas_io, fobj, 1, -1);
    const Program& prog = ws->ctx->prog;
    const int dim = 1 << prog.n;
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
    if (buf == AQC_BUF_Z && ensure_z_full(ws, false)) return 1;
    touch_buf(ws, buf);
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
    if (buf == AQC_BUF_Z && ensure_z_full(ws, false)) return 1;   // (lanes of Z are rewritten, not all of it)
    touch_buf(ws, buf);
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

}  // extern "C"
