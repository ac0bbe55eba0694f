// The dense stages of the sparse-lhs sweep on a VIRTUAL register (grad_of_dot_product, core_operations.py:823-1019, for the
// lhs states every surrogate objective sweeps from: objective_base.py:42-255).
//
// After its first stage (local address bits L0) the lhs state of a lane is  w = psi (x) |e>:  psi = the stage's gates applied to the
// basis index on L0 (one tile of W), e = the basis index on the other bits F.  Let T be the qubits the gates of the LATER stages
// touch, Cb = T n L0, Us = L0 \ T.  Those gates act on T alone, so for every sub-stage k of theirs
//     w_k[u, i_T] = sum_c M_k[i_T, c] psi[u, c],        M_k = U_k ... U_1 M_0,   M_0[i_T, c] = [i_T = (c on Cb, e on T n F)]
// and the 16 x 16 matrix the gradient walk needs,  R_k = sum z_k w_k^H  over everything but the sub-stage's register bits, is
//     R_k = sum_{rest of i_T, c} Y_k M_k^H,             Y_k = U_k ... U_1 Y_0,   Y_0[i_T, c] = sum_u conj(psi[u, c]) z_0[u, i_T]
// (z_0: z as it enters those stages = the checkpoint the mirrored V^H keeps in ZW).  M and Y are states of |T| + |Cb| virtual
// qubits -- T, plus one spectator per qubit of Cb -- and R_k is exactly what the sweep kernels form from a (w, z) pair: the later
// stages become an ordinary sweep of the gates' sub-circuit on that small register, after ONE pass over z (aqc_project.hip).
// At the headline (16 qubits, L0 = 0..11, T = 8..15) the register has 12 qubits: 1/16 of the elements, and the second stage of
// the sweep -- more than half of an evaluation on the sparse route -- becomes one memory-bound pass plus a tile per lane.
// Exact: the same sums in another order.  Two basis states per lane: one item (M, Y pair) per tile of the first stage they occupy.
#include "aqc_ws.h"

#include <algorithm>

using namespace aqc;

namespace aqc {

namespace {
int popc(uint64_t v) { return __builtin_popcountll(v); }
unsigned deposit_bits(unsigned v, const std::vector<int>& bits) {
    unsigned o = 0;
    for (size_t j = 0; j < bits.size(); ++j)
        if (v >> j & 1) o |= 1u << bits[j];
    return o;
}
}  // namespace

void proj_plan(aqc_ws* ws, int low_bits) {
    ProjRoute& pr = ws->proj;
    pr.ok = false;
    const Program& prog = ws->ctx->prog;
    const DevPlan& p = ws->sweep;
    if (env_int("AQC_PROJECTED", 1) == 0 || !ws->sparse_enabled || !ws->inv_mirrored || !p.v3 || ws->col_bits != 0 || ws->ncols != 1 ||
        p.plan.stages.size() < 2 || prog.n > 30)
        return;
    const int n = prog.n;
    uint64_t L0 = 0, T = 0;
    for (int b : p.plan.stages[0].bits) L0 |= 1ull << b;
    pr.rest.clear();
    for (size_t s = 1; s < p.plan.stages.size(); ++s)
        for (int gi : p.plan.stages[s].ops) {
            const GateGroup& g = prog.groups[gi];
            pr.rest.push_back(gi);
            T |= 1ull << g.q0;
            if (g.q1 >= 0) T |= 1ull << g.q1;
        }
    if (pr.rest.empty()) return;
    const uint64_t all = (1ull << n) - 1, F = all & ~L0, Cb = T & L0, Us = L0 & ~T;
    pr.t = popc(T); pr.cb = popc(Cb); pr.us = popc(Us);
    pr.nv = pr.t + pr.cb;
    // the pass over z reads runs of 16 elements (address bits 0..3 among the summed ones); the register must be worth it
    if ((Us & 15) != 15 || pr.t < 4 || pr.cb > 6 || pr.nv + 2 > n || pr.nv > 20) return;
    pr.nvp = std::max(pr.nv, 8);
    pr.kv = std::min(pr.nvp, p.k);
    pr.ntiles_v = 1 << (pr.nvp - pr.kv);
    pr.first_subs = p.h_stages[0].nsubs;
    std::vector<int> tbits, cbits, usbits;
    std::vector<int> vq(n, -1);
    for (int b = 0; b < n; ++b) {
        if (T >> b & 1) { vq[b] = (int)tbits.size(); tbits.push_back(b); }
        if (Cb >> b & 1) cbits.push_back(b);
        if (Us >> b & 1) usbits.push_back(b);
    }
    pr.vprog = prog;   // group indices, thetas and slots stay the real ones; only the qubits move
    for (int gi : pr.rest) {
        GateGroup& g = pr.vprog.groups[gi];
        g.q0 = vq[g.q0];
        if (g.q1 >= 0) g.q1 = vq[g.q1];
    }
    Plan best;
    for (int lb = std::max(low_bits, 2); lb >= 2; --lb) {
        Plan cand = make_plan(pr.vprog, 0, pr.kv, lb, false, &pr.rest, pr.nvp);
        if (best.stages.empty() || cand.stages.size() < best.stages.size()) best = cand;
    }
    // (a narrow sub-stage search: the virtual stages are a small share of an evaluation, planning them must not cost what planning the real ones does)
    lower_plan(pr.vprog, best, pr.vsw, 4, true, true, false, env_int("AQC_PROJECTED_BEAM", 8));
    if (!pr.vsw.v3 || !check_plan(pr.vprog, pr.vsw.plan, &pr.rest).empty()) return;
    lower_plan(pr.vprog, mirror_plan(pr.vsw.plan), pr.vinv, 4, false, true, true);
    if (!pr.vinv.v3 || pr.vinv.h_subs3.size() != pr.vsw.h_subs3.size()) return;
    for (DevPlan* q : {&pr.vsw, &pr.vinv})
        for (DevStage& ds : q->h_stages) {   // one more non-local bit: the item (first or second tile of the lane's lhs state)
            if (ds.nub >= 32) return;
            ds.ubits[ds.nub++] = pr.nvp;
            ds.ntiles *= 2;
        }
    pr.l0_mask = (unsigned)L0;
    pr.ff_mask = (unsigned)(F & ~T);
    pr.cb_mask = (unsigned)Cb;
    pr.tf_mask = (unsigned)(T & F);
    pr.h_tab.clear();
    for (unsigned i = 0; i < (1u << pr.t); ++i) pr.h_tab.push_back(deposit_bits(i, tbits));
    for (unsigned k = 0; k < (1u << pr.us); ++k) pr.h_tab.push_back(deposit_bits(k, usbits));
    for (unsigned c = 0; c < (1u << pr.cb); ++c) pr.h_tab.push_back(deposit_bits(c, cbits));
    for (unsigned c = 0; c < (1u << pr.cb); ++c) {   // the same as an index on the T bits
        const unsigned addr = deposit_bits(c, cbits);
        unsigned it = 0;
        for (size_t j = 0; j < tbits.size(); ++j)
            if (addr >> tbits[j] & 1) it |= 1u << j;
        pr.h_tab.push_back(it);
    }
    pr.ok = true;
    if (env_int("AQC_VERBOSE", 0))
        fprintf(stderr, "aqc_hip: projected route: the sweep's stages after the first run on %d virtual qubits (%d touched, %d shared with the first "
                "stage) instead of %d: %zu stage(s) of 2^%d tiles, %zu sub-stages\n", pr.nv, pr.t, pr.cb, n, pr.vsw.h_stages.size(), pr.kv,
                pr.vsw.h_subs3.size());
}

int proj_alloc(aqc_ws* ws) {
    ProjRoute& pr = ws->proj;
    if (!pr.ok) return 0;
    const size_t B = (size_t)ws->batch;
    if (upload_plan(pr.vsw) || upload_plan(pr.vinv)) return 1;
    const size_t nsubs = std::max<size_t>(pr.vsw.h_subs3.size(), 1);
    HIP_OK(hipMalloc((void**)&pr.vsw.d_umat, sizeof(double) * B * nsubs * 12 * 64));
    HIP_OK(hipMalloc((void**)&pr.vinv.d_umat, sizeof(double) * B * nsubs * 12 * 64));
    HIP_OK(hipMalloc((void**)&pr.vsw.d_rpart, sizeof(double2) * B * nsubs * (2 * (size_t)pr.ntiles_v) * 256));
    const size_t vbytes = sizeof(double2) * B * (2ull << pr.nvp);
    HIP_OK(hipMalloc((void**)&pr.vm, vbytes));
    pr.vy_copies = pr.us > 8 && pr.us <= 10 && pr.cb <= 4 ? 1 << (pr.us - 8) : 1;
    HIP_OK(hipMalloc((void**)&pr.vy, vbytes * pr.vy_copies));
    HIP_OK(hipMemsetAsync(pr.vm, 0, vbytes, ws->stream));   // (entries beyond 2^nv -- a register padded to 8 qubits -- stay zero for good)
    HIP_OK(hipMemsetAsync(pr.vy, 0, vbytes * pr.vy_copies, ws->stream));
    pr.cpart_shares = 0;
    const long want_wgs = env_int("AQC_PROJECTED_FUSED_WGS", 512);
    if (pr.us <= 10 && pr.cb <= 4 && (long)B < want_wgs) {   // the fused pass of a small batch splits its walk over the touched bits (launch_project_fused)
        int shares = 1;
        const int cap = env_int("AQC_PROJECTED_FUSED_MAX_SHARES", 64);   // (1: never split -- what large batches run; tests)
        while ((long)B * pr.vy_copies * shares < want_wgs && 2 * shares <= (1 << (pr.t - 4)) && 2 * shares <= cap) shares *= 2;
        if (shares > 1) {
            HIP_OK(hipMalloc((void**)&pr.cpart, sizeof(double2) * (size_t)shares * 2 * B * (16ull << pr.us)));
            pr.cpart_shares = shares;
        }
    }
    HIP_OK(hipMalloc((void**)&pr.vme, vbytes));
    HIP_OK(hipMemsetAsync(pr.vme, 0, vbytes, ws->stream));
    HIP_OK(hipMalloc((void**)&pr.d_tab, sizeof(unsigned) * pr.h_tab.size()));
    HIP_OK(hipMemcpy(pr.d_tab, pr.h_tab.data(), sizeof(unsigned) * pr.h_tab.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&pr.d_items, sizeof(TileItem) * 2 * B * pr.ntiles_v));
    HIP_OK(hipMalloc((void**)&pr.d_count, sizeof(int)));
    HIP_OK(hipMemsetAsync(pr.d_count, 0, sizeof(int), ws->stream));
    HIP_OK(hipMalloc((void**)&pr.d_lane_parts, sizeof(int) * B));
    HIP_OK(hipMemsetAsync(pr.d_lane_parts, 0, sizeof(int) * B, ws->stream));
    return 0;
}

void proj_free(aqc_ws* ws) {
    ProjRoute& pr = ws->proj;
    for (DevPlan* vp : {&pr.vsw, &pr.vinv}) {
        DevPlan& v = *vp;
        for (void* q : {(void*)v.d_stages, (void*)v.d_ops, (void*)v.d_subs, (void*)v.d_mops, (void*)v.d_subs3, (void*)v.d_grps, (void*)v.d_umat, (void*)v.d_rpart})
            if (q) (void)hipFree(q);
        v.d_stages = nullptr; v.d_ops = nullptr; v.d_subs = nullptr; v.d_mops = nullptr; v.d_subs3 = nullptr; v.d_grps = nullptr;
        v.d_umat = nullptr; v.d_rpart = nullptr;
    }
    for (void* q : {(void*)pr.vm, (void*)pr.vy, (void*)pr.vme, (void*)pr.cpart, (void*)pr.d_tab, (void*)pr.d_items, (void*)pr.d_count, (void*)pr.d_lane_parts})
        if (q) (void)hipFree(q);
    pr.vm = pr.vy = pr.vme = nullptr; pr.cpart = nullptr; pr.d_tab = nullptr; pr.d_items = nullptr; pr.d_count = nullptr; pr.d_lane_parts = nullptr;
    pr.ok = false;
}

bool sweep_route_projected(const aqc_ws* ws, bool sparse) { return sparse && ws->proj.ok; }

static ProjArgs proj_args(aqc_ws* ws);
// the virtual lhs pattern and the item list of the virtual launches: rebuilt when the first-stage list changed (always inside a
// captured graph, whose replays follow a support the host does not see; always with several virtual stages, which work in place on vm)
static int ensure_pattern(aqc_ws* ws, const ProjArgs& a) {
    ProjRoute& pr = ws->proj;
    const bool keep = !ws->capturing && pr.vsw.h_stages.size() == 1 && pr.init_buf == ws->sw_items_buf && pr.init_buf >= 0 &&
                      pr.init_version == ws->sw_items_version;
    if (keep) return 0;
    ProfScope ps(ws, AQC_K_MISC);
    HIP_OK(launch_project_init(a, ws->stream));
    pr.init_buf = ws->capturing ? -1 : ws->sw_items_buf;
    pr.init_version = ws->sw_items_version;
    return 0;
}

// what every launch of the route shares: the first-stage items, the bit masks, the virtual buffers' bookkeeping
static ProjArgs proj_args(aqc_ws* ws) {
    ProjRoute& pr = ws->proj;
    ProjArgs a;
    memset(&a, 0, sizeof a);
    a.lane_stride = ws->lane_elems;
    a.items = ws->d_sw_items;
    a.nitems = ws->d_sw_counts;
    a.lane_parts = ws->d_sw_lane_parts;
    const DevStage& s0 = ws->sweep.h_stages[0];
    a.nub0 = s0.nub;
    for (int i = 0; i < s0.nub; ++i) a.ubits0[i] = s0.ubits[i];
    a.ff_mask = pr.ff_mask; a.cb_mask = pr.cb_mask; a.tf_mask = pr.tf_mask;
    a.nvp = pr.nvp;
    a.off_t = pr.d_tab;
    a.off_cb = pr.d_tab + (1u << pr.t) + (1u << pr.us);
    a.it_of_c = a.off_cb + (1u << pr.cb);
    a.off_us = pr.d_tab + (1u << pr.t);
    a.us_bits = pr.us;
    a.t = pr.t; a.cb = pr.cb; a.ntiles_v = pr.ntiles_v;
    a.vm = pr.vm;
    a.vitems = pr.d_items; a.vcount = pr.d_count; a.vlane_parts = pr.d_lane_parts;
    a.batch = ws->batch;
    return a;
}

// after the sweep's first stage (W holds psi on the listed tiles, ZW the checkpoint): the projection, then the virtual stages
int run_projected_stages(aqc_ws* ws) {
    ProjRoute& pr = ws->proj;
    DevPlan& v = pr.vsw;
    if (ws->proj_y0_ready) {   // run_vdag_projected of this call has left M_0 and Y_0 (from the target) on the virtual register
        ws->proj_y0_ready = false;
    } else {
        ProjArgs a = proj_args(ws);
        if (ensure_pattern(ws, a)) return 1;
        // Y_0[i_T, c] = sum_u conj(psi[u, c]) z[u, i_T]:  Y = the checkpoint, k = u, keep = i_T, S = psi in W, out = the virtual z
        a.y = ws->bufs[AQC_BUF_ZW];
        a.s = ws->bufs[AQC_BUF_W];
        a.out = pr.vy;
        a.s_virtual = 0; a.out_virtual = 1;
        a.staged = 1;   // (proj_plan: address bits 0..3 are among the summed ones)
        a.keep_bits = pr.t; a.k_bits = pr.us;
        const unsigned* off_us = pr.d_tab + (1u << pr.t);
        a.y_keep = ProjMap{pr.d_tab, 0};
        a.y_k = ProjMap{off_us, 0};
        a.s_k = ProjMap{off_us, 0};
        a.s_c = ProjMap{a.off_cb, 0};
        a.o_keep = ProjMap{nullptr, 0};
        a.o_c = ProjMap{nullptr, pr.t};
        ProfScope ps(ws, AQC_K_PROJECT);
        HIP_OK(launch_project(a, ws->stream));
    }
    const size_t m = v.h_stages.size();
    const int ntiles = 2 * pr.ntiles_v;
    for (size_t s = 0; s < m; ++s) {
        Stage3Args a;
        memset(&a, 0, sizeof a);
        a.stage = v.h_stages[s];
        a.subs = v.d_subs3;
        a.umat = v.d_umat;
        a.nsubs_total = (int)v.h_subs3.size();
        a.lane_stride = 2ull << pr.nvp;
        a.ntiles = ntiles;
        a.batch = ws->batch;
        a.in0 = pr.vm; a.in1 = pr.vy; a.out0 = pr.vm; a.out1 = pr.vy;
        a.store_out = s + 1 < m ? 3 : 0;
        a.items = pr.d_items; a.nitems = pr.d_count; a.max_items = 2 * ws->batch * pr.ntiles_v;
        a.rpart = v.d_rpart;
        a.nparts = ntiles;
        a.chunk = 0;
        if (a.stage.nsubs > 0) stage3_first_offsets(a, v.h_subs3[a.stage.sub_begin]);
        ProfScope ps(ws, AQC_K_SWEEP_VIRTUAL);
        HIP_OK(launch_sweep3(ntiles, ws->batch, pr.kv, ws->stream, a));
    }
    return 0;
}

// the gradient entries of the virtual plan's gate groups: its walk rides in the launch of the real plan's (which stops after its first stage)
RgradSecond projected_rgrad_plan(aqc_ws* ws) {
    ProjRoute& pr = ws->proj;
    DevPlan& v = pr.vsw;
    RgradSecond r;
    memset(&r, 0, sizeof r);
    r.subs = v.d_subs3; r.grps = v.d_grps; r.rpart = v.d_rpart; r.lane_parts = pr.d_lane_parts; r.umat = v.d_umat;
    r.nparts = 2 * pr.ntiles_v;
    r.nsubs_total = r.count = (int)v.h_subs3.size();
    return r;
}

// ---- the objective's V^H by projection ------------------------------------------------------------------------------------------
// One-call evaluations whose lhs state is ONE basis state per lane with coefficient 1, known to the host (aqc_ws_set_basis), and
// whose gather indices lie in the lane's first-stage tile or differ from the basis index only on the other bits (the flip states
// of the surrogate objectives): nothing of Z = V^H y is read outside the lhs tile, so the stages of V^H are not run at all.
//     C[tile e][u, c] = sum_{i_T} conj(M_end[i_T, c]) y[u, i_T]      M_end = (later stages' gates) M_0: the tile of (later stages)^H y
//     Y_end[i_T, c]   = sum_u conj(psi[u, c]) y[u, i_T]              psi = (first stage's gates)|e_L0>:  Y_0 = (later stages)^H Y_end
// -- two passes over y (the target is read twice, V^H's first stage read and wrote the whole state and ran 4 sub-stages over it) --,
// then V^H's last stage on the lhs tile alone (Z there), the sweep's first stage as ever, the virtual stages from (M_0, Y_0), and
// <g|V^H y> = sum_c Y_0[(c, g on T n F), c] for the gather indices outside the tile.
bool vdag_route_projected(aqc_ws* ws, int x_buf) {
    ProjRoute& pr = ws->proj;
    if (!pr.ok || !ws->proj_vdag_enabled || ws->capturing || (long long)ws->lane_elems * ws->batch < ws->proj_vdag_min_elems) return false;
    const unsigned long long key[3] = {(unsigned long long)x_buf, ws->supp_version[x_buf], ws->gather_gen};
    if (key[0] == ws->projb_key[0] && key[1] == ws->projb_key[1] && key[2] == ws->projb_key[2]) return ws->projb_ok;
    for (int i = 0; i < 3; ++i) ws->projb_key[i] = key[i];
    ws->projb_ok = false;
    const std::vector<long long>& el = ws->combo_last_elem[x_buf];
    const std::vector<double>& cf = ws->combo_last_coef[x_buf];
    const size_t B = (size_t)ws->batch;
    if (el.size() != 2 * B || cf.size() != 4 * B || (ws->gather_count > 0 && ws->h_gather.size() != (size_t)ws->gather_count)) return false;
    const unsigned fmask = ~pr.l0_mask & (unsigned)((1ull << ws->ctx->prog.n) - 1);
    for (size_t b = 0; b < B; ++b) {
        if (el[2 * b] < 0 || el[2 * b + 1] >= 0 || cf[4 * b] != 1.0 || cf[4 * b + 1] != 0.0) return false;
        const unsigned e = (unsigned)el[2 * b];
        for (long long gl : ws->h_gather) {
            const unsigned d = e ^ (unsigned)gl;
            if ((d & fmask) == 0) continue;                                // inside the lane's tile
            if ((d & pr.l0_mask) != 0 || (d & pr.ff_mask) != 0) return false;   // another psi, or another slice of y
        }
    }
    ws->projb_ok = true;
    return true;
}

static int virtual_apply(aqc_ws* ws, DevPlan& v, const double2* src, double2* dst) {
    ProjRoute& pr = ws->proj;
    const size_t m = v.h_stages.size();
    const int ntiles = 2 * pr.ntiles_v;
    for (size_t s = 0; s < m; ++s) {
        Stage3Args a;
        memset(&a, 0, sizeof a);
        a.stage = v.h_stages[s];
        a.subs = v.d_subs3;
        a.umat = v.d_umat;
        a.nsubs_total = (int)v.h_subs3.size();
        a.lane_stride = 2ull << pr.nvp;
        a.ntiles = ntiles;
        a.batch = ws->batch;
        a.in0 = s == 0 ? src : dst;
        a.out0 = dst;
        a.items = pr.d_items; a.nitems = pr.d_count; a.max_items = 2 * ws->batch * pr.ntiles_v;
        ProfScope ps(ws, AQC_K_APPLY_VIRTUAL);
        HIP_OK(launch_apply3(ntiles, ws->batch, pr.kv, ws->stream, a));
    }
    return 0;
}

int run_vdag_projected(aqc_ws* ws, int x_buf) {   // the caller has asked vdag_route_projected, sweep_sparse_prepare
    ProjRoute& pr = ws->proj;
    DevPlan& p = ws->sweep;
    if (ensure_umat(ws, ws->inv)) return 1;
    if (!ws->d_sw_items || !ws->w_clean) return fail("objective by projection without the sparse route's preparation");
    if (ensure_sweep_items(ws, x_buf)) return 1;
    {   // psi: the first stage's gates on the basis index, on the listed tiles (x -> W)
        Stage3Args a;
        memset(&a, 0, sizeof a);
        a.stage = p.h_stages[0];
        a.subs = p.d_subs3;
        a.umat = p.d_umat;
        a.nsubs_total = (int)p.h_subs3.size();
        a.lane_stride = ws->lane_elems;
        a.ntiles = p.ntiles;
        a.batch = ws->batch;
        a.in0 = ws->bufs[x_buf];
        a.out0 = ws->bufs[AQC_BUF_W];
        a.items = ws->d_sw_items; a.nitems = ws->d_sw_counts; a.max_items = 2 * ws->batch;
        ProfScope ps(ws, AQC_K_APPLY_LIST);
        HIP_OK(launch_apply3(p.ntiles, ws->batch, p.k, ws->stream, a));
    }
    ProjArgs a = proj_args(ws);
    if (ensure_pattern(ws, a)) return 1;
    if (virtual_apply(ws, pr.vsw, pr.vm, pr.vme)) return 1;   // M_end
    const unsigned* off_us = pr.d_tab + (1u << pr.t);
    if (ws->proj_fused_enabled && pr.us <= 10 && pr.cb <= 4 && (pr.us <= 8 || pr.vy_copies == 1 << (pr.us - 8))) {   // both products from one fetch of the target
        ProjArgs q = a;
        q.part_stride = (size_t)ws->batch * (2ull << pr.nvp);
        q.cpart = pr.cpart; q.cpart_shares = pr.cpart_shares;
        q.y = ws->bufs[AQC_BUF_Y]; q.s = ws->bufs[AQC_BUF_W];
        ProfScope ps(ws, AQC_K_PROJECT);
        HIP_OK(launch_project_fused(q, pr.vme, ws->bufs[AQC_BUF_ZW], pr.vy, ws->stream));
    } else {
    {   // Y_end = proj(y)
        ProjArgs q = a;
        q.y = ws->bufs[AQC_BUF_Y]; q.s = ws->bufs[AQC_BUF_W]; q.out = pr.vy;
        q.s_virtual = 0; q.out_virtual = 1; q.staged = 1;
        q.keep_bits = pr.t; q.k_bits = pr.us;
        q.y_keep = ProjMap{pr.d_tab, 0}; q.y_k = ProjMap{off_us, 0};
        q.s_k = ProjMap{off_us, 0}; q.s_c = ProjMap{a.off_cb, 0};
        q.o_keep = ProjMap{nullptr, 0}; q.o_c = ProjMap{nullptr, pr.t};
        ProfScope ps(ws, AQC_K_PROJECT);
        HIP_OK(launch_project(q, ws->stream));
    }
    {   // the lhs tile of (later stages)^H y, into ZW
        ProjArgs q = a;
        q.y = ws->bufs[AQC_BUF_Y]; q.s = pr.vme; q.out = ws->bufs[AQC_BUF_ZW];
        q.s_virtual = 1; q.out_virtual = 0; q.staged = 0;
        q.keep_bits = pr.us; q.k_bits = pr.t;
        q.y_keep = ProjMap{off_us, 0}; q.y_k = ProjMap{pr.d_tab, 0};
        q.s_k = ProjMap{nullptr, 0}; q.s_c = ProjMap{nullptr, pr.t};
        q.o_keep = ProjMap{off_us, 0}; q.o_c = ProjMap{a.off_cb, 0};
        ProfScope ps(ws, AQC_K_PROJECT);
        HIP_OK(launch_project(q, ws->stream));
    }
    }
    if (virtual_apply(ws, pr.vinv, pr.vy, pr.vy)) return 1;   // Y_0
    {   // V^H's last stage on the lhs tiles: ZW -> Z
        DevPlan& iv = ws->inv;
        Stage3Args s3;
        memset(&s3, 0, sizeof s3);
        s3.stage = iv.h_stages.back();
        s3.subs = iv.d_subs3;
        s3.umat = iv.d_umat;
        s3.nsubs_total = (int)iv.h_subs3.size();
        s3.lane_stride = ws->lane_elems;
        s3.ntiles = iv.ntiles;
        s3.batch = ws->batch;
        s3.in0 = ws->bufs[AQC_BUF_ZW];
        s3.out0 = ws->bufs[AQC_BUF_Z];
        s3.items = ws->d_sw_items; s3.nitems = ws->d_sw_counts; s3.max_items = 2 * ws->batch;
        ProfScope ps(ws, AQC_K_APPLY_LIST);
        HIP_OK(launch_apply3(iv.ntiles, ws->batch, iv.k, ws->stream, s3));
    }
    vdag_restricted_state_after(ws, x_buf);
    ws->ckpt_valid = false;         // ZW holds the lhs tiles of the checkpoint only
    ws->z_gather_gen = ~0ull;       // Z covers the lhs tiles, not the gather set: a later gather completes it first
    ws->z_from_y = true;
    ws->proj_y0_ready = true;
    return 0;
}

// the registered gather of a Z that run_vdag_projected has left: the indices inside the lhs tiles from Z, the others from the virtual z
int proj_fix_amplitudes(aqc_ws* ws, int x_buf) {
    ProjRoute& pr = ws->proj;
    if (results_guard(ws)) return 1;
    ProfScope ps(ws, AQC_K_MISC);
    ProjArgs a = proj_args(ws);
    HIP_OK(launch_project_amps(a, ws->d_index, ws->gather_count, ws->d_combo_prev[x_buf], ws->d_small, pr.vy, ws->bufs[AQC_BUF_Z], ws->stream));
    return 0;
}

}  // namespace aqc
