// C ABI (include/aqc_hip.h): V / V^H launches, the w/z sweep and the one-call evaluation.
//
// Matrix-core family, two routes for the sweep (grad_of_dot_product, core_operations.py:823-1019):
//   dense   every stage over all tiles of w and z, in place on the scratch pair (W, ZW) -- any lhs state, any V^H y.
//   sparse  the lhs state is a combination of <= 2 basis states per lane (aqc_ws_set_basis / aqc_ws_set_combo / the device
//           L-BFGS: what every surrogate objective sweeps from, objective_base.py:42-255, objective_lhs_sur_max.py:147-191) and
//           Z = V^H y was produced here, by the MIRRORED plan (the sweep's stages walked backwards).  Then
//           * during the sweep's first stage w is zero outside the tile(s) of the stage that hold the basis indices: the
//             stage runs over those tiles only (a device-resident item list; everywhere else W <- U 0 = 0 and R = Z 0^H = 0
//             exactly, so nothing is lost -- the buffer W is kept zero outside the listed tiles);
//           * z after the sweep's first stage is (stage-0 gates) V^H y = the state V^H had BEFORE its last stage, which the
//             mirrored V^H left in ZW (the "checkpoint"): the second stage reads it from there, and the first stage neither
//             computes nor stores z outside its listed tiles.
//           At the headline (16 qubits, 2 stages of 7 + 4 sub-stages, x = |0>) the first stage shrinks from 16 tiles per lane
//           to one: 7/11 of the sweep's matrix work is gone.
#include "aqc_ws.h"

#include <algorithm>

using namespace aqc;

namespace aqc {

// family 3: the 16 x 16 unitaries of the plan's sub-stages for the coefficients in use
// Jobs are laid out [V^H | sweep | V]: the objective+gradient path (V^H then the sweep) is built by one launch.
int ensure_umat(aqc_ws* ws, DevPlan& p) {
    if (!p.v3 || p.u_valid) return 0;
    const int T = ws->ctx->prog.num_thetas();
    const int ninv = ws->inv.v3 && !ws->ujobs_mirror ? (int)ws->inv.h_subs3.size() : 0;
    int nsw = ws->sweep.v3 ? (int)ws->sweep.h_subs3.size() : 0;
    const int nfwd = ws->fwd.v3 ? (int)ws->fwd.h_subs3.size() : 0;
    const int nvs = ws->proj.ok ? (int)ws->proj.vsw.h_subs3.size() : 0;   // (the virtual sweep's jobs follow the sweep's: one launch builds both)
    nsw += nvs;
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
    ws->ckpt_valid = false;   // ZW (and Z) belong to the previous thetas
    ws->z_from_y = false;
    ws->proj_y0_ready = false;
    if (ws->fwd.v3 && ws->inv.v3 && ws->sweep.v3 && !ws->need_coef) return 0;   // the matrix-core path reads the thetas directly
    ProfScope ps(ws, AQC_K_COEF);
    HIP_OK(launch_coef(ws->d_thetas, ws->d_coef, prog.n, prog.num_blocks, prog.tpb, prog.tail_blocks, ws->batch, ws->stream));
    return 0;
}

// somebody other than the V^H / sweep pair below is about to write ALL of buffer `buf`
void touch_buf(aqc_ws* ws, int buf) {
    if (buf == AQC_BUF_ZW && !ws->z_full) (void)ensure_z_full(ws, false);   // the checkpoint goes away: complete Z while it is there
    if (buf == AQC_BUF_Z || buf == AQC_BUF_ZW) ws->ckpt_valid = false;
    if (buf == AQC_BUF_Y || buf == AQC_BUF_Z) ws->z_from_y = false;
    ws->proj_y0_ready = false;
    if (buf == AQC_BUF_Z) ws->z_full = true;
    if (buf == AQC_BUF_W) ws->w_clean = false;
}

namespace {

#ifdef AQC_TUNING   // AQC_STAMPS=1: mean cycles per phase of the workgroups of one launch, on stderr (tuning builds only)
unsigned long long* g_stamps = nullptr;
int stamps_begin(aqc_ws* ws, Stage3Args& a, size_t nwg) {
    if (env_int("AQC_STAMPS", 0) == 0 || nwg > 65536) return 0;
    if (!g_stamps) HIP_OK(hipMalloc((void**)&g_stamps, sizeof(unsigned long long) * 65536 * kStampSlots));
    HIP_OK(hipMemsetAsync(g_stamps, 0, sizeof(unsigned long long) * nwg * kStampSlots, ws->stream));
    a.stamps = g_stamps;
    return 0;
}
int stamps_apply(aqc_ws* ws, const Stage3Args& a, size_t s, size_t nwg) {
    if (!a.stamps) return 0;
    std::vector<unsigned long long> h(nwg * kStampSlots);
    HIP_OK(hipStreamSynchronize(ws->stream));
    HIP_OK(hipMemcpy(h.data(), g_stamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double load = 0, loop = 0, store = 0, bar = 0;
    const int ns = a.stage.nsubs;
    for (size_t w = 0; w < nwg; ++w) {
        const unsigned long long* t = h.data() + w * kStampSlots;
        load += (double)(t[1] - t[0]); loop += (double)(t[2] - t[1]); store += (double)(t[3] - t[2]);
        for (int i = 0; i < ns && 5 + i < kStampSlots; ++i) bar += (double)(t[5 + i] - t[4 + i]);
    }
    fprintf(stderr, "aqc_hip stamps: V/V^H stage %zu (%d sub-stages, %zu workgroups): load %.0f + sub-stage loop %.0f (%.0f per sub-stage, of which "
            "waiting at its barrier %.0f) + store %.0f cycles per workgroup\n", s, ns, nwg, load / nwg, loop / nwg, loop / nwg / std::max(ns, 1),
            bar / nwg / std::max(ns, 1), store / nwg);
    return 0;
}
int stamps_sweep(aqc_ws* ws, const Stage3Args& a, size_t s, size_t nwg) {
    if (!a.stamps) return 0;
    std::vector<unsigned long long> h(nwg * kStampSlots);
    HIP_OK(hipStreamSynchronize(ws->stream));
    HIP_OK(hipMemcpy(h.data(), g_stamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    const int ns = a.stage.nsubs;
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
    return 0;
}
#endif

Stage3Args stage3_args(aqc_ws* ws, const DevPlan& p, size_t s) {
    Stage3Args a;
    memset(&a, 0, sizeof a);
    a.stage = p.h_stages[s];
    a.subs = p.d_subs3;
    a.umat = p.d_umat;
    a.nsubs_total = (int)p.h_subs3.size();
    a.lane_stride = ws->lane_elems;
    a.ntiles = p.ntiles;
    a.batch = ws->batch;
    return a;
}

}  // namespace

// V^H into Z by the mirrored plan keeps the state before its last stage in ZW (see the head of this file)
static bool keeps_checkpoint(const aqc_ws* ws, bool inverse, int src_buf, int dst_buf) {
    return inverse && ws->inv_mirrored && ws->inv.v3 && dst_buf == AQC_BUF_Z && src_buf != AQC_BUF_ZW && ws->inv.h_stages.size() >= 2;
}
void apply_state_after(aqc_ws* ws, bool inverse, int src_buf, int dst_buf) {   // host-side state a V / V^H leaves (also after a graph replay)
    touch_buf(ws, dst_buf);
    if (keeps_checkpoint(ws, inverse, src_buf, dst_buf)) { touch_buf(ws, AQC_BUF_ZW); ws->ckpt_valid = true; }
}

int run_apply(aqc_ws* ws, bool inverse, int src_buf, int dst_buf) {
    DevPlan& p = inverse ? ws->inv : ws->fwd;
    const Program& prog = ws->ctx->prog;
    if (src_buf == AQC_BUF_Z && ensure_z_full(ws, true)) return 1;
    if (p.v3) {
        if (ensure_umat(ws, p)) return 1;
        const bool keep = keeps_checkpoint(ws, inverse, src_buf, dst_buf);
        const size_t m = p.h_stages.size();
        for (size_t s = 0; s < m; ++s) {
            Stage3Args a = stage3_args(ws, p, s);
            const int mid = keep ? AQC_BUF_ZW : dst_buf;   // where the stages before the last one work
            a.in0 = s == 0 ? ws->bufs[src_buf] : ws->bufs[mid];
            a.out0 = s + 1 == m ? ws->bufs[dst_buf] : ws->bufs[mid];
#ifdef AQC_TUNING
            const size_t nwg = (size_t)p.ntiles * ws->batch;
            if (stamps_begin(ws, a, nwg)) return 1;
#endif
            {
                ProfScope ps(ws, AQC_K_APPLY);
                HIP_OK(launch_apply3(p.ntiles, ws->batch, p.k, ws->stream, a));
            }
#ifdef AQC_TUNING
            if (stamps_apply(ws, a, s, nwg)) return 1;
#endif
        }
        apply_state_after(ws, inverse, src_buf, dst_buf);
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
    apply_state_after(ws, inverse, src_buf, dst_buf);
    return 0;
}

// ---- V^H where the objective reads it ------------------------------------------------------------------------------
// An objective+gradient evaluation reads Z = V^H y in two places: the amplitudes <state_i|V^H y> of the registered gather set
// (objective_lhs_sur_max.py:99-106) and -- sparse route -- the first-stage tiles of the sweep that hold the lhs state.  The
// last stage of the mirrored V^H therefore runs over those tiles only (at the headline 5 of 16 per lane: the tile of |0> and
// of its four flips on qubits 12..15); everything before it is needed in full (it is the second sweep stage's z).  Z is
// completed on demand -- the checkpoint in ZW is all it takes -- as long as the thetas have not changed.
bool vdag_route_restricted(const aqc_ws* ws, int x_buf) {
    return ws->lazy_z_enabled && sweep_route_sparse(ws, x_buf, true) && 2 + ws->gather_count <= kMaxTileCands;
}
static Stage3Args last_vdag_stage(aqc_ws* ws) {
    DevPlan& p = ws->inv;
    Stage3Args a = stage3_args(ws, p, p.h_stages.size() - 1);
    a.in0 = ws->bufs[AQC_BUF_ZW];
    a.out0 = ws->bufs[AQC_BUF_Z];
    return a;
}
// support_in_gather_set: the lhs state is picked among the registered gather indices (surrogate objective: |state_0> and the
// leading flip state) -- the gather set alone names the tiles, and the list only changes when that set does
int run_vdag_restricted(aqc_ws* ws, int x_buf, bool support_in_gather_set) {   // Y -> Z; the caller has asked vdag_route_restricted and sweep_sparse_prepare
    DevPlan& p = ws->inv;
    if (ensure_umat(ws, p)) return 1;
    const size_t m = p.h_stages.size();
    const size_t per_lane = 2 + (size_t)ws->gather_count;
    if (!ws->d_vd_items || ws->vd_items_cap < per_lane * ws->batch) return fail("objective V^H inside a captured graph without its preparation");
    for (size_t s = 0; s + 1 < m; ++s) {
        Stage3Args a = stage3_args(ws, p, s);
        a.in0 = s == 0 ? ws->bufs[AQC_BUF_Y] : ws->bufs[AQC_BUF_ZW];
        a.out0 = ws->bufs[AQC_BUF_ZW];
        ProfScope ps(ws, AQC_K_APPLY);
        HIP_OK(launch_apply3(p.ntiles, ws->batch, p.k, ws->stream, a));
    }
    const bool gather_only = support_in_gather_set && ws->gather_count > 0;
    const unsigned long long key[3] = {gather_only ? ~0ull - 1 : (unsigned long long)x_buf, gather_only ? 0ull : ws->supp_version[x_buf], ws->gather_gen};
    if (ws->capturing || key[0] != ws->vd_key[0] || key[1] != ws->vd_key[1] || key[2] != ws->vd_key[2]) {   // (a static list is built once)
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_tile_items(p.h_stages[m - 1], gather_only ? nullptr : ws->d_combo_prev[x_buf], 2, ws->gather_count > 0 ? ws->d_index : nullptr,
                                 ws->gather_count, ws->batch, ws->d_vd_items, ws->d_sw_counts + 2, nullptr, nullptr, nullptr, nullptr, ws->stream));
        for (int i = 0; i < 3; ++i) ws->vd_key[i] = ws->capturing ? ~0ull : key[i];   // (a captured graph rebuilds it on every replay)
    }
    Stage3Args a = last_vdag_stage(ws);
    a.items = ws->d_vd_items;
    a.nitems = ws->d_sw_counts + 2;
    a.max_items = (int)(per_lane * ws->batch);
    ws->sw_lists_built |= 2;
    {
        ProfScope ps(ws, AQC_K_APPLY_LIST);
        HIP_OK(launch_apply3(p.ntiles, ws->batch, p.k, ws->stream, a));
    }
    vdag_restricted_state_after(ws, x_buf);
    return 0;
}
void vdag_restricted_state_after(aqc_ws* ws, int x_buf) {
    apply_state_after(ws, true, AQC_BUF_Y, AQC_BUF_Z);
    ws->z_full = false;
    ws->z_gather_gen = ws->gather_gen;
    ws->z_x_buf = x_buf;
    ws->z_x_version = ws->supp_version[x_buf];
}
int ensure_z_full(aqc_ws* ws, bool reader) {
    if (ws->z_full) return 0;
    if (ws->capturing) return fail("BUF_Z is partial inside a captured graph");
    if (ws->ckpt_valid && ws->inv.u_valid) {   // the last stage once more, over every tile (its inputs are all in ZW)
        Stage3Args a = last_vdag_stage(ws);
        ProfScope ps(ws, AQC_K_APPLY);
        HIP_OK(launch_apply3(ws->inv.ntiles, ws->batch, ws->inv.k, ws->stream, a));
        ws->z_full = true;
        return 0;
    }
    if (!reader) { ws->z_full = true; return 0; }   // a writer of parts of Z takes the buffer over
    if (ws->z_from_y && ws->inv.u_valid) {   // (objective by projection: no checkpoint -- the whole V^H once more, from Y)
        ws->z_full = true;
        return run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z);
    }
    return fail("BUF_Z holds V^H y only on the tiles the last one-call evaluation read, and the thetas (or ZW) have changed since: "
                "run aqc_ws_apply(inverse) for the whole vector");
}

void drop_graphs(aqc_ws* ws) {
    for (auto& kv : ws->graphs) (void)hipGraphExecDestroy(kv.second);
    ws->graphs.clear();
}

// ---- the sparse route --------------------------------------------------------------------------------------------
// Route of the next sweep from x_buf.  will_vdag: a V^H from Y into Z precedes it inside the same call (one-call
// evaluations decide before they enqueue anything: the decision is part of the key of their captured graph).
bool sweep_route_sparse(const aqc_ws* ws, int x_buf, bool will_vdag) {
    const DevPlan& p = ws->sweep;
    if (!ws->sparse_enabled || !p.v3 || !ws->inv_mirrored || p.h_stages.size() < 2) return false;
    if (!ws->combo_valid[x_buf] || !ws->d_combo_prev[x_buf]) return false;          // support of the lhs state known on the device
    if (!(will_vdag ? keeps_checkpoint(ws, true, AQC_BUF_Y, AQC_BUF_Z) : (ws->ckpt_valid || ws->proj_y0_ready))) return false;   // z of stage 1 available in ZW (or projected already)
    return (long)p.ntiles * ws->batch >= ws->sparse_min_items;   // (fewer items than CUs: a stage takes one item's time either way)
}
// The last sub-stage of the last stage is taken from its inputs alone (R = U (Z W^H) U^H, see sweep_mfma_kernel) unless it is also the
// FIRST sub-stage of a persistent stage, whose operands sit in the prefetch registers in the other layout, and unless the stage is
// long: the variant of the kernel that does it (explicit copies of the sub-stage loop) runs the other sub-stages 3 % slower, the
// saving is 2/3 of ONE sub-stage.  Returns the sub-stage's index over all stages, or -1.
int sweep_r_only_sub(const aqc_ws* ws) {
    const DevPlan& p = ws->sweep;
    if (!p.v3 || !ws->r_only_enabled || p.h_stages.empty()) return -1;
    const DevStage& last = p.h_stages.back();
    if (last.nsubs < 1 || last.nsubs > ws->r_only_max_subs || (p.k >= 12 && last.nsubs == 1)) return -1;
    return last.sub_begin + last.nsubs - 1;
}
// Inside a stage the same knowledge goes further (any number of stages, either route): see sweep_mfma_kernel<K, false, true>.
bool sweep_skips_zero_w(const aqc_ws* ws, int x_buf) {
    return ws->skipw_enabled && ws->sweep.v3 && ws->combo_valid[x_buf] && ws->d_combo_prev[x_buf] != nullptr;
}
// Allocations and one-off clears of the sparse route: everything that must not sit inside a captured graph.
int sweep_sparse_prepare(aqc_ws* ws) {
    const DevPlan& p = ws->sweep;
    const int B = ws->batch;
    if (!ws->d_sw_items) {
        HIP_OK(hipMalloc((void**)&ws->d_sw_items, sizeof(TileItem) * 2 * B));
        HIP_OK(hipMalloc((void**)&ws->d_sw_clear, sizeof(TileItem) * 2 * B));
        HIP_OK(hipMalloc((void**)&ws->d_sw_counts, sizeof(int) * 4));
        HIP_OK(hipMemsetAsync(ws->d_sw_counts, 0, sizeof(int) * 4, ws->stream));
        HIP_OK(hipMalloc((void**)&ws->d_sw_lane_parts, sizeof(int) * B));
        HIP_OK(hipMalloc((void**)&ws->d_sw_prev_tiles, sizeof(int) * 2 * B));
        ws->w_clean = false;
    }
    if (p.h_stages.size() >= 3 && !ws->w2 && !ws->proj.ok) {   // (the projected route runs the later stages on its own small register)   // stages from the second one on work on their own pair: W stays zero outside the listed
        HIP_OK(hipMalloc((void**)&ws->w2, sizeof(double2) * (size_t)B * ws->lane_elems));   // tiles, ZW keeps the checkpoint
        HIP_OK(hipMalloc((void**)&ws->zw2, sizeof(double2) * (size_t)B * ws->lane_elems));
    }
    const size_t vd_need = (2 + (size_t)ws->gather_count) * B;
    if (ws->lazy_z_enabled && ws->vd_items_cap < vd_need) {
        HIP_OK(hipStreamSynchronize(ws->stream));
        if (ws->d_vd_items) HIP_OK(hipFree(ws->d_vd_items));
        ws->d_vd_items = nullptr; ws->vd_items_cap = 0;
        HIP_OK(hipMalloc((void**)&ws->d_vd_items, sizeof(TileItem) * vd_need));
        ws->vd_items_cap = vd_need;
        ws->vd_key[0] = ws->vd_key[1] = ws->vd_key[2] = ~0ull;
    }
    if (!ws->w_clean) {
        HIP_OK(hipMemsetAsync(ws->bufs[AQC_BUF_W], 0, sizeof(double2) * (size_t)B * ws->lane_elems, ws->stream));
        HIP_OK(hipMemsetAsync(ws->d_sw_prev_tiles, 0xff, sizeof(int) * 2 * B, ws->stream));   // -1: W holds no tile of an earlier list
        ws->w_clean = true;
        ws->sw_items_buf = -1;
    }
    return 0;
}
void sweep_state_after(aqc_ws* ws, bool sparse, bool replayed) {   // host-side state a sweep leaves (also after a graph replay)
    if (sparse) {
        if (replayed) ws->sw_items_buf = -1;   // the replay rebuilt the list from whatever the support was: rebuild when asked next
    } else {
        ws->w_clean = false;
        ws->ckpt_valid = false;                // the dense route works in place on (W, ZW)
    }
}

}  // namespace aqc

extern "C" {

int aqc_ws_apply(aqc_ws* ws, int inverse, int src_buf, int dst_buf) {
    if (check_buf(ws, src_buf) || check_buf(ws, dst_buf)) return 1;
    ws->combo_valid[dst_buf] = false;
    if (ensure_coef(ws)) return 1;
    HIP_OK(hipSetDevice(ws->device));
    return run_apply(ws, inverse != 0, src_buf, dst_buf);
}

int aqc_ws_sweep_r_only_sub(aqc_ws* ws) { return ws ? sweep_r_only_sub(ws) : -1; }

int aqc_ws_grad(aqc_ws* ws, int block_from, int block_to, int front_layer) {
    return aqc_ws_grad_from(ws, AQC_BUF_X, block_from, block_to, front_layer);
}

int aqc_ws_grad_from(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer) {
    return grad_from_impl(ws, x_buf, block_from, block_to, front_layer, false);
}

}  // extern "C"

namespace aqc {

// tiles of the first stage that hold the lhs state: a device-side list (the support may have been chosen on the device),
// rebuilt when the support changed; tiles of the previous list that the new one drops are zeroed in W
int ensure_sweep_items(aqc_ws* ws, int x_buf) {
    const DevPlan& p = ws->sweep;
    if (ws->capturing || ws->sw_items_buf != x_buf || ws->sw_items_version != ws->supp_version[x_buf]) {
        ProfScope ps(ws, AQC_K_MISC);
        HIP_OK(launch_tile_items(p.h_stages[0], ws->d_combo_prev[x_buf], 2, nullptr, 0, ws->batch, ws->d_sw_items, ws->d_sw_counts,
                                 ws->d_sw_lane_parts, ws->d_sw_prev_tiles, ws->d_sw_clear, ws->d_sw_counts + 1, ws->stream));
        HIP_OK(launch_clear_tiles(p.h_stages[0], ws->bufs[AQC_BUF_W], ws->lane_elems, ws->d_sw_clear, ws->d_sw_counts + 1, 2 * ws->batch, ws->stream));
        ws->sw_items_buf = x_buf;
        ws->sw_items_version = ws->supp_version[x_buf];
        ws->sw_lists_built |= 1;
    }
    return 0;
}

// support_in_gather_set: the lhs state was chosen on the device among the registered gather indices (surrogate objective)
int grad_from_impl(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer, bool support_in_gather_set) {
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
        const size_t m = p.h_stages.size();
        const bool sparse = sweep_route_sparse(ws, x_buf, false);
        const bool skipw = sweep_skips_zero_w(ws, x_buf);
        const bool projected = sweep_route_projected(ws, sparse);   // the stages after the first on the virtual register (aqc_ws_project.cpp)
        int r_only_sub = skipw || projected ? -1 : sweep_r_only_sub(ws);   // (the zero-w variant of the kernel has no R-only form)
        // objective by projection: psi is in W already and nobody reads the first stage's z': ITS last sub-stage is the R-only one
        const bool first_stage_r_only = projected && ws->proj_y0_ready && ws->r_only_enabled && p.h_stages[0].nsubs >= 2 &&
                                        p.h_stages[0].nsubs <= ws->r_only_max_subs;
        if (first_stage_r_only) r_only_sub = p.h_stages[0].sub_begin + p.h_stages[0].nsubs - 1;
        // a partial Z covers the sparse route's reads when its tiles were chosen for this lhs state (or for a gather set the
        // state was picked from); anything else reads all of Z
        if (!ws->z_full && !(sparse && ((support_in_gather_set && ws->z_gather_gen == ws->gather_gen) ||
                                        (ws->z_x_buf == x_buf && ws->z_x_version == ws->supp_version[x_buf]))) && ensure_z_full(ws, true))
            return 1;
        if (sparse) {
            if (!ws->capturing && sweep_sparse_prepare(ws)) return 1;
            if (!ws->d_sw_items || !ws->w_clean || (m >= 3 && !ws->w2 && !projected)) return fail("sparse sweep inside a captured graph without its preparation");
            if (ensure_sweep_items(ws, x_buf)) return 1;
        }
        for (size_t s = 0; s < (projected ? 1 : m); ++s) {
            Stage3Args a = stage3_args(ws, p, s);
            if (sparse) {
                // stage 0: (x, Z) on the listed tiles -> W there; stage 1: (W, checkpoint in ZW) -> the second pair; then in place
                double2* w2 = m >= 3 ? ws->w2 : nullptr;
                double2* z2 = m >= 3 ? ws->zw2 : nullptr;
                a.in0 = s == 0 ? ws->bufs[x_buf] : (s == 1 ? ws->bufs[AQC_BUF_W] : w2);
                a.in1 = s == 0 ? ws->bufs[AQC_BUF_Z] : (s == 1 ? ws->bufs[AQC_BUF_ZW] : z2);
                a.out0 = s == 0 ? ws->bufs[AQC_BUF_W] : w2;
                a.out1 = s == 0 ? nullptr : z2;
                a.store_out = s + 1 < m ? (s == 0 ? 1 : 3) : 0;
                if (s == 0) { a.items = ws->d_sw_items; a.nitems = ws->d_sw_counts; a.max_items = 2 * ws->batch; }
                if (s == 0 && first_stage_r_only) { a.r_only_last = 1; a.store_out = 0; }
            } else {
                a.in0 = s == 0 ? ws->bufs[x_buf] : ws->bufs[AQC_BUF_W];
                a.in1 = s == 0 ? ws->bufs[AQC_BUF_Z] : ws->bufs[AQC_BUF_ZW];
                a.out0 = ws->bufs[AQC_BUF_W];
                a.out1 = ws->bufs[AQC_BUF_ZW];
                a.store_out = s + 1 < m ? 3 : 0;
            }
            if (skipw && !a.items) a.supp = ws->d_combo_prev[x_buf];
            if (s + 1 == m && r_only_sub >= 0) a.r_only_last = 1;
            a.rpart = p.d_rpart;
            a.chunk = sweep3_chunk(p.ntiles, ws->batch, p.k);
            a.nparts = sweep3_nparts(p.ntiles, ws->batch, p.k);
            if (a.stage.nsubs > 0) stage3_first_offsets(a, p.h_subs3[a.stage.sub_begin]);
#ifdef AQC_TUNING
            const size_t nwg = (size_t)p.ntiles * ws->batch;
            a.debug = env_int("AQC_DEBUG_SKIP", 0);
            if (stamps_begin(ws, a, nwg)) return 1;
#endif
            {
                ProfScope ps(ws, a.items ? AQC_K_SWEEP_LIST : AQC_K_SWEEP);
                HIP_OK(launch_sweep3(p.ntiles, ws->batch, p.k, ws->stream, a));
            }
#ifdef AQC_TUNING
            if (stamps_sweep(ws, a, s, nwg)) return 1;
#endif
        }
        if (projected && run_projected_stages(ws)) return 1;
        sweep_state_after(ws, sparse, false);
        ProfScope ps(ws, AQC_K_FINALIZE);
        RgradSecond vwalk;   // projected route: the virtual plan's walk in the same launch (two launches: 54 + 54 us at the headline, one: ~70)
        if (projected) vwalk = projected_rgrad_plan(ws);
        HIP_OK(launch_rgrad(p.d_subs3, p.d_grps, prog.entangler, ws->d_thetas, prog.num_thetas(), p.d_rpart, p.ntiles, nsubs, ws->d_partial,
                            ws->nslots, block_from, block_to, front_layer ? 1 : 0, ws->batch, ws->stream,
                            ws->grads_direct ? ws->d_slot_theta : nullptr, ws->d_grads, ws->mirror_grads,
                            ws->gather_rides ? GatherJob{ws->bufs[AQC_BUF_Z], ws->lane_elems, ws->d_index, ws->gather_count, ws->d_small, ws->mirror_small}
                                             : GatherJob{nullptr, 0, nullptr, 0, nullptr, nullptr},
                            sweep3_nparts(p.ntiles, ws->batch, p.k), sweep3_chunk(p.ntiles, ws->batch, p.k),
                            sparse ? p.h_stages[0].nsubs : 0, sparse ? ws->d_sw_lane_parts : nullptr, r_only_sub, p.d_umat,
                            projected ? p.h_stages[0].nsubs : -1, projected ? &vwalk : nullptr));
#ifdef AQC_TUNING
        if (env_int("AQC_STAMPS", 0) != 0) { HIP_OK(hipStreamSynchronize(ws->stream)); rgrad_print_stamps(nsubs); }
#endif
        if (!ws->grads_direct)   // some theta collects two slots (2nd-order Trotter half-layers, core_operations.py:966-968)
            HIP_OK(launch_finalize(ws->d_partial, ws->d_theta_slots, ws->d_slot_ntiles, ws->d_grads, prog.num_thetas(), ws->nslots,
                                   1, prog.n, prog.tpb, block_from, block_to, front_layer ? 1 : 0, ws->batch, ws->stream, ws->mirror_grads));
        return 0;
    }
    if (ensure_z_full(ws, true)) return 1;
    touch_buf(ws, AQC_BUF_W);
    touch_buf(ws, AQC_BUF_ZW);
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

}  // namespace aqc

extern "C" {

// Z = V^H Y where the objective reads it, the registered gather (if any), the sweep from x_buf: enqueued, not waited for.
// What a driver that keeps its thetas on the device (aqc_ws_use_theta_set) calls per evaluation, followed by
// aqc_ws_results_async.  Equivalent to aqc_ws_apply(1, Y, Z); aqc_ws_gather_launch(Z); aqc_ws_grad_from(x_buf, ...), except that
// Z may be left partial (completed on demand, see ensure_z_full).
int aqc_ws_objective_launch(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer) {
    if (check_buf(ws, x_buf)) return 1;
    if (x_buf == AQC_BUF_W || x_buf == AQC_BUF_ZW || x_buf == AQC_BUF_Z || x_buf == AQC_BUF_Y) return fail("lhs buffer must be X or X2");
    if (ensure_coef(ws)) return 1;
    HIP_OK(hipSetDevice(ws->device));
    const bool sparse = sweep_route_sparse(ws, x_buf, true);
    if (sparse && sweep_sparse_prepare(ws)) return 1;
    const bool lazy = sparse && vdag_route_restricted(ws, x_buf);
    const bool by_projection = lazy && vdag_route_projected(ws, x_buf);
    if (by_projection) { if (run_vdag_projected(ws, x_buf)) return 1; }
    else if (lazy) { if (run_vdag_restricted(ws, x_buf)) return 1; }
    else if (run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)) return 1;
    if (ws->gather_count > 0 && (by_projection ? proj_fix_amplitudes(ws, x_buf) : aqc_ws_gather_launch(ws, AQC_BUF_Z))) return 1;
    return grad_from_impl(ws, x_buf, block_from, block_to, front_layer, false);
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
    // route of the sweep (decided before anything is enqueued: new thetas invalidate the checkpoint, a V^H in this call renews it)
    const bool sparse = grads && (do_vdag ? sweep_route_sparse(ws, x_buf, true) : (!thetas && sweep_route_sparse(ws, x_buf, false)));
    if (sparse && sweep_sparse_prepare(ws)) return 1;
    const bool lazy = sparse && do_vdag && vdag_route_restricted(ws, x_buf);   // V^H only where this call (gather, sweep) reads it
    // ... and by two passes over y instead of its stages where the lhs state and the gather set allow (never with a riding gather)
    const bool by_projection = lazy && grads && !(gathered && zero_copy && ws->sweep.v3) && vdag_route_projected(ws, x_buf);
    if (!do_vdag && (gathered || grads) && ensure_z_full(ws, true)) return 1;  // a partial Z left by an earlier call is completed here,
                                                                              // outside whatever graph is captured below
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
        if (do_vdag && (by_projection ? run_vdag_projected(ws, x_buf) : (lazy ? run_vdag_restricted(ws, x_buf) : run_apply(ws, true, AQC_BUF_Y, AQC_BUF_Z)))) return 1;
        // with a gradient in the same call the gather (it only reads Z, which the sweep leaves intact) rides along as one
        // extra workgroup per lane of the gradient-walk kernel: one node less on the single-evaluation critical path
        const bool ride = gathered && grads && zero_copy && ws->sweep.v3;
        if (gathered && !ride) {
            if (by_projection ? proj_fix_amplitudes(ws, x_buf) : aqc_ws_gather_launch(ws, AQC_BUF_Z)) return 1;
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
                                            (long long)ws->gather_count, (long long)(size_t)ws->d_small, (long long)(size_t)ws->h_pin,
                                            (sparse ? 1 : 0) + (lazy ? 2 : 0) + (grads && sweep_skips_zero_w(ws, x_buf) ? 4 : 0) + (by_projection ? 8 : 0),
                                            (long long)(size_t)ws->d_combo_prev[x_buf],
                                            (long long)(size_t)ws->d_vd_items};
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
        ws->ckpt_valid = false;
        if (do_vdag) {
            if (lazy) vdag_restricted_state_after(ws, x_buf); else apply_state_after(ws, true, AQC_BUF_Y, AQC_BUF_Z);
            if (by_projection) { ws->ckpt_valid = false; ws->z_from_y = true; ws->z_gather_gen = ~0ull; }   // (what run_vdag_projected leaves)
        }
        if (grads) sweep_state_after(ws, sparse, true);
        HIP_OK(hipGraphLaunch(it->second, ws->stream));
    } else if (enqueue()) {
        return 1;
    }
    HIP_OK(hipStreamSynchronize(ws->stream));
    if (gathered) memcpy(gathered, pin_sm, sizeof(double2) * nsm);
    if (grads) memcpy(grads, pin_gr, sizeof(double2) * nth);
    return 0;
}

}  // extern "C"
