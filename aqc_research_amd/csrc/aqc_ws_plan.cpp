// C ABI (include/aqc_hip.h): lowering of stage plans to the device tables the kernels read.
#include "aqc_ws.h"

#include <algorithm>

using namespace aqc;

namespace aqc {

namespace {
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

}  // namespace

void lower_plan(const Program& prog, const Plan& plan, DevPlan& out, int reg_bits, bool with_dots, bool mfma, bool presplit, int beam_width) {
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
        if (!presplit) split_substages(prog, out.plan, 4, 1 << 30, beam_width);   // a sub-stage is one 16 x 16 unitary: any number of groups
    } else if (out.v2) {
        int max_ops = reg_bits == 4 ? kMaxOpsPerSub : kMaxOpsPerSub / 2;
        if (prog.entangler == 2) max_ops /= 2;   // CP: two reductions per block
        split_substages(prog, out.plan, reg_bits, max_ops);
    }
    out.k = (int)plan.stages.front().bits.size();
    out.ntiles = 1 << (plan.nbits - out.k);
    uint64_t touched_global = 0;   // address bits some earlier stage had local (sparse-lhs bookkeeping, see DevSub3::skipinfo)
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
            if (local_of[b] < 0) {
                ds.ubits[ds.nub++] = b;
                if (!(touched_global >> b & 1)) ds.fresh_nonlocal |= 1u << b;
            }
        uint32_t touched_local = 0;   // local positions an earlier sub-stage of this stage had as register bits
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
                    {   // fresh bits of this sub-stage (see DevSub3::skipinfo)
                        auto fresh = [&](int local_pos) { return !(touched_global >> st.bits[local_pos] & 1) && !(touched_local >> local_pos & 1); };
                        uint32_t info = 0;
                        for (int i = 0; i < 4 && 4 + i < nc; ++i) {   // group-index bit i <-> chunk bit 4 + i
                            if (fresh(cbits[4 + i])) info |= 1u << i;
                            info |= (uint32_t)cbits[4 + i] << (8 + 4 * i);
                        }
                        for (int i = 0; i < dsub.nbits && i < 4; ++i)
                            if (fresh(dsub.bits[i])) info |= 1u << (4 + i);
                        if (dsub.nbits == 4) info |= (uint32_t)dsub.bits[2] << 24 | (uint32_t)dsub.bits[3] << 28;
                        else info &= ~0xC0u;   // (fewer than 4 register bits: no K-step selection)
                        d3.skipinfo = info;
                        for (int i = 0; i < dsub.nbits; ++i) touched_local |= 1u << dsub.bits[i];
                    }
                    out.h_subs3.push_back(d3);
                }
            }
        }
        out.h_stages.push_back(ds);
        for (int b : st.bits) touched_global |= 1ull << b;
    }
}

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

// The plan of V^H that walks a forward plan backwards: stages in reverse order with the same local bits, sub-stages in reverse
// order with the same register bits, the groups of every sub-stage in reverse order.  Per-qubit program order of the inverse
// program is the reverse of the forward one, so the result is valid whenever the forward plan is (check_plan), and the state
// between its stages j and j + 1 is what the forward plan's stage (m - 1 - j) would be handed from the stage before it.
Plan mirror_plan(const Plan& plan) {
    Plan out = plan;
    out.inverse = !plan.inverse;
    std::reverse(out.stages.begin(), out.stages.end());
    for (Stage& st : out.stages) {
        std::reverse(st.ops.begin(), st.ops.end());
        std::reverse(st.subs.begin(), st.subs.end());
        for (SubStage& sub : st.subs) std::reverse(sub.ops.begin(), sub.ops.end());
    }
    return out;
}

}  // namespace aqc
