#include "aqc_plan.h"

#include <algorithm>
#include <cstdio>
#include <map>
#include <mutex>

namespace aqc {

std::string build_program(int n, int entangler, const int32_t* blocks, int L, bool trotter,
                          bool second_order, Program& out) {
    if (n < 2 || n > 30) return "number of qubits must be in [2, 30]";
    if (entangler < 0 || entangler > 2) return "entangler must be 0 (cx), 1 (cz) or 2 (cp)";
    if (L < 0) return "negative number of blocks";
    if (L > 0 && blocks == nullptr) return "blocks pointer is null";
    if (trotter && entangler != 0) return "Trotter ansatz expects the cx entangler";
    if (second_order && !trotter) return "second_order needs a Trotter ansatz";
    for (int i = 0; i < L; ++i) {
        const int c = blocks[i], t = blocks[L + i];
        if (c < 0 || c >= n || t < 0 || t >= n || c == t) return "not a valid structure of unit-blocks";
    }
    if (trotter && L > 0) {
        // parametric_circuit.py:391-423: layers of triplets (t,c),(c,t),(t,c) on adjacent qubits.
        if (L % (3 * (n - 1)) != 0) return "not a valid Trotterized block layout";
        for (int g = 0; g < L / 3; ++g) {
            const int c0 = blocks[3 * g], t0 = blocks[L + 3 * g];
            const int c1 = blocks[3 * g + 1], t1 = blocks[L + 3 * g + 1];
            const int c2 = blocks[3 * g + 2], t2 = blocks[L + 3 * g + 2];
            if (!(c0 == c2 && t0 == t2 && c0 == t1 && t0 == c1 && c0 == t0 + 1))
                return "not a valid Trotterized block layout";
        }
        if (second_order) {
            for (int i = 0; i < n / 2; ++i) {
                const int c1 = blocks[3 * i + 1], t1 = blocks[L + 3 * i + 1];
                if (!(c1 == 2 * i && t1 == 2 * i + 1)) return "unexpected layout of the leading half-layer";
            }
        }
    }
    if (trotter && L == 0 && second_order) return "second-order Trotter ansatz needs at least one layer";

    out = Program();
    out.n = n;
    out.entangler = entangler;
    out.num_blocks = L;
    out.tpb = entangler == 2 ? 5 : 4;
    out.trotter = trotter;
    out.second_order = second_order;
    out.tail_blocks = (trotter && second_order) ? 3 * (n / 2) : 0;
    out.blocks.assign(blocks, blocks + 2 * (size_t)L);
    for (int q = 0; q < n; ++q) out.groups.push_back({GROUP_FRONT, q, -1, q, 3 * q, -1, 0});
    for (int i = 0; i < L + out.tail_blocks; ++i) {
        const int j = i % L;
        int flags = 0;
        if (trotter && i % 3 == 0) flags |= FLAG_PRE_RZ;
        if (trotter && i % 3 == 2) flags |= FLAG_POST_RZ;
        out.groups.push_back({GROUP_BLOCK, blocks[j], blocks[L + j], n + j, 3 * n + out.tpb * j, j, flags});
    }
    return "";
}

namespace {

struct Sim {
    const Program& prog;
    const std::vector<int>& order;  // remaining group indices in execution order
    int col_bits;
    const std::vector<int>* remap = nullptr;  // optional address bit -> position map (sub-stage planning)
    int max_take = 1 << 30;
    uint64_t group_bits(const GateGroup& g) const {
        auto pos = [&](int q) { const int b = col_bits + q; return remap ? (*remap)[b] : b; };
        uint64_t bits = 1ull << pos(g.q0);
        if (g.q1 >= 0) bits |= 1ull << pos(g.q1);
        return bits;
    }
    // Number of groups executable with local set `mask`; optionally collects them and
    // reports the first group that was skipped although none of its bits was blocked yet.
    int run(uint64_t mask, std::vector<int>* taken, int* first_missing) const {
        uint64_t blocked = 0;
        int count = 0;
        if (first_missing) *first_missing = -1;
        for (int gi : order) {
            const GateGroup& g = prog.groups[gi];
            const uint64_t bits = group_bits(g);
            if (count >= max_take) break;
            if (bits & blocked) {
                blocked |= bits;
            } else if ((bits & mask) == bits) {
                ++count;
                if (taken) taken->push_back(gi);
            } else {
                if (first_missing && *first_missing < 0) *first_missing = gi;
                blocked |= bits;
            }
            if ((blocked & mask) == mask) break;  // nothing local is usable any more
        }
        return count;
    }
};

int popcount64(uint64_t v) { return __builtin_popcountll(v); }

}  // namespace

Plan make_plan(const Program& prog, int col_bits, int tile_bits, int low_bits, bool inverse, const std::vector<int>* subset, int nbits) {
    Plan plan;
    plan.nbits = nbits > 0 ? nbits : col_bits + prog.n;
    plan.col_bits = col_bits;
    plan.inverse = inverse;
    const int k = std::min(std::max(tile_bits, 2), plan.nbits);
    plan.tile_bits = k;
    low_bits = std::max(0, std::min(low_bits, k - 2));

    std::vector<int> order(subset ? subset->size() : prog.groups.size());
    for (size_t i = 0; i < order.size(); ++i) {
        const size_t j = inverse ? order.size() - 1 - i : i;
        order[i] = subset ? (*subset)[j] : (int)j;
    }

    const uint64_t forced = low_bits ? ((1ull << low_bits) - 1) : 0;
    while (!order.empty()) {
        Sim sim{prog, order, col_bits};
        uint64_t best_mask = 0;
        int best_count = -1;
        auto consider = [&](uint64_t m) {
            const int c = sim.run(m, nullptr, nullptr);
            if (c > best_count) { best_count = c; best_mask = m; }
        };
        if (plan.nbits <= k) {
            best_mask = (1ull << plan.nbits) - 1;
            best_count = sim.run(best_mask, nullptr, nullptr);
        } else {
            // (a) greedy growth along program order
            uint64_t m = forced;
            for (;;) {
                int miss = -1;
                sim.run(m, nullptr, &miss);
                if (miss < 0) break;
                const GateGroup& g = prog.groups[miss];
                uint64_t need = 1ull << (col_bits + g.q0);
                if (g.q1 >= 0) need |= 1ull << (col_bits + g.q1);
                if (popcount64(m | need) > k) break;
                m |= need;
            }
            consider(m);
            // (b) forced low bits + a contiguous window of the remaining budget
            for (int a = 0; a < plan.nbits; ++a) {
                uint64_t w = forced;
                for (int b = a; b < plan.nbits && popcount64(w) < k; ++b) w |= 1ull << b;
                consider(w);
            }
        }
        // pad the set to exactly k bits (deterministic: lowest free bits first)
        for (int b = 0; b < plan.nbits && popcount64(best_mask) < k; ++b) best_mask |= 1ull << b;

        Stage st;
        for (int b = 0; b < plan.nbits; ++b)
            if (best_mask >> b & 1) st.bits.push_back(b);
        sim.run(best_mask, &st.ops, nullptr);
        if (st.ops.empty()) {  // cannot happen for k >= low_bits + 2; keep the loop finite anyway
            st.ops.push_back(order.front());
        }
        std::vector<char> done(prog.groups.size(), 0);
        for (int gi : st.ops) done[gi] = 1;
        std::vector<int> rest;
        for (int gi : order)
            if (!done[gi]) rest.push_back(gi);
        order.swap(rest);
        plan.stages.push_back(std::move(st));
    }
    if (plan.stages.empty()) {  // empty program: one stage of identity so that copies still happen
        Stage st;
        for (int b = 0; b < k; ++b) st.bits.push_back(b);
        plan.stages.push_back(st);
    }
    return plan;
}

namespace {

// Beam search for the fewest sub-stages of one stage (matrix-core kernels: a sub-stage costs the same whatever the
// number of gate groups it absorbs, so the count is what matters).  State = set of executed groups; a move = a set of
// `r` register bits (for long stages: only those containing the bits of the first pending group), executing everything
// it can reach in program order.  Returns the chosen register-bit masks, or an empty vector when the stage is too
// large for the search.
std::vector<uint64_t> beam_substages_search(const std::vector<uint64_t>& op_bits, int k, int r, int width);

// The search only sees the stage's groups as bit masks over its local positions, and the stages of a layered ansatz repeat
// (a 240-step Trotter circuit at 20 qubits has 182 stages and a dozen distinct patterns; the V plan and the sweep plan of a
// workspace share all of them): results are remembered per pattern for the life of the process.
std::vector<uint64_t> beam_substages(const std::vector<uint64_t>& op_bits, int k, int r, int width) {
    static std::mutex mu;
    static std::map<std::vector<uint64_t>, std::vector<uint64_t>> memo;
    std::vector<uint64_t> key = op_bits;
    key.push_back(((uint64_t)k << 32) | ((uint64_t)r << 16) | (uint64_t)width);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    std::vector<uint64_t> out = beam_substages_search(op_bits, k, r, width);
    std::lock_guard<std::mutex> lock(mu);
    if (memo.size() > 4096) memo.clear();
    memo.emplace(std::move(key), out);
    return out;
}

std::vector<uint64_t> beam_substages_search(const std::vector<uint64_t>& op_bits, int k, int r, int width) {
    const int nops = (int)op_bits.size();
    if (nops == 0 || k <= r || k > 16 || (long long)nops * nops > 4000000) return {};
    std::vector<uint64_t> masks;
    for (uint64_t m = 0; m < (1ull << k); ++m)
        if (popcount64(m) == r) masks.push_back(m);
    struct State { std::vector<char> done; int ndone; int parent; uint64_t mask; uint64_t hash; };
    std::vector<std::vector<State>> levels(1);
    levels[0].push_back({std::vector<char>(nops, 0), 0, -1, 0, 0});
    auto advance = [&](const std::vector<char>& done, uint64_t mask, std::vector<char>& out) {
        uint64_t blocked = 0;
        int count = 0;
        out = done;
        for (int i = 0; i < nops; ++i) {
            if (done[i]) continue;
            const uint64_t b = op_bits[i];
            if (b & blocked) blocked |= b;
            else if ((b & mask) == b) { out[i] = 1; ++count; }
            else blocked |= b;
            if ((blocked & mask) == mask) break;
        }
        return count;
    };
    for (int depth = 0; depth < 4 * nops + 4; ++depth) {
        const std::vector<State>& cur = levels.back();
        int finished = -1;
        for (size_t i = 0; i < cur.size(); ++i)
            if (cur[i].ndone == nops) { finished = (int)i; break; }
        if (finished >= 0) {   // walk back
            std::vector<uint64_t> out;
            int idx = finished;
            for (int d = (int)levels.size() - 1; d > 0; --d) { out.push_back(levels[d][idx].mask); idx = levels[d][idx].parent; }
            std::reverse(out.begin(), out.end());
            return out;
        }
        std::vector<State> next;
        std::vector<char> tmp;
        for (size_t i = 0; i < cur.size(); ++i) {
            int first = 0;
            while (cur[i].done[first]) ++first;
            const uint64_t need = nops > 160 ? op_bits[first] : 0;
            for (uint64_t m : masks) {
                if ((m & need) != need) continue;
                const int c = advance(cur[i].done, m, tmp);
                if (c == 0) continue;
                uint64_t h = 1469598103934665603ull;
                for (int q = 0; q < nops; ++q) h = (h ^ (uint64_t)tmp[q]) * 1099511628211ull;
                bool dup = false;
                for (const State& s : next)
                    if (s.hash == h && s.ndone == cur[i].ndone + c && s.done == tmp) { dup = true; break; }
                if (!dup) next.push_back({tmp, cur[i].ndone + c, (int)i, m, h});
            }
        }
        if (next.empty()) return {};
        std::stable_sort(next.begin(), next.end(), [](const State& a, const State& b) { return a.ndone > b.ndone; });
        if ((int)next.size() > width) next.resize(width);
        levels.push_back(std::move(next));
    }
    return {};
}

}  // namespace

void split_substages(const Program& prog, Plan& plan, int reg_bits, int max_ops, int beam_width) {
    for (Stage& st : plan.stages) {
        st.subs.clear();
        const int k = (int)st.bits.size();
        const int r = std::min(reg_bits, k);
        std::vector<int> local_of(plan.nbits, 0);
        for (int j = 0; j < k; ++j) local_of[st.bits[j]] = j;
        std::vector<int> order = st.ops;
        // unlimited groups per sub-stage (matrix-core kernels): search for the fewest sub-stages first
        std::vector<uint64_t> chosen;
        if (max_ops >= (1 << 20)) {
            Sim sim{prog, order, plan.col_bits, &local_of};
            std::vector<uint64_t> op_bits;
            for (int gi : order) op_bits.push_back(sim.group_bits(prog.groups[gi]));
            chosen = beam_substages(op_bits, k, r, beam_width);
        }
        size_t next_choice = 0;
        while (!order.empty()) {
            Sim sim{prog, order, plan.col_bits, &local_of, max_ops};
            uint64_t best = 0;
            int best_count = -1;
            auto consider = [&](uint64_t m) {
                const int c = sim.run(m, nullptr, nullptr);
                if (c > best_count) { best_count = c; best = m; }
            };
            if (next_choice < chosen.size()) {
                consider(chosen[next_choice++]);
            } else if (k <= r) {
                consider((1ull << k) - 1);
            } else {
                uint64_t m = 0;
                for (;;) {  // greedy growth along program order
                    int miss = -1;
                    sim.run(m, nullptr, &miss);
                    if (miss < 0) break;
                    const uint64_t need = sim.group_bits(prog.groups[miss]);
                    if (popcount64(m | need) > r) break;
                    m |= need;
                }
                consider(m);
                for (int a = 0; a + r <= k; ++a) consider(((1ull << r) - 1) << a);  // contiguous windows
            }
            for (int b = 0; b < k && popcount64(best) < r; ++b) best |= 1ull << b;
            SubStage sub;
            for (int b = 0; b < k; ++b)
                if (best >> b & 1) sub.bits.push_back(b);
            sim.run(best, &sub.ops, nullptr);
            if (sub.ops.empty()) sub.ops.push_back(order.front());  // unreachable for r >= 2
            std::vector<char> done(prog.groups.size(), 0);
            for (int gi : sub.ops) done[gi] = 1;
            std::vector<int> rest;
            for (int gi : order)
                if (!done[gi]) rest.push_back(gi);
            order.swap(rest);
            st.subs.push_back(std::move(sub));
        }
        // execution order of the stage is now the concatenation of its sub-stages
        st.ops.clear();
        for (const SubStage& sub : st.subs) st.ops.insert(st.ops.end(), sub.ops.begin(), sub.ops.end());
    }
}

std::string check_plan(const Program& prog, const Plan& plan, const std::vector<int>* subset) {
    const int G = (int)prog.groups.size();
    std::vector<int> seen(G, 0);
    std::vector<int> last_on_bit(plan.nbits, plan.inverse ? G : -1);
    for (const Stage& st : plan.stages) {
        uint64_t mask = 0;
        for (int b : st.bits) mask |= 1ull << b;
        if ((int)st.bits.size() != std::min(plan.tile_bits, plan.nbits)) return "stage with wrong number of local bits";
        for (int gi : st.ops) {
            if (gi < 0 || gi >= G) return "group index out of range";
            if (seen[gi]++) return "group scheduled twice";
            const GateGroup& g = prog.groups[gi];
            int bits[2] = {plan.col_bits + g.q0, g.q1 >= 0 ? plan.col_bits + g.q1 : -1};
            for (int b : bits) {
                if (b < 0) continue;
                if (!(mask >> b & 1)) return "group uses a non-local bit";
                if (plan.inverse ? gi > last_on_bit[b] : gi < last_on_bit[b]) return "per-qubit order violated";
                last_on_bit[b] = gi;
            }
        }
    }
    std::vector<int> want(G, subset ? 0 : 1);
    if (subset)
        for (int gi : *subset) {
            if (gi < 0 || gi >= G) return "subset index out of range";
            want[gi] = 1;
        }
    for (int i = 0; i < G; ++i)
        if (seen[i] != want[i]) return want[i] ? "group not scheduled" : "group outside the subset scheduled";
    for (const Stage& st : plan.stages) {
        if (st.subs.empty()) continue;
        std::vector<int> local_of(plan.nbits, -1);
        for (size_t j = 0; j < st.bits.size(); ++j) local_of[st.bits[j]] = (int)j;
        std::vector<int> flat;
        for (const SubStage& sub : st.subs) {
            uint64_t m = 0;
            for (int b : sub.bits) m |= 1ull << b;
            for (int gi : sub.ops) {
                const GateGroup& g = prog.groups[gi];
                const int p0 = local_of[plan.col_bits + g.q0], p1 = g.q1 >= 0 ? local_of[plan.col_bits + g.q1] : p0;
                if (!(m >> p0 & 1) || !(m >> p1 & 1)) return "sub-stage group uses a non-register bit";
                flat.push_back(gi);
            }
        }
        if (flat != st.ops) return "sub-stages do not reproduce the stage order";
    }
    return "";
}

}  // namespace aqc
