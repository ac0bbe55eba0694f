// Experiment: how many sub-stages (4 register bits each) does a stage need?  Greedy (aqc_plan.cpp) vs beam search.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#include "../../aqc_research_amd/csrc/aqc_plan.h"
using namespace aqc;

struct Ctx {
    const Program* prog; const Plan* plan; std::vector<int> local_of; std::vector<int> ops;  // stage ops in order
    std::vector<uint64_t> bits;  // per op: mask of local bits
};
// run: given done flags, mask -> list of newly executable op indices (positions in ops)
static std::vector<int> run(const Ctx& c, const std::vector<char>& done, uint64_t mask) {
    uint64_t blocked = 0; std::vector<int> out;
    for (size_t i = 0; i < c.ops.size(); ++i) {
        if (done[i]) continue;
        const uint64_t b = c.bits[i];
        if (b & blocked) { blocked |= b; continue; }
        if ((b & mask) == b) out.push_back((int)i); else blocked |= b;
    }
    return out;
}
static int beam(const Ctx& c, int k, int width) {
    struct St { std::vector<char> done; int ndone; };
    std::vector<St> cur{{std::vector<char>(c.ops.size(), 0), 0}};
    int steps = 0;
    const int total = (int)c.ops.size();
    std::vector<uint64_t> masks;
    for (uint64_t m = 0; m < (1ull << k); ++m) if (__builtin_popcountll(m) == 4) masks.push_back(m);
    while (true) {
        for (auto& s : cur) if (s.ndone == total) return steps;
        ++steps;
        std::map<std::vector<char>, int> seen;
        std::vector<St> next;
        for (auto& s : cur) {
            // candidate masks: only those containing the bits of the first not-done, unblocked op (must make progress on frontier)
            for (uint64_t m : masks) {
                auto ex = run(c, s.done, m);
                if (ex.empty()) continue;
                St n = s; for (int i : ex) { n.done[i] = 1; } n.ndone += (int)ex.size();
                if (seen.count(n.done)) continue;
                seen[n.done] = 1; next.push_back(std::move(n));
            }
        }
        std::sort(next.begin(), next.end(), [](const St& a, const St& b) { return a.ndone > b.ndone; });
        if ((int)next.size() > width) next.resize(width);
        cur.swap(next);
        if (cur.empty()) return -1;
    }
}
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 16, L = argc > 2 ? atoi(argv[2]) : 40, colbits = argc > 3 ? atoi(argv[3]) : 0, kind = argc > 4 ? atoi(argv[4]) : 0;
    int k = argc > 5 ? atoi(argv[5]) : 12, width = argc > 6 ? atoi(argv[6]) : 64;
    std::vector<int32_t> blocks;
    bool trotter = false, second = false;
    if (kind == 2) {  // trotter-like: L = layers
        std::vector<std::pair<int,int>> base;
        std::vector<std::pair<int,int>> pairs; for (int s = 0; s < 2; ++s) for (int i = s; i < n - 1; i += 2) pairs.push_back({i, i + 1});
        for (int i = 0; i < L * (n - 1); ++i) base.push_back(pairs[i % pairs.size()]);
        std::vector<int32_t> c, t;
        for (auto& p : base) { c.push_back(p.second); t.push_back(p.first); c.push_back(p.first); t.push_back(p.second); c.push_back(p.second); t.push_back(p.first); }
        blocks = c; blocks.insert(blocks.end(), t.begin(), t.end()); L = (int)c.size(); trotter = true; second = true;
    } else {
        std::vector<std::pair<int,int>> pairs; for (int s = 0; s < 2; ++s) for (int i = s; i < n - 1; i += 2) pairs.push_back({i, i + 1});
        blocks.resize(2 * L);
        for (int i = 0; i < L; ++i) {
            std::pair<int,int> p;
            if (kind == 1) { int off = (n % 2 == 0) ? (i / (n / 2)) % 2 : 0; p = {(2 * i + off) % n, (2 * i + off + 1) % n}; } else p = pairs[i % pairs.size()];
            blocks[i] = p.first; blocks[L + i] = p.second;
        }
    }
    Program prog;
    std::string e = build_program(n, 0, blocks.data(), L, trotter, second, prog);
    if (!e.empty()) { printf("err %s\n", e.c_str()); return 1; }
    for (int inv = 0; inv < 2; ++inv) {
        Plan p = make_plan(prog, colbits, k, 3, inv);
        split_substages(prog, p, 4, 1 << 30);
        int greedy = 0, best = 0;
        for (auto& st : p.stages) {
            greedy += (int)st.subs.size();
            Ctx c; c.prog = &prog; c.plan = &p; c.local_of.assign(p.nbits, -1);
            for (size_t j = 0; j < st.bits.size(); ++j) c.local_of[st.bits[j]] = (int)j;
            // original program order within the stage (not the sub-stage order)
            std::vector<int> ops = st.ops; std::sort(ops.begin(), ops.end()); if (inv) std::reverse(ops.begin(), ops.end());
            c.ops = ops;
            for (int gi : ops) { const GateGroup& g = prog.groups[gi]; uint64_t b = 1ull << c.local_of[colbits + g.q0]; if (g.q1 >= 0) b |= 1ull << c.local_of[colbits + g.q1]; c.bits.push_back(b); }
            int b = beam(c, (int)st.bits.size(), width);
            printf("  inv=%d stage ops=%zu greedy=%zu beam=%d\n", inv, st.ops.size(), st.subs.size(), b);
            best += b;
        }
        printf("inv=%d total greedy=%d beam=%d (groups %zu)\n", inv, greedy, best, prog.groups.size());
    }
}
